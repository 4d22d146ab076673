// mg_roll.h — k_roll7: the step / fused-rollout kernel for the reference's default observation, the 7x7x3 egocentric view
// (MiniGridEnv.step + gen_obs, minigrid_env.py:525-650), round 3.  It replaces k_step<0, true, ...> for every level.
//
// What changed against k_step's 7x7 path (mg_step.h) and why -- measured there: one wave per SIMD at 65 536 envs, 824 VALU per
// wave-step, 179 VGPRs and 20 KB of LDS per wave (occupancy 2), the wave parked on LDS round trips 43 % of its cycles:
//  * TIME SPLIT.  A workgroup is still 64 consecutive envs, but NW (1, 2 or 4) wavefronts share them: wave w produces the outputs
//    of steps [split[w], split[w+1]) of the launch.  Dynamics are a few dozen instructions per step, the observation several
//    hundred, so wave w first replays steps 0 .. split[w]-1 SILENTLY (actions, dynamics, resets on its private copy of the 64 grids;
//    no observation, no stores) and then runs its own steps in full.  No barrier after the prologue, no inter-wave traffic: the
//    waves are independent instruction streams on (normally) four different SIMDs, i.e. four times the wavefronts for the same
//    batch without four lanes per env replicating every step's work.  The last wave owns the final state and writes it back.
//  * The observation leaves the lane-per-env domain as CELL CODES, not as bytes: each lane stages its env's 49 one-byte codes
//    (already masked by process_vis, in output order) in LDS (3.1 KB per wave instead of the 9.4 KB byte stream), and the encode
//    runs in OUTPUT space: lane u of a round takes dword u of the code stream (four cells), does four code -> (type, colour, state)
//    lookups and three byte permutes and stores the 12 bytes [12 u, 12 u + 12) of the wave's contiguous observation stream with one
//    store (obs7_quad; obs7_chunk, 16 bytes per lane with a phase, for a ragged last workgroup).
//    Nothing is assembled per env in registers (k_step held 37 packed dwords + 10 copy-out quads per lane).
//  * The view is handled as seven LINES of seven codes (2 VGPRs each) end to end: orientation by v_perm_b32 (byte reversal, the
//    8 x 8 byte transpose between "line = view column" and "line = view row"), opacity rows by v_dot4_u32_u8, process_vis rows
//    by carry propagation (an occluded fill to the right is (((t + g) ^ t) & t) | g; to the left the same on bit-reversed
//    words), visibility applied as byte masks.  No per-cell select, no per-cell bit insert.
// Every function of the observation pipeline is host-callable: mg_selftest_obs7 runs it on the CPU for the test-suite
// (tests/test_abi_cpu.py compares it with the oracle over rollouts); the primitives' device forms are checked on the GPU.
//
// Round 4: how a workgroup's waves share a fused launch now depends on the level (the host picks; mg_api.hip launch_step):
//  * LOG SPLIT (the 7x7 view of the ring levels, three or four waves): one DYNAMICS wave runs every step's action / transition / scalar outputs
//    once and logs (pose, changed cell, reset) per env and step in an LDS ring; the ENCODE waves keep private grid copies current from the log
//    and produce the observations, step j by encode wave j mod (NW - 1).  The time split's silent replays are gone.
//  * STAGED SPLIT (DynamicObstacles, the sentence levels, FullyObs): ONE copy of the grids; the dynamics wave also stages what the encode needs
//    -- the 49 codes per env (DynamicObstacles: its step is the per-lane placement loop of mg_dynobs.h; the sentence levels: the verifier of
//    mg_verify.h runs in the same wave) or a copy of its image-order stream (FullyObs) -- into a small ring of stagings; the other wave(s) run
//    only the output-space encode and the stores.
//  * the TIME SPLIT above remains for two-wave configurations that ask for it (MG_ROLL_SPLIT=0 / MG_FULL_SPLIT=0 / MG_SENT_SPLIT=0: A/B), the
//    SHARED ENCODE for one-step launches (Env.step: one wave steps, all waves of the workgroup encode that step).
#pragma once
#include "mg_step.h"
#include "mg_verify.h"
#include "mg_dynobs.h"

// analysis aid: -DMG_ISA_MARKS puts section comments into the device ISA (profiles/isa_stats.py --marks); never in the product build
#if defined(MG_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define MG_MARK(name) asm volatile("; ##MARK " name ::: "memory")
#else
#define MG_MARK(name) do { } while (0)
#endif

// -DMG_ATTRIBUTION -DMG_SPIN_COUNTS builds only (profiles/spin_counts.py): how often the waves of the log split wait for each other -- iterations of the
// dynamics wave's flow-control loop (kind 0) and of the encode waves' wait for the next log entry (kind 1), summed into the statistics buffer's scratch
// words 8 / 9 (mg_debug_stamps).  Measured in round 5 (profiles/r5/spin_counts.txt): at 65 536 Empty envs the dynamics wave waits 3-4 iterations per
// step (~5 % of its time), each encode wave ~1 (~1-2 %); at 32 768 envs the dynamics wave never waits and the encode waves ~6-7 % -- neither side idles:
// what paces a full chip is the SIMDs' shared issue (the same build runs 1.80 us per step at 32 768 envs = two waves per SIMD, 2.93 at 65 536 = four).
// (The counters themselves cost the attribution build 0.7 us per step: a separate switch.)
#if defined(MG_ATTRIBUTION) && defined(MG_SPIN_COUNTS) && defined(__HIP_DEVICE_COMPILE__)
#define MG_SPIN_DECL uint32_t mg_spins_[2] = { 0u, 0u }
#define MG_SPIN_COUNT(k) (mg_spins_[k]++)
#define MG_SPIN_REPORT(k) do { if (lane == 0 && mg_spins_[k]) atomicAdd(&P.counters[8 + (k)], (unsigned long long)mg_spins_[k]); } while (0)
#else
#define MG_SPIN_DECL do { } while (0)
#define MG_SPIN_COUNT(k) do { } while (0)
#define MG_SPIN_REPORT(k) do { } while (0)
#endif

// LDS words addressed by their LDS offset (the dynamic LDS of k_roll7 starts at LDS address 0): the inter-wave counters of the split loops.
// (tests/emu compiles these sources for the host: there the LDS is an ordinary array)
// MG_WAVE_ORDER: the DS operations of one wave execute in order, so between "the wave wrote" and "the wave (or a polling neighbour) reads" only
// the compiler has to be kept from reordering.  (tests/emu: a lane runs ahead of its neighbours between cross-lane operations -- a wave barrier there)
// MG_LOCKSTEP: places that rely on the wave executing in lockstep with nothing for the compiler to be told (empty in the product build).
#ifndef MG_EMU
#define MG_WAVE_ORDER() asm volatile("" ::: "memory")
#define MG_LOCKSTEP() do { } while (0)
#else
#define MG_WAVE_ORDER() emu_wave_barrier()
#define MG_LOCKSTEP() emu_wave_barrier()
#endif
// MG_MASKED_READS: obs7_view reads a view line as three aligned dwords; where the line leaves the grid those dwords hold a NEIGHBOUR env's cells
// (or a guard band) and every such byte is replaced by a wall before use.  The thread-sanitizer build of tests/emu is told that these reads are
// deliberate (another lane may be writing its own grid at that moment: the value is never used); empty everywhere else.
#if defined(MG_EMU) && defined(MG_EMU_TSAN)
extern "C" void AnnotateIgnoreReadsBegin(const char* file, int line);
extern "C" void AnnotateIgnoreReadsEnd(const char* file, int line);
#define MG_MASKED_READS_BEGIN(outside) const bool mg_masked_ = (outside); if (mg_masked_) AnnotateIgnoreReadsBegin(__FILE__, __LINE__)
#define MG_MASKED_READS_END() if (mg_masked_) AnnotateIgnoreReadsEnd(__FILE__, __LINE__)
#else
#define MG_MASKED_READS_BEGIN(outside) do { } while (0)
#define MG_MASKED_READS_END() do { } while (0)
#endif
#ifndef MG_EMU
#define MG_LDS_VU32 __attribute__((address_space(3))) volatile uint32_t
#define MG_LDS_AT(off) ((MG_LDS_VU32*)(uintptr_t)(uint32_t)(off))
#else
#define MG_LDS_VU32 ::emu::SyncWord             /* (release store / acquire load: what "volatile LDS word + in-order DS operations" means to a host compiler) */
#define MG_LDS_AT(off) ((MG_LDS_VU32*)(smem + (off)))
#endif

namespace mg {

// ---- VALU primitives with host equivalents ----
// v_perm_b32: byte i of the result = byte sel.byte[i] (0..7) of {hi:lo}; selector 0x0c gives 0x00, 0x0d and above 0xff
MG_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(hi, lo, sel);
#else
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xFFu;
    const uint32_t b = s < 8u ? (uint32_t)(v >> (8u * s)) & 0xFFu : (s == 0x0Cu ? 0u : 0xFFu);
    r |= b << (8 * i);
  }
  return r;
#endif
}
// v_dot4_u32_u8: sum of the four byte products + c
MG_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_udot4(a, b, c, false);
#else
  for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
  return c;
#endif
}
MG_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x);
#else
  uint32_t r = 0;
  for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
  return r;
#endif
}
MG_HD uint32_t mul24(uint32_t a, uint32_t b) {          // both < 2^24: the full-rate 24-bit multiply
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}
// ((w >> 8 n) & 0xff) << 2 -- the byte offset of entry `byte n of w` in a table of dwords -- as ONE instruction: an SDWA shift that
// reads its operand through a byte select (the compiler emits an extract and a shift-add).  `two` = a register holding 2.
#if defined(__HIP_DEVICE_COMPILE__)
#define MG_BYTE_X4(w, n, two) ({ uint32_t _r; asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #n : "=v"(_r) : "v"(two), "v"(w)); _r; })
#else
#define MG_BYTE_X4(w, n, two) ((((w) >> (8 * (n))) & 0xFFu) << (two))
#endif
// four bits -> four bytes of 0x00 / 0xff (bit i -> byte i)
MG_HD uint32_t expand4(uint32_t bits) {
  return perm_b32(0u, 0u, 0x0C0C0C0Cu | (mul24(bits & 15u, 0x00204081u) & 0x01010101u));
}

// One row of Grid.process_vis (core/grid.py:291-328) like vis_row (mg_device.h), with the two occluded fills done by carry
// propagation instead of Kogge-Stone steps: adding the seeds g (a subset of the transparent cells t) to t ripples a carry from
// every seed through the run of 1s above it, so (t + g) ^ t marks each seed's run up to and including the first opaque cell,
// "& t" drops that cell, "| g" restores seeds that sat inside another seed's ripple.  The fill toward lower indices is the
// same on bit-reversed words.  tests/test_abi_cpu.py checks all 2^14 (m, t) pairs against vis_row and the literal loops.
MG_HD void vis_row_carry(uint32_t m, uint32_t t, uint32_t* m_out, uint32_t* up_out) {
  const uint32_t g = m & t;
  const uint32_t fr = (((t + g) ^ t) & t) | g;
  const uint32_t tr = brev32(t), gr = brev32(g);
  const uint32_t fl = brev32((((tr + gr) ^ tr) & tr) | gr);
  const uint32_t s1 = fr & 0x3Fu;                 // sweep-1 sources i = 0..5: light i+1 here, i and i+1 above
  const uint32_t s2 = (fr | fl) & 0x7Eu;          // sweep-2 sources i = 6..1: light i-1 here, i and i-1 above
  const uint32_t s1s = s1 << 1, s2s = s2 >> 1;
  *m_out = m | s1s | s2s;
  *up_out = s1 | s1s | s2 | s2s;
}

// Seven lines of seven cell codes: line t, byte j; lo = bytes 0..3, hi = bytes 4..6 (top byte 0).
struct View7 { uint32_t lo[7], hi[7]; };

// 4 x 4 byte transpose: c_j.byte[t] = r_t.byte[j]
MG_HD void transpose4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3) {
  const uint32_t p0 = perm_b32(r1, r0, 0x05010400u), p1 = perm_b32(r1, r0, 0x07030602u);   // bytes of r0 / r1 interleaved
  const uint32_t p2 = perm_b32(r3, r2, 0x05010400u), p3 = perm_b32(r3, r2, 0x07030602u);
  c0 = perm_b32(p2, p0, 0x05040100u); c1 = perm_b32(p2, p0, 0x07060302u);
  c2 = perm_b32(p3, p1, 0x05040100u); c3 = perm_b32(p3, p1, 0x07060302u);
}
// B.line[j].byte[t] = A.line[t].byte[j] (the eighth line / byte is zero)
MG_HD void view7_transpose(const View7& A, View7& B) {
  uint32_t x;
  transpose4(A.lo[0], A.lo[1], A.lo[2], A.lo[3], B.lo[0], B.lo[1], B.lo[2], B.lo[3]);
  transpose4(A.lo[4], A.lo[5], A.lo[6], 0u, B.hi[0], B.hi[1], B.hi[2], B.hi[3]);
  transpose4(A.hi[0], A.hi[1], A.hi[2], A.hi[3], B.lo[4], B.lo[5], B.lo[6], x);
  transpose4(A.hi[4], A.hi[5], A.hi[6], 0u, B.hi[4], B.hi[5], B.hi[6], x);
  (void)x;
}

typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));

// gen_obs up to the encode (minigrid_env.py:597-632, core/grid.py:110-143, 291-328): the agent's 7x7 view as 49 cell codes in
// image order k = vx * 7 + vy, invisible cells already 0 (= the code of "unseen"), the agent's own cell showing what it carries
// (:623-630) -- written to `codes` (49 bytes, any alignment; the 50th byte is left alone).
// View cell (vx, vy) is world cell agent + f (6 - vy) + r (vx - 3), f = DIR_TO_VEC[dir], r = (-f.y, f.x).  World x is the
// contiguous direction of the grid, so the seven LINES read are world rows: the view's columns when the agent faces +-x
// (line t = vx, byte = vy), its rows when it faces +-y (line t = vy, byte = vx); bytes reversed when the view index runs
// against world x (east, south).  Cells outside the grid read a neighbour env's cells or a guard band and are replaced by walls.
MG_HD void obs7_view(const Agent& a, const uint8_t* mygrid, int W, int H, bool see_through, View7& O) {
  const uint32_t d = a.dir;
  const int ax = (int)a.x, ay = (int)a.y;
  const bool horiz = (d & 1u) == 0u, rev = d < 2u;
  const int row_base = ay + (d == 0u ? -3 : d == 2u ? 3 : d == 1u ? 6 : -6);   // world row of line 0 ...
  const int sy = (d == 0u || d == 3u) ? 1 : -1;                                 // ... and the step to the next line
  const int x0 = ax - (d == 0u ? 0 : d == 2u ? 6 : 3);                          // world x of a line's first byte in memory
  // validity, in memory order: byte m is world x0 + m, line t is world row row_base + sy t (neither range is ever empty)
  const int mlo = max(0, -x0), mhi = min(6, W - 1 - x0);
  const uint32_t mm = ((2u << mhi) - 1u) & ~((1u << mlo) - 1u);
  const int tlo = sy > 0 ? max(0, -row_base) : max(0, row_base - (H - 1));
  const int thi = sy > 0 ? min(6, H - 1 - row_base) : min(6, row_base);
  const uint32_t lm = ((2u << thi) - 1u) & ~((1u << tlo) - 1u);
  const uint32_t bm_lo = expand4(mm), bm_hi = expand4(mm >> 4) & 0x00FFFFFFu;
  const uint32_t sel_lo = rev ? 0x03040506u : 0x03020100u, sel_hi = rev ? 0x0C000102u : 0x0C060504u;
  const uint32_t WALL4 = CELL_WALL_GREY * 0x01010101u;
  // A line is read as the three ALIGNED dwords that hold its seven bytes and shifted into place: one unaligned 8-byte LDS access
  // takes ~70 cycles of the CU's LDS pipe on gfx950 (the lanes are served one by one; measured, profiles/r3/lds_notes.md), three
  // aligned dword reads take ~6.  (mygrid is 4-byte aligned: LDS carve-up and GS are multiples of 4.)
  const int off0 = row_base * W + x0, lstep = sy * W;
  View7 N;                                                     // natural orientation
#pragma unroll
  for (int t = 0; t < 7; t++) {
    const int off = off0 + t * lstep;
    const uint32_t* ap = (const uint32_t*)(mygrid + (off & ~3));
    const uint32_t sh = (uint32_t)off & 3u;
    MG_MASKED_READS_BEGIN((off & ~3) < 0 || (off & ~3) + 12 > W * H);
    const uint32_t d0 = ap[0], d1 = ap[1], d2 = ap[2];
    MG_MASKED_READS_END();
    const uint32_t rlo = funnel_bytes(d1, d0, sh), rhi = funnel_bytes(d2, d1, sh);
    const uint32_t on = 0u - ((lm >> t) & 1u);
    const uint32_t ml = bm_lo & on, mh = bm_hi & on;
    const uint32_t qlo = (rlo & ml) | (WALL4 & ~ml), qhi = (rhi & mh) | (WALL4 & ~mh);
    N.lo[t] = perm_b32(qhi, qlo, sel_lo);
    N.hi[t] = perm_b32(qhi, qlo, sel_hi);
  }
  // rows of the view for everyone: line = vy, byte = vx
  View7 R, X;
  view7_transpose(N, X);
#pragma unroll
  for (int t = 0; t < 7; t++) { R.lo[t] = horiz ? X.lo[t] : N.lo[t]; R.hi[t] = horiz ? X.hi[t] : N.hi[t]; }
  // the agent's own cell (vx 3, vy 6) shows what it carries, or nothing
  R.lo[6] = (R.lo[6] & 0x00FFFFFFu) | ((a.carry ? a.carry : (uint32_t)CELL_EMPTY) << 24);
  if (!see_through) {
    // process_vis, rows bottom-up; the visibility of row vy as a byte mask over its codes
    uint32_t m = 1u << 3;
#pragma unroll
    for (int j = 6; j >= 0; j--) {
      // transparency bits of the row: bit vx = !(code & OPAQUE_BIT)
      const uint32_t tl = (~R.lo[j] & 0x80808080u) >> 7, th = (~R.hi[j] & 0x00808080u) >> 7;
      const uint32_t tb = udot4(th, 0x00402010u, udot4(tl, 0x08040201u, 0u));
      uint32_t vr, up;
      vis_row_carry(m, tb, &vr, &up);
      R.lo[j] &= expand4(vr);
      R.hi[j] &= expand4(vr >> 4);
      m = up;
    }
  }
  // image order: line = vx, byte = vy
  view7_transpose(R, O);
}

// The 49 codes of an env (seven lines of seven bytes, image order) as the 13 dwords D[i] = bytes [4 i, 4 i + 4) of its code string
// (D[12]: one byte).  Every D[i] takes its bytes from at most two neighbouring line registers: one v_perm_b32 each.
MG_HD void view7_pack(const View7& O, uint32_t D[13]) {
  // the code string is lo0 (4 bytes) hi0 (3) lo1 hi1 ... : byte 7 t + j is line t byte j
#define MG_B(k) ((k) % 7 < 4 ? O.lo[(k) / 7] : O.hi[(k) / 7])                 /* register holding code byte k */
#define MG_I(k) ((uint32_t)((k) % 7 < 4 ? (k) % 7 : (k) % 7 - 4))             /* its byte index inside that register */
#pragma unroll
  for (int i = 0; i < 12; i++) {
    const int k = 4 * i;
    // bytes k .. k+3: the register of byte k is "lo" of the perm, the register of byte k+3 is "hi" (the same one when they coincide)
    const uint32_t rl = MG_B(k), rh = MG_B(k + 3);
    uint32_t sel = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const bool in_lo = ((k + b) / 7 == k / 7) && (((k + b) % 7 < 4) == (k % 7 < 4));
      sel |= (in_lo ? MG_I(k + b) : 4u + MG_I(k + b)) << (8 * b);
    }
    D[i] = perm_b32(rh, rl, sel);
  }
  D[12] = (O.hi[6] >> 16) & 0xFFu;
#undef MG_B
#undef MG_I
}

// Lane `lane` stages its env's code string at byte 49 * lane of the wave's code stream with ALIGNED dword stores (StreamEmit: the
// string shifted by the env's byte phase; the last, partial dword completed with the first bytes of the next lane's string, next0).
MG_HD void obs7_stage(const uint32_t D[13], uint32_t next0, int lane, uint32_t* stream) {
  StreamEmit em;
  em.setup(stream, (uint32_t)(lane * VIEW_CELLS), (uint32_t)VIEW_CELLS);
  em.first(D[0]);
#pragma unroll
  for (int i = 1; i < 12; i++) em.put(D[i]);
  em.put_last(D[12], next0);
}

// Grid.encode(vis_mask) (core/grid.py:244-268) in OUTPUT space: the 16 bytes [16 c, 16 c + 16) of a contiguous observation stream
// whose cell g (= env * 49 + k) has its code at codes[g] and its three bytes at 3 g.  16 c = 3 q + ph with ph = c mod 3, so the
// chunk is bytes ph .. ph + 15 of the 18 bytes of cells q .. q + 5.  `slut` = cell code -> type | colour << 8 | state << 16.
// (Reads the three aligned dwords from codes[q & ~3] on: the staging buffer carries 16 bytes of slack.)
MG_HD void obs7_chunk(uint32_t c, const uint8_t* codes, const uint32_t* slut, uint32_t out[4]) {
  const uint32_t q = mul24(c, 0xAAAB0u) >> 17;                 // 16 c / 3 (exact for 16 c < 2^16; the full-rate 24-bit multiply)
  const uint32_t ph8 = (16u * c - mul24(q, 3u)) * 8u;
  const uint32_t* cw = (const uint32_t*)codes + (q >> 2);      // three aligned dwords, shifted: see obs7_view on unaligned LDS accesses
  const uint32_t w0 = cw[0], w1 = cw[1], w2 = cw[2], qs = q & 3u;
  const uint32_t wl = funnel_bytes(w1, w0, qs), wh = funnel_bytes(w2, w1, qs);
  const uint32_t two = 2u;
#if defined(__HIP_DEVICE_COMPILE__)
  // the table as an LDS byte address (an integer): the SDWA result IS the ds_read address, no pointer arithmetic in between
  typedef __attribute__((address_space(3))) const uint32_t lds_u32;
  // (k_roll7 keeps the table at LDS address 0 -- the start of its dynamic LDS, it has no static LDS -- and checks that at launch)
#define MG_LUT(off) (*(lds_u32*)(uintptr_t)(off))
#else
  const uint8_t* lut = (const uint8_t*)slut;
#define MG_LUT(off) (*(const uint32_t*)(lut + (off)))
#endif
  const uint32_t t0 = MG_LUT(MG_BYTE_X4(wl, 0, two)), t1 = MG_LUT(MG_BYTE_X4(wl, 1, two)), t2 = MG_LUT(MG_BYTE_X4(wl, 2, two));
  const uint32_t t3 = MG_LUT(MG_BYTE_X4(wl, 3, two)), t4 = MG_LUT(MG_BYTE_X4(wh, 0, two)), t5 = MG_LUT(MG_BYTE_X4(wh, 1, two));
#undef MG_LUT
  // the 18 bytes of the six triples as dwords p0..p4, one byte permute each
  const uint32_t p0 = perm_b32(t1, t0, 0x04020100u), p1 = perm_b32(t2, t1, 0x05040201u), p2 = perm_b32(t3, t2, 0x06050402u),
                 p3 = perm_b32(t5, t4, 0x04020100u), p4 = t5 >> 8;
#if defined(__HIP_DEVICE_COMPILE__)
  out[0] = __funnelshift_r(p0, p1, ph8); out[1] = __funnelshift_r(p1, p2, ph8);
  out[2] = __funnelshift_r(p2, p3, ph8); out[3] = __funnelshift_r(p3, p4, ph8);
#else
  const uint32_t p[5] = { p0, p1, p2, p3, p4 };
  for (int i = 0; i < 4; i++) out[i] = ph8 ? (p[i] >> ph8) | (p[i + 1] << (32u - ph8)) : p[i];
#endif
}

// The same encode per QUAD of cells: the 12 bytes [12 u, 12 u + 12) of the stream are the triples of the four codes in dword u of the
// code stream -- one aligned code dword, four lookups, three byte permutes, no phase: 4 u cells never straddle anything.  A round of 64
// lanes writes 768 contiguous bytes with one 12-byte store per lane.  Needs a stream of 4 k cells (a full workgroup: 64 envs).
// (in two halves, so that a loop can have the lookups of the next quad in flight while it packs this one)
MG_HD void obs7_quad_lookup(uint32_t w, const uint32_t* slut, uint32_t t[4]) {
  const uint32_t two = 2u;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) const uint32_t lds_u32;
#define MG_LUT(off) (*(lds_u32*)(uintptr_t)(off))
#else
  const uint8_t* lut = (const uint8_t*)slut;
#define MG_LUT(off) (*(const uint32_t*)(lut + (off)))
#endif
  t[0] = MG_LUT(MG_BYTE_X4(w, 0, two)); t[1] = MG_LUT(MG_BYTE_X4(w, 1, two)); t[2] = MG_LUT(MG_BYTE_X4(w, 2, two)); t[3] = MG_LUT(MG_BYTE_X4(w, 3, two));
#undef MG_LUT
}
MG_HD void obs7_quad_pack(const uint32_t t[4], uint32_t out[3]) {
  out[0] = perm_b32(t[1], t[0], 0x04020100u); out[1] = perm_b32(t[2], t[1], 0x05040201u); out[2] = perm_b32(t[3], t[2], 0x06050402u);
}
MG_HD void obs7_quad(uint32_t u, const uint8_t* codes, const uint32_t* slut, uint32_t out[3]) {
  uint32_t t[4];
  obs7_quad_lookup(((const uint32_t*)codes)[u], slut, t);
  obs7_quad_pack(t, out);
}
// compile-time choice of the encode of full workgroups (the chunk form stays for ragged ones): 0 builds the round-3 chunk encode for A/B runs
#ifndef MG_ENCODE_QUADS
#define MG_ENCODE_QUADS 1
#endif

struct Out12 { uint32_t x, y, z; };                    // 4-byte aligned: one global_store_dwordx3
// The trajectory is written once and not read again by the kernel: NONTEMPORAL stores (`nt`: streamed through L2 instead of staying
// resident as dirty lines) -- measured in round 4 (profiles/r4/ab_nt.txt): Empty-8x8 x 65 536 2.36 -> 2.24 us per step, DoorKey-8x8 x 262 144
// 10.95 -> 8.85: the batches that run in several rounds of workgroups re-read grids and spare episodes through the same L2.
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// ... while a SHORT burst that follows an idle stream is absorbed by the write-back caches (a single 20-step launch of the headline, 220 MB
// against 256 MB of Infinity Cache: 2.15 us per step with plain stores, 2.39 nontemporal), and a one-step launch's observation is read by
// the consumer right away.  The host decides per launch (StepParams::nt, mg_api.hip `launch_step`: bytes written since the stream was last
// known idle); the scalars always take plain stores (nontemporal there measured slower: profiles/r4/ab_nt2.txt).
// A/B builds only (profiles/variant_build.py ... -DMG_OBS_STORE_AUX=<aux>): the nontemporal instantiation's observation stores as BUFFER stores with
// explicit cache-policy bits (aux: 1 = sc0, 2 = nt, 16 = sc1; sc1 = write-through, the line is dropped from the XCD's L2 -- MI355X_MICROARCH.md); the
// product build (-1) keeps the flat stores below
#ifndef MG_OBS_STORE_AUX
#define MG_OBS_STORE_AUX -1
#endif
// base: wave-uniform (a workgroup's observation block of one step), off: this lane's byte offset inside it
MG_D void store12(uint8_t* base, uint32_t off, const Out12& v, bool nt) {
#if MG_OBS_STORE_AUX >= 0 && defined(__HIP_DEVICE_COMPILE__)
  if (nt) {
    u32x3_t w; w.x = v.x; w.y = v.y; w.z = v.z;
    __builtin_amdgcn_raw_buffer_store_b96(w, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000), (int)off, 0, MG_OBS_STORE_AUX);
    return;
  }
#endif
  uint8_t* p = base + off;
  if (nt) { u32x3_t w; w.x = v.x; w.y = v.y; w.z = v.z; __builtin_nontemporal_store(w, (u32x3_t*)p); }
  else *(Out12*)p = v;
}
MG_D void store16(uint8_t* base, uint32_t off, const uint4& v, bool nt) {
#if MG_OBS_STORE_AUX >= 0 && defined(__HIP_DEVICE_COMPILE__)
  if (nt) {
    u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(w, __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000), (int)off, 0, MG_OBS_STORE_AUX);
    return;
  }
#endif
  uint8_t* p = base + off;
  if (nt) { u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, (u32x4_t*)p); }
  else *(uint4*)p = v;
}
// NQ quads by threads l0, l0 + STRIDE, ...: software-pipelined -- every code dword first, then the lookups of quad it + 1 are issued before
// quad it is packed and stored
#ifndef MG_ALIGN_ROUNDS
#define MG_ALIGN_ROUNDS 1        // (0: A/B builds -- every workgroup's rounds counted from its block's first quad, as before round 6)
#endif
// LINE-ALIGNED ROUNDS (round 6).  A workgroup's block of a step's observation tensor is 64 * 147 = 9 408 bytes = 73.5 cache lines: every other workgroup's block
// starts 64 bytes into a 128-byte line, and with rounds counted from the block's first quad each of its 768-byte rounds had a half line at both ends.  `shift`
// (0, or 16 quads = 192 bytes for a block that starts mid-line: 12 * 16 = 64 mod 128) moves the SHORT round to the front: quads [0, 16) first, then full rounds from
// quad 16 on, each starting on a line.  Measured on the attribution build before it was built (blocks moved down to their line start, MG_EXP bit 16384,
// profiles/r6/attribution_store_alignment.txt): Empty-8x8 x 65 536 2.18-2.25 -> 2.06-2.09 us per step.  (Needs NQ = a multiple of STRIDE + 16.)
template <int STRIDE, int NQ, bool NT>
MG_D void encode_quads(int l0, const uint8_t* codes, const uint32_t* slut, uint8_t* obase, bool do_store = true, int shift = 0) {
  constexpr int NIT = (NQ + STRIDE - 1) / STRIDE, REM = NQ - STRIDE * (NIT - 1);     // REM quads in the short round
  uint32_t cw[NIT], tq[2][4];
  // round it of this lane: the full rounds first (from quad `shift` on), the short round last in program order -- its quads are the block's first ones when shifted
  auto quad_of = [&](int it) { return it + 1 < NIT ? shift + STRIDE * it + l0 : min((shift ? 0 : STRIDE * (NIT - 1)) + l0, NQ - 1); };
#pragma unroll
  for (int it = 0; it < NIT; it++) cw[it] = ((const uint32_t*)codes)[quad_of(it)];
  obs7_quad_lookup(cw[0], slut, tq[0]);
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int u = quad_of(it);
    if (it + 1 < NIT) obs7_quad_lookup(cw[it + 1], slut, tq[(it + 1) & 1]);
    uint32_t o3[3];
    obs7_quad_pack(tq[it & 1], o3);
    Out12 v; v.x = o3[0]; v.y = o3[1]; v.z = o3[2];
    if ((it + 1 < NIT || l0 < REM) && do_store) store12(obase, (uint32_t)u * 12u, v, NT);
  }
}

constexpr int ROLL_CODES_BYTES = 64 * VIEW_CELLS + 16;        // one wave's code staging (+ slack for the 8-byte accesses)
constexpr int ROLL_MAX_WAVES = 4;
#ifndef MG_DPRIO
#define MG_DPRIO 1
#endif
#ifndef MG_ROLL_LOG_STEPS
#define MG_ROLL_LOG_STEPS 8       // (the inter-wave protocol's stress build, tests/test_gpu_lds_protocol.py, makes it 2: the dynamics wave then waits on the encode waves in nearly every step)
#endif
constexpr int ROLL_LOG_STEPS = MG_ROLL_LOG_STEPS;             // split mode: entries of the dynamics wave's step log (a ring in LDS; power of two)
static_assert(ROLL_LOG_STEPS >= 2 && (ROLL_LOG_STEPS & (ROLL_LOG_STEPS - 1)) == 0, "the step log is a power-of-two ring");
constexpr int ROLL_LOG_SYNC_BYTES = 64;                       // ... behind its progress counters
// -DMG_SCAL_BY_ENCODE=1 (A/B builds, profiles/variant_build.py; round 6): the 16-byte scalar record of a step stored by the step's ENCODE wave instead of
// the dynamics wave.  The store-only model of this launch shape (profiles/microbench/rollstore2.hip, profiles/r6/rollstore2.txt) runs 2.27 us per step with
// the scalars stored by a wave that is up to eight steps ahead of the observations and 2.05 with the scalars stored next to their observations; the kernel
// does not follow: Empty-8x8 x 65 536 2.19-2.23 -> 2.18 us per step, DoorKey-8x8 x 262 144 8.15 -> 8.46-8.49 (profiles/r6/ab_scalars_by_encode_wave.txt).
// Measured, not adopted; the whole GPU suite is green with it.
#ifndef MG_SCAL_BY_ENCODE
#define MG_SCAL_BY_ENCODE 0
#endif
constexpr int ROLL_LOG_BYTES = ROLL_LOG_SYNC_BYTES + ROLL_LOG_STEPS * 64 * 8 + (MG_SCAL_BY_ENCODE ? ROLL_LOG_STEPS * 64 * 2 : 0);   // (pose, delta) per env and entry (+ the step count the reward is priced on)
#ifndef MG_RG_WPE
// waves per SIMD the register allocation of k_roll7<GG_ROOMGRID> (the 7x7 view) aims at: 3 = 151 VGPRs, 4 = 128 (eight dwords spilled).  With 151 only
// three of a CU's four workgroup slots were usable (12 waves): round 6, profiles/r6/ab_roomgrid_waves_per_simd.txt -- KeyCorridorS3R3 x 131 072 22.0 -> 24.5 G,
// Unlock 20.0 -> 21.2, GoToRedBall x 65 536 12.7 -> 16.3, x 131 072 15.6 -> 18.1 (x 32 768, two workgroups per CU: unchanged)
#define MG_RG_WPE 4
#endif
#ifndef MG_LR_WPE
// ... and of k_roll7<GG_LIGHT> / <GG_ROOMS> (the 7x7 view, not STAGED): 3 = 159 / 140 VGPRs, 4 = 128 (profiles/r6/ab_light_rooms_waves_per_simd.txt, x 131 072:
// Fetch-8x8-N3 19.8 -> 24.5 G, BabyAI-PickupDist 14.8 -> 16.7, PutNextLocal 11.4 -> 13.2, OpenRedDoor 18.4 -> 21.5; Memory / LockedRoom unchanged)
#define MG_LR_WPE 4
#endif
#ifndef MG_DYN_WPE
// waves per SIMD the register allocation of k_roll7<GG_DYNOBS> aims at: 4 = 128 VGPRs (three spilled, outside the placement loop) against 149.
// Measured (profiles/r4/dynobs_waves_sweep2.txt, 65 536 envs): 16x16 12.5 us per step against 18.2, 8x8 12.3 against 17.8, Random-6x6 17.6 against 24.5
#define MG_DYN_WPE 4
#endif
#ifndef MG_INSTR_LDS
// the sentence levels' instruction records staged in LDS for the launch (1) or left in global memory (0).  Measured after the verifier was
// restated (mg_verify.h; profiles/r4/bosslevel_records_ab.txt): BossLevel x 131 072 73.3 us per step with the records in LDS, 64.7 in global
// memory -- the 21 KB of records leave two single-wave workgroups per CU instead of four, and the verifier now makes few enough accesses
// for that to cost more than their latency.  Both forms pass the sentence levels' GPU tests; the LDS form stays for A/B builds.
#define MG_INSTR_LDS 0
#endif
constexpr int ROLL_INSTR_STRIDE = INSTR_WORDS + 1;            // k_roll7<GG_SENTENCE>: u64 words between the envs' instruction records in LDS (odd: conflict-free 8-byte reads)
constexpr int ROLL_DSPLIT_RING = 4;                           // k_roll7<GG_DYNOBS / GG_SENTENCE>, split: code stagings between the dynamics wave and the encode waves (P.dring: a power of two up to this)

// LDS carve-up (bytes) of a k_roll7 workgroup, computed by the host (mg_api.hip roll_layout) and passed in StepParams:
//   [0, 1024) code -> triple table | guard | NW private copies of the 64 grids (GS bytes per env) | guard | NW code stagings |
//   shadow grids (every env's next spare episode) | shadow agent / aux words | the caller's actions [T][64]
// StepParams: off_grid = first private grid copy, off_T = first code staging, off_shadow / off_spr / off_act as in k_step;
// split[w] = first step wave w produces (split[NW] = T).

// FullyObsWrapper.observation (wrappers.py:419-426) in the same kernel (FULL): the observation is the WHOLE grid, image[x][y] =
// encode(grid[y][x]) with the agent's cell = (10, 0, dir).  There is nothing to gather or to mask, so the per-env work of a step is the
// dynamics alone, provided the encode finds its input ready: every wave keeps, next to its row-major copy of the 64 grids (what the
// dynamics index), a second image of them in IMAGE order -- one contiguous code stream, W*H bytes per env, k = x*H + y -- that follows
// the grids cell by cell (the one dirty cell of a step; a reset copies the shadow spare's image stream).  A step patches the agent's
// cell into the stream, runs the same output-space encode as the 7x7 view over it (obs7_quad: lane u = bytes [12 u, 12 u + 12) of the
// wave's observations) and restores the cell.  Round 2 ran FullyObs with four lanes per env replicating the dynamics (k_step<1,.,4>).
MG_HD void image_stream_build(const uint8_t* g, uint8_t* gt, int W, int H) {         // gt[x*H + y] = g[y*W + x]
  for (int x = 0; x < W; x++)
    for (int y = 0; y < H; y++) gt[x * H + y] = g[y * W + x];
}
// GG_DYNOBS (round 4): DynamicObstacles with the draws of its step() and reset() inside the step loop -- RNG = the env streams' type, one
// stream per lane in registers for the whole launch (mg_dynobs.h).  The level has no spare ring: an env whose episode ended is redrawn in
// place by its own lane; the encode waves of the split follow the dynamics wave's grids through the obstacle LIST each step logs.
// ONE (round 5): the ONE-STEP specialisation -- Env.step() and the reset observation (T = 1).  A one-step launch runs through its code exactly once, so what it
// costs beyond the launch itself is largely instruction fetch: the general kernel carries four loop shapes (time split, log split, staged split, encode
// waves), each with its own copy of the dynamics / observation code, and the shadow-spare staging of fused launches.  ONE compiles all of that out: one
// wave steps, the workgroup's waves share the encode (`share`) or it is the only wave; a spare episode comes straight from the ring in HBM.
// STAGED (round 6): the staged split for the 7x7 view of the BIG grids (more than 256 cells: the 22 x 22 mazes, MultiRoom's 25 x 25, ...) of every rule
// group.  A private copy of the 64 grids per wave is 32-41 KB there, so those levels ran ONE wave per workgroup -- 4 (3) workgroups per CU, one wave per
// SIMD, every latency of the step exposed.  With one copy per workgroup (the dynamics wave's, which also stages the step's 49 codes per env) a second
// wave takes the output-space encode and the stores, exactly as for the sentence levels (same 22 x 22 grids) since round 4.
template <int GG, bool FULL, bool NT, class RNG = Pcg64Stream, bool ONE = false, bool STAGED = false>
__global__ void __launch_bounds__(64 * ROLL_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(STAGED ? 2 : (gg_group(GG) == GG_NONE && !FULL) ? 4 : (gg_group(GG) == GG_ROOMGRID && !FULL) ? MG_RG_WPE : gg_group(GG) == GG_DYNOBS ? MG_DYN_WPE : gg_group(GG) == GG_SENTENCE ? 2 : ((gg_group(GG) == GG_LIGHT || gg_group(GG) == GG_ROOMS) && !FULL) ? MG_LR_WPE : 3, 8))) k_roll7(const StepParams P) {
  static_assert(!STAGED || (!FULL && !ONE && gg_group(GG) != GG_DYNOBS && gg_group(GG) != GG_SENTENCE), "STAGED: the 7x7 view of the ring levels (the others stage by themselves)");
  static_assert(gg_group(GG) != GG_DYNOBS || !FULL, "DynamicObstacles' in-loop path is built for the 7x7 view");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int NW = nthreads >> 6;
  const int wg = blockIdx.x;      // (an XCD-contiguous remap of the workgroups and a wait after every store were measured: no gain, profiles/r4/ab_store_variants.txt)
  // A workgroup's envs: 64 (one per lane), or 32 when the batch would otherwise leave the chip half empty (P.epw; lanes 32 .. 63 idle:
  // twice the workgroups, and this kernel is bound by the latency of its per-wave chains long before it is bound by lanes)
  const int EPW = P.epw;
  const int env0 = wg * EPW, e = env0 + lane;
  const bool active = lane < EPW && e < P.N;
  const int nvalid = min(EPW, P.N - env0);
  const int W = P.W, H = P.H, CS = P.CS, GS = P.GS;
  const size_t N = (size_t)P.N;
  uint32_t* slut = (uint32_t*)smem;
  // obs7_quad / obs7_chunk address the table by absolute LDS offsets: it must sit at LDS address 0 (no static LDS in this kernel)
#ifndef MG_EMU
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem != 0u) __builtin_trap();
#endif
  // share (one-step launches, Env.step): there is nothing to split in time, so ONE wave runs the step up to the staged codes and ALL waves
  // of the workgroup share the output-space encode behind one barrier (a quarter of the 784 cell quads each); the stepping wave owns the state.
  // Which wave steps rotates with the workgroup index (P.share - 1 = the shift): the workgroups resident on one CU then step on different
  // SIMDs instead of all on the one that holds every workgroup's wave 0.
  const bool share = P.share != 0;
  const int sw = (share && P.share < 16) ? (int)(((uint32_t)wg >> (P.share - 1)) & (uint32_t)(NW - 1)) : 0;
  // split (fused launches with three or four waves, round 4): wave 0 runs the DYNAMICS of every step once and logs them, waves 1.. keep
  // their own grids current from the log and produce the observations, step j by encode wave j mod (NW - 1) -- see the loops below
  const bool split_mode = !ONE && !share && P.split_mode != 0;      // (FullyObs: only ever the staged-codes split below -- the host sets nothing else)
  // Which wave is the dynamics wave rotates with the workgroup index (P.split_mode - 1 = the shift): a workgroup's wave i lands on SIMD i,
  // so with wave 0 everywhere one SIMD of a CU would carry the dynamics waves of all its workgroups -- the longest instruction stream of
  // the four -- and pace the launch (measured: profiles/r4/split_rotation.txt)
  const int dw = split_mode ? (int)(((uint32_t)wg >> (P.split_mode - 1)) % (uint32_t)NW) : 0;
  const int ek = split_mode ? (wave - dw - 1 + NW) % NW : 0;        // encode wave index 0 .. NW - 2 (split mode)
  // DynamicObstacles, the sentence levels and FullyObs split differently (the staged split, see the loops below): ONE copy of the grids, the
  // dynamics wave's, which also stages every step's codes (FullyObs: a copy of its image-order stream)
  const bool dsplit = (gg_group(GG) == GG_DYNOBS || gg_group(GG) == GG_SENTENCE || FULL || STAGED) && split_mode;
  const int mycopy = (share || dsplit) ? 0 : wave;
  uint8_t* sgrid = smem + P.off_grid + mycopy * (EPW * GS);          // this wave's private copy of the workgroup's grids
  uint8_t* scodes = smem + P.off_T + (dsplit ? 0 : split_mode ? min(ek, NW - 2) : mycopy) * P.codes_stride;   // the wave's code stream (FULL: its image-order stream of the 64 grids)
  const int cells = P.cells, OBE = FULL ? cells * 3 : PARTIAL_OBS_BYTES;                           // observation bytes per env
  uint8_t* sshadow = smem + P.off_shadow;
  uint64_t* sspr = (uint64_t*)(smem + P.off_spr) + lane * 2;
  uint8_t* sact = smem + P.off_act;
  uint64_t* sinstr = (uint64_t*)(smem + P.off_instr);                 // GG_SENTENCE: the 64 envs' instruction records, ROLL_INSTR_STRIDE words apart
  const bool last_wave = share ? wave == sw : split_mode ? wave == dw : wave == NW - 1;    // the wave that owns the final state
  // (split[] is read with compile-time indices: a register-indexed read of a kernel argument is a load from the argument segment)
  const int sp_lo = wave == 0 ? P.split[0] : wave == 1 ? P.split[1] : wave == 2 ? P.split[2] : P.split[3];
  const int sp_hi = wave == 0 ? P.split[1] : wave == 1 ? P.split[2] : wave == 2 ? P.split[3] : P.split[4];
  const int j_begin = share ? 0 : sp_lo, j_end = share ? (wave == sw ? P.T : 0) : sp_hi;
  const bool reset_enabled = P.autoreset_next_step || P.phase == PHASE_OBSERVE;
  const bool goto_rule = (gg_group(GG) == GG_ROOMGRID && (MG_RULE(GG, P) == RULE_GOTO || MG_RULE(GG, P) == RULE_GOTOOBJ || MG_RULE(GG, P) == RULE_PUTNEAR)) ||
                         (gg_group(GG) == GG_ROOMS && (MG_RULE(GG, P) == RULE_GOTO_BIG || MG_RULE(GG, P) == RULE_PUTNEXT || MG_RULE(GG, P) == RULE_OPENDOOR));

  // ---- prologue: every load up front (see k_step: no global load may sit in the step loop), and every INDEPENDENT load issued before
  // the first one is waited for: a one-step launch (Env.step) is a chain of memory round trips and little else
  // (profiles/r3/unfused_anatomy.txt: 5.0 of 9.8 us were a launch without any step work; a kernel that only loads 4 MB into LDS takes 3.55)
  // (lane-level conditions are kept OUT of the loads -- clamped indices instead: a load under a divergent branch is waited for at the
  // branch's end, which made the prologue a chain of eight round trips)
  const int ec = min(e, P.N - 1);
  const uint64_t rec_ld = P.agent[ec];
  EnvRegs S;
  Agent& a = S.a;
  const uint64_t tg_ld = (goto_rule || gg_group(GG) == GG_DYNOBS) ? P.aux[ec] : 0ull;     // (GG_DYNOBS: the obstacle list)
  RNG rng;                                                                        // GG_DYNOBS: this env's stream
  if constexpr (gg_group(GG) == GG_DYNOBS) rng.load(P.rng, N, (size_t)ec);
  const uint32_t h_ld = P.head ? P.head[ec] : 0u;
  const uint32_t mask_ld = P.obs_mask ? (uint32_t)P.obs_mask[ec] : 1u;
  const bool stage_acts = P.phase == PHASE_STEP && P.act_src == ACT_SRC_BUFFER;
  const bool act_mine = tid < P.T * 64 && (tid & 63) < EPW && env0 + (tid & 63) < P.N;    // one-step launches: this lane's share of the caller's actions
  const uint32_t act_ld = stage_acts ? load_action(P, min(env0 + (tid & 63), P.N - 1), min(tid >> 6, P.T - 1)) : 0u;
  S.shadow_left = (uint32_t)P.use_shadow;
  const int cpe = CS >> 4, nchunks = nvalid * cpe;
  {
    // the 64 grids, 16 B per lane, coalesced.  Time split: each wave stages its own private copy (the redundant reads hit L2); share: ONE
    // copy, staged by all the threads of the workgroup.  Four loads in flight per lane, then the four LDS writes.
    const uint4* live = (const uint4*)(P.grid + (size_t)env0 * CS);
    const bool loads = share || split_mode || j_end > 0;             // wave-uniform
    const int l0 = (share || dsplit) ? tid : lane, lstride = (share || dsplit) ? nthreads : 64;
    auto stage4 = [&](int base) {
      uint4 gv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) gv[k] = live[min(base + l0 + k * lstride, nchunks - 1)];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = base + l0 + k * lstride;
        if (c < nchunks) {
          const uint32_t ce = ((uint32_t)c * P.cpe_magic) >> 20, part = (uint32_t)c - ce * (uint32_t)cpe;
          uint32_t* dst = (uint32_t*)(sgrid + ce * GS + part * 16);
          dst[0] = gv[k].x; dst[1] = gv[k].y; dst[2] = gv[k].z; dst[3] = gv[k].w;
        }
      }
    };
    // (the first group outside the loop: a loop header waits for every load in flight, the lane's agent record and ring position included)
    if (loads) {
      stage4(0);
      for (int base = 4 * lstride; base < nchunks; base += 4 * lstride) stage4(base);
    }
  }
  const uint64_t rec = active ? rec_ld : 0ull;
  S.targets = active ? tg_ld : 0ull;
  S.h = active ? h_ld : 0u;
  const uint32_t mask_byte = active ? mask_ld : 0u;
  const uint32_t act0 = act_ld;
  // shared, read-only after the barrier: the decode table, the shadow spares, the caller's actions
  for (int k = tid; k < 256; k += nthreads) slut[k] = cell_triple((uint32_t)k);
  if constexpr (gg_group(GG) == GG_DYNOBS) for (int k = tid; k < CS; k += nthreads) smem[P.off_tmpl + k] = (uint8_t)dynobs_template_cell(k, W, H, P.w_magic);
  if constexpr (gg_group(GG) == GG_SENTENCE && MG_INSTR_LDS) {
    // the workgroup's instruction records: consecutive in global memory (INSTR_WORDS u64 per env), 8 bytes per lane, coalesced
    const uint64_t* gi = P.instr + (size_t)env0 * INSTR_WORDS;
    for (int k = tid; k < nvalid * INSTR_WORDS; k += nthreads) {
      const uint32_t ce = ((uint32_t)k * 1639u) >> 16, w = (uint32_t)k - ce * (uint32_t)INSTR_WORDS;      // k / 40 for k < 64 * 40
      sinstr[ce * ROLL_INSTR_STRIDE + w] = gi[k];
    }
  }
  if (split_mode && tid < ROLL_LOG_SYNC_BYTES / 4) ((uint32_t*)(smem + P.off_log))[tid] = (tid >= 1 && tid < NW) ? 0u : (tid == 0 ? 0u : 0xFFFFFFFFu);   // [0] logged, [1 + k] consumed by encode wave k (absent waves: never behind)
  // the next use_shadow (1 or 2) spare episodes of every env: a batch may take up to cb >= 2 per env, so ring slots head and head + 1 are drawn
  // (an env's ring position is lane ce's S.h: every wave has loaded its own copy of the 64 heads)
  if constexpr (!ONE)
  for (int set = 0; set < P.use_shadow; set++) {
    if (wave == 0 && active) {
      const size_t se = (size_t)((S.h + (uint32_t)set) & P.ring_mask) * N + (size_t)e;
      uint64_t* sp = (uint64_t*)((uint8_t*)sspr + set * P.spr_stride);
      sp[0] = P.spare_agent[se];
      sp[1] = goto_rule ? P.spare_aux[se] : 0ull;
    }
    for (int c0 = 0; c0 < nchunks; c0 += nthreads) {                 // (uniform trip count: every lane takes part in the shuffle)
      const int c = c0 + tid;
      const bool in = c < nchunks;
      const uint32_t ce = in ? ((uint32_t)c * P.cpe_magic) >> 20 : 0u, part = (uint32_t)c - ce * (uint32_t)cpe;
      const uint32_t hce = (uint32_t)__shfl((int)S.h, (int)ce);
      if (in) {
        const uint32_t slot = P.head ? ((hce + (uint32_t)set) & P.ring_mask) : 0u;
        const uint4 s = ((const uint4*)(P.spare_grid + ((size_t)slot * N + (size_t)env0 + ce) * CS))[part];
        uint32_t* d2 = (uint32_t*)(sshadow + set * P.shadow_stride + ce * GS + part * 16);
        d2[0] = s.x; d2[1] = s.y; d2[2] = s.z; d2[3] = s.w;
      }
    }
  }
  if (stage_acts) {
    if (act_mine) sact[tid] = (uint8_t)act0;
    for (int k = tid + nthreads; k < P.T * 64; k += nthreads) {       // (fused launches with caller actions: the rest of the T x 64 block)
      const int j = k >> 6, l = k & 63;
      if (l < EPW && env0 + l < P.N) sact[k] = (uint8_t)load_action(P, env0 + l, j);
    }
  }
  // (Round 6 built what VERDICT r5 asked for here -- every wave of the workgroup drawing its share of the launch's T x 64 Philox actions into LDS up front,
  // the loop reading one byte, so that no Philox block sits on the dynamics wave -- and measured it: Empty-8x8 x 65 536 2.31 us per step against 2.15-2.22
  // with the draw inside the loop, DoorKey / GoToRedBall / LavaCrossing indifferent (profiles/r6/ab_action_staging_not_adopted.txt): the 2 KB of LDS and the
  // longer prologue cost more than ~25 VALU instructions per step on one wave.  Removed again.)
  if constexpr (FULL) {
    // the shadow spares' image stream (shared, built by wave 0 from the staged shadow grids)
    if (!ONE && P.use_shadow) {
      __syncthreads();
      for (int set = wave; set < P.use_shadow; set += NW)
        if (active) image_stream_build(sshadow + set * P.shadow_stride + lane * GS, smem + P.off_shadow_gt + set * P.codes_stride + lane * cells, W, H);
    }
  }
  __syncthreads();
  if constexpr (FULL) { if ((dsplit ? wave == dw : j_end > 0) && active) image_stream_build(sgrid + lane * GS, scodes + lane * cells, W, H); MG_LDS_SYNC(); }

  a = agent_unpack(rec);
  const uint32_t h_in = S.h;
  const bool maskok = mask_byte != 0u;
  uint8_t* mygrid = sgrid + lane * GS;
  S.cur = S.targets;
  if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTO && (a.flags & FLAG_TARGETS_STALE)) {
    const uint32_t desc = goto_desc(P, a.mission);
    S.cur = 0;
    for (int k = 0; k < P.cells; k++) S.cur |= (uint64_t)((uint32_t)mygrid[k] == desc) << k;
  }
  const uint32_t o_scal = (uint32_t)P.off_reward + (uint32_t)e * 16u;      // this env's mg_step_scalars inside a step record (16 bytes, ABI 3)
  S.rec_dirty = false; S.aux_dirty = false; S.wb_all = false; S.errbits = 0;
  uint32_t fin_total = 0, errs_mine = 0;
  uint32_t pw[4] = { 0, 0, 0, 0 };
  LaneCtx C;
  C.e = e; C.el = lane; C.sub = 0; C.active = active; C.lead = true; C.reset_enabled = reset_enabled; C.maskok = maskok; C.goto_rule = goto_rule;
  C.mygrid = mygrid; C.myshadow = sshadow + lane * GS; C.sspr = sspr;
  const bool see_through = P.see_through != 0 || MG_EXPBIT(P, 1);
  // GG_DYNOBS: the obstacle list (byte i = cell of obstacle i), what this launch did to it, episodes redrawn in the loop
  uint64_t obst = active ? tg_ld : 0ull;
  bool obst_dirty = false;
  uint32_t ngen = 0;
  // DynamicObstaclesEnv's draws for one step of the 64 envs (mg_dynobs.h): `regen` lanes redraw their env's episode in place (reset():
  // the constant grid copied from the template, one env at a time by the whole wave, then the agent and the obstacles placed by the lane),
  // `move` lanes move their obstacles (step(), before MiniGridEnv.step) -- all under one loop of placement tries
  auto dyn_draws = [&](bool regen, bool move, uint32_t flags_after_regen) __attribute__((always_inline)) {
    if constexpr (gg_group(GG) == GG_DYNOBS) {
      unsigned long long rm = __ballot(regen);
      if (rm) {
        const uint32_t* tm = (const uint32_t*)(smem + P.off_tmpl);
        while (rm) {
          const int b = __ffsll((long long)rm) - 1;
          rm &= rm - 1ull;
          uint32_t* gb = (uint32_t*)(sgrid + b * GS);
          for (int k = lane; k < (CS >> 2); k += 64) gb[k] = tm[k];
        }
        MG_WAVE_ORDER();                                 // (DS operations of a wave execute in order: the lanes below read what the wave wrote)
      }
      if (move) {
        // `not_clear` (dynamicobstacles.py:142-144): what is in front of the agent BEFORE the obstacles move
        const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
        const uint32_t F = ((unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H) ? (uint32_t)mygrid[fy * W + fx] : (uint32_t)CELL_WALL_GREY;
        const bool not_clear = F != CELL_EMPTY && cell_type(F) != T_GOAL;
        a.flags = (a.flags & ~FLAG_NOT_CLEAR) | (not_clear ? FLAG_NOT_CLEAR : 0u);
      }
      if constexpr (RNG::kEpisodic) if (regen) rng.begin_episode();
      uint32_t ax = a.x, ay = a.y, adir = a.dir;
      bool failed = false, changed = false;
      if (!MG_EXPBIT(P, 512))            // (attribution builds: the step without its placement loop)
        dynobs_place(rng, mygrid, W, H, P.w_magic, P.dyn_n, regen, move, P.dyn_sx, P.dyn_sy, P.dyn_sdir, ax, ay, adir, obst, failed, changed);
      if (regen) {
        a.x = ax; a.y = ay; a.dir = adir; a.carry = 0; a.step = 0; a.mission = 0; a.flags = flags_after_regen;
        S.rec_dirty = true; S.wb_all = true; obst_dirty = true;
        if (failed) S.errbits |= ERR_GENERATOR;                        // the reference's reset() raises RecursionError
        if (last_wave) ngen++;
      }
      if (changed) { S.wb_all = true; obst_dirty = true; }
    }
  };
  MG_SPIN_DECL;
  // -DMG_SPIN_BOUND=<polls> builds (profiles/r6_protocol_bound.sh; never the product library): every inter-wave spin below gives up after that many
  // polls, raises error word ERR_WORD_SPIN (the next mg_sync fails) and carries on with whatever it finds -- the launch then ENDS (every loop is
  // bounded by T), so a regression of the LDS protocol is a failing test instead of a hung lease (VERDICT r5 "next" #7).  The product build keeps
  // the unbounded form: a counter in the hottest loop of the dynamics wave for a failure that needs corrupted LDS.
#if defined(MG_SPIN_BOUND)
#define MG_SPIN_POLL(n) if (++(n) > (uint32_t)(MG_SPIN_BOUND)) { if (lane == 0) P.err[ERR_WORD_SPIN] = 1u; break; }
#else
#define MG_SPIN_POLL(n) do { } while (0)
#endif
  uint32_t spin_polls = 0u; (void)spin_polls;
  constexpr bool nt = NT;                                            // this launch's observation stores are nontemporal (see store12; the host picks the instantiation)

  // COOPERATIVE SPARE FETCH (round 6; the levels that stage no shadow spares: the big grids and the sentence levels).  An env that resets takes its next
  // episode from the ring in HBM; env_transition's take_spare does that per lane -- CS / 16 dependent 16-byte loads in ONE lane (40 for MultiRoom's 25 x 25)
  // under a divergent branch the other 63 lanes wait at: with 64 envs per wave and a reset in 0.1-0.8 % of the env-steps that is every second to tenth step
  // of the dynamics wave.  Here the WAVE fetches the grid -- lane c takes 16-byte piece c: one load instruction, one round trip -- and, fused launches, one
  // step AHEAD: RESET_PENDING is raised by the step that ends the episode and honoured by the next one, so the load is issued at the end of step j and
  // its result is written into the LDS grid at the start of step j + 1, behind step j's observation (up to two envs per step; a third falls back to the
  // per-lane copy).  The first step of a launch fetches and commits in place.
  constexpr bool COOP = STAGED || gg_group(GG) == GG_SENTENCE || ONE;
  const bool coop = COOP && gg_group(GG) != GG_DYNOBS && P.use_shadow == 0 && !P.static_gen && P.head != nullptr && cpe > 8 && cpe <= 64 && !MG_EXPBIT(P, 4096);
  uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = make_uint4(0, 0, 0, 0);
  unsigned long long pf_mask = 0ull;
  uint64_t pf_agent = 0ull, pf_aux = 0ull;                 // every pending lane's own spare record (a lane-level load, issued with the grids')
  bool pf_rec = false;
  auto coop_fetch = [&]() __attribute__((always_inline)) {
    if constexpr (COOP) if (coop) {
      const bool pend = active && (a.flags & FLAG_RESET_PENDING) && reset_enabled && maskok && S.shadow_left == 0u && !MG_EXPBIT(P, 64);
      unsigned long long m = __ballot(pend);
      pf_mask = 0ull;
      pf_rec = pend;
      if (m) {
        const uint32_t slot = S.h & P.ring_mask;
        const int cl = min(lane, cpe - 1);
        const int b0 = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const size_t se0 = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)slot, b0) * N + (size_t)(env0 + b0);
        // (no lane condition around the load -- a load under a divergent branch is waited for at the branch's end --: the lanes with nothing pending read the first pending env's record)
        const size_t se_mine = pend ? (size_t)slot * N + (size_t)e : se0;
        pf_agent = P.spare_agent[se_mine];
        pf_aux = goto_rule ? P.spare_aux[se_mine] : 0ull;
        pf0 = ((const uint4*)(P.spare_grid + se0 * CS))[cl];
        pf_mask = 1ull << b0;
        if (m) {
          const int b1 = __ffsll((long long)m) - 1;
          const size_t se1 = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)slot, b1) * N + (size_t)(env0 + b1);
          pf1 = ((const uint4*)(P.spare_grid + se1 * CS))[cl];
          pf_mask |= 1ull << b1;
        }
      }
    }
  };
  auto coop_commit = [&]() __attribute__((always_inline)) {
    if constexpr (COOP) {
      C.spare_in_lds = false;
      C.spare_rec_pf = coop && pf_rec; C.pf_agent = pf_agent; C.pf_aux = pf_aux;
      pf_rec = false;
      if (coop && pf_mask) {
        unsigned long long m = pf_mask;
        const int b0 = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        if (lane < cpe) { uint32_t* d = (uint32_t*)(sgrid + b0 * GS + lane * 16); d[0] = pf0.x; d[1] = pf0.y; d[2] = pf0.z; d[3] = pf0.w; }
        if (m) {
          const int b1 = __ffsll((long long)m) - 1;
          if (lane < cpe) { uint32_t* d = (uint32_t*)(sgrid + b1 * GS + lane * 16); d[0] = pf1.x; d[1] = pf1.y; d[2] = pf1.z; d[3] = pf1.w; }
        }
        C.spare_in_lds = ((pf_mask >> lane) & 1ull) != 0ull;
        pf_mask = 0ull;
        MG_WAVE_ORDER();                                   // (DS operations of a wave execute in order: the lanes below read what the wave wrote)
      }
    }
  };

  // ---- the pieces of a step ----
  struct StepOut { uint32_t act_in, term, trunc; double reward; uint64_t sent0, sent1; bool show_taken; };
  // action + MiniGridEnv.step / reset on this wave's copy of the grids (+ the sentence levels' verifier)
  auto dynamics = [&](int j, StepOut& o) {
    MG_MARK("action");
    uint32_t act = A_DONE;
    if (P.phase == PHASE_STEP) {
      if (P.act_src == ACT_SRC_PHILOX) {
        const uint32_t t = P.t0 + (uint32_t)j;
        if (j == 0 || (t & 3u) == 0u) philox_action_block(P, e, t >> 2, pw);
        const uint32_t w = (t & 3u) == 0u ? pw[0] : (t & 3u) == 1u ? pw[1] : (t & 3u) == 2u ? pw[2] : pw[3];
        act = (uint32_t)(((uint64_t)w * 7u) >> 32);
      } else act = sact[j * 64 + lane];
    }
    o.act_in = act;
    if constexpr (gg_group(GG) == GG_LIGHT) if (MG_RULE(GG, P) == RULE_MEMORY && act == A_PICKUP) act = A_TOGGLE;    // MemoryEnv.step (memory.py:151-153)
    if constexpr (gg_group(GG) == GG_NONE) if (MG_RULE(GG, P) == RULE_DYNOBS && act >= 3u) act = A_LEFT;             // "Invalid action" (dynamicobstacles.py:137-139)
    if constexpr (gg_group(GG) == GG_DYNOBS) if (act >= 3u) act = A_LEFT;
    o.reward = 0.0; o.term = 0; o.trunc = 0; o.sent0 = 0; o.sent1 = 0;
    S.errbits = 0;
    if constexpr (gg_group(GG) == GG_DYNOBS) {
      if (P.phase == PHASE_STEP) {
        // NEXT_STEP autoreset: the env whose episode the previous step ended is redrawn now and comes out FRESH (this step only observes it,
        // like an env the host's live refill redrew before the launch); everyone else's obstacles move
        const bool pend = (a.flags & FLAG_RESET_PENDING) != 0u;
        dyn_draws(active && pend && reset_enabled && maskok, active && !(a.flags & (FLAG_RESET_PENDING | FLAG_FRESH)), FLAG_FRESH);
      }
    }
    // the sentence levels: the hot words of the env's instruction record, requested BEFORE the step's own work so that the verifier below finds
    // them arrived (mg_verify.h InstrWords; an env that takes a new episode in this step does not verify, and its record is replaced below)
    InstrWords IWd;
    if constexpr (gg_group(GG) == GG_SENTENCE) if (active && P.phase == PHASE_STEP && !MG_EXPBIT(P, 2048))
      IWd.load(MG_INSTR_LDS ? sinstr + lane * ROLL_INSTR_STRIDE : P.instr + (size_t)e * INSTR_WORDS);
    MG_MARK("transition");
    if (j == 0) coop_fetch();                                // (later steps: issued at the end of the step before)
    coop_commit();
    if (!MG_EXPBIT(P, 16)) env_transition<GG, 1>(P, C, S, act, o.reward, o.term, o.trunc);
    if constexpr (COOP) { C.spare_in_lds = false; C.spare_rec_pf = false; }
    MG_MARK("after_transition");
    if constexpr (gg_group(GG) == GG_DYNOBS) if (P.autoreset_same_step && P.phase == PHASE_STEP) {
      // Gymnasium's SAME_STEP autoreset: the step that ended the episode also redraws the env; the observation below is the new episode's
      // first, reward / terminated / truncated stay the ended one's
      dyn_draws(active && (o.term | o.trunc) != 0u, false, 0u);
    }
    if constexpr (gg_group(GG) == GG_SENTENCE) if (active) {
      // The sentence levels' verifier inside the step loop (round 2 ran it as a second kernel after every one-step launch): one lane per env on
      // the env's instruction record, which no other wave touches (one wave per workgroup: the time split would replay it); the grid it looks
      // at is the LDS copy.  (The record can be staged in LDS for the launch as well -- MG_INSTR_LDS above; what made the verifier 66 of the 90 us of
      // a BossLevel step at 131 072 envs, profiles/r4/bosslevel_attr.txt, was its instruction count under divergence, not its loads: mg_verify.h.)
      uint64_t* I = MG_INSTR_LDS ? sinstr + lane * ROLL_INSTR_STRIDE : P.instr + (size_t)e * INSTR_WORDS;
      if (a.flags & FLAG_NEW_EPISODE) {
        // the spare taken in this step brings its instruction record (its ring slot = the head before the take); head itself is
        // published at launch end, so no refill can have touched the slot
        const uint64_t* src = P.spare_instr + ((size_t)((S.h - 1u) & P.ring_mask) * N + (size_t)e) * INSTR_WORDS;
        for (int k = 0; k < INSTR_WORDS; k++) I[k] = src[k];
        a.flags &= ~FLAG_NEW_EPISODE;
      } else if (P.phase == PHASE_STEP) {
        uint32_t max_steps = 0, verr = 0;
        // (attribution builds, MG_EXP bit 2048: the step without its verifier -- nothing ever succeeds, the episode limit still comes from the record)
        uint32_t status = R_CONTINUE;
        if (MG_EXPBIT(P, 2048)) max_steps = (uint32_t)(I[0] >> 39) & 0xFFFFu;
        else status = verify_action(I, IWd, mygrid, W, H, a, o.act_in, max_steps, verr, P.done_actions);
        S.errbits |= verr;
        o.term = status != R_CONTINUE; o.trunc = a.step >= max_steps;
        o.reward = status == R_SUCCESS ? reward_exact(a.step, (int)max_steps) : 0.0;
        if ((o.term | o.trunc) && (P.autoreset_next_step || P.autoreset_same_step)) { a.flags |= FLAG_RESET_PENDING; S.rec_dirty = true; }
      }
      o.sent0 = I[IW_MISSION]; o.sent1 = I[IW_MISSION + 1];
    }
    if constexpr (gg_group(GG) == GG_SENTENCE) if (P.autoreset_same_step && P.phase == PHASE_STEP) {
      // Gymnasium's SAME_STEP autoreset for the sentence levels (round 4): their episodes end in the verifier, i.e. after env_transition's own
      // SAME_STEP branch; the envs the verifier just ended take their next episode now (env_transition in reset-only mode), their instruction
      // record comes with it, and the observation below is the new episode's first -- reward / terminated / truncated stay the ended one's
      LaneCtx C2 = C;
      C2.reset_enabled = true; C2.reset_only = true;
      double r2 = 0.0; uint32_t t2 = 0, u2 = 0;
      env_transition<GG, 1>(P, C2, S, A_DONE, r2, t2, u2);
      if (active && (a.flags & FLAG_NEW_EPISODE)) {
        uint64_t* I = MG_INSTR_LDS ? sinstr + lane * ROLL_INSTR_STRIDE : P.instr + (size_t)e * INSTR_WORDS;
        const uint64_t* src = P.spare_instr + ((size_t)((S.h - 1u) & P.ring_mask) * N + (size_t)e) * INSTR_WORDS;
        for (int k = 0; k < INSTR_WORDS; k++) I[k] = src[k];
        a.flags &= ~FLAG_NEW_EPISODE;
        o.sent0 = I[IW_MISSION]; o.sent1 = I[IW_MISSION + 1];
      }
    }
    if constexpr (!ONE) if (P.phase == PHASE_STEP && j + 1 < P.T) coop_fetch();
    o.show_taken = false;
    if constexpr (gg_group(GG) == GG_ROOMS) if (MG_RULE(GG, P) == RULE_PUTNEXT && active && (a.flags & FLAG_SHOW_TAKEN)) {
      // PutNext(start_carrying): the episode's first core observation shows the object where it was and empty hands (see k_step)
      o.show_taken = true;
      a.flags &= ~FLAG_SHOW_TAKEN; S.rec_dirty = true;
    }
  };
  auto slot_of = [&](int j) { int s_ = P.slot0 - j; s_ += s_ < 0 ? P.S : 0; return s_; };   // (T <= S and slot0 < S: at most one wrap; a loop here compiled to a scalar division)
  // reward / terminated / truncated / direction / mission id / action of step j -> its trajectory slot
  auto store_scalars = [&](int slot_out, const StepOut& o, bool mine = true) {
    MG_MARK("scalars");
    errs_mine |= S.errbits;
    if (P.phase == PHASE_STEP) fin_total += (uint32_t)__popcll(__ballot(active && (o.term | o.trunc)));
    uint8_t* ob = P.out + (size_t)slot_out * P.slot_bytes;
    if (active && mine && !MG_EXPBIT(P, 8)) {
      // {reward f64 | terminated, truncated, direction, action u8 | mission id u16 | 0}: ONE 16-byte store per env (six partial-line
      // stores cost the dynamics wave 0.3 of the 2.4 us of a 65 536-env step: profiles/r4/attribution_split.txt)
      uint4 v;
      v.x = (uint32_t)__double2loint(o.reward); v.y = (uint32_t)__double2hiint(o.reward);
      v.z = o.term | (o.trunc << 8) | (a.dir << 16) | (o.act_in << 24);
      v.w = a.mission & 0xFFFFu;
      *(uint4*)(ob + o_scal) = v;
      if constexpr (gg_group(GG) == GG_SENTENCE) { uint64_t* sp = (uint64_t*)(ob + P.off_sentence) + (size_t)e * 2; sp[0] = o.sent0; sp[1] = o.sent1; }
    }
  };
  // gen_obs of the 64 envs as they stand in this wave's grids -> the observation of trajectory slot slot_out:
  // 49 codes per env (lane = env), then the encode in output space (lane = four cells = 12 bytes)
  // (codes / parts: the code staging to use and which half to run -- 1 = stage the codes, 2 = encode them; the staged split runs the
  // halves in different waves, everything else passes its own staging and 3)
  auto observe = [&](int slot_out, const Agent& av, bool show_taken, uint32_t taken_idx, uint32_t taken_code, uint8_t* codes_arg, int parts) {
    uint8_t* const codes = (FULL && parts == 3) ? scodes : codes_arg;   // (FullyObs encodes its own image-order stream, or -- parts 2 -- a staged copy of one)
    if constexpr (gg_group(GG) == GG_ROOMS) if (show_taken) mygrid[taken_idx] = (uint8_t)taken_code;
    MG_MARK("codes");
    uint32_t gt_pos = 0, gt_old = 0;
    if constexpr (FULL) {
      // the agent's own cell reads (10, 0, dir) in the observation: patched into the stream for the encode, restored after it
      gt_pos = (uint32_t)(lane * cells) + av.x * (uint32_t)H + av.y;
      if (active && parts == 3) { gt_old = codes[gt_pos]; codes[gt_pos] = (uint8_t)(T_AGENT_MARK | (av.dir << 4)); }
    } else if ((parts & 1) && !MG_EXPBIT(P, 4)) {
      View7 O;
      obs7_view(av, mygrid, W, H, see_through, O);
      uint32_t D[13];
      view7_pack(O, D);
      // (every lane takes part in the shuffle: a lane that is masked off reads back as 0 -- lane 62 would lose env 63's first bytes)
      const uint32_t next0 = (uint32_t)__shfl_down((int)D[0], 1);
      obs7_stage(D, next0, lane, (uint32_t*)codes);
    }
    MG_MARK("codes_end");
    if constexpr (gg_group(GG) == GG_ROOMS) if (show_taken) mygrid[taken_idx] = (uint8_t)CELL_EMPTY;
    MG_LDS_SYNC();
    MG_MARK("chunks");
    if ((parts & 2) && !MG_EXPBIT(P, 2) && !share) {
      uint8_t* obase = P.obs + (size_t)slot_out * P.obs_stride + (size_t)wg * P.obs_wg_stride;   // 64 * OBE is a multiple of 16
      // (attribution builds, MG_EXP bit 256: every workgroup's observations go to a 1.2 MB window that stays in L2 -- the same store
      // instructions without the HBM write stream: is the observation stream's cost its issue or its bandwidth?)
      if (MG_EXPBIT(P, 256)) obase = P.obs + (size_t)(wg & 127) * P.obs_wg_stride;
      // (attribution builds, MG_EXP bit 16384: every workgroup's block moved down to the 128-byte line it starts in -- an odd workgroup's block starts 64 bytes into a
      // line, so each of its 768-byte rounds has a half line at both ends; the moved block overlaps its neighbour's tail: timing only)
      if (MG_EXPBIT(P, 16384)) obase -= (size_t)(obase - P.obs) & 127u;
      const int nbytes = nvalid * OBE;
      const int nvec = nbytes >> 4;
#if MG_ENCODE_QUADS
      if (!FULL && nvalid == 64) {
        // 784 cell quads: thirteen rounds, the last one 16 lanes wide
        // (a block starts mid-line when wg * 64 * 147 = 64 mod 128, i.e. for the odd workgroups of a 64-env launch shape; the tensor itself starts on a 256-byte boundary)
        encode_quads<64, 64 * VIEW_CELLS / 4, NT>(lane, codes, slut, obase, !MG_EXPBIT(P, 32), (MG_ALIGN_ROUNDS && ((uintptr_t)obase & 127u) == 64u) ? 16 : 0);
      } else if (!FULL && nvalid == 32) {
        encode_quads<64, 32 * VIEW_CELLS / 4, NT>(lane, codes, slut, obase, !MG_EXPBIT(P, 32));      // 32-env workgroups: 392 quads, seven rounds
      } else if (FULL && nvalid == 64) {
        // 16 * cells quads (64 * cells / 4), rounds of 64 in BLOCKS of four: the four code dwords first, then the sixteen lookups, then four packs and stores --
        // a round by itself is two dependent LDS latencies (code, then table) in front of every store, and FullyObs runs two waves per SIMD: nothing hides them
        // (round 6; LavaCrossing 9 x 9: 20.25 rounds per step in ONE wave).  Rounds past the end re-read the last quad and store nothing.
        const int nq = 16 * cells;
        // (line-aligned rounds as in encode_quads: a block that starts mid-line -- odd workgroups of an odd cell count -- writes its first 16 quads by themselves)
        const int qs = (MG_ALIGN_ROUNDS && ((uintptr_t)obase & 127u) == 64u) ? 16 : 0;
        if (qs && lane < qs) {
          uint32_t o3[3];
          obs7_quad((uint32_t)lane, codes, slut, o3);
          Out12 v; v.x = o3[0]; v.y = o3[1]; v.z = o3[2];
          if (!MG_EXPBIT(P, 32)) store12(obase, (uint32_t)lane * 12u, v, nt);
        }
        for (int u0 = qs + lane; u0 < nq; u0 += 256) {
          uint32_t cw4[4], t4[4][4];
#pragma unroll
          for (int r = 0; r < 4; r++) cw4[r] = ((const uint32_t*)codes)[min(u0 + 64 * r, nq - 1)];
#pragma unroll
          for (int r = 0; r < 4; r++) obs7_quad_lookup(cw4[r], slut, t4[r]);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int u = u0 + 64 * r;
            uint32_t o3[3];
            obs7_quad_pack(t4[r], o3);
            Out12 v; v.x = o3[0]; v.y = o3[1]; v.z = o3[2];
            if (u < nq && !MG_EXPBIT(P, 32)) store12(obase, (uint32_t)u * 12u, v, nt);
          }
        }
      } else {
#else
      if (!FULL && nvalid == 64) {
        constexpr int NCH = 64 * PARTIAL_OBS_BYTES / 16, NIT = (NCH + 63) / 64;   // 588 chunks: ten rounds, the last one 12 lanes wide
#pragma unroll 2
        for (int it = 0; it < NIT; it++) {
          const int c = lane + 64 * it;
          uint32_t o4[4];
          obs7_chunk((uint32_t)(it == NIT - 1 ? min(c, NCH - 1) : c), codes, slut, o4);
          uint4 v; v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3];
          if (it < NIT - 1 || c < NCH) store16(obase, (uint32_t)c * 16u, v, nt);
        }
      } else if (FULL && nvalid == 64) {
        const int nch = 12 * cells;                                               // 64 * 3 * cells / 16 chunks
        for (int c = lane; c < nch; c += 64) {
          uint32_t o4[4];
          obs7_chunk((uint32_t)c, codes, slut, o4);
          uint4 v; v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3];
          store16(obase, (uint32_t)c * 16u, v, nt);
        }
      } else {
#endif
        // the ragged last workgroup of a batch: whole chunks, then the stream's last bytes one by one
#pragma unroll 1
        for (int c = lane; c <= nvec; c += 64) {
          uint32_t o4[4];
          obs7_chunk((uint32_t)c, codes, slut, o4);
          if (c < nvec) { uint4 v; v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3]; store16(obase, (uint32_t)c * 16u, v, nt); }
          else for (int b = 0; b < (nbytes & 15); b++) obase[(nvec << 4) + b] = (uint8_t)(o4[b >> 2] >> (8 * (b & 3)));
        }
      }
    }
    if constexpr (FULL) if (parts == 3) MG_LOCKSTEP();
    if constexpr (FULL) if (active && !share && parts == 3) codes[gt_pos] = (uint8_t)gt_old;     // (in order behind the chunk reads)
    MG_MARK("step_end");
    // (no wait here: the LDS pipe is in order, so the next step's staging writes cannot pass this step's chunk reads)
  };

  // FullyObs: the image-order stream follows the grids.  A reset replaces a whole grid: the WAVE re-images the envs that took a spare, one
  // env at a time, lane k doing cell k (a lane re-imaging its own env cell by cell would make the whole wave walk W*H cells in
  // every step in which any env resets -- under a random policy on a lava level that is nearly every step).
  auto full_follow = [&]() __attribute__((always_inline)) {
    if constexpr (FULL) {
      unsigned long long rm = __ballot(active && S.ev_reset != 0u);
      if (rm) {
        const unsigned long long from_shadow = __ballot(active && S.ev_reset == 1u), from_set1 = __ballot(active && S.ev_reset == 1u && S.ev_shadow == 1u);
        MG_LDS_SYNC();                                             // the lanes' grid writes of this step are done
        while (rm) {
          const int b = __ffsll((long long)rm) - 1;
          rm &= rm - 1ull;
          const uint8_t* sgt = smem + P.off_shadow_gt + (int)((from_set1 >> b) & 1ull) * P.codes_stride + b * cells;
          const uint8_t* gb = sgrid + b * GS;
          uint8_t* gt = scodes + b * cells;
          if ((from_shadow >> b) & 1ull) { for (int k = lane; k < cells; k += 64) gt[k] = sgt[k]; }
          else for (int k = lane; k < cells; k += 64) {
            const uint32_t x = ((uint32_t)k * P.h_magic) >> 16, y = (uint32_t)k - x * (uint32_t)H;      // k = x * H + y
            gt[k] = gb[y * (uint32_t)W + x];
          }
        }
        MG_LDS_SYNC();
      }
      if (active && S.ev_dirty_idx >= 0) {
        const uint32_t y = ((uint32_t)S.ev_dirty_idx * P.w_magic) >> 16, x = (uint32_t)S.ev_dirty_idx - y * (uint32_t)W;
        scodes[lane * cells + x * H + y] = (uint8_t)S.ev_dirty_code;
      }
    }
  };

  if (ONE || !split_mode) {
    // ---- every wave steps for itself: one wave (NW = 1, or `share`), or the TIME SPLIT (wave w replays steps 0 .. split[w]-1 silently) ----
    if constexpr (ONE) {
      // (host invariant, launch_roll_*: a one-step launch has ONE stepping wave -- nw == 1 or `share`; a time split with j_begin = j_end = 1 in the
      // other waves must not store the step again: ADVICE r5)
      if (j_begin == 0 && j_end > 0) {
        StepOut o;
        dynamics(0, o);
        full_follow();
        const int slot_out = slot_of(0);
        store_scalars(slot_out, o);
        Agent av = a;
        if (o.show_taken) av.carry = 0;
        observe(slot_out, av, o.show_taken, (uint32_t)(S.targets & 0xFFFFull), a.carry, scodes, 3);
      }
    } else
    for (int j = 0; j < j_end; j++) {
      const bool emit = j >= j_begin;                                  // wave-uniform: silent replay before the wave's own steps
      StepOut o;
      dynamics(j, o);
      full_follow();
      if (!emit) continue;
      const int slot_out = slot_of(j);
      store_scalars(slot_out, o);
      Agent av = a;
      if (o.show_taken) av.carry = 0;
      observe(slot_out, av, o.show_taken, (uint32_t)(S.targets & 0xFFFFull), a.carry, scodes, 3);
    }
  } else if constexpr (ONE) {
    // (never: the one-step kernel has no split)
  } else if constexpr (gg_group(GG) == GG_DYNOBS || gg_group(GG) == GG_SENTENCE || FULL || STAGED) {
    // ---- DynamicObstacles and the sentence levels (and, STAGED, the big grids of the other levels), split: the dynamics wave also STAGES every step's codes (gather, orientation, visibility -- the part of gen_obs
    // that needs the grid), into a ring of ROLL_DSPLIT_RING code stagings; the other waves only run the output-space encode and the stores, step
    // j by encode wave j mod (NW - 1).  The level's step is its placement loop (a 128-bit multiply per try, ~16 tries deep for the unluckiest
    // of 64 lanes; profiles/r4/dynobs_attr_first.txt: 26 of 30 us), which the ~150 instructions of the staging do not lengthen noticeably --
    // and ONE copy of the grids per workgroup instead of one per wave lets four workgroups share a CU at 16 x 16 instead of two, i.e. every
    // workgroup of a 65 536-env batch is resident at once.
    // The sentence levels (round 4, one encode wave): their step is ~5 000 dependent instructions in ONE wave per workgroup (the verifier's record
    // is that wave's), four workgroups per CU -- one wave per SIMD at ~10 cycles per instruction.  Handing the output-space encode (a quarter of
    // the instructions) to a second wave shortens the chain and puts a second wave on every SIMD without a second copy of the 22 x 22 grids.
    // FullyObs (round 4): what is staged is a COPY of the dynamics wave's image-order stream with the agents' cells patched in (64 x W*H bytes,
    // a handful of 16-byte LDS moves per lane) -- the time split's second wave replayed every step's dynamics, resets and re-imaging included.
    typedef MG_LDS_VU32 lds_vu32;
    lds_vu32* sync = MG_LDS_AT(P.off_log);                 // [0] = steps staged, [1 + k] = steps encode wave k has written out
    uint8_t* ring = smem + P.off_T + (FULL ? P.codes_stride : 0);              // (FullyObs: behind the dynamics wave's own stream)
    const int NE = NW - 1;
    if (wave == dw) {
      __builtin_amdgcn_s_setprio(MG_DPRIO);
      int kq = 0; uint32_t mq = 0;                                              // (j - RING) mod NE and div NE: who consumed the staging about to be reused
      for (int j = 0; j < P.T; j++) {
        StepOut o;
        dynamics(j, o);
        store_scalars(slot_of(j), o);
        if (j >= P.dring) {
          spin_polls = 0u;
          while ((uint32_t)__builtin_amdgcn_readfirstlane((int)sync[1 + kq]) < mq + 1u) { __builtin_amdgcn_s_sleep(1); MG_SPIN_POLL(spin_polls); }
          if (++kq == NE) { kq = 0; mq++; }
        }
        MG_WAVE_ORDER();
        if constexpr (FULL) {
          full_follow();
          // the stream as this step's observation shows it: the agent's own cell reads (10, 0, dir) (wrappers.py:422-424)
          const uint32_t gt_pos = (uint32_t)(lane * cells) + a.x * (uint32_t)H + a.y;
          uint32_t gt_old = 0;
          if (!MG_EXPBIT(P, 8192)) {           // (attribution builds: the step without its staging copy -- what does the copy cost the dynamics wave?)
          if (active) { gt_old = scodes[gt_pos]; scodes[gt_pos] = (uint8_t)(T_AGENT_MARK | (a.dir << 4)); }
          MG_LDS_SYNC();
          const uint4* src = (const uint4*)scodes;
          uint4* dst = (uint4*)(ring + (j & (P.dring - 1)) * P.codes_stride);
          for (int c = lane; c < 4 * cells; c += 64) dst[c] = src[c];            // 64 * cells bytes = 4 * cells 16-byte pieces
          MG_LDS_SYNC();
          if (active) scodes[gt_pos] = (uint8_t)gt_old;
          }
        } else {
          // (PutNext(start_carrying): the episode's first observation shows the object where it was and empty hands -- the codes are staged that way)
          Agent av = a;
          if (o.show_taken) av.carry = 0;
          observe(0, av, o.show_taken, (uint32_t)(S.targets & 0xFFFFull), a.carry, ring + (j & (P.dring - 1)) * P.codes_stride, 1);
        }
        // (DS operations of one wave execute in order: the counter cannot become visible before the codes)
        MG_WAVE_ORDER();
        sync[0] = (uint32_t)(j + 1);
      }
    } else {
      const int k = ek;
      uint32_t done = 0;
      Agent av = agent_unpack(0ull);
      for (int j = k; j < P.T; j += NE) {
        spin_polls = 0u;
        while ((uint32_t)__builtin_amdgcn_readfirstlane((int)sync[0]) <= (uint32_t)j) { __builtin_amdgcn_s_sleep(1); MG_SPIN_POLL(spin_polls); }
        MG_WAVE_ORDER();
        observe(slot_of(j), av, false, 0u, 0u, ring + (j & (P.dring - 1)) * P.codes_stride, 2);
        MG_LDS_SYNC();                                                          // the staging's last read has returned
        MG_WAVE_ORDER();
        sync[1 + k] = ++done;
      }
      return;        // (nothing to report, no state to write back: the dynamics wave owns both)
    }
  } else if (wave == dw) {
    // ---- DYNAMICS wave (round 4): every step's action, transition and scalar outputs, and ONE record per env and step for the encode waves --
    // the pose the view needs and what the step did to the grid -- in a ring of ROLL_LOG_STEPS entries in LDS.  The time split made every wave
    // replay the dynamics of the steps before its own (1.3 silent steps per produced step: a third of all instructions of a launch,
    // profiles/r4/attribution.txt); here the dynamics run once.
    // (an LDS-typed pointer: through a generic `volatile uint32_t*` the compiler emits FLAT loads / stores with system scope, which also
    // count on vmcnt; the dynamic LDS starts at LDS address 0, checked above)
    typedef MG_LDS_VU32 lds_vu32;
    lds_vu32* sync = MG_LDS_AT(P.off_log);                 // [0] = steps logged, [1 + k] = entries encode wave k has consumed
    uint2* logbuf = (uint2*)(smem + P.off_log + ROLL_LOG_SYNC_BYTES);
    // the dynamics wave is the one serial chain everything else waits for: it wins issue arbitration against the encode waves on its SIMD
    // (profiles/r4/ab_prio.txt: 2.42 -> 2.28 us per 65 536-env step)
    __builtin_amdgcn_s_setprio(MG_DPRIO);
    uint16_t* log3 = (uint16_t*)(smem + P.off_log + ROLL_LOG_SYNC_BYTES + ROLL_LOG_STEPS * 64 * 8);
    for (int j = 0; j < P.T; j++) {
      StepOut o;
      dynamics(j, o);
      // SCALARS BY THE STEP'S ENCODE WAVE (round 6).  The 16-byte scalar record of a step was stored here, by the dynamics wave -- up to ROLL_LOG_STEPS
      // steps ahead of the observations, i.e. into trajectory slots the observation stream reaches microseconds later: two write streams 10 MB apart.
      // profiles/microbench/rollstore2.hip (this launch shape, stores only): 2.27 us per step that way, 2.05 with the step's scalars stored by the wave
      // that stores its observations (= the observation stream alone).  The log entry carries what the record needs beyond the pose: terminated,
      // truncated, the action, and the reward as a CODE -- 0 = 0.0, 1 = reward_exact(step count, max_steps) (the step count rides in a third,
      // 16-bit log word), 2 = -1.0, 3 = anything else: that lane's record is stored here, as before, and the encode wave leaves it alone.
      // (the mission id is followed by the encode waves themselves: it changes with the episode only)
      uint32_t rcode = 3u;
      if constexpr (MG_SCAL_BY_ENCODE) { rcode = 0u;
      if (__builtin_bit_cast(uint64_t, o.reward) != 0ull)
        rcode = __builtin_bit_cast(uint64_t, o.reward) == __builtin_bit_cast(uint64_t, reward_exact(a.step, P.max_steps)) ? 1u : o.reward == -1.0 ? 2u : 3u;
      if (o.act_in > 7u) rcode = 3u;                                   // (an unknown action from the caller: the record as it always was; the call raises anyway)
      }
      store_scalars(slot_of(j), o, rcode == 3u);
      // pose: x | y << 8 | dir << 16 | terminated << 18 | truncated << 19 | action << 20 | reward code bit 0 << 23 | what the agent's cell shows (the carried object; nothing under show_taken) << 24
      const uint32_t pose = a.x | (a.y << 8) | (a.dir << 16) | (o.term << 18) | (o.trunc << 19) | ((o.act_in & 7u) << 20) | ((rcode & 1u) << 23) | ((o.show_taken ? 0u : a.carry) << 24);
      // delta: [0:10) cell, [10:18) code, [18:20) reset kind (1 = staged shadow spare, 2 = ring slot in HBM), [20] shadow set, [21] show_taken
      // (cell / code = where the taken object is drawn for this one observation), [22:30) ring slot, [30] the cell changed for good
      uint32_t delta;
      if (o.show_taken) delta = (uint32_t)(S.targets & 0x3FFull) | (a.carry << 10) | (1u << 21);
      else delta = S.ev_dirty_idx >= 0 ? ((uint32_t)S.ev_dirty_idx | (S.ev_dirty_code << 10) | (1u << 30)) : 0u;
      delta |= ((S.ev_reset & 3u) << 18) | ((S.ev_reset == 1u ? S.ev_shadow & 1u : 0u) << 20) | (((S.h - 1u) & P.ring_mask & 0xFFu) << 22) | ((rcode >> 1) << 31);
      if (j >= ROLL_LOG_STEPS) {
        // flow control: entry j reuses the slot of entry j - ROLL_LOG_STEPS, which every encode wave must have consumed
        const uint32_t need = (uint32_t)(j - ROLL_LOG_STEPS + 1);
        spin_polls = 0u;
        while (true) {
          const uint32_t p0 = sync[1], p1 = sync[2], p2 = sync[3];
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)min(p0, min(p1, p2))) >= need) break;
          __builtin_amdgcn_s_sleep(1);
          MG_SPIN_COUNT(0);
          MG_SPIN_POLL(spin_polls);
        }
      }
      logbuf[(j & (ROLL_LOG_STEPS - 1)) * 64 + lane] = make_uint2(pose, active ? delta : 0u);
      if constexpr (MG_SCAL_BY_ENCODE) log3[(j & (ROLL_LOG_STEPS - 1)) * 64 + lane] = (uint16_t)a.step;
      // (DS operations of one wave execute in order: the counter cannot become visible before the entry; the compiler must keep that order)
      MG_WAVE_ORDER();
      sync[0] = (uint32_t)(j + 1);
    }
    MG_SPIN_REPORT(0);
  } else {
    // ---- ENCODE waves: wave k + 1 keeps its own copy of the 64 grids current from the log (a byte write per step, a reset now and then)
    // and produces the observations of steps j = k, k + NE, k + 2 NE, ...
    typedef MG_LDS_VU32 lds_vu32;
    lds_vu32* sync = MG_LDS_AT(P.off_log);
    const uint2* logbuf = (const uint2*)(smem + P.off_log + ROLL_LOG_SYNC_BYTES);
    const int k = ek, NE = NW - 1;
    int mine = k;                                                      // next step this wave produces
    const uint16_t* log3 = (const uint16_t*)(smem + P.off_log + ROLL_LOG_SYNC_BYTES + ROLL_LOG_STEPS * 64 * 8);
    uint32_t mission = a.mission;                                      // this lane's env's mission id: the episode's (followed through the resets below)
    for (int j = 0; j < P.T; j++) {
      spin_polls = 0u;
      while ((uint32_t)__builtin_amdgcn_readfirstlane((int)sync[0]) <= (uint32_t)j) { __builtin_amdgcn_s_sleep(1); MG_SPIN_COUNT(1); MG_SPIN_POLL(spin_polls); }
      MG_WAVE_ORDER();
      const uint2 rec2 = logbuf[(j & (ROLL_LOG_STEPS - 1)) * 64 + lane];
      const uint32_t st16 = MG_SCAL_BY_ENCODE ? log3[(j & (ROLL_LOG_STEPS - 1)) * 64 + lane] : 0u;      // (read before the entry is reported consumed below)
      const uint32_t delta = rec2.y;
      const uint32_t rk = (delta >> 18) & 3u;
      if (__ballot(rk != 0u)) {
        if (rk == 1u) {
          const uint32_t* s4 = (const uint32_t*)(C.myshadow + ((delta >> 20) & 1u) * (uint32_t)P.shadow_stride);
          lds_copy_dwords((uint32_t*)mygrid, s4, CS >> 2);
          if constexpr (MG_SCAL_BY_ENCODE) mission = agent_unpack(*(const uint64_t*)((const uint8_t*)C.sspr + ((delta >> 20) & 1u) * (uint32_t)P.spr_stride)).mission;
        } else if (rk == 2u) {
          // the env's second reset of this launch: its spare comes straight from the ring in HBM (loads and their wait stay inside this branch)
          const size_t se = (size_t)((delta >> 22) & 0xFFu) * N + (size_t)e;
          if constexpr (MG_SCAL_BY_ENCODE) mission = agent_unpack(P.spare_agent[se]).mission;
          const uint4* src = (const uint4*)(P.spare_grid + se * CS);
          for (int c = 0; c < (CS >> 4); c++) {
            const uint4 v = src[c];
            uint32_t* d4 = (uint32_t*)(mygrid + c * 16);
            d4[0] = v.x; d4[1] = v.y; d4[2] = v.z; d4[3] = v.w;
          }
        }
      }
      if (delta & (1u << 30)) mygrid[delta & 0x3FFu] = (uint8_t)(delta >> 10);
      MG_WAVE_ORDER();
#if !defined(MG_SPIN_NEGATIVE_CONTROL)      // (the bounded-spin build's negative control: an encode wave that never reports progress -- the dynamics wave's bound must fire)
      sync[1 + k] = (uint32_t)(j + 1);                                 // (in order behind the entry's read)
#endif
      if (j != mine) continue;
      mine += NE;
      Agent av;
      av.x = rec2.x & 0xFFu; av.y = (rec2.x >> 8) & 0xFFu; av.dir = (rec2.x >> 16) & 3u; av.carry = rec2.x >> 24;
      av.step = 0; av.flags = 0; av.mission = 0;
      {
        // the step's scalar record, ahead of its observation rounds (see the dynamics wave; mg_step_scalars: {reward f64 | terminated, truncated,
        // direction, action u8 | mission id u16 | 0}, ONE 16-byte store per env)
        const uint32_t rcode = ((rec2.x >> 23) & 1u) | ((delta >> 31) << 1);
        if (MG_SCAL_BY_ENCODE && active && rcode != 3u && !MG_EXPBIT(P, 8)) {
          double rew = 0.0;
          if (rcode == 1u) rew = reward_exact(st16, P.max_steps); else if (rcode == 2u) rew = -1.0;
          uint4 v;
          v.x = (uint32_t)__double2loint(rew); v.y = (uint32_t)__double2hiint(rew);
          v.z = ((rec2.x >> 18) & 1u) | (((rec2.x >> 19) & 1u) << 8) | (av.dir << 16) | (((rec2.x >> 20) & 7u) << 24);
          v.w = mission & 0xFFFFu;
          *(uint4*)(P.out + (size_t)slot_of(j) * P.slot_bytes + o_scal) = v;
        }
      }
      observe(slot_of(j), av, ((delta >> 21) & 1u) != 0u, delta & 0x3FFu, (delta >> 10) & 0xFFu, scodes, 3);
    }
    MG_SPIN_REPORT(1);
    return;          // (nothing to report, no state to write back: the dynamics wave owns both -- and the env state is dead on this path)
  }

  if (share) {
    // the one step's observation, encoded by every wave of the workgroup from the stepping wave's code stream
    __syncthreads();
    const uint8_t* codes0 = smem + P.off_T;
    uint8_t* obase = P.obs + (size_t)P.slot0 * P.obs_stride + (size_t)wg * P.obs_wg_stride;
    const int nbytes = nvalid * OBE, nvec = nbytes >> 4;
    if (MG_ENCODE_QUADS && !FULL && nvalid == 64 && nthreads == 64 * ROLL_MAX_WAVES) {
      if (!MG_EXPBIT(P, 2)) encode_quads<64 * ROLL_MAX_WAVES, 64 * VIEW_CELLS / 4, false>(tid, codes0, slut, obase, true, (MG_ALIGN_ROUNDS && ((uintptr_t)obase & 127u) == 64u) ? 16 : 0);     // four rounds, one of them 16 threads wide
    } else if (MG_ENCODE_QUADS && !FULL && nvalid == 32 && nthreads == 64 * ROLL_MAX_WAVES) {
      if (!MG_EXPBIT(P, 2)) encode_quads<64 * ROLL_MAX_WAVES, 32 * VIEW_CELLS / 4, false>(tid, codes0, slut, obase);
    } else if (MG_ENCODE_QUADS && nvalid == 64) {
      const int nq = 16 * (FULL ? cells : VIEW_CELLS);
      if (!MG_EXPBIT(P, 2))
        for (int u = tid; u < nq; u += nthreads) {
          uint32_t o3[3];
          obs7_quad((uint32_t)u, codes0, slut, o3);
          Out12 v; v.x = o3[0]; v.y = o3[1]; v.z = o3[2];
          store12(obase, (uint32_t)u * 12u, v, nt);
        }
    } else if (!MG_EXPBIT(P, 2))
      for (int c = tid; c <= nvec; c += nthreads) {
        uint32_t o4[4];
        obs7_chunk((uint32_t)c, codes0, slut, o4);
        if (c < nvec) { uint4 v; v.x = o4[0]; v.y = o4[1]; v.z = o4[2]; v.w = o4[3]; store16(obase, (uint32_t)c * 16u, v, nt); }
        else for (int b = 0; b < (nbytes & 15); b++) obase[(nvec << 4) + b] = (uint8_t)(o4[b >> 2] >> (8 * (b & 3)));
      }
  }

  // ---- launch end.  Device errors and finished episodes: every wave for the steps it produced; state: the last wave ----
  if (errs_mine && active) report_errors(P.err, errs_mine);
  if (fin_total && lane == 0) atomicAdd(&P.counters[STAT_EPISODES + wg], (unsigned long long)fin_total);
  if (!last_wave) return;
  if constexpr (gg_group(GG) == GG_DYNOBS)         // episodes drawn inside the loop count like the generator kernels' (mg_get_counters sums the slots)
    if (ngen) atomicAdd(&P.counters[(size_t)P.stat_gen_off + 2u * (((uint32_t)wg * 64u + (uint32_t)lane) & (STAT_GEN_SLOTS - 1u))], (unsigned long long)ngen);
  if constexpr (gg_group(GG) == GG_ROOMGRID) if (MG_RULE(GG, P) == RULE_GOTO) {
    const uint32_t fl = (a.flags & ~FLAG_TARGETS_STALE) | (S.cur != S.targets ? FLAG_TARGETS_STALE : 0u);
    if (fl != a.flags) { a.flags = fl; S.rec_dirty = true; }
  }
  if constexpr (gg_group(GG) == GG_SENTENCE && MG_INSTR_LDS) {
    // the records go back the way they came (this wave is the only one that touched them; its own LDS writes are in order before these reads)
    uint64_t* gi = P.instr + (size_t)env0 * INSTR_WORDS;
    for (int k = lane; k < nvalid * INSTR_WORDS; k += 64) {
      const uint32_t ce = ((uint32_t)k * 1639u) >> 16, w = (uint32_t)k - ce * (uint32_t)INSTR_WORDS;
      gi[k] = sinstr[ce * ROLL_INSTR_STRIDE + w];
    }
  }
  if (active) {
    if (S.rec_dirty) P.agent[e] = agent_pack(a);
    if (goto_rule && S.aux_dirty) P.aux[e] = S.targets;
    if constexpr (gg_group(GG) == GG_DYNOBS) {
      if (obst_dirty) P.aux[e] = obst;
      if (P.phase == PHASE_STEP) rng.store(P.rng, N, (size_t)e);
    }
    // (sentence levels: k_verify publishes head, after it copied the consumed slot's instruction record)
    if (S.h != h_in && !(gg_group(GG) == GG_NONE && MG_RULE(GG, P) == RULE_SENTENCE)) P.head[e] = S.h;
  }
  {
    const unsigned long long wb = __ballot(active && S.wb_all);        // envs whose whole live grid changed (new episode, fused launch)
    if (wb) {
      uint4* live = (uint4*)(P.grid + (size_t)env0 * CS);
      for (int c = lane; c < nchunks; c += 64) {
        const uint32_t ce = ((uint32_t)c * P.cpe_magic) >> 20, part = (uint32_t)c - ce * (uint32_t)cpe;
        if ((wb >> ce) & 1ull) {
          const uint32_t* s = (const uint32_t*)(sgrid + ce * GS + part * 16);
          uint4 v; v.x = s[0]; v.y = s[1]; v.z = s[2]; v.w = s[3];
          live[c] = v;
        }
      }
    }
  }
  if (P.seg_count) {
    const bool want = active && (P.live_gen ? ((a.flags & FLAG_RESET_PENDING) != 0u && P.phase == PHASE_STEP) : (S.h != h_in));
    const unsigned long long m = __ballot(want);
    if (gg_group(GG) == GG_DYNOBS && P.live_gen == 2) {
      // (in-loop redraws: every earlier request was served by this launch's first step -- the list becomes the envs waiting NOW)
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (want) P.seg[(size_t)wg * P.seg_cap + rank] = (uint32_t)e;
      if (lane == 0) P.seg_count[wg] = (uint32_t)__popcll(m);
    } else
    if (m) {
      // (the segment's fill count is read here, where a request is filed -- in the prologue it was one more round trip before the grids)
      uint32_t qn = uni32(P.seg_count[wg]);
      const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (want && qn + rank < (uint32_t)P.seg_cap) P.seg[(size_t)wg * P.seg_cap + qn + rank] = (uint32_t)e;
      qn = min(qn + (uint32_t)__popcll(m), (uint32_t)P.seg_cap);
      if (lane == 0) P.seg_count[wg] = qn;
    }
  }
}

}  // namespace mg
