// mg_genlane.h — k_refill_lane: the spare-episode refill with ONE LANE PER EPISODE (round 4).
//
// k_refill (mg_genk.h) spends a whole wavefront on one episode: its control flow is wave-uniform and the 64 lanes help where they can
// (jumped-ahead draws, speculative rejection sampling, ballots over the cells), but most of a generator is a sequential scalar program, and
// a wave-instruction is spent per scalar step.  Measured in round 4 (profiles/r4/gotoredball_attr2.txt): BabyAI-GoToRedBall x 32 768 under a
// random policy ends 573 episodes per step, ~4 000 wave-instructions each -- 82 % of the instructions the chip issues per step (5.0 us per step,
// 1.96 us with the resets switched off); LavaCrossing FullyObs x 131 072: 6.5 us against 3.4.  Here every lane runs the generator for its
// own request -- the SAME generator source (mg_gen.h: the single-room levels are templated on the grid type; LaneGrid = a private byte
// grid per lane) on the lane's own numpy PCG64 / Philox stream (Pcg64Stream / PhiloxStream, mg_rng.h: the per-lane forms of the wave-cooperative
// WavePcg64 / WavePhilox, same SoA words) -- so a wavefront draws 64 episodes at once and a wave-instruction is spent per scalar step of 64
// generators.  Rejection loops and whole-map retries make the lanes diverge (a lane idles while its neighbours retry); it is off the step
// stream's critical path (three refill batches of slack, mg_api.hip).  Bit-exact like k_refill: same draw order, same Lemire / masked-rejection
// arithmetic, same ring protocol (claim epoch, tail .. head + R - 1 in stream order, the stream words before each slot in rng_snap).
// Levels it serves: lane_gen_kind() below (the single-room levels incl. every BASELINE.json config); the others keep k_refill.
#pragma once
#include "mg_genk.h"

namespace mg {

// Which levels the lane kernels serve (round 5: measured on MI355X, profiles/r5/lane_wide_*, ab_lane_generators_product_build.txt,
// ab_packed_lane_refill.txt; G env-steps/s, lane refill against the wavefront-per-episode refill of the SAME library):
//   KeyCorridorS3R3 x 131 072   3.21 -> 14.78     Unlock 17.6 -> 20.1     UnlockPickup 21.1 -> 22.7     BlockedUnlockPickup 22.8 -> 23.4
//   BabyAI-GoTo x 131 072       1.66 ->  1.80 (a few busy lanes per wave) / 1.42 (packed: whole waves of busy lanes)
//   BossLevel x 131 072         4.58 ->  3.07 / 1.76       MultiRoom-N6 x 65 536   2.40 -> 2.18 / 2.47
// REFILL on lanes: the single-room levels (FN 0), the Unlock family (FN 2) and KeyCorridor (FN 5).  The maze / sentence levels and MultiRoom keep
// k_refill: their long reachability floods and whole-map retries diverge across a wave's lanes (a wave of 64 maze generators runs 4-7 ms, a cooperative
// wavefront 0.8 ms per episode), and a refill is latency-critical -- one generator stream, batch after batch.
// DIRECT generation on lanes: EVERY level (one kernel per generator function, lane_fn_of_kind: one kernel with every generator needs 512 VGPRs and
// spills): reset(seed) and its ring fill draw one episode per env and slot -- every lane busy, throughput counts, and lanes win everywhere (ring fill of
// BabyAI-GoTo x 131 072: 30.4 -> 8.0 ms per slot; BossLevel 25.3 -> 11.8; MultiRoom-N6 2.0 -> 1.8).  Which path a handle takes: mg_api.hip mg_create.
// (MG_LANE_WIDE is accepted for old scripts and changes nothing any more: every lane kernel is in the product build.)
#ifndef MG_LANE_WIDE
#define MG_LANE_WIDE 0
#endif
MG_HD bool lane_gen_kind_base(int kind) {
  return kind == 0 || kind == 1 || kind == 2 || kind == 3 || kind == 4 || kind == 5 || kind == 6 || kind == 7 ||
         kind == 16 || kind == 17 || kind == 18 || kind == 19 || kind == 20;
}
// the levels besides the single-room ones whose REFILL runs on lanes: FN 2 (gen_unlock_family) and FN 5 (gen_keycorridor)
MG_HD bool lane_gen_kind_product_fn(int kind) { return kind == 9 || kind == 10 || kind == 11 || kind == 14 || kind == 30; }
// the levels whose per-lane generator exists (GoToDoor 8 .. KeyCorridor 14, LockedRoom 21, Playground 22, MultiRoom 23, the BabyAI levels 24-53):
// every level but DynamicObstacles, whose episodes are drawn inside the step kernel (mg_dynobs.h)
MG_HD bool lane_gen_kind_wide(int kind) { return lane_gen_kind_base(kind) || (kind >= 8 && kind <= 14) || (kind >= 21 && kind <= 53); }
// (every level that has lane kernels in the build; which of them USE their lane kernels is the host's choice: mg_api.hip mg_create)
MG_HD bool lane_gen_kind(int kind) { return lane_gen_kind_wide(kind); }
MG_HD int lane_grid_stride(int CS) { return CS + 4; }                 // odd dword stride: the 64 lanes' grids start in different LDS banks
constexpr int LANE_INSTR_STRIDE = INSTR_WORDS + 1;                    // u64 per lane: the sentence levels' instruction record under construction
// LDS of one generating wavefront: 64 private grids (+ 64 instruction records, sentence levels of the wide build)
MG_HD int lane_gen_lds_bytes(int CS, bool sentence) { return 64 * lane_grid_stride(CS) + (sentence ? 64 * LANE_INSTR_STRIDE * 8 : 0); }

// The wide build's kernels are instantiated PER GENERATOR FUNCTION (FN below; 0 = the product kernel with the single-room levels' switch): one kernel
// carrying every generator needs 512 VGPRs and spills (measured at compile time, profiles/r4/lane_wide_build.txt) -- a lane-per-episode kernel lives on
// its occupancy.  gen_babyai_levels (ten levels behind one run-time switch: 512 VGPRs and 1 KB of scratch even alone) is instantiated per LEVEL:
// FN = 100 + kind, the kind a compile-time constant.
MG_HD int lane_fn_of_kind_all(int kind) {            // every level that has a per-lane generator (the host self-test runs them all)
  if (lane_gen_kind_base(kind)) return 0;
  switch (kind) {
    case 8: return 1; case 9: case 10: case 11: return 2; case 12: return 3; case 13: return 4; case 14: case 30: return 5; case 21: return 6;
    case 22: return 7; case 23: return 19; case 24: case 25: case 27: return 8; case 26: return 9; case 28: return 10; case 29: return 11; case 31: return 12;
    case 32: return 13; case 33: case 34: case 35: return 14; case 50: case 51: case 52: return 17; case 53: return 18;
    default: return kind >= 36 && kind <= 45 ? 100 + kind : kind >= 46 && kind <= 49 ? 16 : -1;
  }
}
// the kernel THIS build launches for a level (-1: the level keeps k_refill / k_generate)
MG_HD int lane_fn_of_kind(int kind) { return lane_gen_kind(kind) ? lane_fn_of_kind_all(kind) : -1; }

// iw: the lane's instruction record (sentence levels), st: LevelGen's locked_room words {as the previous episode left it, scratch} -- WIDE only.
// FN = 0: every generator the build serves behind a run-time switch (the product kernel; WIDE: the host selftest); FN > 0: that function alone.
#define MG_LANE_FN(n) if constexpr (FN == 0 || FN == (n))
template <class R, bool WIDE = false, int FN = 0>
MG_HD void generate_episode_lane(R& rng, LaneGrid& g, const GenParams& P, GenResult& out, uint64_t* iw = nullptr, uint32_t* st = nullptr) {
  out.ax = out.ay = 1; out.dir = 0; out.mission = 0; out.retries = 0; out.failed = false; out.aux = 0;
  if constexpr (WIDE) {
    switch (P.kind) {
      case 8: MG_LANE_FN(1) gen_gotodoor(rng, g, P, out); return;
      case 9: MG_LANE_FN(2) gen_unlock_family(rng, g, P, out, 0); return;
      case 10: MG_LANE_FN(2) gen_unlock_family(rng, g, P, out, 1); return;
      case 11: MG_LANE_FN(2) gen_unlock_family(rng, g, P, out, 2); return;
      case 12: MG_LANE_FN(3) gen_redbluedoors(rng, g, P, out); return;
      case 13: MG_LANE_FN(4) gen_memory(rng, g, P, out); return;
      case 14: MG_LANE_FN(5) gen_keycorridor(rng, g, P, out); return;
      case 30: MG_LANE_FN(5) { gen_keycorridor(rng, g, P, out); out.mission = 2u; } return;     // BabyAI KeyCorridor (other.py:252-272): "pick up the ball"
      case 21: MG_LANE_FN(6) gen_lockedroom(rng, g, P, out); return;
      case 22: MG_LANE_FN(7) gen_playground(rng, g, P, out); return;
      case 23: MG_LANE_FN(19) gen_multiroom(rng, g, P, out); return;
      case 24: case 25: case 27: MG_LANE_FN(8) gen_pickup_level(rng, g, P, out); return;
      case 26: MG_LANE_FN(9) gen_openreddoor(rng, g, P, out); return;
      case 28: MG_LANE_FN(10) gen_findobj(rng, g, P, out); return;
      case 29: MG_LANE_FN(11) gen_unlocklocal(rng, g, P, out); return;
      case 31: MG_LANE_FN(12) gen_obstructedmaze(rng, g, P, out); return;
      case 32: MG_LANE_FN(13) gen_putnear(rng, g, P, out); return;
      case 33: case 34: case 35: MG_LANE_FN(14) gen_babyai_maze(rng, g, P, out); return;
      case 36: case 37: case 38: case 39: case 40: case 41: case 42: case 43: case 44: case 45: 
        if constexpr (FN == 0) gen_babyai_levels(rng, g, P, out);
        else if constexpr (FN >= 136 && FN <= 145) { GenParams Pk = P; Pk.kind = FN - 100; gen_babyai_levels(rng, g, Pk, out); }
        return;
      case 46: case 47: case 48: case 49: MG_LANE_FN(16) gen_babyai_put_open(rng, g, P, out); return;
      case 50: case 51: case 52: MG_LANE_FN(17) gen_babyai_seq(rng, g, P, out, iw); return;
      case 53: MG_LANE_FN(18) gen_levelgen(rng, g, P, out, iw, st); return;
      default: break;
    }
    if constexpr (FN != 0) { out.failed = true; return; }
  }
  switch (P.kind) {
    case 0: gen_empty(rng, g, P, out); return;
    case 1: gen_doorkey(rng, g, P, out); return;
    case 2: gen_crossing(rng, g, P, out); return;
    case 3: case 16: case 17: case 18: case 19: gen_goto_lane(rng, g, P, out); return;      // (gen_goto as one loop of draws: mg_gen.h)
    case 4: gen_lavagap(rng, g, P, out); return;
    case 5: gen_distshift(rng, g, P, out); return;
    case 6: gen_fourrooms(rng, g, P, out); return;
    case 7: gen_fetch(rng, g, P, out); return;
    case 20: gen_gotoobject(rng, g, P, out); return;
    default: out.failed = true; return;
  }
}

// one lane: the episode of env e for ring slot `slot`
#undef MG_LANE_FN

// (host-callable: mg_selftest_generate runs it on the CPU, ring slot by ring slot, against the oracle -- tests/test_generators_cpu.py)
template <class R, int FN = 0>
MG_HD void generate_one_lane(const GenArgs& A, int e, uint32_t slot, LaneGrid& g, uint64_t* iw = nullptr) {
  const size_t N = (size_t)A.N;
  const size_t se = (size_t)slot * N + (size_t)e;
  R rng;
  rng.load(A.rng, N, (size_t)e);
  if (A.rng_snap) {
    uint64_t* snap = A.rng_snap + (size_t)slot * 5u * N + (size_t)e;
#pragma unroll
    for (int k = 0; k < 5; k++) snap[(size_t)k * N] = A.rng[(size_t)k * N + (size_t)e];
  }
  if constexpr (R::kEpisodic) rng.begin_episode();
  GenResult out;
  out.gstate = 0; out.stuck = 0; out.carry = 0; out.resume = 0;
  if constexpr (FN != 0) {
    // LevelGen's locked_room: what this env's previous episode left (generate_one, mg_genk.h); the value before this slot's episode is kept for ring restarts
    uint32_t st[2] = { 0u, 0u };
    if (A.gstate) { st[0] = st[1] = A.gstate[e]; if (A.gsnap) A.gsnap[se] = st[0]; }
    if (A.dst_instr) for (int k = 0; k < INSTR_WORDS; k++) iw[k] = 0ull;
    generate_episode_lane<R, true, FN>(rng, g, A.gp, out, iw, st);
    if (A.gstate) A.gstate[e] = out.gstate;
    if (A.dst_instr) for (int k = 0; k < INSTR_WORDS; k++) A.dst_instr[se * INSTR_WORDS + (size_t)k] = iw[k];
  } else
  generate_episode_lane(rng, g, A.gp, out);
  rng.store(A.rng, N, (size_t)e);
  // the grid: CS bytes per (slot, env), cells past W*H zero
  {
    const int cells = g.W * g.H;
    for (int k = cells; k < A.CS; k++) g.p[k] = 0;
    uint32_t* dst = (uint32_t*)(A.dst_grid + se * A.CS);
    const uint32_t* src = (const uint32_t*)g.p;
    for (int k = 0; k < (A.CS >> 2); k += 4) {
      uint32_t t0 = src[k], t1 = src[k + 1], t2 = src[k + 2], t3 = src[k + 3];
      uint4 v; v.x = t0; v.y = t1; v.z = t2; v.w = t3;
      *(uint4*)(dst + k) = v;
    }
  }
  Agent ag; ag.x = out.ax; ag.y = out.ay; ag.dir = out.dir; ag.carry = 0; ag.step = 0; ag.mission = out.mission; ag.flags = 0;
  if constexpr (FN != 0) {          // (generate_one's record: PutNext's start_carrying, RoomGrid.place_agent's endless loop)
    ag.carry = out.carry;
    ag.flags = (out.carry ? FLAG_SHOW_TAKEN : 0u) | ((out.stuck && A.stuck_mode == 0) ? FLAG_STUCK : 0u);
  }
  A.dst_agent[se] = agent_pack(ag);
  if (A.dst_aux) A.dst_aux[se] = out.aux;
  if (out.failed || (FN != 0 && out.stuck && A.stuck_mode == 1)) report_errors(A.err, (uint32_t)ERR_GENERATOR);
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned long long* st = A.counters + A.stat_gen_off + 2u * ((blockIdx.x * 64u + threadIdx.x) & (STAT_GEN_SLOTS - 1u));
  atomicAdd(&st[0], 1ull);
  if (out.retries) atomicAdd(&st[1], (unsigned long long)out.retries);
#else
  // (host form: mg_selftest_generate -- and tests/emu, where the lanes of a launch are fibers and the thread sanitizer watches: atomics there too)
  unsigned long long* st = A.counters + A.stat_gen_off;
  __atomic_fetch_add(&st[0], 1ull, __ATOMIC_RELAXED); __atomic_fetch_add(&st[1], (unsigned long long)out.retries, __ATOMIC_RELAXED);
#endif
}

// A.wps wavefronts per request segment (= per 64-env step workgroup): wave w of a segment serves its requests w, w + wps, w + 2 wps, ... one
// per lane.  Few requests per wave on purpose: the lanes of a wave wait for each other in every rejection loop and every whole-map retry
// (the wave runs as long as its unluckiest lane), and a generator is a chain of dependent 128-bit multiplies -- many short waves overlap,
// one wave with 36 diverging lanes does not (GoToRedBall x 32 768: 10.1 us per step with one wave per segment, see profiles/r4/lane_refill.txt).
// (FN: the wide build's kernels, one per generator function -- lane_fn_of_kind; 0 = the product kernel)
// one refill request on one lane: env e, every ring slot from tail to head + R - 1 in stream order
// How many of an env's `want` free ring slots one request draws now (GenArgs::slot_cap, mg_genk.h): all of them without a cap or once the ring is more
// than half empty; otherwise a quarter of them, at least slot_cap -- the backlog of an env that consumes c spares per batch settles near 4 c, every lane's
// chain near c, instead of the Poisson tail of the 64 lanes' own last batch.
MG_HD uint32_t refill_slots(const GenArgs& A, uint32_t want) {
  if (!A.slot_cap || want > (A.ring_mask + 1u) / 2u) return want;
  const uint32_t q = (want + 3u) >> 2;
  return min(want, max(A.slot_cap, q));
}
template <class R, int FN>
MG_D void refill_lane_request(const GenArgs& A, int e, LaneGrid& g, uint64_t* iw) {
  const uint32_t old = atomicMax(&A.claim[e], A.epoch);
  if (old >= A.epoch) return;                                 // another request of this batch already covers the env
  uint32_t h = A.head[e] + A.ring_mask + 1u;                  // every slot below head + R is free to fill
  uint32_t t = A.tail[e];
  if (h - t > A.ring_mask + 1u) { report_errors(A.err, (uint32_t)ERR_GENERATOR); return; }   // ring bookkeeping broken: never spin
  h = t + refill_slots(A, h - t);                              // (GenArgs::slot_cap: the rest with the env's next request)
  while (t != h) {
    generate_one_lane<R, FN>(A, e, t & A.ring_mask, g, iw);
    t++;
  }
  A.tail[e] = t;
}

template <class R, int FN = 0>
__global__ void __launch_bounds__(64) k_refill_lane(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const int sidx = (int)(blockIdx.x / (uint32_t)A.wps), w = (int)(blockIdx.x % (uint32_t)A.wps);
  const int cnt = (int)A.seg_count[sidx];
  if (w >= cnt) return;
  LaneGrid g;
  g.p = smem + lane * lane_grid_stride(A.CS); g.W = A.gp.W; g.H = A.gp.H; g.lane = lane; g.nonempty = 0; g.walls = 0;
  uint64_t* iw = nullptr;
  if constexpr (FN != 0) iw = (uint64_t*)(smem + 64 * lane_grid_stride(A.CS)) + lane * LANE_INSTR_STRIDE;
  const uint32_t* seg = A.seg + (size_t)sidx * A.seg_cap;
  for (int k = w + lane * A.wps; k < cnt; k += 64 * A.wps) refill_lane_request<R, FN>(A, (int)seg[k], g, iw);
}

// PACKED refill (round 5).  k_refill_lane above gives every request SEGMENT (the 64 envs of a step workgroup) its own wavefronts, so a wave holds as many
// busy lanes as its segment filed requests -- a handful for the short-episode single-room levels it was tuned on, ONE OR TWO for the levels with long
// episodes and big grids (BabyAI-GoTo x 131 072: ~5 requests per segment and batch): a lane-per-episode kernel with one busy lane is a very slow scalar
// core (measured with the wide build: BossLevel 4.6 -> 3.1 G, MultiRoom-N6 2.4 -> 2.2).  Here the batch's requests are first numbered across the
// segments (k_seg_scan: exclusive prefix sums of the segment counts) and wavefront b serves requests [b lpw, (b + 1) lpw): every lane of a generating
// wave is busy, a refill of ten thousand episodes is a few hundred wavefronts instead of eight thousand, and the chip is left to the step kernel.
// lpw (lanes per wave, <= 64) trades lane utilisation against how long a wave waits for its unluckiest lane.
constexpr int SEG_SCAN_THREADS = 256;                        // one workgroup; launch with SEG_SCAN_THREADS * 4 bytes of dynamic LDS
template <int UNUSED = 0>      // (a template only so that the header can be included by several translation units)
__global__ void __launch_bounds__(SEG_SCAN_THREADS) k_seg_scan(const uint32_t* seg_count, uint32_t* seg_off, int nseg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* part = (uint32_t*)smem;
  const int tid = (int)threadIdx.x;
  const int per = (nseg + SEG_SCAN_THREADS - 1) / SEG_SCAN_THREADS, lo = min(nseg, tid * per), hi = min(nseg, lo + per);
  uint32_t sum = 0;
  for (int k = lo; k < hi; k++) sum += seg_count[k];
  part[tid] = sum;
  __syncthreads();
  for (int d = 1; d < SEG_SCAN_THREADS; d <<= 1) {           // Hillis-Steele inclusive scan of the partial sums
    const uint32_t v = tid >= d ? part[tid - d] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;                            // exclusive
  for (int k = lo; k < hi; k++) { seg_off[k] = run; run += seg_count[k]; }
  if (tid == SEG_SCAN_THREADS - 1) seg_off[nseg] = part[SEG_SCAN_THREADS - 1];
}

template <class R, int FN = 0>
__global__ void __launch_bounds__(64) k_refill_lane_packed(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const uint32_t total = A.seg_off[A.nseg];
  if ((uint32_t)blockIdx.x * (uint32_t)A.lpw >= total) return;
  if (A.burst_min && total < A.burst_min) return;              // (burst hybrid: a small batch is k_refill's, mg_genk.h)
  LaneGrid g;
  g.p = smem + lane * lane_grid_stride(A.CS); g.W = A.gp.W; g.H = A.gp.H; g.lane = lane; g.nonempty = 0; g.walls = 0;
  uint64_t* iw = nullptr;
  if constexpr (FN != 0) iw = (uint64_t*)(smem + 64 * lane_grid_stride(A.CS)) + lane * LANE_INSTR_STRIDE;
  for (uint32_t base = (uint32_t)blockIdx.x * (uint32_t)A.lpw; base < total; base += gridDim.x * (uint32_t)A.lpw) {
    const uint32_t q = base + (uint32_t)lane;
    if (lane >= A.lpw || q >= total) continue;
    int lo = 0, hi = A.nseg;                                 // the segment of request q: the last s with seg_off[s] <= q
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.seg_off[mid] <= q) lo = mid; else hi = mid; }
    const int e = (int)A.seg[(size_t)lo * A.seg_cap + (q - A.seg_off[lo])];
    refill_lane_request<R, FN>(A, e, g, iw);
  }
}

// Direct generation with one lane per env (explicit reset(seed=...): the live episode, then the ring slots; mg_set_rng): lane l of workgroup b
// draws env 64 b + l.  The destination pointers are pre-offset to the ring slot by the host (gen_args), like k_generate's.
template <class R, int FN = 0>
__global__ void __launch_bounds__(64) k_generate_lane(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const int e = (int)blockIdx.x * 64 + lane;
  if (e >= A.N) return;
  if (A.mask && !A.mask[e]) return;
  LaneGrid g;
  g.p = smem + lane * lane_grid_stride(A.CS); g.W = A.gp.W; g.H = A.gp.H; g.lane = lane; g.nonempty = 0; g.walls = 0;
  uint64_t* iw = nullptr;
  if constexpr (FN != 0) iw = (uint64_t*)(smem + 64 * lane_grid_stride(A.CS)) + lane * LANE_INSTR_STRIDE;
  generate_one_lane<R, FN>(A, e, 0u, g, iw);
}

}  // namespace mg
