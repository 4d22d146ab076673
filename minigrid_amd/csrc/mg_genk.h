// mg_genk.h — the episode-generator kernels: one wavefront draws one episode (the reference's _gen_grid, restated in mg_gen.h)
// into the spare-episode ring (k_refill, second stream) or directly (k_generate: reset(seed=...), mg_set_rng).  See mg_step.h
// for the ring's role on the step path.
#pragma once
#include "mg_device.h"
#include "mg_gen.h"
#include "mg_rng.h"

namespace mg {

// ======================================================================================================
// Episode generation (the reference's _gen_grid, see mg_gen.h): one wavefront draws one episode.
// ======================================================================================================
struct GenArgs {
  GenParams gp;
  uint8_t* dst_grid; uint64_t* dst_agent;            // slot 0 of the destination (ring or live state)
  uint64_t* rng; uint64_t* rng_snap;                 // rng_snap != null: save the pre-draw state of slot s there first
  uint64_t* dst_aux;                                 // auxiliary word of the generated episode (GenResult.aux) or null
  uint64_t* dst_instr;                               // sentence levels: [slot][N][INSTR_WORDS] instruction records, or null
  uint32_t* gstate; uint32_t* gsnap;                 // LevelGen: generator state carried across episodes [N]; as it was before slot s [R][N]
  const uint8_t* mask;                               // direct mode: optional per-env mask
  uint32_t* err; unsigned long long* counters;
  int N, CS;
  int cap_words;                                     // draw-buffer capacity per generating wave (LDS), in words
  int stat_gen_off;                                  // first generator statistics slot in `counters`
  int live;                                          // 1: requests are regenerated IN PLACE (dst = live state): only
                                                     //    envs still flagged RESET_PENDING are drawn, and come out FRESH
  int stuck_mode;                                    // an episode whose drawing met RoomGrid.place_agent's endless loop (GenResult.stuck):
                                                     //    0 = mark the record FLAG_STUCK (ring slots: the error is due when the episode is
                                                     //    taken), 1 = report ERR_GENERATOR now (the live state), 2 = neither (redrawn, accepted)
  // refill mode (k_refill): request segments of one batch, ring bookkeeping
  const uint32_t* seg; uint32_t* seg_count; int seg_cap;
  int wps;                                           // generating workgroups (one wavefront each) per request segment
  const uint32_t* head; uint32_t* tail; uint32_t* claim; uint32_t epoch; uint32_t ring_mask;
  // packed lane refill (k_refill_lane_packed, mg_genlane.h): exclusive prefix sums of the nseg request counts (seg_off[nseg] = all requests of the batch),
  // lanes used per generating wavefront
  const uint32_t* seg_off; int nseg; int lpw;
  // burst hybrid (round 5): a batch with at least burst_min requests (a synchronized truncation burst: every env of a long-episode level at once) is
  // served by k_refill_lane_packed -- dense lanes, throughput --, a smaller one by k_refill -- cooperative wavefronts, latency; both kernels are
  // launched and the one whose case it is not returns at once (0 = off: no such check)
  uint32_t burst_min;
  // lane refills (round 6): a request draws max(slot_cap, a quarter) of its env's free ring slots while the ring is at least half full (0 = all of them;
  // refill_slots, mg_genlane.h).  A lane draws its
  // env's free slots one after the other (one stream), so a wavefront runs as long as its longest chain -- BabyAI-GoToRedBall consumes 2.3 spares per env
  // and batch on average, the unluckiest of 64 lanes 7-10 -- and what the refill costs the chip is wavefronts x longest chain.  The ring is 256 deep: a slot
  // left for the env's next request (it resets again soon: that is why it had so many) is drawn then; nothing reads a ring's fill level but the kernels.
  uint32_t slot_cap;
};


#ifdef MG_DEBUG_TIMING
// tuning aid (never built into the product library): cycle stamps of the first wave of block 0 -> counters[4..]
#define MG_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) A.counters[4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define MG_STAMP(k) do { } while (0)
#endif

MG_D uint64_t pick5(const uint64_t w[5], uint32_t k) { return k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : k == 3 ? w[3] : w[4]; }

constexpr int GEN_SBASE_BYTES = (int)GEN_SBASE_ENTRIES * 16;
constexpr int GEN_SCRATCH_BYTES = 64;   // generator state that must survive a restart from a checkpoint (MultiRoom's room lists), at the end
constexpr int GEN_INSTR_BYTES = INSTR_WORDS * 8;   // sentence levels: the instruction record under construction, after the scratch words
MG_HD int gen_wave_lds_bytes(int CS, int cap_words, bool sentence = false) {
  return CS + GEN_SBASE_BYTES + (cap_words + 4) * 4 + GEN_SCRATCH_BYTES + (sentence ? GEN_INSTR_BYTES : 0);
}

// wave-cooperative: all 64 lanes of one wave call this with the same `e` and ring slot; `lds` = gen_wave_lds_bytes() of LDS
template <int GG, class RNG>
MG_D void generate_one(const GenArgs& A, RNG& rng, int e, uint32_t slot, uint32_t flags_out, uint32_t lane, uint8_t* lds) {
  const size_t N = (size_t)A.N;
  const size_t se = (size_t)slot * N + (size_t)e;      // index of (slot, env) in the [R][N] arrays

  uint8_t* mygrid = lds;
  MG_STAMP(1);
#if defined(MG_GEN_ATTR) && defined(__HIP_DEVICE_COMPILE__)
  GenAttr ga0; ga0.t = __builtin_readcyclecounter(); ga0.attempts = 0;
  for (int k = 0; k < MG_GA_N; k++) ga0.ph[k] = 0;
#endif
  rng.load(A.rng, N, (size_t)e, lds + A.CS);
  MG_STAMP(2);
  if (A.rng_snap && lane < 5u) A.rng_snap[(size_t)slot * 5u * N + lane * N + (size_t)e] = pick5(rng.w_in, lane);
  GridRef g{ mygrid, A.gp.W, A.gp.H, (int)lane };
  for (int k = A.gp.W * A.gp.H + (int)lane; k < A.CS; k += 64) mygrid[k] = 0;
  GenResult out;
  const int scratch0 = gen_wave_lds_bytes(A.CS, A.cap_words) - GEN_SCRATCH_BYTES;
  if (A.gstate) {
    // LevelGen's locked_room: the generator state this env's previous episode left (mg_gen.h gen_levelgen); the value before this
    // slot's episode is kept, like rng_snap, for restarts of the ring
    const uint32_t gs = uni32(A.gstate[e]);
    if (lane == 0) { ((uint32_t*)(mygrid + scratch0))[0] = gs; if (A.gsnap) A.gsnap[se] = gs; }
  }
  out.gstate = 0; out.stuck = 0;
#if defined(MG_GEN_ATTR) && defined(__HIP_DEVICE_COMPILE__)
  out.ga = ga0;
#endif
  // draw-budget loop: buffer `budget` draws, run the generator.  A pass that ran out of draws restarts from its
  // last checkpoint (GoToRedBall: the start of the current whole-map attempt) with a fresh buffer, or -- no
  // checkpoint passed -- is replayed from the start (same draws, same path) with twice the budget.  One refill
  // covers every DoorKey/Crossing episode; GoToRedBall (about 60 draws per attempt, 15.6 % of attempts rejected)
  // starts with three.
  const uint32_t cap = ((uint32_t)A.cap_words / RNG::kRefillWords) * RNG::kRefillWords;
  // (round 6: the multi-room levels start with what an attempt typically draws -- a 3 x 3 maze attempt ~250 words (5 attempts per BabyAI-GoTo episode),
  // a MultiRoom-N6 chain search > 1000 words per episode: with one refill of 128 nearly every attempt ran twice, once into the end of the buffer and
  // once more from its checkpoint: profiles/r6/refill_attribution_*_before.txt "prologue")
  const int kd = A.gp.kind;
  const uint32_t want0 = kd == 3 ? 384u : kd == 53 ? 512u : (kd >= 33 && kd <= 35) ? 768u : kd == 23 ? 512u : (kd >= 21 && kd <= 52) ? 256u : 1u;
  const uint32_t budget0 = ((want0 + RNG::kRefillWords - 1u) / RNG::kRefillWords) * RNG::kRefillWords;
  uint32_t budget = budget0, retries_before = 0;
  out.resume = 0;
  for (;;) {
    out.carry = 0;
    budget = min(budget, cap);
    while (rng.limit < rng.off + budget) rng.refill();
    MG_STAMP(3);
    MG_GA(out, 0);
    rng.begin_pass();
    // The generator parameters are made opaque per pass: otherwise every switch case's loop-invariant set-up is
    // hoisted out of this (rarely repeated) loop and all of it is live at once -- 160+ VGPRs instead of < 70.
    GenParams gp = A.gp;
    gp.scratch_off = gen_wave_lds_bytes(A.CS, A.cap_words) - GEN_SCRATCH_BYTES;
    gp.instr_off = gp.scratch_off + GEN_SCRATCH_BYTES;
#ifndef MG_EMU      // (the host emulator of tests/emu compiles these sources as plain C++: no register-class constraints there)
    asm volatile("" : "+s"(gp.kind), "+s"(gp.W), "+s"(gp.H), "+s"(gp.start_x), "+s"(gp.start_y), "+s"(gp.start_dir));
    asm volatile("" : "+s"(gp.num_crossings), "+s"(gp.obstacle_cell), "+s"(gp.num_dists), "+s"(gp.strip2_row), "+s"(gp.room_size), "+s"(gp.random_length), "+s"(gp.scratch_off));
#endif
    g.W = gp.W; g.H = gp.H;
#ifndef MG_EMU
    asm volatile("" : "+v"(g.p), "+v"(g.lane));
#endif
    generate_episode<GG>(rng, g, gp, out);
    MG_STAMP(4);
    out.retries += retries_before;
    if (!rng.dead()) break;
    if (rng.ck != 0) { retries_before = out.retries; rng.rebase_to_checkpoint(); budget = budget0; out.resume = 1; continue; }
    if (budget >= cap) { out.failed = true; break; }
    budget *= 2u;
  }
  uint64_t w[5];
  rng.final_words(w);
  MG_STAMP(5);
  if (lane < 5u) A.rng[lane * N + (size_t)e] = pick5(w, lane);
  MG_WAVE_LDS_SYNC();
  uint4* dst = (uint4*)(A.dst_grid + se * A.CS);
  for (int k = (int)lane; k < (A.CS >> 4); k += 64) dst[k] = ((const uint4*)mygrid)[k];
  if (A.dst_instr && lane < (uint32_t)INSTR_WORDS) A.dst_instr[se * INSTR_WORDS + lane] = ((const uint64_t*)(mygrid + scratch0 + GEN_SCRATCH_BYTES))[lane];
  if (lane == 0) {
    Agent ag; ag.x = out.ax; ag.y = out.ay; ag.dir = out.dir; ag.carry = out.carry; ag.step = 0; ag.mission = out.mission;
    ag.flags = flags_out | (out.carry ? FLAG_SHOW_TAKEN : 0u) | ((out.stuck && A.stuck_mode == 0) ? FLAG_STUCK : 0u);
    A.dst_agent[se] = agent_pack(ag);
    if (A.dst_aux) A.dst_aux[se] = out.aux;
    if (A.gstate) A.gstate[e] = out.gstate;
    if (out.failed || (out.stuck && A.stuck_mode == 1)) report_errors(A.err, (uint32_t)ERR_GENERATOR);
#if defined(MG_GEN_ATTR) && defined(__HIP_DEVICE_COMPILE__)
    MG_GA(out, 7);
    for (int k = 0; k < MG_GA_N; k++) if (out.ga.ph[k]) atomicAdd(&A.counters[4 + k], (unsigned long long)out.ga.ph[k]);
    atomicAdd(&A.counters[4 + MG_GA_N], (unsigned long long)out.ga.attempts);
    atomicAdd(&A.counters[4 + MG_GA_N + 1], 1ull);
#endif
    unsigned long long* st = A.counters + A.stat_gen_off + 2u * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (STAT_GEN_SLOTS - 1u));
    atomicAdd(&st[0], 1ull);                                         // (mostly) private slot per generating wave
    if (out.retries) atomicAdd(&st[1], (unsigned long long)out.retries);
  }
  MG_STAMP(6);
  MG_WAVE_LDS_SYNC();
}

// Direct launch over all envs (optionally masked): explicit reset(seed=...), mg_set_rng.  4 generating waves per workgroup.
// The destination pointers are pre-offset to the ring slot by the host.
constexpr int GEN_THREADS = 256;
template <int GGEN, class RNG>
__global__ void __launch_bounds__(GEN_THREADS) k_generate(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t lane = threadIdx.x & 63u;
  const int wave = (int)(threadIdx.x >> 6);
  MG_STAMP(0);
  RNG rng;
  rng.prefetch(lane);
  uint8_t* lds = smem + wave * gen_wave_lds_bytes(A.CS, A.cap_words, A.dst_instr != nullptr);
  const int nwaves = (int)gridDim.x * (GEN_THREADS / 64);
  for (int e = (int)blockIdx.x * (GEN_THREADS / 64) + wave; e < A.N; e += nwaves) {
    if (A.mask && !uni32(A.mask[e])) continue;
    generate_one<GGEN, RNG>(A, rng, e, 0u, 0u, lane, lds);
  }
}

// Refill launch (second stream): workgroup b serves the request segment of step-wave b -- the envs of that 64-env group
// that took a spare out of their ring during the batch.  A request is an env id; an env may be listed more than once
// (several launches of one batch), the first wave to raise claim[e] to this batch's epoch serves it: it draws episodes
// into the consumed slots tail .. head-1 in stream order.  head[] may already be ahead of what the batch consumed
// (later step launches run concurrently): those slots are free as well, and drawing them early is harmless.
// live = 1 (DynamicObstacles, same stream, right before the step launch): requests are the envs whose episode ended;
// they are redrawn IN PLACE if they are still waiting for a reset, and come out FRESH (observed, not stepped).
// Launch geometry: ONE generating wavefront per workgroup, A.wps workgroups per request segment (workgroup b serves requests
// b % wps, b % wps + wps, ... of segment b / wps).  Single-wave workgroups keep the LDS footprint at one draw buffer (5 KB), so
// a CU holds 32 generating waves; multi-wave workgroups held their whole allocation until the slowest wave finished and
// capped the chip at ~1000 concurrent generations (LavaCrossing refill: 210 us -> measured in profiles/r2).
template <int GGEN, class RNG>
__global__ void __launch_bounds__(64) k_refill(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t lane = threadIdx.x;
  const int sidx = (int)(blockIdx.x / (uint32_t)A.wps), wave = (int)(blockIdx.x % (uint32_t)A.wps);
  const int cnt = (int)uni32(A.seg_count[sidx]);
  if (wave >= cnt) return;
  if (A.burst_min && uni32(A.seg_off[A.nseg]) >= A.burst_min) return;      // a burst: the packed lane refill launched beside this kernel serves it
  RNG rng;
  rng.prefetch(lane);
  uint8_t* lds = smem;
  const uint32_t* seg = A.seg + (size_t)sidx * A.seg_cap;
  for (int k = wave; k < cnt; k += A.wps) {
    const int e = (int)uni32(seg[k]);
    uint32_t old = 0;
    if (lane == 0) old = atomicMax(&A.claim[e], A.epoch);
    if (uni32(old) >= A.epoch) continue;                       // another request of this batch already covers the env
    if (A.live) {
      const uint32_t fl = (uint32_t)(uni64(A.dst_agent[e]) >> 48) & 0xFFu;
      if (!(fl & FLAG_RESET_PENDING)) continue;                // an explicit reset() has drawn this env in the meantime
      generate_one<GGEN, RNG>(A, rng, e, 0u, FLAG_FRESH, lane, lds);
      continue;
    }
    const uint32_t h = uni32(A.head[e]) + A.ring_mask + 1u;    // every slot below head + R is free to fill
    uint32_t t = uni32(A.tail[e]);
    if (h - t > A.ring_mask + 1u) { if (lane == 0) report_errors(A.err, (uint32_t)ERR_GENERATOR); continue; }   // ring bookkeeping broken: never spin
    while (t != h) {
      generate_one<GGEN, RNG>(A, rng, e, t & A.ring_mask, 0u, lane, lds);
      t++;
    }
    if (lane == 0) A.tail[e] = t;
  }
  // (the host clears the segment counters on the same stream after this launch; the set is reused QSETS batches later)
}

}  // namespace mg
