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

MG_HD bool lane_gen_kind(int kind) {
  return kind == 0 || kind == 1 || kind == 2 || kind == 3 || kind == 4 || kind == 5 || kind == 6 || kind == 7 ||
         kind == 16 || kind == 17 || kind == 18 || kind == 19 || kind == 20;
}
MG_HD int lane_grid_stride(int CS) { return CS + 4; }                 // odd dword stride: the 64 lanes' grids start in different LDS banks

template <class R>
MG_HD void generate_episode_lane(R& rng, LaneGrid& g, const GenParams& P, GenResult& out) {
  out.ax = out.ay = 1; out.dir = 0; out.mission = 0; out.retries = 0; out.failed = false; out.aux = 0;
  switch (P.kind) {
    case 0: gen_empty(rng, g, P, out); return;
    case 1: gen_doorkey(rng, g, P, out); return;
    case 2: gen_crossing(rng, g, P, out); return;
    case 3: case 16: case 17: case 18: case 19: gen_goto(rng, g, P, out); return;
    case 4: gen_lavagap(rng, g, P, out); return;
    case 5: gen_distshift(rng, g, P, out); return;
    case 6: gen_fourrooms(rng, g, P, out); return;
    case 7: gen_fetch(rng, g, P, out); return;
    case 20: gen_gotoobject(rng, g, P, out); return;
    default: out.failed = true; return;
  }
}

// one lane: the episode of env e for ring slot `slot`
template <class R>
MG_D void generate_one_lane(const GenArgs& A, int e, uint32_t slot, LaneGrid& g) {
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
  A.dst_agent[se] = agent_pack(ag);
  if (A.dst_aux) A.dst_aux[se] = out.aux;
  if (out.failed) report_errors(A.err, (uint32_t)ERR_GENERATOR);
  unsigned long long* st = A.counters + A.stat_gen_off + 2u * ((blockIdx.x * 64u + threadIdx.x) & (STAT_GEN_SLOTS - 1u));
  atomicAdd(&st[0], 1ull);
  if (out.retries) atomicAdd(&st[1], (unsigned long long)out.retries);
}

// A.wps wavefronts per request segment (= per 64-env step workgroup): wave w of a segment serves its requests w, w + wps, w + 2 wps, ... one
// per lane.  Few requests per wave on purpose: the lanes of a wave wait for each other in every rejection loop and every whole-map retry
// (the wave runs as long as its unluckiest lane), and a generator is a chain of dependent 128-bit multiplies -- many short waves overlap,
// one wave with 36 diverging lanes does not (GoToRedBall x 32 768: 10.1 us per step with one wave per segment, see profiles/r4/lane_refill.txt).
template <class R>
__global__ void __launch_bounds__(64) k_refill_lane(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const int sidx = (int)(blockIdx.x / (uint32_t)A.wps), w = (int)(blockIdx.x % (uint32_t)A.wps);
  const int cnt = (int)A.seg_count[sidx];
  if (w >= cnt) return;
  LaneGrid g;
  g.p = smem + lane * lane_grid_stride(A.CS); g.W = A.gp.W; g.H = A.gp.H; g.lane = lane; g.nonempty = 0; g.walls = 0;
  const uint32_t* seg = A.seg + (size_t)sidx * A.seg_cap;
  for (int k = w + lane * A.wps; k < cnt; k += 64 * A.wps) {
    const int e = (int)seg[k];
    const uint32_t old = atomicMax(&A.claim[e], A.epoch);
    if (old >= A.epoch) continue;                               // another request of this batch already covers the env
    const uint32_t h = A.head[e] + A.ring_mask + 1u;            // every slot below head + R is free to fill
    uint32_t t = A.tail[e];
    if (h - t > A.ring_mask + 1u) { report_errors(A.err, (uint32_t)ERR_GENERATOR); continue; }   // ring bookkeeping broken: never spin
    while (t != h) {
      generate_one_lane<R>(A, e, t & A.ring_mask, g);
      t++;
    }
    A.tail[e] = t;
  }
}

// Direct generation with one lane per env (explicit reset(seed=...): the live episode, then the ring slots; mg_set_rng): lane l of workgroup b
// draws env 64 b + l.  The destination pointers are pre-offset to the ring slot by the host (gen_args), like k_generate's.
template <class R>
__global__ void __launch_bounds__(64) k_generate_lane(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const int e = (int)blockIdx.x * 64 + lane;
  if (e >= A.N) return;
  if (A.mask && !A.mask[e]) return;
  LaneGrid g;
  g.p = smem + lane * lane_grid_stride(A.CS); g.W = A.gp.W; g.H = A.gp.H; g.lane = lane; g.nonempty = 0; g.walls = 0;
  generate_one_lane<R>(A, e, 0u, g);
}

}  // namespace mg
