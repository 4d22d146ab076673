// mg_genmr.h — MultiRoom (envs/multiroom.py:118-300) on lanes WITHOUT a private grid per lane (round 6).
//
// The lane-per-episode kernels of mg_genlane.h give every lane a byte grid in LDS: 64 x 644 B = 41 KB per generating wavefront at 25 x 25, i.e. three
// generating wavefronts per CU -- one per SIMD, every latency of the chain search exposed -- and, worse, almost no LDS left for the step kernel's
// workgroups (48 KB each) that run beside a refill: profiles/r6/kernel_stats_multiroom_call7_*.txt, a 32-step launch takes 319 us beside a refill against
// 201 us alone.  A MultiRoom episode needs no grid while it is DRAWN: the chain search looks at room rectangles only, and the two placements that follow
// (place_agent in the first room, place_obj(Goal) in the last) ask one thing of the grid -- is this cell None -- which the rooms' WALL ROW MASKS answer
// (bit x of row y: a wall or a door, i.e. not None; 25 words per lane).  So here a lane draws its episode into registers + 128 bytes of LDS, and the
// WAVE rasterises the 64 episodes one after the other -- lane c produces the 16-byte piece c of the grid from two row masks and stores it straight to
// the ring in HBM, coalesced.  8.2 KB of LDS per generating wavefront.
// Same draws in the same order as gen_multiroom (mg_gen.h), whose lane form stays the host-checked restatement (tests/test_generators_cpu.py) and the
// fallback for shapes this file does not take (mr_lanes_ok).
#pragma once
#include "mg_genlane.h"

namespace mg {

MG_HD int mr_rows_stride(int H) { return H | 1; }                                   // words per lane, odd: the lanes' rows y lie in different banks
constexpr int MR_CUR_STRIDE = 7;                                                     // the chain under construction: 6 words per lane (+ 1: odd stride)
MG_HD int mr_lane_lds_bytes(int H) { return 64 * MR_CUR_STRIDE * 4 + 64 * mr_rows_stride(H) * 4; }
// rows are 32-bit masks, a 16-cell piece must not span more than two rows, rooms are packed with 5-bit coordinates (mr_pack)
MG_HD bool mr_lanes_ok(const GenParams& P, int CS) { return P.kind == 23 && P.W >= 15 && P.W <= 31 && P.H >= 4 && P.H <= 32 && CS <= 1024 && P.num_dists <= 6 && P.room_size <= 15; }

struct MrEp {
  uint32_t room[6];        // the chain (mr_pack words)
  uint32_t n;              // rooms in it
  uint32_t doors;          // 4 bits per door idx 1 .. 5: COLOR_TO_IDX of the door | 8 = a later room's wall was drawn over it
  uint32_t dc_lo, dc_hi;   // the doors' cells (y * W + x), 10 bits each: idx 1 2 3 | idx 4 5
  uint32_t ax, ay, dir, gcell;
  uint32_t failed;
};

// _gen_grid for one lane, from the room count on; `cur` / `rows`: the lane's 6 + H words of LDS
template <class R>
MG_D void mr_draw_lane(R& rng, const GenParams& P, uint32_t* cur, uint32_t* rows, MrEp& E) {
  const int W = P.W, H = P.H;
  LaneGrid g; g.p = (uint8_t*)cur; g.W = W; g.H = H; g.lane = 0; g.nonempty = 0; g.walls = 0;
  const int num_rooms = rand_int(rng, P.num_crossings, P.num_dists + 1);             // multiroom.py:121
  int nbest = 0;
  mr_search_lane(rng, g, P, num_rooms, E.room, nbest);                               // :123-141 (mg_gen.h)
  E.n = (uint32_t)nbest; E.failed = 0u;
  // the rooms' walls as row masks (:153-165; a door sits on a wall: same bit)
  for (int y = 0; y < H; y++) rows[y] = 0u;
#pragma unroll
  for (int k = 0; k < 6; k++) if (k < nbest) {
    const uint32_t r = E.room[k];
    const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
    const uint32_t span = ((1u << sx) - 1u) << tx, edges = (1u << tx) | (1u << (tx + sx - 1));
    for (int j = 0; j < sy; j++) rows[ty + j] |= (j == 0 || j == sy - 1) ? span : edges;
  }
  // the entry doors' colours, in room order (:167-179) -- and whether a LATER room's wall is drawn over the door (the reference draws room by room)
  uint32_t prev = 6u, doors = 0u, dlo = 0u, dhi = 0u;
#pragma unroll
  for (int idx = 1; idx < 6; idx++) if (idx < nbest) {
    const uint32_t k = (uint32_t)rand_int(rng, 0, prev < 6u ? 5 : 6);                // sorted(COLOR_NAMES - {prevDoorColor})
    const uint32_t c = (prev < 6u && k >= prev) ? k + 1u : k;
    prev = c;
    const int ex = (int)((E.room[idx] >> 18) & 31u), ey = (int)((E.room[idx] >> 23) & 31u);
    bool cov = false;
#pragma unroll
    for (int j = idx + 1; j < 6; j++) if (j < nbest) {
      const uint32_t r = E.room[j];
      const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
      const bool inside = ex >= tx && ex < tx + sx && ey >= ty && ey < ty + sy;
      cov = cov || (inside && (ex == tx || ex == tx + sx - 1 || ey == ty || ey == ty + sy - 1));
    }
    doors |= (color_from_sorted(c) | (cov ? 8u : 0u)) << (4 * (idx - 1));
    const uint32_t cell = (uint32_t)(ey * W + ex);
    if (idx <= 3) dlo |= cell << (10 * (idx - 1)); else dhi |= cell << (10 * (idx - 4));
  }
  E.doors = doors; E.dc_lo = dlo; E.dc_hi = dhi;
  // place_agent(roomList[0].top, roomList[0].size) (:184; minigrid_env.py:313-395: the cell must be None -- no wall bit --, no try limit), then the direction
  {
    const uint32_t r = E.room[0];
    const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
    const int hx = min(tx + sx, W), hy = min(ty + sy, H);
    int x = 0, y = 0; bool ok = false;
    for (int tries = 0; tries < (1 << 16) && !ok; tries++) {
      x = rand_int(rng, tx, hx); y = rand_int(rng, ty, hy);
      ok = ((rows[y] >> x) & 1u) == 0u;
    }
    if (!ok) E.failed = 1u;
    E.ax = (uint32_t)x; E.ay = (uint32_t)y;
    E.dir = (uint32_t)rand_int(rng, 0, 4);
  }
  // place_obj(Goal(), roomList[-1].top, roomList[-1].size) (:187): None, and not the agent's cell
  {
    const uint32_t r = nbest == 1 ? E.room[0] : nbest == 2 ? E.room[1] : nbest == 3 ? E.room[2] : nbest == 4 ? E.room[3] : nbest == 5 ? E.room[4] : E.room[5];
    const int tx = (int)(r & 31u), ty = (int)((r >> 5) & 31u), sx = (int)((r >> 10) & 15u), sy = (int)((r >> 14) & 15u);
    const int hx = min(tx + sx, W), hy = min(ty + sy, H);
    int x = 0, y = 0; bool ok = false;
    for (int tries = 0; tries < (1 << 16) && !ok; tries++) {
      x = rand_int(rng, tx, hx); y = rand_int(rng, ty, hy);
      ok = ((rows[y] >> x) & 1u) == 0u && !(x == (int)E.ax && y == (int)E.ay);
    }
    if (!ok) E.failed = 1u;
    E.gcell = (uint32_t)(y * W + x);
  }
}

// one lane: the episode of env e for ring slot `slot`, everything but the grid (generate_one_lane's bookkeeping, mg_genlane.h)
template <class R>
MG_D void mr_generate_one_lane(const GenArgs& A, int e, uint32_t slot, uint32_t* cur, uint32_t* rows, MrEp& E) {
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
  mr_draw_lane(rng, A.gp, cur, rows, E);
  rng.store(A.rng, N, (size_t)e);
  Agent ag; ag.x = E.ax; ag.y = E.ay; ag.dir = E.dir; ag.carry = 0; ag.step = 0; ag.mission = 0; ag.flags = 0;
  A.dst_agent[se] = agent_pack(ag);
  if (A.dst_aux) A.dst_aux[se] = 0ull;
  if (E.failed) report_errors(A.err, (uint32_t)ERR_GENERATOR);
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned long long* st = A.counters + A.stat_gen_off + 2u * ((blockIdx.x * 64u + threadIdx.x) & (STAT_GEN_SLOTS - 1u));
  atomicAdd(&st[0], 1ull);
#else
  unsigned long long* st = A.counters + A.stat_gen_off;
  __atomic_fetch_add(&st[0], 1ull, __ATOMIC_RELAXED);
#endif
}

MG_D uint32_t mr_bytes_of_bits(uint32_t bits4) { return (((bits4 & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu; }     // bit i -> byte i = 0x00 / 0xff

// The wave draws the grids of the lanes in `mask` (their episodes in E, their row masks in LDS) into ring slot entries se (per lane), one after the other:
// lane c = the 16-byte piece c of the CS-byte grid.  Cells past W * H are zero (generate_one_lane).
MG_D void mr_raster_wave(const GenArgs& A, unsigned long long mask, const MrEp& E, size_t se, const uint32_t* rows_wave, int lane) {
  const int W = A.gp.W, H = A.gp.H, cells = W * H, cpe = A.CS >> 4, stride = mr_rows_stride(H);
  const uint32_t w_magic = (65536u + (uint32_t)W - 1u) / (uint32_t)W;
  const int k0 = lane * 16;
  const int y0 = (int)(((uint32_t)k0 * w_magic) >> 16), x0 = k0 - y0 * W;
  const uint32_t vmask = k0 + 16 <= cells ? 0xFFFFu : k0 >= cells ? 0u : ((1u << (cells - k0)) - 1u);
  const uint32_t WALL4 = (uint32_t)CELL_WALL_GREY * 0x01010101u, EMPTY4 = (uint32_t)CELL_EMPTY * 0x01010101u;
  MG_WAVE_LDS_SYNC();                                        // the lanes' row masks are written
  while (mask) {
    const int b = __ffsll((long long)mask) - 1;
    mask &= mask - 1ull;
    const uint32_t n = lane32(E.n, (uint32_t)b), doors = lane32(E.doors, (uint32_t)b), dlo = lane32(E.dc_lo, (uint32_t)b), dhi = lane32(E.dc_hi, (uint32_t)b);
    const uint32_t gcell = lane32(E.gcell, (uint32_t)b);
    const size_t seb = (size_t)lane32((uint32_t)se, (uint32_t)b) | ((size_t)lane32((uint32_t)((uint64_t)se >> 32), (uint32_t)b) << 32);
    const uint32_t* rb = rows_wave + b * stride;
    const uint32_t m0 = y0 < H ? rb[min(y0, H - 1)] : 0u, m1 = y0 + 1 < H ? rb[min(y0 + 1, H - 1)] : 0u;
    const uint32_t bits = ((m0 >> x0) | (m1 << (W - x0))) & 0xFFFFu;           // (W - x0 in 1 .. W: x0 < W <= 32, and W >= 15 keeps 16 cells within two rows)
    uint32_t d[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t wm = mr_bytes_of_bits(bits >> (4 * q)), vm = mr_bytes_of_bits(vmask >> (4 * q));
      d[q] = ((wm & WALL4) | (~wm & EMPTY4)) & vm;
    }
    auto patch = [&](uint32_t cell, uint32_t code) {
      const uint32_t i = cell - (uint32_t)k0;
      if (i < 16u) {
        const uint32_t sh = (i & 3u) * 8u, q = i >> 2;
#pragma unroll
        for (int t = 0; t < 4; t++) d[t] = q == (uint32_t)t ? ((d[t] & ~(0xFFu << sh)) | (code << sh)) : d[t];
      }
    };
#pragma unroll
    for (int idx = 1; idx < 6; idx++) {
      const uint32_t nib = (doors >> (4 * (idx - 1))) & 15u;
      const uint32_t cell = idx <= 3 ? (dlo >> (10 * (idx - 1))) & 1023u : (dhi >> (10 * (idx - 4))) & 1023u;
      if ((uint32_t)idx < n && !(nib & 8u)) patch(cell, make_cell(T_DOOR_CLOSED, nib & 7u));
    }
    patch(gcell, (uint32_t)CELL_GOAL);
    if (lane < cpe) { uint4 v; v.x = d[0]; v.y = d[1]; v.z = d[2]; v.w = d[3]; ((uint4*)(A.dst_grid + seb * A.CS))[lane] = v; }
  }
  MG_WAVE_LDS_SYNC();                                        // (the next round's lanes overwrite their rows)
}

// direct generation: lane l of workgroup b draws env 64 b + l (k_generate_lane's role)
template <class R>
__global__ void __launch_bounds__(64) k_generate_lane_mr(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const int e = (int)blockIdx.x * 64 + lane;
  const bool go = e < A.N && (!A.mask || A.mask[min(e, A.N - 1)]);
  uint32_t* cur = (uint32_t*)smem + lane * MR_CUR_STRIDE;
  uint32_t* rows_wave = (uint32_t*)smem + 64 * MR_CUR_STRIDE;
  MrEp E{};
  if (go) mr_generate_one_lane<R>(A, e, 0u, cur, rows_wave + lane * mr_rows_stride(A.gp.H), E);
  mr_raster_wave(A, __ballot(go), E, (size_t)min(e, A.N - 1), rows_wave, lane);
}

// packed refill (k_refill_lane_packed's role and protocol: requests numbered across the segments, claim epoch, slots tail .. head + R - 1 in stream order)
template <class R>
__global__ void __launch_bounds__(64) k_refill_lane_packed_mr(const GenArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = (int)threadIdx.x;
  const uint32_t total = A.seg_off[A.nseg];
  if ((uint32_t)blockIdx.x * (uint32_t)A.lpw >= total) return;
  if (A.burst_min && total < A.burst_min) return;              // (burst hybrid: a small batch is k_refill's, mg_genk.h)
  uint32_t* cur = (uint32_t*)smem + lane * MR_CUR_STRIDE;
  uint32_t* rows_wave = (uint32_t*)smem + 64 * MR_CUR_STRIDE;
  const size_t N = (size_t)A.N;
  for (uint32_t base = (uint32_t)blockIdx.x * (uint32_t)A.lpw; base < total; base += gridDim.x * (uint32_t)A.lpw) {      // (wave-uniform)
    const uint32_t q = base + (uint32_t)lane;
    int e = 0; uint32_t t = 0, h = 0; bool work = false;
    if (lane < A.lpw && q < total) {
      int lo = 0, hi = A.nseg;                                 // the segment of request q: the last s with seg_off[s] <= q
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.seg_off[mid] <= q) lo = mid; else hi = mid; }
      e = (int)A.seg[(size_t)lo * A.seg_cap + (q - A.seg_off[lo])];
      const uint32_t old = atomicMax(&A.claim[e], A.epoch);
      if (old < A.epoch) {                                     // (else: another request of this batch already covers the env)
        h = A.head[e] + A.ring_mask + 1u;                      // every slot below head + R is free to fill
        t = A.tail[e];
        if (h - t > A.ring_mask + 1u) report_errors(A.err, (uint32_t)ERR_GENERATOR);   // ring bookkeeping broken: never spin
        else work = true;
        if (work) h = t + refill_slots(A, h - t);               // (GenArgs::slot_cap)
      }
    }
    while (__ballot(work && t != h)) {                         // (wave-uniform: a round draws one ring slot of every lane that has one left)
      const bool go = work && t != h;
      const uint32_t slot = t & A.ring_mask;
      MrEp E{};
      if (go) mr_generate_one_lane<R>(A, e, slot, cur, rows_wave + lane * mr_rows_stride(A.gp.H), E);
      mr_raster_wave(A, __ballot(go), E, (size_t)slot * N + (size_t)e, rows_wave, lane);
      if (go) t++;
    }
    if (work) A.tail[e] = t;
  }
}

}  // namespace mg
