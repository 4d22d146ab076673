// mg_kernels_aux.h — the kernels around the step / generator kernels of mg_kernels.h: ring bookkeeping, the sentence levels' verifier,
// DynamicObstacles' obstacle moves, seeding, the RGB blit, the state exchange.  Included by mg_api.hip only (the non-template
// __global__ functions here must be compiled into exactly one translation unit).
#pragma once
#include "mg_kernels.h"
#include "mg_roll.h"
#include "mg_verify.h"

namespace mg {

// mg_get_rng: the reference env's stream position "now" = the state before its next unconsumed spare was drawn
__global__ void k_gather_rng(const uint64_t* rng_snap, const uint32_t* head, uint32_t ring_mask, uint64_t* out, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const size_t s = head ? (size_t)(head[e] & ring_mask) : 0;
  for (int k = 0; k < 5; k++) out[(size_t)k * N + e] = rng_snap[(s * 5 + k) * (size_t)N + e];
}

// reset(seed=...): the ring of the selected envs restarts (head = 0; the host then draws all R slots, tail = R)
__global__ void k_gstate_restore(uint32_t* gstate, const uint32_t* gsnap, const uint32_t* head, const uint8_t* mask, uint32_t R, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  gstate[e] = gsnap[(size_t)(head[e] & (R - 1u)) * (size_t)N + (size_t)e];
}
__global__ void k_ring_restart(uint32_t* head, uint32_t* tail, const uint8_t* mask, uint32_t R, int N, uint32_t* gstate, const uint32_t* gsnap) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  // LevelGen's generator state goes back to what it was after the LIVE episode was drawn (= before the next spare, like mg_get_rng)
  if (gstate) gstate[e] = gsnap[(size_t)(head[e] & (R - 1u)) * (size_t)N + (size_t)e];
  head[e] = 0u; tail[e] = R;
}

// ======================================================================================================
// k_verify: RoomGridLevel.step's second half for the sentence levels (roomgrid_level.py:87-104): update_objs_poss after a drop,
// instrs.verify(action), and -- because max_steps is per episode there (:71-85) -- truncation and the reward.  Runs after every
// k_step launch of such a level (one step per launch) on the state k_step left in HBM; one lane per env.  k_step itself applies
// the action, encodes the observation and takes the spare episode at a reset; it reports reward 0 / terminated 0 / truncated 0.
// ======================================================================================================
struct VerifyParams {
  const uint8_t* grid; uint64_t* agent; uint64_t* instr; const uint64_t* spare_instr;
  uint32_t* head; uint32_t ring_mask;
  uint8_t* rec;                      // the step record k_step just wrote
  size_t off_reward, off_term, off_trunc, off_action, off_sentence;   // fields of env 0's mg_step_scalars (16 bytes per env)
  uint32_t* err;
  int N, W, H, CS, phase, autoreset_next_step;
  int done_actions;                  // verifier.py's use_done_actions
};
__global__ void k_verify(const VerifyParams V) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= V.N) return;
  Agent a = agent_unpack(V.agent[e]);
  uint64_t* I = V.instr + (size_t)e * INSTR_WORDS;
  uint64_t* sent = (uint64_t*)(V.rec + V.off_sentence) + (size_t)e * 2;
  if (a.flags & FLAG_NEW_EPISODE) {
    // k_step took the env's next spare episode in this launch (reset, autoreset): its instruction record comes with it
    // head is published HERE, after the copy: k_refill (generator stream, possibly serving an earlier batch right now) draws into every
    // slot below head + R, so the slot must not count as consumed while its record is still being read
    const uint32_t h = V.head[e];
    const uint64_t* src = V.spare_instr + ((size_t)(h & V.ring_mask) * (size_t)V.N + (size_t)e) * INSTR_WORDS;
    uint64_t m0 = 0, m1 = 0;
    for (int k = 0; k < INSTR_WORDS; k++) { const uint64_t w = src[k]; I[k] = w; if (k == IW_MISSION) m0 = w; if (k == IW_MISSION + 1) m1 = w; }
    a.flags &= ~FLAG_NEW_EPISODE;
    V.agent[e] = agent_pack(a);
    sent[0] = m0; sent[1] = m1;
    __threadfence();
    V.head[e] = h + 1u;
    return;
  }
  sent[0] = I[IW_MISSION]; sent[1] = I[IW_MISSION + 1];
  if (V.phase != PHASE_STEP) return;
  uint32_t max_steps = 0, errbits = 0;
  const uint32_t status = verify_action(I, V.grid + (size_t)e * V.CS, V.W, V.H, a, (uint32_t)V.rec[V.off_action + (size_t)e * 16], max_steps, errbits, V.done_actions);
  const uint32_t term = status != R_CONTINUE, trunc = a.step >= max_steps;
  *(double*)(V.rec + V.off_reward + (size_t)e * 16) = status == R_SUCCESS ? reward_exact(a.step, (int)max_steps) : 0.0;
  V.rec[V.off_term + (size_t)e * 16] = (uint8_t)term;
  V.rec[V.off_trunc + (size_t)e * 16] = (uint8_t)trunc;
  if ((term | trunc) && V.autoreset_next_step) { a.flags |= FLAG_RESET_PENDING; V.agent[e] = agent_pack(a); }
  if (errbits) report_errors(V.err, errbits);
}

// DynamicObstaclesEnv.step, the part before MiniGridEnv.step (dynamicobstacles.py:141-157): remember whether the
// front cell is occupied, then move every obstacle, in list order, to a random free cell of its 3x3 neighbourhood
// (place_obj with max_tries=100 on the ENV's stream; an obstacle that finds no place stays).  This level's step consumes the
// stream, so it is kept out of k_step's register budget.  One wavefront per MOVE_EPB envs: their grids are staged
// into LDS with 16 B/lane coalesced loads (env stride CS + 4: an odd dword stride), lane l works on env l's copy -- the
// rejection-sampling chain (draw, look at the cell, draw again) runs at LDS latency instead of one HBM round trip per try --
// and the grids go back with coalesced 16 B stores.
constexpr int MOVE_EPB = 16;
template <class RNG>
__global__ void __launch_bounds__(64) k_move_obstacles(uint8_t* grid, uint64_t* agent, uint64_t* rng, uint64_t* obst, int N, int W, int H, int CS,
                                                       int n_obst, int epb) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // The per-env chain (PCG64 draw -> cell test -> next draw) is latency-bound and sequential: MOVE_EPB = 16 envs per wavefront
  // (the other lanes only help with the staging) puts four wavefronts on every SIMD at 65 536 envs instead of one.
  const int lane = (int)threadIdx.x, env0 = (int)blockIdx.x * epb, nvalid = min(epb, N - env0);
  const int GS = CS + 4, cpe = CS >> 4, nchunks = nvalid * cpe;
  uint4* live = (uint4*)(grid + (size_t)env0 * CS);
  for (int c = lane; c < nchunks; c += 64) {
    const int ce = c / cpe, part = c - ce * cpe;
    const uint4 v = live[c];
    uint32_t* d = (uint32_t*)(smem + ce * GS + part * 16);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const int e = env0 + lane;
  bool moved = false;
  if (lane < nvalid) {
    Agent a = agent_unpack(agent[e]);
    if (!(a.flags & (FLAG_RESET_PENDING | FLAG_FRESH))) {            // (otherwise: no step for this env in the coming launch)
      uint8_t* g = smem + lane * GS;
      const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
      const uint32_t F = ((unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H) ? (uint32_t)g[fy * W + fx] : (uint32_t)CELL_WALL_GREY;
      const bool not_clear = F != CELL_EMPTY && cell_type(F) != T_GOAL;
      RNG r;
      r.load(rng, (size_t)N, (size_t)e);
      uint64_t o = obst[e];
      for (int i = 0; i < n_obst; i++) {
        const int idx = (int)((o >> (8 * i)) & 0xFF), oy = idx / W, ox = idx - oy * W;
        const int topx = max(ox - 1, 0), topy = max(oy - 1, 0), hx = min(topx + 3, W), hy = min(topy + 3, H);
        int tries = 0, nx = -1, ny = -1;
        for (;;) {
          if (tries > 100) break;                                       // RecursionError, swallowed by `except Exception`
          tries++;
          const int x = rand_int(r, topx, hx), y = rand_int(r, topy, hy);
          if (g[y * W + x] != CELL_EMPTY) continue;
          if (x == (int)a.x && y == (int)a.y) continue;
          nx = x; ny = y;
          break;
        }
        if (nx >= 0) {
          g[ny * W + nx] = (uint8_t)CELL_BALL_BLUE;
          g[idx] = (uint8_t)CELL_EMPTY;
          o = (o & ~(0xFFull << (8 * i))) | ((uint64_t)(ny * W + nx) << (8 * i));
          moved = true;
        }
      }
      r.store(rng, (size_t)N, (size_t)e);
      obst[e] = o;
      a.flags = (a.flags & ~FLAG_NOT_CLEAR) | (not_clear ? FLAG_NOT_CLEAR : 0u);
      agent[e] = agent_pack(a);
    }
  }
  const unsigned long long wb = __ballot(moved);
  __syncthreads();
  if (wb)
    for (int c = lane; c < nchunks; c += 64) {
      const int ce = c / cpe, part = c - ce * cpe;
      if ((wb >> ce) & 1ull) {
        const uint32_t* sp = (const uint32_t*)(smem + ce * GS + part * 16);
        uint4 v; v.x = sp[0]; v.y = sp[1]; v.z = sp[2]; v.w = sp[3];
        live[c] = v;
      }
    }
}

// gymnasium.Env.reset(seed=s): np_random = Generator(PCG64(SeedSequence(s)))  (minigrid_env.py:125)
template <class RNG>
__global__ void k_seed(uint64_t* rng, const uint64_t* seeds, const uint8_t* mask, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  RNG r;
  r.seed(seeds[e]);
  r.store(rng, (size_t)N, (size_t)e);
}

// mark envs for an explicit reset() that continues their stream (consumes the spare)
__global__ void k_mark_pending(uint64_t* agent, const uint8_t* mask, int N) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  Agent a = agent_unpack(agent[e]);
  a.flags |= FLAG_RESET_PENDING;
  agent[e] = agent_pack(a);
}

// ======================================================================================================
// k_render: RGBImgObsWrapper / RGBImgPartialObsWrapper (wrappers.py:287-380) = Grid.render (grid.py:200-242): the
// frame is a mosaic of pre-rendered tiles (mg_tiles.h).  Input: k_step's tile map (one byte per cell = tile key * 2 +
// highlight) and, for the full render, the agent record; output: [N][Ht*ts][Wt*ts][3] bytes.
//
// HBM-write bound (9-12 KB written per env against ~60 B read), so the kernel is organised around the store stream:
//  * A workgroup's EPW consecutive frames are ONE contiguous byte range, dealt out as 16 B chunks, thread t taking
//    chunks t, t + T, t + 2T, ... with T a multiple of the chunks per "period" (R pixel rows, R the smallest count
//    whose dwords divide by 4).  A thread's position inside its period -- which tile columns and which dword of the
//    tile row its four dwords come from -- is therefore loop-invariant; per chunk only the period index is
//    decomposed into env / tile row / pixel row, incrementally and with 24-bit multiplies.
//  * Tiles are read from LDS: the 102 agent-free tiles are staged once per workgroup (which then loops over groups
//    of EPW envs), the one agent tile of each env (cell kind x direction x highlight) once per env.
// Measured (profiles/r1_final/render_*.txt): the kernel runs at the speed of its own bare store loop; the write order
// (contiguous per workgroup vs. all workgroups sweeping adjacent frames) made no difference on MI355X.
// ======================================================================================================
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int RENDER_MAX_THREADS = 1024;          // 256 per workgroup while the LDS footprint lets >= 4 workgroups share a CU, else 1024
constexpr int STATIC_TILES = 2 * TILE_KEYS;          // [key][highlight]

struct RenderParams {
  const uint8_t* tilemap; const uint64_t* agent;
  const uint32_t* atlas_static;    // [key][hl][ts][ts*3/4] dwords
  const uint32_t* atlas_agent;     // [key][dir][hl][...]
  uint4* out;
  int N, Wt, Ht, cells, ts, full, epw, ngroups;
  int tile_dw, tdw_row, rowdw, R, cpp, ppe, t_active, pp;      // see above; ppe = periods per env, pp = periods per sweep
  int log2R; uint32_t magic_ts, magic_tdw;                     // R = 1 << log2R; magic_x = ceil(2^16 / x)
  int off_map;                                                 // LDS: [atlas dwords | u16 tile offsets per cell]
};

__global__ void __launch_bounds__(RENDER_MAX_THREADS) k_render(const RenderParams R) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* satlas = (uint32_t*)smem;
  uint16_t* smap = (uint16_t*)(smem + R.off_map);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = (int)blockDim.x;
  for (int i = tid; i < STATIC_TILES * R.tile_dw; i += nthreads) satlas[i] = R.atlas_static[i];

  // loop-invariant position of this thread's four dwords inside a period
  const bool worker = tid < R.t_active;
  const int cidx = tid % R.cpp, p0 = tid / R.cpp;
  int txj[4], srcj[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int dw = cidx * 4 + j, dr = dw / R.rowdw, col = dw - dr * R.rowdw;
    txj[j] = col / R.tdw_row;
    srcj[j] = dr * R.tdw_row + (col - txj[j] * R.tdw_row);
  }
  const int img_chunks = R.ppe * R.cpp;

  for (int g = blockIdx.x; g < R.ngroups; g += gridDim.x) {
    const int env0 = g * R.epw, nv = min(R.epw, R.N - env0);
    __syncthreads();                                                 // the previous group's blit is done with smap / the agent tiles
    for (int el = wave; el < nv; el += nthreads >> 6) {
      const int env = env0 + el;
      // the agent's cell: POV = bottom centre facing up (minigrid_env.py:659-663); full = its position and direction
      int cell = (R.Ht - 1) * R.Wt + (R.Wt >> 1), dir = 3;
      if (R.full) {
        const Agent a = agent_unpack(R.agent[env]);
        cell = (int)a.y * R.Wt + (int)a.x; dir = (int)a.dir;
      }
      const uint8_t* tm = R.tilemap + (size_t)env * R.cells;
      uint16_t* m = smap + el * R.cells;
      for (int k = lane; k < R.cells; k += 64) m[k] = (uint16_t)__umul24((uint32_t)tm[k], (uint32_t)R.tile_dw);
      MG_WAVE_LDS_SYNC();
      const uint32_t tb = (__umul24((uint32_t)m[cell], R.magic_tdw) >> 16);          // the tile byte under the agent
      const uint32_t* src = R.atlas_agent + (size_t)(((tb >> 1) * 4u + (uint32_t)dir) * 2u + (tb & 1u)) * R.tile_dw;
      uint32_t* dst = satlas + (STATIC_TILES + el) * R.tile_dw;
      for (int k = lane; k < R.tile_dw; k += 64) dst[k] = src[k];
      MG_WAVE_LDS_SYNC();
      if (lane == 0) m[cell] = (uint16_t)((STATIC_TILES + el) * R.tile_dw);
    }
    __syncthreads();
    if (worker) {
      // period p = p0, p0 + pp, ...: (env, period inside the env) advance by constant steps with one conditional
      // wrap; the rest is 24-bit multiplies of small numbers (full rate), no division
      u32x4* out = (u32x4*)R.out + (size_t)env0 * img_chunks + (uint32_t)(p0 * R.cpp + cidx);
      const uint32_t ostep = (uint32_t)(R.pp * R.cpp);
      const int total = nv * R.ppe, d_el = R.pp / R.ppe, d_pr = R.pp - d_el * R.ppe;
      int el = p0 / R.ppe, pr = p0 - el * R.ppe;
      int mb = el * R.cells;
      const int d_mb = d_el * R.cells;
#pragma unroll 2
      for (int p = p0; p < total; p += R.pp) {
        const uint32_t row0 = (uint32_t)pr << R.log2R;
        const uint32_t ty = __umul24(row0, R.magic_ts) >> 16;
        const uint32_t rowoff = __umul24(row0 - __umul24(ty, (uint32_t)R.ts), (uint32_t)R.tdw_row);
        const uint16_t* m = smap + mb + __umul24(ty, (uint32_t)R.Wt);
        u32x4 v;
        v.x = satlas[(uint32_t)m[txj[0]] + rowoff + srcj[0]];
        v.y = satlas[(uint32_t)m[txj[1]] + rowoff + srcj[1]];
        v.z = satlas[(uint32_t)m[txj[2]] + rowoff + srcj[2]];
        v.w = satlas[(uint32_t)m[txj[3]] + rowoff + srcj[3]];
        *out = v;
        out += ostep;
        pr += d_pr; mb += d_mb;
        if (pr >= R.ppe) { pr -= R.ppe; mb += R.cells; }
      }
    }
  }
}

// The same mosaic for ANY tile size (the reference's wrappers take tile_size as an argument, wrappers.py:299, 346): one thread per
// pixel, tiles read from the atlas in global memory.  The general fallback -- tile sizes 4 / 8 / 12 / 16 run k_render above.
struct RenderGenericParams {
  const uint8_t* tilemap; const uint64_t* agent; const uint8_t* atlas; uint8_t* out;
  int N, Wt, Ht, cells, ts, full;
};
__global__ void k_render_generic(const RenderGenericParams R) {
  const size_t ppe = (size_t)R.Ht * R.ts * R.Wt * R.ts;                  // pixels per env
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)R.N * ppe) return;
  const size_t env = i / ppe;
  const uint32_t p = (uint32_t)(i - env * ppe), roww = (uint32_t)(R.Wt * R.ts);
  const uint32_t py = p / roww, px = p - py * roww, ty = py / (uint32_t)R.ts, tx = px / (uint32_t)R.ts;
  const int cell = (int)(ty * (uint32_t)R.Wt + tx);
  int acell = (R.Ht - 1) * R.Wt + (R.Wt >> 1), dir = 3;              // POV: bottom centre, facing up (minigrid_env.py:659-663)
  if (R.full) {
    const Agent a = agent_unpack(R.agent[env]);
    acell = (int)a.y * R.Wt + (int)a.x; dir = (int)a.dir;
  }
  const uint32_t tm = R.tilemap[env * (size_t)R.cells + cell];
  const uint32_t tile = cell == acell ? (uint32_t)STATIC_TILES + (((tm >> 1) * 4u + (uint32_t)dir) * 2u + (tm & 1u)) : tm;
  const size_t tb = (size_t)R.ts * R.ts * 3;
  const uint8_t* src = R.atlas + tile * tb + ((size_t)(py - ty * (uint32_t)R.ts) * R.ts + (px - tx * (uint32_t)R.ts)) * 3;
  uint8_t* dst = R.out + i * 3;
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// ======================================================================================================
// State exchange on the device (mg_get_state / mg_set_state): Grid.encode() layout (N, W, H, 3) <-> the one-byte-per-cell
// row-major grids, and the (N, 8) i32 agent records <-> the packed u64 records.  One thread per (env, cell) / per env.
// ======================================================================================================
__global__ void k_state_encode(const uint8_t* grid, const uint64_t* agent, uint8_t* out_grid, int32_t* out_agent, int N, int W, int H, int CS) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cells = (size_t)W * H;
  if (i >= (size_t)N * cells) return;
  const size_t n = i / cells;
  const int k = (int)(i - n * cells), x = k / H, y = k - x * H;             // output order image[x][y]
  const uint32_t tri = cell_triple(grid[n * CS + (size_t)y * W + x]);
  uint8_t* p = out_grid + i * 3;
  p[0] = (uint8_t)tri; p[1] = (uint8_t)(tri >> 8); p[2] = (uint8_t)(tri >> 16);
  if (k == 0) {
    const Agent ag = agent_unpack(agent[n]);
    int32_t* o = out_agent + n * 8;
    o[0] = (int32_t)ag.x; o[1] = (int32_t)ag.y; o[2] = (int32_t)ag.dir;
    o[3] = ag.carry ? (int32_t)(cell_triple(ag.carry) & 0xFF) : 0;
    o[4] = ag.carry ? (int32_t)((cell_triple(ag.carry) >> 8) & 0xFF) : 0;
    o[5] = (int32_t)ag.step; o[6] = (int32_t)(ag.flags & FLAG_RESET_PENDING); o[7] = (int32_t)ag.mission;
  }
}
__global__ void k_state_decode(const uint8_t* in_grid, const int32_t* in_agent, uint8_t* grid, uint64_t* agent, uint32_t* bad, int N, int W, int H, int CS) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t cells = (size_t)W * H;
  if (i >= (size_t)N * cells) return;
  const size_t n = i / cells;
  const int k = (int)(i - n * cells), x = k / H, y = k - x * H;
  const uint8_t* p = in_grid + i * 3;
  grid[n * CS + (size_t)y * W + x] = (uint8_t)cell_from_triple(p[0], p[1], p[2]);
  if (k == 0) {
    const int32_t* o = in_agent + n * 8;
    if (o[0] < 0 || o[0] >= W || o[1] < 0 || o[1] >= H || (unsigned)o[2] > 3u || o[5] < 0 || o[5] > 65535 || (unsigned)o[7] > 16383u) { *bad = 1u; return; }
    Agent ag;
    ag.x = (uint32_t)o[0]; ag.y = (uint32_t)o[1]; ag.dir = (uint32_t)o[2];
    ag.carry = o[3] ? cell_from_triple((uint32_t)o[3], (uint32_t)o[4], 0) : 0u;
    if (ag.carry == CELL_EMPTY) ag.carry = 0;
    ag.step = (uint32_t)o[5]; ag.flags = o[6] ? FLAG_RESET_PENDING : 0u; ag.mission = (uint32_t)o[7];
    agent[n] = agent_pack(ag);
    for (int c = (int)cells; c < CS; c++) grid[n * CS + c] = 0;
  }
}
// the auxiliary word is not part of the exchanged state: it is re-derived from the injected grid.  mode 1: GoToInstr's tracked
// positions / target_pos = the cells holding the described object (desc from the mission id, see k_step); mode 2:
// DynamicObstacles' obstacle list, rebuilt in cell-index order
__global__ void k_aux_rebuild(const uint8_t* grid, const uint64_t* agent, uint64_t* aux, int N, int cells, int CS, int mode, int rule_div, int rule_cell) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const uint8_t* g = grid + (size_t)n * CS;
  uint64_t w = 0;
  if (mode == 1) {
    const uint32_t mis = agent_unpack(agent[n]).mission, m18 = mis % 18u;
    const uint32_t desc = rule_div == 0 ? (uint32_t)rule_cell
                        : rule_div == 1 ? make_cell(T_BALL, mis ? (uint32_t)C_BLUE : (uint32_t)C_RED)
                                        : make_cell(T_KEY + m18 % 3u, color_from_sorted(m18 / 3u));
    for (int c = 0; c < cells && c < 64; c++) if (g[c] == desc) w |= 1ull << c;
  } else if (mode == 3) {
    w = ~0ull;                                       // RULE_GOTO_BIG: no stale tracked position; PutNext: not used once an episode runs
  } else if (mode == 4) {
    // OpenDoor: the described doors.  A colour description follows from the mission id; a location description ("the door on
    // your left") was resolved against the agent's pose at reset and is not part of the exchanged state: the env's set is kept.
    const uint32_t mis = agent_unpack(agent[n]).mission;
    if (mis >= 6u) return;
    w = 1ull << color_from_sorted(mis);
  } else {
    int k = 0;
    for (int c = 0; c < cells && k < 8; c++) if (cell_type(g[c]) == T_BALL) w |= (uint64_t)c << (8 * k++);
  }
  aux[n] = w;
}

}  // namespace mg
