// mg_api.hip — C ABI of libminigrid_hip.so (see include/minigrid_hip.h).  Host side: buffers, launches, state
// exchange.  No CPU fallback exists: every entry point that computes needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cmath>
#include <cstddef>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/minigrid_hip.h"
#include "mg_kernels.h"
#include "mg_kernels_aux.h"
#include "mg_roll.h"
#include "mg_launch.h"
#include "mg_genlane.h"
#include "mg_knobs.h"
#include "mg_host.h"

#ifndef MG_GOTO_TU
#define MG_GOTO_TU 1          // (0: A/B builds -- the GoTo levels on k_roll7<GG_ROOMGRID> like the rest of their rule group, as before round 6)
#endif

using namespace mg;

static thread_local std::string g_create_error;

// Host waits poll the stream / event for a while before they block: a blocking hipStreamSynchronize sleeps on an interrupt and
// wakes up tens of microseconds late, which is most of a driver-sized run (one 20-step launch = 60 us) and of every Env.step().
static hipError_t wait_stream(hipStream_t st) {
  for (int i = 0; i < 20000; i++) {
    const hipError_t q = hipStreamQuery(st);
    if (q != hipErrorNotReady) return q;
  }
  return hipStreamSynchronize(st);
}
static hipError_t wait_event(hipEvent_t ev) {
  for (int i = 0; i < 20000; i++) {
    const hipError_t q = hipEventQuery(ev);
    if (q != hipErrorNotReady) return q;
  }
  return hipEventSynchronize(ev);
}

// Spare-episode ring and refill batches (see mg_kernels.h).  A BATCH is a run of consecutive step launches whose
// refill requests are served by ONE k_refill launch on the generator stream.  With R ring slots per env and at most
// cb = R/4 spares consumed per env and batch (a reset call consumes one; among n consecutive step calls at most
// ceil(n/2) do: a step that ends an episode sits between two consuming calls), a slot consumed in batch b is needed again
// in batch b+4 at the earliest, so the first launch of batch b waits (event) for the refill of batch b-4: the generator
// has three whole batches to finish, and never delays a step launch unless it falls that far behind.
constexpr int REFILL_LAG = 4;               // batch b waits for refill(b - REFILL_LAG)
constexpr int QSETS = REFILL_LAG + 1;       // request-segment sets / event pairs in rotation

struct mg_env {
  mg_config cfg;
  int device = 0;
  hipStream_t stream = nullptr, gen_stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_step[QSETS] = {}, ev_gen[QSETS] = {};
  hipEvent_t ev_live = nullptr, ev_fill = nullptr;   // reset(seed=...): the ring is redrawn on the generator stream
  bool fill_pending = false, fill_no_wait = false;
  // geometry
  int N = 0, W = 0, H = 0, cells = 0, CS = 0, GS = 0, obs_bytes = 0;
  int map_bytes = 0;          // bytes per env k_step writes: obs_bytes, or the tile map k_render expands (RGB modes)
  int off_grid = 0, off_shadow = 0, off_spr = 0, off_act = 0, off_trow = 0, off_T = 0, lds_bytes = 0;
  int lpe = 1, epw = 64;      // lanes per env in k_step (1 or 4), envs per wavefront = 64 / lpe
  bool fast7 = false;         // the default 7x7 partial view: k_roll7 (mg_roll.h) instead of k_step
  bool fast_full = false;     // FullyObs on grids whose two images fit the LDS: k_roll7<., true>
  int roll_nw = 1;            // wavefronts per 64-env workgroup in fused k_roll7 launches (1, 2 or 4: time split)
  bool lane_gen = false;      // the refills run one lane per episode (k_refill_lane: the single-room levels; MG_LANE_GEN=0: the wave-per-episode k_refill)
  bool lane_direct = false;   // direct generation (reset(seed): the live episode and the ring fill) runs one lane per env (k_generate_lane): every level with a lane generator
  bool dyn_inloop = false;    // DynamicObstacles, default 7x7 view: k_roll7<GG_DYNOBS> draws the level's moves and resets inside the step loop (mg_dynobs.h; MG_DYN_INLOOP=0: the round-3 launches)
  bool full_split = true;     // FullyObs (k_roll7<., true>): the dynamics wave + encode waves over staged copies of its image-order stream (MG_FULL_SPLIT=0: the two-wave time split)
  bool roll_split_on = true;  // MG_ROLL_SPLIT (read when the observation configuration is made): 0 = the round-3 time split at every width
  int roll_shadows = 1;       // spare episodes per env staged in LDS by a fused k_roll7 launch (2 unless the level draws nothing)
  int roll_guard = 0;
  int nwaves = 0;             // k_step workgroups (one wavefront of epw envs each) = refill request segments
  bool static_gen = false;
  bool live_gen = false;      // DynamicObstacles: step() consumes the stream => resets are drawn right before the step launch
  int rule = RULE_NONE, rule_cell = 0, rule_div = 1;
  int rule_group = GG_NONE;   // which level rules the k_step variant carries (mg_gen.h groups)
  // spare ring
  int R = 1, cb = 1;          // ring slots per env; spares one env may consume per batch
  int seg_cap = 64;
  uint32_t batch_id = 0; int batch_obs = 0, batch_steps = 0; bool batch_open = false;
  int max_fused = 1;          // steps per fused launch
  // device buffers
  uint8_t *grid = nullptr, *spare_grid = nullptr;
  bool sentence = false;       // BabyAI levels with an instruction tree / object identity: instruction records + k_verify
  uint64_t *instr = nullptr, *spare_instr = nullptr; uint32_t *gstate = nullptr, *gsnap = nullptr; size_t off_sentence = 0;
  uint64_t *agent = nullptr, *spare_agent = nullptr, *rng = nullptr, *rng_snap = nullptr, *rng_tmp = nullptr, *seeds = nullptr;
  uint32_t *head = nullptr, *tail = nullptr, *claim = nullptr, *seg = nullptr, *seg_count = nullptr;
  uint32_t* seg_off = nullptr;    // packed lane refill: nwaves + 1 prefix sums behind the segment counts (the same allocation)
  bool lane_packed = false;       // the level's refill numbers its requests across the segments and fills whole wavefronts (mg_genlane.h)
  int lane_lpw = 64;              // ... with this many busy lanes each
  long long lane_burst_min = 0;   // burst hybrid of the wavefront-per-episode levels: batches of at least this many requests refill on packed lanes (0 = off)
  uint8_t *mask = nullptr, *actions = nullptr;
  uint64_t *aux = nullptr, *spare_aux = nullptr;   // auxiliary word per env: DynamicObstacles obstacle list / GoTo targets
  bool goto_kind = false;
  uint8_t* tilemap = nullptr; uint32_t* atlas = nullptr;   // RGB modes: k_step's output / the tile atlas (mg_tiles.h)
  uint8_t* st_grid = nullptr; int32_t* st_agent = nullptr;   // state-exchange staging (mg_get_state / mg_set_state), on first use
  uint32_t* claim_bad() { return err + 4; }                // last word of the error buffer: mg_set_state's validation flag
  volatile uint32_t* err_host = nullptr;                   // the error words live in mapped pinned host memory: the kernels store into
                                                           // them (only when something is wrong), the host reads them after a stream sync
                                                           // without a device-to-host copy (a 4-byte copy costs ~10 us per mg_sync)
  RenderParams render;        // ... and k_render's launch geometry
  bool render_generic = false; RenderGenericParams render_g;   // any other tile size: k_render_generic
  int render_lds = 0, render_blocks = 0, render_threads = 256;
  bool rgb = false;
  // trajectory ring: S slots of { obs | reward | terminated | truncated | direction | mission | action }
  uint8_t* out = nullptr;
  mg_step_scalars* h_scal = nullptr;   // pinned host staging of one slot's scalars (mg_copy_slot), on first use
  int S = 1;
  size_t slot_bytes = 0, record_bytes = 0, off_reward = 0, off_term = 0, off_trunc = 0, off_dir = 0, off_mission = 0, off_action = 0;
  uint32_t* err = nullptr;
  unsigned long long* counters = nullptr;
  size_t ncounters = 0;
  Knobs k;                                       // the environment's A/B switches and debugging aids, read once at mg_create (mg_knobs.h)
  bool staged_big = false;                       // k_roll7's STAGED instantiation: big grids, one copy of the grids per workgroup
  int k_epw = 64;                                // envs per k_roll7 workgroup of the 7x7 view (64 | 32): decides the LDS carve-up (roll_layout)
  bool mask_sparse = false;                      // the reset in progress is masked and resets fewer than an eighth of the envs
  uint64_t env_steps = 0;     // env-steps executed (host-side count: N per step)
  unsigned long long burst_bytes = 0;   // output bytes of the launches enqueued since the step stream was last known idle (launch_step: nontemporal stores)
  uint64_t stat_base[3] = { 0, 0, 0 };   // episodes / maps / retries counted before the last mg_set_obs_config (its statistics slots are re-made)
  uint32_t launches = 0;      // k_step launches so far
  uint32_t t = 0;             // rollout step counter (Philox action counter)
  // every device buffer of the handle: freed in mg_destroy; with MG_GUARD=1 each one sits between two pattern-filled red zones
  struct Alloc { void* user; void* base; size_t bytes; const char* name; };
  std::vector<Alloc> allocs;
  std::string last_error;
};

static int fail(mg_env* env, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  if (env) env->last_error = buf; else g_create_error = buf;
  return code;
}
#define HIP_TRY(env, call)                                                                         \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess) return fail(env, MG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); \
  } while (0)

static GenParams gen_params(const mg_env* e) {
  GenParams g = gen_params_of(e->cfg);
  g.W = e->W; g.H = e->H;
  return g;
}

// ---- launches -------------------------------------------------------------------------------------------
// GenArgs common to the direct generator launches (slot: ring slot to fill, < 0 = the live state) and the refill launches
static GenArgs gen_args(mg_env* e, int slot) {
  GenArgs A;
  const bool to_spare = slot >= 0;
  const size_t N = (size_t)e->N, s = to_spare ? (size_t)slot : 0;
  A.gp = gen_params(e);
  A.dst_grid = to_spare ? e->spare_grid + s * N * e->CS : e->grid;
  A.dst_agent = to_spare ? e->spare_agent + s * N : e->agent;
  A.rng = e->rng;
  A.rng_snap = to_spare ? e->rng_snap + s * 5 * N : nullptr;
  A.dst_aux = (e->live_gen && !to_spare) ? e->aux : (e->goto_kind ? (to_spare ? e->spare_aux + s * N : e->aux) : nullptr);
  A.dst_instr = e->sentence ? (to_spare ? e->spare_instr + s * N * INSTR_WORDS : e->instr) : nullptr;
  A.gstate = e->sentence ? e->gstate : nullptr; A.gsnap = (e->sentence && to_spare) ? e->gsnap + s * N : nullptr;
  A.mask = nullptr;
  A.err = e->err; A.counters = e->counters;
  A.N = e->N; A.CS = e->CS; A.stat_gen_off = STAT_EPISODES + e->nwaves;
  // (round 6: 1024 like the refill's -- the 8 KB buffer of 2048 words held k_generate at four waves per SIMD: LDS, not registers, bounded its occupancy)
  A.cap_words = e->sentence ? 4864 : 1024;   // LevelGen: up to 100 tries per description (levelgen.py:113-155); 38 PCG refills (GEN_SBASE_ENTRIES)
  A.live = 0;
  // LevelGen, num_crossings bit 10: an episode whose drawing met RoomGrid.place_agent's endless loop is redrawn and accepted
  A.stuck_mode = (e->cfg.env_kind == MG_ENV_LEVELGEN && ((e->cfg.num_crossings >> 10) & 1)) ? 2 : (to_spare ? 0 : 1);
  A.seg = nullptr; A.seg_count = nullptr; A.seg_cap = e->seg_cap; A.wps = 1;
  A.seg_off = nullptr; A.nseg = 0; A.lpw = 64; A.burst_min = 0u; A.slot_cap = 0u;
  A.head = e->head; A.tail = e->tail; A.claim = e->claim; A.epoch = 0; A.ring_mask = (uint32_t)(e->R - 1);
  return A;
}

// Direct generator launch over all envs (optionally masked), on the main stream.
static int launch_generate(mg_env* e, int slot, const uint8_t* d_mask, hipStream_t st = nullptr) {
  if (!st) st = e->stream;
  GenArgs A = gen_args(e, slot);
  A.mask = d_mask;
  const int wpb = GEN_THREADS / 64;
  const int blocks = std::min((e->N + wpb - 1) / wpb, 8192);
  const size_t lds = (size_t)wpb * gen_wave_lds_bytes(e->CS, A.cap_words, e->sentence);
  // one translation unit per generator group (mg_gen.h): a level's generator kernel carries only its group's generators
  const int gg = gen_group_of_kind(e->cfg.env_kind);
  const bool philox = e->cfg.rng_mode == MG_RNG_PHILOX;
  // (a SPARSE masked reset of a level whose refill stays on cooperative wavefronts -- the mazes, MultiRoom, the sentence levels -- keeps k_generate: a lane
  // wave with one or two busy lanes is a very slow scalar core, ~5 ms per call against ~0.8; a seeded reset issues R + 1 such launches.  ADVICE r5)
  if (e->lane_direct && !(d_mask && e->mask_sparse && !e->lane_gen)) {             // one lane per env (mg_genlane.h)
    if (!launch_generate_lane(philox, dim3((e->N + 63) / 64), (size_t)lane_gen_lds_bytes(e->CS, e->sentence), st, A))
      return fail(e, MG_ERR_INVALID, "internal: no lane generator kernel for env_kind %d", e->cfg.env_kind);
  }
  else
  MG_GEN_DISPATCH(launch_generate_, gg, philox, dim3(blocks), lds, st, A);
  HIP_TRY(e, hipGetLastError());
  return MG_OK;
}

// Refill launch for request-segment set `set`: one workgroup per step-wave segment.
static int launch_refill(mg_env* e, int set, uint32_t epoch, bool live, hipStream_t st) {
  GenArgs A = live ? gen_args(e, -1) : gen_args(e, 0);
  // the ring slot is chosen per request on the device: pass slot 0's pointers, generate_one offsets them
  A.live = live ? 1 : 0;
  A.seg = e->seg + (size_t)set * e->nwaves * e->seg_cap;
  A.seg_count = e->seg_count + (size_t)set * e->nwaves;
  A.epoch = epoch;
  A.cap_words = e->sentence ? 4864 : 1024;    // a pass that runs out of buffered draws restarts from its checkpoint / doubles
  A.wps = std::max(1, e->epw / 4);            // a GoToRedBall batch files ~EPW/7 requests per segment (Poisson: some segments twice that)
  // The generator stream outranks the step stream, and every generating wavefront is a workgroup of its own: on the big-grid levels -- long episodes,
  // few requests per segment, a step kernel that needs every SIMD -- sixteen of them per segment mostly start, find nothing to do and still
  // delay the step kernel's workgroups.  Measured (profiles/r4/bosslevel_generator2.txt, refill_wps_other_levels.txt): BossLevel x 131 072 50.0 us
  // per step with 16, 33.4 with 2 (28.7 with the deeper ring below); BabyAI-GoTo x 131 072 122.8 with 16, 76.1 with 4; MultiRoom-N6 equal;
  // KeyCorridorS3R3 (7 x 7 cells, short episodes: it needs the generator's throughput) 47.6 with 16, 52.5 with 4 -- hence by grid size.
  if (e->sentence) A.wps = 2;
  else if (e->cells > 256) A.wps = 4;
  const size_t lds = (size_t)gen_wave_lds_bytes(e->CS, A.cap_words, e->sentence);
  const int gg = gen_group_of_kind(e->cfg.env_kind);
  const bool philox = e->cfg.rng_mode == MG_RNG_PHILOX;
  const dim3 rgrid(e->nwaves * A.wps);
  // (lane refills: GenArgs::slot_cap -- a request draws a quarter of its env's free slots, at least two; MG_LANE_CAP: the minimum, 0 = every free slot as
  // before.  GoToRedBall x 32 768: 2.76 us per step without, 2.39-2.55 with a fixed cap of 2-4: profiles/r6/ab_lane_cap.txt)
  if (!live && (e->lane_gen || e->lane_burst_min > 0) && e->R >= 16) A.slot_cap = e->k.lane_cap >= 0 ? (uint32_t)e->k.lane_cap : 2u;
  if (!live && e->lane_gen) {
    // one LANE per episode (mg_genlane.h): a wavefront per request segment, each lane drawing its own request's episodes
    bool ok;
    if (e->lane_packed) {
      // the batch's requests numbered across the segments (k_seg_scan), lane_lpw of them per wavefront; a grid-stride loop covers any count
      A.seg_off = e->seg_off; A.nseg = e->nwaves; A.lpw = e->lane_lpw;
      const unsigned blocks = (unsigned)std::min<long long>(((long long)e->N + A.lpw - 1) / A.lpw, 16384);
      ok = launch_refill_lane_packed(philox, dim3(blocks), (size_t)lane_gen_lds_bytes(e->CS, e->sentence), st, A);
    } else {
      A.wps = 2;                               // (round 6, profiles/r6/ab_lane_wps.txt: GoToRedBall x 32 768 2.7 us per step with 2, 3.2 with 3, 3.6 with 4, 4.3 with 1 -- what a refill costs is wavefronts x longest lane chain)
      ok = launch_refill_lane(philox, dim3(e->nwaves * A.wps), (size_t)lane_gen_lds_bytes(e->CS, e->sentence), st, A);
    }
    if (!ok) return fail(e, MG_ERR_INVALID, "internal: no lane refill kernel for env_kind %d (packed %d)", e->cfg.env_kind, (int)e->lane_packed);
  } else {
    if (!live && e->lane_burst_min > 0) {
      // burst hybrid: the packed lane refill takes a batch of >= lane_burst_min requests (every env of a long-episode level truncating at once),
      // k_refill everything smaller; both are launched, the request count (k_seg_scan) decides on the device
      A.seg_off = e->seg_off; A.nseg = e->nwaves; A.lpw = e->lane_lpw; A.burst_min = (uint32_t)e->lane_burst_min;
      const unsigned blocks = (unsigned)std::min<long long>(((long long)e->N + A.lpw - 1) / A.lpw, 16384);
      if (!launch_refill_lane_packed(philox, dim3(blocks), (size_t)lane_gen_lds_bytes(e->CS, e->sentence), st, A))
        return fail(e, MG_ERR_INVALID, "internal: no packed lane refill kernel for env_kind %d", e->cfg.env_kind);
    }
    MG_GEN_DISPATCH(launch_refill_, gg, philox, rgrid, lds, st, A);
  }
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemsetAsync(A.seg_count, 0, (size_t)e->nwaves * sizeof(uint32_t), st));
  return MG_OK;
}

// the ring redraw a seeded reset left running on the generator stream: whatever consumes spares or reads stream positions waits for it
static int await_ring_fill(mg_env* e) {
  if (!e->fill_pending) return MG_OK;
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_fill, 0));
  e->fill_pending = false;
  return MG_OK;
}

static bool uses_ring(const mg_env* e) { return !e->static_gen && !e->live_gen; }

// close the open batch: its refill runs on the generator stream as soon as the batch's last step launch has finished
static int close_batch(mg_env* e) {
  if (!uses_ring(e) || !e->batch_open) return MG_OK;
  const int set = (int)(e->batch_id % QSETS);
  HIP_TRY(e, hipEventRecord(e->ev_step[set], e->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->gen_stream, e->ev_step[set], 0));
  int rc = launch_refill(e, set, e->batch_id + 1u, false, e->gen_stream);
  if (rc) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_gen[set], e->gen_stream));
  e->batch_id++; e->batch_open = false; e->batch_obs = 0; e->batch_steps = 0;
  return MG_OK;
}

// account a launch (an OBSERVE launch, or T step calls) to the open batch, closing / opening batches as needed
static int batch_admit(mg_env* e, int phase, int T) {
  if (!uses_ring(e)) return MG_OK;
  // (a seeded reset's own OBSERVE launch takes no spare -- the live episode was drawn in place -- and does not wait: fill_no_wait)
  if (!e->fill_no_wait) { int rc = await_ring_fill(e); if (rc) return rc; }
  const int add_obs = phase == PHASE_OBSERVE ? 1 : 0, add_steps = phase == PHASE_STEP ? T : 0;
  // (SAME_STEP autoreset: every step can take a spare, not every other one)
  const int div = e->cfg.autoreset_mode == MG_AUTORESET_SAME_STEP ? 1 : 2;
  if (e->batch_open && (e->batch_obs + add_obs) + (e->batch_steps + add_steps + div - 1) / div > e->cb) {
    int rc = close_batch(e);
    if (rc) return rc;
  }
  if (!e->batch_open) {
    if (e->batch_id >= (uint32_t)REFILL_LAG)
      HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_gen[(e->batch_id - REFILL_LAG) % QSETS], 0));
    e->batch_open = true;
  }
  e->batch_obs += add_obs; e->batch_steps += add_steps;
  return MG_OK;
}

// Entry points that read or overwrite spares / stream positions first bring every ring up to date: all requests
// served (tail = head + R for every env), and the main stream ordered after the generator stream.
static int flush_refills(mg_env* e) {
  { int rc = await_ring_fill(e); if (rc) return rc; }
  if (!uses_ring(e)) return MG_OK;
  int rc = close_batch(e);
  if (rc) return rc;
  if (e->batch_id > 0) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_gen[(e->batch_id - 1) % QSETS], 0));
  return MG_OK;
}

// LDS carve-up of a k_roll7 workgroup with nw wavefronts (mg_roll.h): table | guard | nw private grid copies | guard | nw code
// stagings | shadow grids | shadow agent / aux words | caller-supplied actions
// code stagings between the dynamics wave and the encode waves of the DynamicObstacles / sentence-level split (mg_roll.h): four, or two where the
// 22 x 22 grids of the sentence levels leave no more (38.75 KB per workgroup = four workgroups per CU)
static int roll_dring(const mg_env* e) {
  return e->k.dring ? e->k.dring : (e->sentence || e->fast_full || e->staged_big) ? 2 : ROLL_DSPLIT_RING;
}
struct RollLayout { int off_grid, off_codes, codes_stride, off_shadow, shadow_stride, off_shadow_gt, off_spr, off_act, off_log, off_tmpl, off_instr, total; };
// split: wave 0 = the dynamics wave (no code staging of its own), + the step log ring (mg_roll.h)
static RollLayout roll_layout(const mg_env* e, int nw, bool with_actions, bool split = false) {
  RollLayout L;
  // (DynamicObstacles in the loop, split: ONE copy of the grids -- the dynamics wave's, which stages the codes itself -- and a ring of stagings)
  const bool dsplit = split && (e->dyn_inloop || (e->sentence && e->fast7) || e->fast_full || e->staged_big);
  L.off_grid = 1024 + e->roll_guard;
  const int epw = (e->fast7 && !e->fast_full) ? e->k_epw : 64;            // envs per workgroup (the 7x7 view: 64, or 32 -- mg_env::k_epw)
  L.off_codes = (L.off_grid + (dsplit ? 1 : nw) * epw * e->GS + e->roll_guard + 15) & ~15;
  const int ncodes = dsplit ? roll_dring(e) + (e->fast_full ? 1 : 0) : split ? nw - 1 : nw;   // (FullyObs: the dynamics wave's own image-order stream + the ring of staged copies)
  // per wave: the 7x7 view's code staging, or (FullyObs) the image-order stream of its 64 grids
  L.codes_stride = e->fast_full ? ((64 * e->cells + 16 + 15) & ~15) : ROLL_CODES_BYTES;
  // the shadow sets (the next one or two spare episodes of every env): grids, (FullyObs) their image streams, agent / aux words
  const int K = e->dyn_inloop ? 0 : e->roll_shadows;                   // (DynamicObstacles in the loop: no spare ring, nothing to stage)
  L.shadow_stride = (epw * e->GS + 15) & ~15;
  L.off_shadow = L.off_codes + ncodes * L.codes_stride;
  L.off_shadow_gt = L.off_shadow + K * L.shadow_stride;
  L.off_spr = L.off_shadow_gt + (e->fast_full ? K * L.codes_stride : 0);
  L.off_act = L.off_spr + K * 64 * 16;
  L.off_log = L.off_act + (with_actions ? MAX_FUSED_STEPS * 64 : 0);
  L.off_tmpl = L.off_log + (dsplit ? ROLL_LOG_SYNC_BYTES : split ? ROLL_LOG_BYTES : 0);   // k_roll7<GG_DYNOBS>: the level's constant grid
  L.off_instr = (L.off_tmpl + (e->dyn_inloop ? e->CS : 0) + 15) & ~15;          // k_roll7<GG_SENTENCE>: the workgroup's instruction records
  L.total = L.off_instr + ((e->sentence && e->fast7 && MG_INSTR_LDS) ? 64 * ROLL_INSTR_STRIDE * 8 : 0);
  return L;
}
static int roll_lds_bytes(const mg_env* e, int nw, bool with_actions, bool split = false) { return roll_layout(e, nw, with_actions, split).total; }
// Fused launches of the 7x7 view with three or four waves per workgroup run SPLIT (one dynamics wave + encode waves, mg_roll.h) instead of the
// time split: the dynamics of a step run once instead of once per wave that has not reached it yet.  With two waves the time split wins
// (one encode wave would carry every observation alone); FullyObs and the sentence levels keep their round-3 shapes.  MG_ROLL_SPLIT=0: A/B.
static bool roll_split_ok(const mg_env* e, int nw) {
  if (e->fast_full) return e->roll_split_on && e->full_split && nw >= 2;
  return e->roll_split_on && e->fast7 && nw >= ((e->dyn_inloop || e->sentence || e->staged_big) ? 2 : 3);
}

static void fill_step_params(mg_env* e, StepParams& P, int phase) {
  P.grid = e->grid; P.agent = e->agent; P.aux = e->aux;
  P.spare_grid = e->spare_grid; P.spare_agent = e->spare_agent; P.spare_aux = e->spare_aux;
  P.head = uses_ring(e) ? e->head : nullptr; P.ring_mask = (uint32_t)(e->R - 1);
  const int set = e->live_gen ? 0 : (int)(e->batch_id % QSETS);
  P.seg = e->static_gen ? nullptr : e->seg + (size_t)set * e->nwaves * e->seg_cap;
  P.seg_count = e->static_gen ? nullptr : e->seg_count + (size_t)set * e->nwaves;
  P.seg_cap = e->seg_cap;
  P.actions = e->actions; P.act_dtype = MG_ACT_U8; P.act_src = ACT_SRC_BUFFER; P.action_seed = 0; P.t0 = 0; P.staged = 0;
  P.obs_mask = nullptr;
  P.instr = e->instr; P.spare_instr = e->spare_instr; P.off_sentence = e->off_sentence;
  P.out = e->out; P.slot_bytes = e->slot_bytes;
  P.obs = e->rgb ? e->tilemap : e->out; P.obs_stride = e->rgb ? 0ull : (unsigned long long)e->slot_bytes;
  P.epw = e->epw;
  {
    // use_done_actions applies to the levels built on RoomGridLevel (instrs.verify at the end of their step, roomgrid_level.py:87-104)
    const int k = e->cfg.env_kind;
    const bool babyai = k == MG_ENV_GOTO_REDBALL || (k >= MG_ENV_GOTO_REDBALLGREY && k <= MG_ENV_GOTO_LOCAL) || (k >= MG_ENV_PICKUPDIST && k <= MG_ENV_BABYAI_KEYCORRIDOR) ||
                        (k >= MG_ENV_BABYAI_GOTO && k <= MG_ENV_LEVELGEN);
    P.done_actions = (e->cfg.babyai_done_actions != 0 && babyai) ? (e->cfg.babyai_done_actions == 2 ? 2 : 1) : 0;      // (2: AndInstr's enum-identity branch, mg_verify.h)
  }
  P.obs_wg_stride = (unsigned long long)e->epw * (unsigned long long)e->map_bytes;
  P.rng = e->rng; P.dyn_n = std::min(e->cfg.num_dists, 8); P.dyn_sx = e->cfg.agent_start_x; P.dyn_sy = e->cfg.agent_start_y; P.dyn_sdir = e->cfg.agent_start_dir;
  P.off_tmpl = 0; P.off_instr = 0; P.dring = ROLL_DSPLIT_RING; P.stat_gen_off = STAT_EPISODES + e->nwaves;
  P.off_reward = e->off_reward; P.off_term = e->off_term; P.off_trunc = e->off_trunc; P.off_dir = e->off_dir;
  P.off_mission = e->off_mission; P.off_action = e->off_action;
  P.T = 1; P.slot0 = 0; P.S = e->S;
  P.err = e->err; P.counters = e->counters;
  P.N = e->N; P.W = e->W; P.H = e->H; P.CS = e->CS; P.GS = e->GS; P.cells = e->cells; P.max_steps = e->sentence ? 65535 : e->cfg.max_steps;   // sentence levels: per-episode limit, applied by k_verify
  P.see_through = e->cfg.see_through_walls; P.rule = e->rule; P.rule_cell = e->rule_cell; P.rule_div = e->rule_div;
  P.autoreset_next_step = e->cfg.autoreset_mode == MG_AUTORESET_NEXT_STEP;
  P.autoreset_same_step = e->cfg.autoreset_mode == MG_AUTORESET_SAME_STEP;
  P.share = 0; P.split_mode = 0; P.off_log = 0; P.nt = 0; P.codes_stride = ROLL_CODES_BYTES; P.off_shadow_gt = 0; P.w_magic = 0; P.h_magic = 0; P.shadow_stride = 0; P.spr_stride = 0;
  P.phase = phase; P.static_gen = e->static_gen; P.live_gen = e->live_gen ? 1 : 0; P.use_shadow = 0;
  P.off_grid = e->off_grid; P.off_shadow = e->off_shadow; P.off_spr = e->off_spr; P.off_act = e->off_act; P.off_trow = e->off_trow;
  P.off_T = e->off_T; P.OBE = e->map_bytes;
  P.rgb_full = e->cfg.obs_mode == MG_OBS_RGB; P.rgb_highlight = e->cfg.rgb_highlight != 0;
  P.view = e->cfg.agent_view_size; P.no_death_mask = e->cfg.no_death_mask; P.death_cost = e->cfg.death_cost;
  const uint32_t cpe = (uint32_t)(e->CS >> 4);
  P.cpe_magic = ((1u << 20) + cpe - 1) / cpe;
  P.env_base = e->cfg.env_index_base;
#if defined(MG_ATTRIBUTION)
  P.exp = e->k.exp;   // attribution builds only (mg_roll.h MG_EXPBIT)
#else
  P.exp = 0;
#endif
}

static int launch_step(mg_env* e, StepParams& P) {
  // grid = one 64-lane workgroup (one autonomous wavefront) per 64 consecutive envs; P.T steps per launch
  if (e->live_gen) {
    // (1) draw, in place, the episodes of the envs the previous launch left RESET_PENDING (they come out FRESH and
    //     are only observed by this launch); (2) before a real step, move the obstacles of everyone else
    // (running (1) on the generator stream beside (2) was measured in round 3 -- profiles/r3/dynobs_overlap.txt, 16x16 x 65 536: 72.5 us per step side by
    // side, 71.5 one after the other: both kernels are bound by the SIMDs' issue rate -- and the switch was deleted in round 6: one stream)
    // DynamicObstacles in the loop: a STEP launch redraws the envs waiting for their autoreset itself, at its first step (mg_dynobs.h), and
    // REPLACES the request list with the envs its last step left waiting (P.live_gen = 2) -- no redraw launch, no counter reset between step
    // launches; an OBSERVE launch (reset) still finds exactly the waiting envs listed
    const bool inloop_step = e->dyn_inloop && P.phase == PHASE_STEP;
    if (inloop_step) P.live_gen = 2;
    if (e->launches > 0 && !inloop_step) {
      int rc = launch_refill(e, 0, e->launches + 1u, true, e->stream);
      if (rc) return rc;
    }
    if (P.phase == PHASE_STEP && !e->dyn_inloop) {       // (in the loop: k_roll7<GG_DYNOBS> moves the obstacles itself)
      const int epb = MOVE_EPB;
      const int nb = (e->N + epb - 1) / epb;
      const size_t mlds = (size_t)epb * (size_t)(e->CS + 4);          // the wave's staged grids (DynamicObstacles: at most 16 x 16)
      if (e->cfg.rng_mode == MG_RNG_PHILOX)
        hipLaunchKernelGGL(k_move_obstacles<PhiloxStream>, dim3(nb), dim3(64), mlds, e->stream, e->grid, e->agent, e->rng, e->aux,
                           e->N, e->W, e->H, e->CS, e->cfg.num_dists, epb);
      else
        hipLaunchKernelGGL(k_move_obstacles<Pcg64Stream>, dim3(nb), dim3(64), mlds, e->stream, e->grid, e->agent, e->rng, e->aux,
                           e->N, e->W, e->H, e->CS, e->cfg.num_dists, epb);
      HIP_TRY(e, hipGetLastError());
    }
  }
  { int rc = batch_admit(e, P.phase, P.T); if (rc) return rc; }
  {
    // the batch this launch files its refill requests under (batch_admit may have moved on to the next one)
    const int set = e->live_gen ? 0 : (int)(e->batch_id % QSETS);
    if (!e->static_gen) { P.seg = e->seg + (size_t)set * e->nwaves * e->seg_cap; P.seg_count = e->seg_count + (size_t)set * e->nwaves; }
  }
  // fused launches stage every env's next spare episode in its LDS shadow slot at launch start
  P.use_shadow = (P.T > 1 && !e->live_gen) ? 1 : 0;
  const size_t lds = (size_t)(P.act_src == ACT_SRC_BUFFER && P.phase == PHASE_STEP ? e->lds_bytes : e->off_act);
  dim3 grid(e->nwaves);
  const int mode = e->cfg.obs_mode == MG_OBS_FULL ? 1 : e->cfg.obs_mode == MG_OBS_SYMBOLIC ? 3 : e->cfg.obs_mode == MG_OBS_ONEHOT ? 2
                 : (e->cfg.obs_mode == MG_OBS_RGB || e->cfg.obs_mode == MG_OBS_RGB_PARTIAL) ? 4 : 0;
  const int gg = e->rule_group;
  bool launched = false;
  if (e->fast7 || e->fast_full) {
    // wave w of a workgroup produces steps [split[w], split[w + 1]) after replaying the steps before them silently: the split that
    // equalises the waves' work for a silent step costing `ratio` of a full one (x_{w+1} = x_w (1 - ratio) + x_1)
    int nw = std::min(e->roll_nw, std::max(1, P.T));
    const bool in_loop_verify = e->sentence && e->fast7;       // k_roll7<GG_SENTENCE>: one wave per workgroup (the record is shared state)
    // (the split of the sentence levels: the stepping / verifying wave + ONE encode wave over one copy of the grids, mg_roll.h)
    if (in_loop_verify) nw = (P.T > 1 && roll_split_ok(e, 2)) ? 2 : 1;
    // one-step launches (Env.step): four waves share the encode of the one step (k_roll7 `share`); one private grid copy
    // (P.share: 1..15 = the stepping wave is (workgroup >> (value - 1)) & 3, 16 = always wave 0 -- what was measured best, profiles/r3/unfused_share.txt)
    const int share_mode = 16;
    const bool share_ok = true;
    // (only while the batch leaves wave slots free: at 4 096 workgroups the three waiting waves per workgroup cost more than the shared
    // encode saves -- Empty-8x8 x 65 536: 9.4 us per step against 10.1; DoorKey-8x8 x 262 144: 33.0 against 23.8, profiles/r3/unfused_share.txt)
    const bool share = P.T == 1 && share_ok && P.phase == PHASE_STEP && e->nwaves <= 2048;
    P.share = share ? share_mode : 0;
    const double ratio = 0.12;                                 // (a silent replayed step costs ~0.12 of a produced one: profiles/r3/sweep_nw_ratio_quads.txt)
    double geo = 0.0, pw = 1.0;
    for (int w = 0; w < nw; w++) { geo += pw; pw *= 1.0 - ratio; }
    const double x1 = (double)P.T / geo;
    double x = 0.0;
    P.split[0] = 0;
    for (int w = 1; w < nw; w++) { x = x * (1.0 - ratio) + x1; P.split[w] = std::min(P.T - (nw - w), std::max(P.split[w - 1] + 1, (int)std::lround(x))); }
    for (int w = nw; w <= ROLL_MAX_WAVES; w++) P.split[w] = P.T;
    const bool split = !share && P.T > 1 && roll_split_ok(e, nw);
    const bool dsplit_cfg = e->dyn_inloop || (e->sentence && e->fast7) || e->fast_full || e->staged_big;
    P.staged = (split && e->staged_big) ? 1 : 0;
    (void)dsplit_cfg;
    const bool acts = P.act_src == ACT_SRC_BUFFER && P.phase == PHASE_STEP;
    const RollLayout L = roll_layout(e, nw, acts, split);
    // split_mode - 1 = the shift that picks the dynamics wave: wave (workgroup >> shift) % nw (9 = shift 8: profiles/r4/split_rotation.txt; 31 = always wave 0)
    const int drot = 9;
    // (DynamicObstacles in the loop: always wave 0 -- three waves per workgroup rotate over a CU's four SIMDs by themselves; 12.5 us per step against
    // 15.3 with the rotation, profiles/r4/dynobs_waves_sweep2.txt)
    P.split_mode = split ? ((e->dyn_inloop || e->sentence || e->fast_full || e->staged_big) ? 31 : drot) : 0; P.dring = roll_dring(e); P.off_log = L.off_log; P.off_tmpl = L.off_tmpl; P.off_instr = L.off_instr;
    {
      // Nontemporal observation stores once the launches enqueued since the stream was last known idle have written more than the write-back
      // caches hold (256 MB of Infinity Cache): a long rollout streams to HBM and leaves L2 to the grids and spare episodes it re-reads
      // (DoorKey-8x8 x 262 144: 10.9 -> 8.8 us per step); a short burst, or a one-step launch whose observation the consumer reads next, is
      // better off absorbed by the caches (one 20-step launch: 2.15 us per step plain, 2.39 nontemporal).  MG_NT_BYTES: the threshold in MB
      // (0 = always nontemporal, negative = never).
      const long long nt_mb = e->k.nt_mb;
      const unsigned long long wr = (unsigned long long)e->N * (unsigned long long)P.T * (unsigned long long)(e->obs_bytes + 16);
      e->burst_bytes += wr;
      P.nt = (nt_mb >= 0 && P.T > 1 && e->burst_bytes > (unsigned long long)nt_mb * 1000000ull) ? 1 : 0;
    }
    if (share) nw = ROLL_MAX_WAVES;          // (layout of one private copy, four waves' worth of threads)
    P.off_grid = L.off_grid; P.off_T = L.off_codes; P.off_shadow = L.off_shadow; P.off_spr = L.off_spr; P.off_act = L.off_act;
    P.codes_stride = L.codes_stride; P.off_shadow_gt = L.off_shadow_gt;
    P.shadow_stride = L.shadow_stride; P.spr_stride = 64 * 16;
    if (P.use_shadow) P.use_shadow = e->roll_shadows;
    P.w_magic = (65536u + (uint32_t)e->W - 1u) / (uint32_t)e->W; P.h_magic = (65536u + (uint32_t)e->H - 1u) / (uint32_t)e->H;
    const bool full = e->fast_full;
    if (e->dyn_inloop) launch_roll_dynobs(e->cfg.rng_mode == MG_RNG_PHILOX, grid, nw, (size_t)L.total, e->stream, P);
    else if (in_loop_verify) launch_roll_sentence(full, grid, nw, (size_t)L.total, e->stream, P);
    // (round 6: the rules that have a unit of their own -- RULE_x alone instead of its whole group: MG_ONE_RULE_UNITS, mg_launch.h; a unit without the STAGED split leaves
    // a staged launch to its group's unit)
#define MG_TRY_UNIT(NAME, GROUP, RULE, STAGED) else if (MG_GOTO_TU && gg == GROUP && e->rule == RULE && (STAGED || !P.staged)) launch_roll_##NAME(full, grid, nw, (size_t)L.total, e->stream, P);
    MG_ONE_RULE_UNITS(MG_TRY_UNIT)
#undef MG_TRY_UNIT
    else if (gg == GG_NONE) launch_roll_none(full, grid, nw, (size_t)L.total, e->stream, P);
    else if (gg == GG_LIGHT) launch_roll_light(full, grid, nw, (size_t)L.total, e->stream, P);
    else if (gg == GG_ROOMGRID) launch_roll_roomgrid(full, grid, nw, (size_t)L.total, e->stream, P);
    else launch_roll_rooms(full, grid, nw, (size_t)L.total, e->stream, P);
    launched = true;
  }
  // one translation unit per rule group (mg_step_*.hip): (MODE, LPE) picks the instantiation inside it
  if (!launched) launched = gg == GG_NONE ? launch_step_none(mode, e->lpe, grid, lds, e->stream, P)
                      : gg == GG_LIGHT ? launch_step_light(mode, e->lpe, grid, lds, e->stream, P)
                      : gg == GG_ROOMGRID ? launch_step_roomgrid(mode, e->lpe, grid, lds, e->stream, P)
                                          : launch_step_rooms(mode, e->lpe, grid, lds, e->stream, P);
  if (!launched) return fail(e, MG_ERR_INVALID, "internal: no k_step variant for mode %d / group %d", mode, gg);
  HIP_TRY(e, hipGetLastError());
  if (e->sentence && !e->fast7) {
    // RoomGridLevel.step's verifier half (per-episode max_steps, the instruction tree, object identity): see k_verify
    // (the default 7x7 view runs it inside k_roll7<GG_SENTENCE>'s step loop instead)
    VerifyParams V;
    V.grid = e->grid; V.agent = e->agent; V.instr = e->instr; V.spare_instr = e->spare_instr;
    V.head = e->head; V.ring_mask = (uint32_t)(e->R - 1);
    V.rec = e->out + (size_t)P.slot0 * e->slot_bytes;
    V.off_reward = e->off_reward; V.off_term = e->off_term; V.off_trunc = e->off_trunc; V.off_action = e->off_action; V.off_sentence = e->off_sentence;
    V.err = e->err; V.N = e->N; V.W = e->W; V.H = e->H; V.CS = e->CS; V.phase = P.phase; V.autoreset_next_step = P.autoreset_next_step; V.done_actions = P.done_actions;
    hipLaunchKernelGGL(k_verify, dim3((e->N + 127) / 128), dim3(128), 0, e->stream, V);
    HIP_TRY(e, hipGetLastError());
  }
  if (e->rgb) {
    if (e->render_generic) {
      const size_t px = (size_t)e->N * e->render_g.Ht * e->render_g.ts * e->render_g.Wt * e->render_g.ts;
      hipLaunchKernelGGL(k_render_generic, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, e->stream, e->render_g);
    } else
      hipLaunchKernelGGL(k_render, dim3(e->render_blocks), dim3(e->render_threads), (size_t)e->render_lds, e->stream, e->render);
    HIP_TRY(e, hipGetLastError());
  }
  e->launches++;
  if (P.phase == PHASE_STEP) e->env_steps += (uint64_t)e->N * (uint64_t)P.T;
  return MG_OK;
}

static int check_device_errors(mg_env* e) {
  uint32_t bits = 0;
  HIP_TRY(e, wait_stream(e->stream));
  e->burst_bytes = 0;                                      // the step stream is idle
  for (int k = 0; k < 4; k++) if (e->err_host[k]) { bits |= 1u << k; e->err_host[k] = 0u; }
  if (e->err_host[ERR_WORD_SPIN]) {                        // (-DMG_SPIN_BOUND builds: a protocol regression shows up as an error, not as a hung GPU)
    e->err_host[ERR_WORD_SPIN] = 0u;
    return fail(e, MG_ERR_HIP, "k_roll7: an inter-wave spin loop ran past MG_SPIN_BOUND polls (the LDS protocol between the dynamics and the encode waves is broken)");
  }
  if (!bits) return MG_OK;
  if (bits & ERR_BAD_ACTION) return fail(e, MG_ERR_BAD_ACTION, "Unknown action: value outside 0..6 (minigrid_env.py:584-585)");
  if (bits & ERR_OOB) return fail(e, MG_ERR_OOB, "front cell outside the grid (core/grid.py:74-78 assert)");
  if (bits & ERR_TRACKED) return fail(e, MG_ERR_TRACKED, "GoToInstr: more than four stale tracked positions between two drop actions");
  return fail(e, MG_ERR_GENERATOR, "map generator exhausted its retry bound, or RoomGrid.place_agent cannot terminate (every free cell of the "
                                   "agent's room faces an object: the reference spins for ever in roomgrid.py:327-332)");
}

// Debug aid for the crash hunt (MG_GUARD=1; profiles/crash_hunt.md): every device buffer gets a 4 KB red zone on both sides, filled
// with a pattern; mg_sync and mg_destroy verify that no kernel wrote into one and name the buffer otherwise.
constexpr size_t GUARD_ZONE = 4096;
constexpr uint8_t GUARD_BYTE = 0xC7;
static bool guard_on() { static const bool on = Knobs::from_env().guard; return on; }
static hipError_t env_alloc(mg_env* e, void** p, size_t bytes, const char* name) {
  const size_t z = guard_on() ? GUARD_ZONE : 0;
  void* base = nullptr;
  hipError_t rc = hipMalloc(&base, bytes + 2 * z + 16);
  if (rc != hipSuccess) return rc;
  if (z) {
    rc = hipMemset(base, GUARD_BYTE, z);
    if (rc == hipSuccess) rc = hipMemset((uint8_t*)base + z + bytes, GUARD_BYTE, z + 16);
    if (rc != hipSuccess) { (void)hipFree(base); return rc; }
  }
  *p = (uint8_t*)base + z;
  e->allocs.push_back({ *p, base, bytes, name });
  return hipSuccess;
}
template <class T>
static hipError_t dalloc_named(mg_env* e, T** p, size_t n, const char* name) { return env_alloc(e, (void**)p, n * sizeof(T), name); }
#define dalloc(p, n) dalloc_named(e, p, n, #p)
static int check_guards(mg_env* e) {
  if (!guard_on()) return MG_OK;
  std::vector<uint8_t> h(GUARD_ZONE + 16);
  for (const auto& a : e->allocs) {
    for (int side = 0; side < 2; side++) {
      const size_t len = side ? GUARD_ZONE + 16 : GUARD_ZONE;
      const uint8_t* src = side ? (const uint8_t*)a.user + a.bytes : (const uint8_t*)a.base;
      if (hipMemcpy(h.data(), src, len, hipMemcpyDeviceToHost) != hipSuccess) return fail(e, MG_ERR_HIP, "guard check: copy failed");
      for (size_t i = 0; i < len; i++)
        if (h[i] != GUARD_BYTE) {
          const long long off = side ? (long long)(a.bytes + i) : (long long)i - (long long)GUARD_ZONE;
          fprintf(stderr, "[libminigrid_hip] MG_GUARD: out-of-bounds write at %s%+lld (buffer of %zu bytes, value 0x%02x)\n", a.name, off, a.bytes, h[i]);
          return fail(e, MG_ERR_HIP, "MG_GUARD: out-of-bounds device write at %s%+lld (buffer of %zu bytes)", a.name, off, a.bytes);
        }
    }
  }
  return MG_OK;
}

// ---- RGB modes: the tile atlas and k_render's launch geometry ------------------------------------------------
static int setup_render(mg_env* e) {
  const int ts = e->cfg.tile_size, V = e->cfg.agent_view_size;
  const bool full = e->cfg.obs_mode == MG_OBS_RGB;
  RenderParams& R = e->render;
  R.N = e->N; R.ts = ts; R.full = full ? 1 : 0;
  R.Wt = full ? e->W : V; R.Ht = full ? e->H : V; R.cells = R.Wt * R.Ht;
  e->render_generic = !(ts % 4 == 0 && ts >= 4 && ts <= 16);
  if (e->render_generic) {
    // any other tile size: the per-pixel kernel, atlas in global memory (the tile bytes need not be whole dwords)
    const size_t tb = (size_t)ts * ts * 3;
    std::vector<uint8_t> all((size_t)TILE_KEYS * 10 * tb), dev((size_t)TILE_KEYS * 10 * tb);
    tiles::render_all(ts, all.data());
    for (int k = 0; k < TILE_KEYS; k++)
      for (int ad = 0; ad < 5; ad++)
        for (int hl = 0; hl < 2; hl++) {
          const uint8_t* src = all.data() + (((size_t)k * 5 + ad) * 2 + hl) * tb;
          const size_t di = ad == 0 ? (size_t)k * 2 + hl : (size_t)STATIC_TILES + ((size_t)k * 4 + (ad - 1)) * 2 + hl;
          memcpy(dev.data() + di * tb, src, tb);
        }
    HIP_TRY(e, env_alloc(e, (void**)&e->atlas, dev.size() + 16, "atlas"));
    HIP_TRY(e, hipMemcpy(e->atlas, dev.data(), dev.size(), hipMemcpyHostToDevice));
    HIP_TRY(e, env_alloc(e, (void**)&e->tilemap, (size_t)e->N * e->map_bytes + 16, "tilemap"));
    HIP_TRY(e, hipMemsetAsync(e->tilemap, 0, (size_t)e->N * e->map_bytes + 16, e->stream));
    RenderGenericParams& G = e->render_g;
    G.tilemap = e->tilemap; G.agent = e->agent; G.atlas = (const uint8_t*)e->atlas; G.out = e->out;
    G.N = e->N; G.Wt = R.Wt; G.Ht = R.Ht; G.cells = R.cells; G.ts = ts; G.full = R.full;
    return MG_OK;
  }
  R.tdw_row = ts * 3 / 4; R.tile_dw = ts * R.tdw_row;
  R.rowdw = R.Wt * R.tdw_row;
  R.R = R.rowdw % 4 == 0 ? 1 : (R.rowdw % 2 == 0 ? 2 : 4);            // ts % 4 == 0, so R divides ts: a period stays inside one tile row
  R.cpp = R.R * R.rowdw / 4;
  R.ppe = R.Ht * ts / R.R;
  // envs per workgroup: as many as keep the LDS footprint (atlas + per-env agent tiles + tile offsets) near 32 KB
  int epw = 16;                     // measured best on MI355X together with 8 (profiles/r1_final/render_sweep.txt); tuning aid: MG_RENDER_EPW
  auto lds_for = [&](int n) { return (STATIC_TILES + n) * R.tile_dw * 4 + ((n * R.cells * 2 + 15) & ~15); };
  while (epw > 8 && lds_for(epw) > 40 * 1024) epw >>= 1;
  R.epw = epw; R.ngroups = (e->N + epw - 1) / epw;
  R.off_map = (STATIC_TILES + epw) * R.tile_dw * 4;
  e->render_lds = lds_for(epw);
  if (e->render_lds > 160 * 1024 || (STATIC_TILES + epw) * R.tile_dw > 65535) return fail(e, MG_ERR_INVALID, "tile atlas too large for the LDS staging");
  R.log2R = R.R == 1 ? 0 : (R.R == 2 ? 1 : 2);
  R.magic_ts = (65536u + (uint32_t)ts - 1u) / (uint32_t)ts;
  R.magic_tdw = (65536u + (uint32_t)R.tile_dw - 1u) / (uint32_t)R.tile_dw;
  for (uint32_t t = 0; t < (uint32_t)STATIC_TILES; t++)
    if (((t * (uint32_t)R.tile_dw * R.magic_tdw) >> 16) != t) return fail(e, MG_ERR_INVALID, "internal: magic_tdw");
  for (uint32_t r = 0; r < (uint32_t)(R.Ht * ts); r++)                  // the multiply-shift division is exact over its whole range
    if (((r * R.magic_ts) >> 16) != r / (uint32_t)ts) return fail(e, MG_ERR_INVALID, "internal: magic_ts");
  // big tiles (16 px: 84 KB of atlas): one 1024-thread workgroup per CU instead of one 256-thread one
  e->render_threads = e->render_lds > 40 * 1024 ? RENDER_MAX_THREADS : 256;
  if (R.cpp > e->render_threads) return fail(e, MG_ERR_INVALID, "frame too wide for k_render");
  R.t_active = e->render_threads / R.cpp * R.cpp;
  R.pp = R.t_active / R.cpp;
  int cus = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, e->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
  const int per_cu = std::max(1, std::min(2048 / e->render_threads, (160 * 1024) / std::max(e->render_lds, 1)));
  e->render_blocks = std::min(R.ngroups, cus * per_cu);              // persistent: the atlas is staged once per workgroup

  // atlas: host-rendered [key][agent][hl] -> device [key][hl] (agent-free) followed by [key][dir][hl]
  const size_t tb = (size_t)ts * ts * 3;
  std::vector<uint8_t> all((size_t)TILE_KEYS * 10 * tb), dev((size_t)TILE_KEYS * 10 * tb);
  tiles::render_all(ts, all.data());
  for (int k = 0; k < TILE_KEYS; k++)
    for (int ad = 0; ad < 5; ad++)
      for (int hl = 0; hl < 2; hl++) {
        const uint8_t* src = all.data() + (((size_t)k * 5 + ad) * 2 + hl) * tb;
        const size_t di = ad == 0 ? (size_t)k * 2 + hl : (size_t)STATIC_TILES + ((size_t)k * 4 + (ad - 1)) * 2 + hl;
        memcpy(dev.data() + di * tb, src, tb);
      }
  HIP_TRY(e, env_alloc(e, (void**)&e->atlas, dev.size(), "atlas"));
  HIP_TRY(e, hipMemcpy(e->atlas, dev.data(), dev.size(), hipMemcpyHostToDevice));
  HIP_TRY(e, env_alloc(e, (void**)&e->tilemap, (size_t)e->N * e->map_bytes + 16, "tilemap"));
  HIP_TRY(e, hipMemsetAsync(e->tilemap, 0, (size_t)e->N * e->map_bytes + 16, e->stream));
  R.tilemap = e->tilemap; R.agent = e->agent;
  R.atlas_static = e->atlas; R.atlas_agent = e->atlas + (size_t)STATIC_TILES * R.tile_dw;
  R.out = (uint4*)e->out;
  if (e->render_lds > 64 * 1024) HIP_TRY(e, hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize, e->render_lds));
  return MG_OK;
}

// What follows from the OBSERVATION part of the configuration (obs_mode, agent_view_size, tile_size, traj_slots): output sizes,
// lanes per env, the kernels' LDS carve-ups, trajectory slots, steps per fused launch.  Used by mg_create and mg_set_obs_config;
// needs the level flags (static_gen / live_gen / sentence) and the ring depth (cb) in place.  Returns an error text or null.
static const char* configure_obs(mg_env* e) {
  const int V = e->cfg.agent_view_size;
  switch (e->cfg.obs_mode) {
    case MG_OBS_FULL: case MG_OBS_SYMBOLIC: e->obs_bytes = e->cells * 3; break;
    case MG_OBS_ONEHOT: e->obs_bytes = V * V * 20; break;
    case MG_OBS_RGB_PARTIAL: e->obs_bytes = V * V * e->cfg.tile_size * e->cfg.tile_size * 3; break;
    case MG_OBS_RGB: e->obs_bytes = e->cells * e->cfg.tile_size * e->cfg.tile_size * 3; break;
    default: e->obs_bytes = V * V * 3; break;
  }
  const bool rgb = e->cfg.obs_mode == MG_OBS_RGB || e->cfg.obs_mode == MG_OBS_RGB_PARTIAL;
  e->rgb = rgb;
  e->map_bytes = !rgb ? e->obs_bytes : (e->cfg.obs_mode == MG_OBS_RGB ? e->cells : V * V);
  {
    // lanes per env: 4 wherever the encode supports it (default 7x7 partial view, FullyObs) -- four times the wavefronts for
    // the same batch (16 envs each), each a quarter of the LDS: the step loop is latency-bound per wave, not issue-bound
    const bool fast7 = e->cfg.obs_mode == MG_OBS_PARTIAL && V == 7;
    const bool fullish = (e->cfg.obs_mode == MG_OBS_FULL || e->cfg.obs_mode == MG_OBS_SYMBOLIC) && e->cells >= 32;
    // measured (profiles/r2/sweep_lpe_*.txt): 4 wins for FullyObs; for the 7x7 view 1 wins once the batch fills the chip with
    // one wave per SIMD (65 536 envs = 1024 waves), below that the extra waves of 4 lanes per env win
    // (the 7x7 view runs k_roll7: one lane per env, more wavefronts through its time split)
    e->fast7 = fast7;
    e->lpe = fullish ? 4 : 1;
    e->epw = 64 / e->lpe;
  }
  e->nwaves = (e->N + e->epw - 1) / e->epw;
  {
    // LDS carve-up of k_step (bytes), per wavefront of epw envs: decode table | guard | staged grids | guard | visibility
    // rows | observation byte stream in output order | shadow slots: every env's next spare episode (grid, agent record,
    // auxiliary word) | the caller's actions for the launch's steps.  The guard bands cover the furthest a view cell can
    // lie outside an env's own grid (V-1 rows + V-1 cells): such reads are masked, they only have to stay inside the allocation.
    const int guard = ((V - 1) * e->W + (V - 1) + 15) & ~15;
    e->off_grid = 1024 + guard;
    e->off_trow = (e->off_grid + e->epw * e->GS + guard + 15) & ~15;
    const bool generic_view = !(e->cfg.obs_mode == MG_OBS_PARTIAL && V == 7) && e->cfg.obs_mode != MG_OBS_FULL && e->cfg.obs_mode != MG_OBS_SYMBOLIC;
    e->off_T = e->off_trow + (generic_view ? e->epw * 32 : 64);   // one u16 per view row and env (generic view encode); staging scratch
    e->off_shadow = e->off_T + ((e->epw * e->map_bytes + 15) & ~15) + 16;
    e->off_spr = e->off_shadow + ((e->epw * e->GS + 15) & ~15);
    e->off_act = e->off_spr + e->epw * 16;
    e->lds_bytes = e->off_act + MAX_FUSED_STEPS * e->epw;   // the actions (at most MAX_FUSED_STEPS steps per launch) only when the caller supplies them
  }
  // Spare episodes per env a fused k_roll7 launch stages in LDS: ONE.  Two were measured (MG_ROLL_SHADOWS=2; profiles/r3/shadows.txt):
  // a second reset of an env within a launch then finds its spare in LDS instead of fetching it from the ring in HBM inside the step
  // loop, but the doubled staging costs more than that saves on every level tried (DoorKey-8x8 x 262 144: 12.3 -> 13.8 us per step,
  // LavaCrossing FullyObs: 13.0 -> 14.0, GoToRedBall: equal).  (Two need cb >= 2: a batch may take that many spares per env.)
  e->roll_shadows = 1;
  // The sentence levels (22 x 22 grids, ONE wave per workgroup: the verifier's record is the wave's) do without: the staged spares are half of the
  // workgroup's LDS (32 of 71 KB), i.e. two single-wave workgroups per CU instead of four, and their episodes are long (a reset in 1.4 % of
  // the wave-steps).  BossLevel x 131 072: 126 -> 89 us per step, x 32 768: 36.8 -> 27.6 (profiles/r4/shadows_bosslevel.txt).
  if (e->sentence) e->roll_shadows = 0;
  // (round 6) ... and so do all the big grids (more than 256 cells: the 22 x 22 mazes, MultiRoom's 25 x 25): one staged spare per env is a second copy of
  // the workgroup's grids -- 32 of 70 KB at 22 x 22 (two resident single-wave workgroups per CU instead of four), 41 of 84 KB at 25 x 25 (one instead of
  // three) -- for episodes of 120-576 steps, i.e. a handful of resets per workgroup and launch, which fetch their spare from the ring in HBM instead.
  // BabyAI-GoTo x 131 072: 4.29 -> 6.03 G env-steps/s, MultiRoom-N6 x 65 536: 2.48 -> 3.93 (profiles/r6/ab_connect_all_shadows.txt); KeyCorridorS3R3 (small
  // grid) is indifferent (23.5 / 23.1).
  if (e->cells > 256) e->roll_shadows = 0;
  // (round 6) TWO for the small levels whose episodes are at most 64 steps long (BabyAI-GoToRedBall and the other single-room GoTo levels: max_steps =
  // room_size^2): with the generators no longer what a GoToRedBall step waits for (gen_goto_lane, mg_gen.h), a second reset of an env within a 32-step launch
  // -- every env ends an episode at least once in two launches -- is worth the second staged set: x 32 768 13.2 -> 14.0 G env-steps/s, three runs each
  // (profiles/r6/ab_shadow_sets_fullyobs_gotoredball.txt; DoorKey-8x8 / Empty-8x8, episodes of hundreds of steps: 31.2 -> 28.9 / 30.6 -> 29.1 with two, as in round 3)
  // -- while the batch leaves the CUs a fourth workgroup slot to give away: the second set is 5.4 KB of a workgroup's LDS (43 instead of 37.6 KB: three resident
  // workgroups per CU instead of four).  profiles/r6/ab_shadow_sets_by_batch_size.txt, two sets against one: x 16 384 7.36 / 7.03 G, x 32 768 13.97 / 13.18, x 49 152
  // 14.7 / 14.3, x 65 536 14.3 / 16.1, x 131 072 16.8 / 18.4 -- so: up to three workgroups per CU (49 152 envs on 256 CUs)
  {
    int cus = 256;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, e->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    if (e->cells <= 64 && e->cfg.max_steps > 0 && e->cfg.max_steps <= 64 && !e->sentence && !e->static_gen && !e->live_gen && e->cb >= 2 &&
        (e->N + 63) / 64 <= 3 * cus) e->roll_shadows = 2;
  }
  if (e->k.roll_shadows == 1) e->roll_shadows = 1;
  if (e->k.roll_shadows == 2 && !e->static_gen && !e->live_gen && e->cb >= 2) e->roll_shadows = 2;
  if (e->k.roll_shadows == 0 && !e->static_gen) e->roll_shadows = 0;    // no staging: every reset fetches its spare from the ring in HBM inside the loop
  e->fast_full = false;
  if (e->cfg.obs_mode == MG_OBS_FULL && e->cells <= 341) {
    // FullyObs through k_roll7<., true>: the row-major grids + their image-order streams, private per wave, and the shadow pair.  Only
    // while one wave's worth fits comfortably (grids up to 16 x 16); larger grids keep k_step with four lanes per env.
    e->fast_full = true;
    e->roll_guard = 16;
    if (roll_lds_bytes(e, 1, true) > 72 * 1024) e->fast_full = false;
    else { e->lpe = 1; e->epw = 64; e->nwaves = (e->N + 63) / 64; }
  }
  e->roll_split_on = e->k.roll_split;
  e->k_epw = e->k.roll_epw;
  e->dyn_inloop = e->live_gen && e->fast7 && !e->fast_full && e->k.dyn_inloop != 0;
  if (e->fast7 || e->fast_full) {
    // k_roll7 (mg_roll.h): NW wavefronts per workgroup, each with a private copy of the 64 grids and its own code staging.  As many
    // as keep three workgroups on a CU (160 KB of LDS): 4 for the 8x8 and 9x9 levels, fewer for the big grids.
    if (e->fast7) e->roll_guard = (6 * e->W + 12 + 15) & ~15;
    // Measured with the quad encode (profiles/r3/sweep_nw_ratio_quads.txt): 4 waves per workgroup at every batch size (Empty-8x8 x
    // 65 536: 2.64 us per step with 4, 2.71 with 3; DoorKey-8x8 x 262 144: 11.3 with 4, 11.8 with 3, 12.1 with 2).  (With the chunk
    // encode 3 waves won above 1 536 workgroups -- sweep_nw_ratio.txt -- : the own step was dearer, the fourth wave's replays bought less.)
    int nw = 4;
    e->full_split = true;
    // FullyObs: two waves -- the dynamics wave + ONE encode wave over staged copies of its image-order stream (round 4; LavaCrossing FullyObs x 131 072:
    // 7.63 us per step, 7.92 with two encode waves, 8.42 with the round-3 time split, whose second wave replayed the dynamics: profiles/r4/lava_split.txt)
    if (e->fast_full) nw = 2;
    while (nw > 1 && roll_lds_bytes(e, nw, true, roll_split_ok(e, nw)) > 53 * 1024) nw--;
    // (round 6) the 9 x 9 levels: four waves are 48 KB of LDS = three workgroups per CU, three waves 38.5 KB = four.  Where that decides whether the WHOLE batch is
    // resident at once -- more than three, at most four workgroups per CU -- three waves win: x 65 536 LavaCrossingS9N1 17.9 -> 20.3 G, SimpleCrossingS9N3 21.5 ->
    // 22.9, MemoryS9 15.2 -> 17.7; larger batches are a wash either way (profiles/r6/ab_9x9_waves_per_workgroup.txt)
    if (nw == 4 && !e->fast_full) {
      int cus = 256;
      { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, e->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
      const long long wgs = ((long long)e->N + 63) / 64;
      if (wgs > 3LL * cus && wgs <= 4LL * cus && roll_lds_bytes(e, 4, false, roll_split_ok(e, 4)) > 40 * 1024 && roll_lds_bytes(e, 3, false, roll_split_ok(e, 3)) <= 40 * 1024) nw = 3;
    }
    // DynamicObstacles in the loop: the dynamics wave + two encode waves over ONE copy of the grids (roll_layout; 31 KB at 16 x 16).  The level's
    // step is its placement loop, so the encode waves idle most of the time: three waves of ~150 VGPRs leave room for four workgroups per CU.
    if (e->dyn_inloop) nw = 3;
    // the big grids (more than 256 cells) of the ring levels: the STAGED split (mg_roll.h) -- the dynamics wave + ONE encode wave over one copy of the grids
    // (a private copy per wave left them one wave per workgroup).
    e->staged_big = e->fast7 && !e->fast_full && !e->sentence && !e->dyn_inloop && !e->static_gen && e->cells > 256;     // (round 6, profiles/r6/ab_staged_threshold.txt: staging from 16 x 16 on is a wash -- DoorKey-16x16 +3 %, ObstructedMaze-Full +6 %, KeyCorridorS6R3 -3 %, ObstructedMaze-2Dlhb -5 % -- and below that a loss: MemoryS11 18.0 -> 12.9 G, KeyCorridorS4R3 20.4 -> 15.1)
    if (e->staged_big) nw = 2;
    if (e->k.roll_nw >= 1 && e->k.roll_nw <= ROLL_MAX_WAVES) nw = e->k.roll_nw;
    e->roll_nw = nw;
    e->lds_bytes = std::max(roll_lds_bytes(e, nw, true), roll_lds_bytes(e, nw, true, roll_split_ok(e, nw)));
    if (e->sentence && e->fast7) e->lds_bytes = std::max(e->lds_bytes, roll_lds_bytes(e, 2, true, true));
  }
  if (e->fast7 && !e->fast_full) {
    // MG_ROLL_EPW=32: 32 envs per k_roll7 workgroup (lanes 32 .. 63 idle) = twice the workgroups for a batch that leaves the chip half
    // empty.  Measured in round 4 and NOT adopted (profiles/r4/epw32.txt): GoToRedBall x 32 768 4.9 us per step against 2.9, Empty-8x8 x 32 768
    // 1.60 against 1.37, x 16 384 1.28 against 1.32 -- the wave-instructions double and the chains do not get shorter.  The switch stays
    // for A/B runs; tests/test_gpu_roll.py keeps the path exact.
    const int epw = e->k_epw;
    e->epw = epw; e->nwaves = (e->N + epw - 1) / epw;
  }
  if (e->lds_bytes > 160 * 1024) return "grid too large for the LDS staging";
  e->seg_cap = e->live_gen ? e->epw : e->epw * 2 * e->cb;  // at most 2*cb launches per batch, one request per env each
  // trajectory slots S: default 32 (fused launches write every step of the launch to its own slot), fewer when one
  // slot is large (RGB frames: a single slot)
  const size_t per_slot = (size_t)e->N * ((size_t)e->obs_bytes + 16);
  // traj_slots > 0: exactly that many; 0: the default; < 0: -traj_slots PREFERRED, halved like the default while the ring would exceed 2 GB
  // (ShardedVecEnv asks for two blocks of max_fused_steps slots this way: ADVICE r3)
  int S = e->cfg.traj_slots > 0 ? e->cfg.traj_slots : e->cfg.traj_slots < 0 ? -e->cfg.traj_slots : 32;
  if (S > 4096) return "traj_slots must be <= 4096";
  if (rgb) S = 1;
  if (e->cfg.traj_slots <= 0) while (S > 1 && per_slot * S > ((size_t)2 << 30)) S >>= 1;
  e->S = S;
  // steps per fused launch: every step of a launch goes to its own slot; ring levels consume at most cb per launch
  // (never more than MAX_FUSED_STEPS = 32: k_step's LDS action staging and Philox blocks are sized for that, whatever S and R are)
  // (the sentence levels fuse only where their verifier runs inside the step loop: the default 7x7 view)
  e->max_fused = (rgb || (e->live_gen && !e->dyn_inloop) || (e->sentence && !e->fast7)) ? 1 : std::min(std::min(S, MAX_FUSED_STEPS), (e->static_gen || e->dyn_inloop) ? MAX_FUSED_STEPS :
                                                                        (e->cfg.autoreset_mode == MG_AUTORESET_SAME_STEP ? 1 : 2) * e->cb);
  return nullptr;
}

static const char* validate_obs_cfg(const mg_config* cfg) {
  if (cfg->agent_view_size < 3 || cfg->agent_view_size > 15 || (cfg->agent_view_size & 1) == 0)
    return "agent_view_size must be odd and in 3..15 (wrappers.py:650-651 asserts odd, >= 3)";
  if (cfg->obs_mode < MG_OBS_PARTIAL || cfg->obs_mode > MG_OBS_RGB) return "unknown obs_mode";
  const bool rgb = cfg->obs_mode == MG_OBS_RGB || cfg->obs_mode == MG_OBS_RGB_PARTIAL;
  if (rgb && (cfg->tile_size < 1 || cfg->tile_size > 64)) return "RGB observations: tile_size must be in 1..64";
  if (cfg->no_death_mask & (1 << T_GOAL)) return "goal cannot be a death cell (wrappers.py:854)";
  // the sentence levels' SAME_STEP autoreset lives in k_roll7<GG_SENTENCE> (the verifier inside the step loop): the default 7x7 view only
  if (cfg->autoreset_mode == MG_AUTORESET_SAME_STEP && cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN &&
      !(cfg->obs_mode == MG_OBS_PARTIAL && cfg->agent_view_size == 7))
    return "SAME_STEP autoreset of the sentence levels is built for the default 7x7x3 observation only (their other observation modes end episodes in k_verify, after the step kernel)";
  // DynamicObstacles' reset draws on the stream its steps consume: SAME_STEP needs the redraw inside the step kernel (k_roll7<GG_DYNOBS>, mg_dynobs.h)
  if (cfg->autoreset_mode == MG_AUTORESET_SAME_STEP && cfg->env_kind == MG_ENV_DYNOBS && !(cfg->obs_mode == MG_OBS_PARTIAL && cfg->agent_view_size == 7))
    return "SAME_STEP autoreset of DynamicObstacles is built for the default 7x7x3 observation only (the other observation modes redraw finished envs between launches)";
  return nullptr;
}

// The buffers whose size follows from the observation configuration (configure_obs): refill request segments (one per step
// workgroup), the trajectory ring, the RGB atlas / tile map, the per-workgroup statistics.  alloc_obs / free_obs bracket them so that
// mg_set_obs_config can swap the observation mode of a live handle without touching its state.
static int alloc_obs(mg_env* e) {
  const size_t N = (size_t)e->N;
  const size_t nseg = (size_t)(e->live_gen ? 1 : QSETS) * e->nwaves;
  HIP_TRY(e, dalloc(&e->seg, nseg * e->seg_cap));
  HIP_TRY(e, dalloc(&e->seg_count, nseg + (size_t)e->nwaves + 1));              // (+ the packed lane refill's prefix sums: seg_off)
  HIP_TRY(e, hipMemsetAsync(e->seg_count, 0, (nseg + (size_t)e->nwaves + 1) * sizeof(uint32_t), e->stream));
  e->seg_off = e->seg_count + nseg;
  {
    // one trajectory slot = one contiguous record: obs | reward | terminated | truncated | direction | mission | action
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // scalars: one 16-byte mg_step_scalars per env (ABI 3), written by the step kernels with ONE store per env
    static_assert(sizeof(mg_step_scalars) == 16, "mg_step_scalars is one 16-byte store");
    e->off_reward = up(N * e->obs_bytes + 16);
    e->off_term = e->off_reward + offsetof(mg_step_scalars, terminated);
    e->off_trunc = e->off_reward + offsetof(mg_step_scalars, truncated);
    e->off_dir = e->off_reward + offsetof(mg_step_scalars, direction);
    e->off_action = e->off_reward + offsetof(mg_step_scalars, action);
    e->off_mission = e->off_reward + offsetof(mg_step_scalars, mission_id);
    e->off_sentence = e->off_reward + up(N * sizeof(mg_step_scalars));   // sentence levels: the mission as data, two u64 per env
    e->record_bytes = e->sentence ? up(e->off_sentence + 16 * N) : e->off_sentence;
    e->slot_bytes = e->record_bytes;
    if (e->slot_bytes >= ((size_t)1 << 32)) return fail(e, MG_ERR_INVALID, "one step record must stay below 4 GB (fewer envs per handle)");
    HIP_TRY(e, dalloc(&e->out, e->slot_bytes * (size_t)e->S));
    HIP_TRY(e, hipMemsetAsync(e->out, 0, e->slot_bytes * (size_t)e->S, e->stream));
  }
  if (e->rgb) { int rc = setup_render(e); if (rc) return rc; }
  e->ncounters = (size_t)STAT_EPISODES + (size_t)e->nwaves + 2 * (size_t)STAT_GEN_SLOTS;
  HIP_TRY(e, dalloc(&e->counters, e->ncounters));
  HIP_TRY(e, hipMemsetAsync(e->counters, 0, e->ncounters * sizeof(unsigned long long), e->stream));
  const int need = e->lds_bytes;
  if (need > 64 * 1024) {
    // the attribute is per function, not per handle: only ever raise it, so that a handle with a smaller LDS need
    // created later cannot make the launches of an earlier, larger one fail (per device; guarded for concurrent creates)
    static std::mutex lds_mu;
    static int lds_max[64] = { 0 };
    std::lock_guard<std::mutex> lk(lds_mu);
    if (need > lds_max[e->device & 63]) {
      HIP_TRY(e, step_max_lds_none(need)); HIP_TRY(e, step_max_lds_light(need)); HIP_TRY(e, step_max_lds_roomgrid(need)); HIP_TRY(e, step_max_lds_rooms(need));
      HIP_TRY(e, roll_max_lds_none(need)); HIP_TRY(e, roll_max_lds_light(need)); HIP_TRY(e, roll_max_lds_roomgrid(need)); HIP_TRY(e, roll_max_lds_rooms(need));
      HIP_TRY(e, roll_max_lds_sentence(need)); HIP_TRY(e, roll_max_lds_dynobs(need)); 
#define MG_UNIT_LDS(NAME, GROUP, RULE, STAGED) HIP_TRY(e, roll_max_lds_##NAME(need));
      MG_ONE_RULE_UNITS(MG_UNIT_LDS)
#undef MG_UNIT_LDS
     
      lds_max[e->device & 63] = need;
    }
  }
  return MG_OK;
}
static void free_obs(mg_env* e) {
  void** ptrs[] = { (void**)&e->seg, (void**)&e->seg_count, (void**)&e->out, (void**)&e->counters, (void**)&e->tilemap, (void**)&e->atlas };
  for (void** pp : ptrs) {
    if (!*pp) continue;
    for (size_t k = 0; k < e->allocs.size(); k++)
      if (e->allocs[k].user == *pp) { (void)hipFree(e->allocs[k].base); e->allocs.erase(e->allocs.begin() + (long)k); break; }
    *pp = nullptr;
  }
}

// DynamicObstacles draws a finished env's next episode in place right before the next step launch, from a request the step that
// ended it filed (one segment per step workgroup).  The requests follow from the agent records (RESET_PENDING): this re-files them
// after the segments were re-made (mg_set_obs_config) or the records replaced (mg_load_state).  host_rec: the N agent records, or
// null to read them from the device.
static int rebuild_live_requests(mg_env* e, const uint64_t* host_rec) {
  const size_t N = (size_t)e->N;
  std::vector<uint64_t> tmp;
  if (!host_rec) {
    tmp.resize(N);
    HIP_TRY(e, hipMemcpy(tmp.data(), e->agent, N * 8, hipMemcpyDeviceToHost));
    host_rec = tmp.data();
  }
  std::vector<uint32_t> seg((size_t)e->nwaves * e->seg_cap, 0u), cnt((size_t)e->nwaves, 0u);
  for (size_t n = 0; n < N; n++)
    if ((host_rec[n] >> 48) & FLAG_RESET_PENDING) {
      const size_t wg = n / (size_t)e->epw;
      if (cnt[wg] < (uint32_t)e->seg_cap) seg[wg * e->seg_cap + cnt[wg]++] = (uint32_t)n;
    }
  HIP_TRY(e, hipMemcpy(e->seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(e, hipMemcpy(e->seg_count, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
  return MG_OK;
}

// ---- C ABI ------------------------------------------------------------------------------------------------
extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }

const char* mg_build_info(void) {
#if defined(MG_ATTRIBUTION)
#define MG_BI_ATTR "1"
#else
#define MG_BI_ATTR "0"
#endif
#define MG_BI_STR2(x) #x
#define MG_BI_STR(x) MG_BI_STR2(x)
#ifdef MG_EMU      // tests/emu: these sources compiled for the host SIMT emulator -- test infrastructure; bench.py and smoke() refuse such a library
  return "attribution=" MG_BI_ATTR ";encode_quads=" MG_BI_STR(MG_ENCODE_QUADS) ";lane_wide=" MG_BI_STR(MG_LANE_WIDE) ";arch=host;emulator=1";
#else
  return "attribution=" MG_BI_ATTR ";encode_quads=" MG_BI_STR(MG_ENCODE_QUADS) ";lane_wide=" MG_BI_STR(MG_LANE_WIDE) ";arch=gfx950";
#endif
}

int mg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* mg_last_error(mg_env* env) { return env ? env->last_error.c_str() : g_create_error.c_str(); }

// Diagnostic aid (MG_ABORT_BACKTRACE=1, set by tests/conftest.py): a SIGABRT anywhere in the process -- the HIP / HSA runtime aborts
// without a message on some queue errors -- first prints the native backtrace of the aborting thread to stderr, then takes the
// previous disposition (Python's faulthandler, or the default core dump).
static struct sigaction g_prev_abrt;
static void on_abort(int sig, siginfo_t* info, void* ctx) {
  static const char msg[] = "\n[libminigrid_hip] SIGABRT -- native backtrace of the aborting thread:\n";
  (void)!write(2, msg, sizeof msg - 1);
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  sigaction(SIGABRT, &g_prev_abrt, nullptr);
  if ((g_prev_abrt.sa_flags & SA_SIGINFO) && g_prev_abrt.sa_sigaction) g_prev_abrt.sa_sigaction(sig, info, ctx);
  else if (g_prev_abrt.sa_handler != SIG_DFL && g_prev_abrt.sa_handler != SIG_IGN && g_prev_abrt.sa_handler) g_prev_abrt.sa_handler(sig);
  raise(SIGABRT);
}
static void install_abort_backtrace() {
  static std::once_flag once;
  std::call_once(once, [] {
    if (!Knobs::from_env().abort_backtrace) return;
    void* warm[4]; (void)backtrace(warm, 4);           // loads libgcc now, not inside the handler
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_abort; sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGABRT, &sa, &g_prev_abrt);
  });
}

int mg_create(const mg_config* cfg, int device, void* stream, mg_env** out) {
  if (!cfg || !out) return fail(nullptr, MG_ERR_INVALID, "null argument");
  install_abort_backtrace();
  *out = nullptr;
  if (cfg->abi_version != MG_ABI_VERSION) return fail(nullptr, MG_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, MG_ABI_VERSION);
  if (cfg->num_envs < 1) return fail(nullptr, MG_ERR_INVALID, "num_envs must be >= 1");
  if (cfg->width < 3 || cfg->height < 3 || cfg->width > 25 || cfg->height > 25)
    return fail(nullptr, MG_ERR_INVALID, "width/height must be in 3..25 (core/grid.py:29-30 asserts >= 3)");
  if (const char* bad = validate_obs_cfg(cfg)) return fail(nullptr, MG_ERR_INVALID, "%s", bad);
  if (cfg->max_steps < 1 || cfg->max_steps > 65535) return fail(nullptr, MG_ERR_INVALID, "max_steps must be in 1..65535");
  if (cfg->autoreset_mode < MG_AUTORESET_NEXT_STEP || cfg->autoreset_mode > MG_AUTORESET_SAME_STEP) return fail(nullptr, MG_ERR_INVALID, "unknown autoreset_mode");
  if (cfg->autoreset_mode == MG_AUTORESET_SAME_STEP && cfg->env_kind == MG_ENV_DYNOBS && Knobs::from_env().dyn_inloop == 0)
    return fail(nullptr, MG_ERR_INVALID, "SAME_STEP autoreset of DynamicObstacles needs the in-loop redraw (MG_DYN_INLOOP=0 switches it off)");
  if (cfg->env_kind < MG_ENV_EMPTY || cfg->env_kind > MG_ENV_LEVELGEN) return fail(nullptr, MG_ERR_INVALID, "unknown env_kind");
  if (cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN) {
    const int st = cfg->room_size - 1, k = cfg->env_kind;
    const int nc = st > 0 ? (cfg->width - 1) / st : 0, nr = st > 0 ? (cfg->height - 1) / st : 0;
    if (cfg->room_size < 4 || cfg->room_size > 8 || (cfg->width - 1) % st || (cfg->height - 1) % st || nc < 1 || nc > 3 || nr < 1 || nr > 3)
      return fail(nullptr, MG_ERR_INVALID, "sentence levels: 1..3 x 1..3 rooms of room_size 4..8");
    if ((k == MG_ENV_OPENTWODOORS || k == MG_ENV_OPENDOORSORDER) && (nc != 3 || nr != 3)) return fail(nullptr, MG_ERR_INVALID, "OpenTwoDoors / OpenDoorsOrder: 3 x 3 rooms");
    if (k == MG_ENV_OPENTWODOORS && (cfg->agent_start_x < -1 || cfg->agent_start_x > 5 || cfg->agent_start_y < -1 || cfg->agent_start_y > 5))
      return fail(nullptr, MG_ERR_INVALID, "OpenTwoDoors: agent_start_x / agent_start_y = first / second door colour (COLOR_NAMES index) or -1");
    if (k == MG_ENV_OPENDOORSORDER && (cfg->num_dists < 2 || cfg->num_dists > 4)) return fail(nullptr, MG_ERR_INVALID, "OpenDoorsOrder: 2..4 doors");
    if (k == MG_ENV_MOVETWOACROSS && (nc != 2 || nr != 1 || cfg->num_dists < 2 || cfg->num_dists > 9)) return fail(nullptr, MG_ERR_INVALID, "MoveTwoAcross: 1 x 2 rooms, 2..9 objects per room");
    if (k == MG_ENV_LEVELGEN && (cfg->num_dists < 0 || cfg->num_dists > 24 || (cfg->num_crossings & 15) == 0 || ((cfg->num_crossings >> 4) & 7) == 0 ||
        (unsigned)cfg->num_crossings > 2047u || cfg->strip2_row < 0 || cfg->strip2_row > 100))
      return fail(nullptr, MG_ERR_INVALID, "LevelGen: num_crossings = action kinds | instr kinds << 4 | locations << 7 | unblocking << 8 | implicit_unlock << 9 | redraw_stuck << 10, strip2_row = locked_room_prob in percent, at most 24 distractors");
  }
  if (cfg->env_kind >= MG_ENV_PUTNEXTLOCAL && cfg->env_kind <= MG_ENV_OPENDOOR) {
    const int st = cfg->room_size - 1, k = cfg->env_kind;
    const int nc = st > 0 ? (cfg->width - 1) / st : 0, nr = st > 0 ? (cfg->height - 1) / st : 0;
    const int want_c = k == MG_ENV_PUTNEXTLOCAL ? 1 : k == MG_ENV_PUTNEXT ? 2 : 3, want_r = (k == MG_ENV_PUTNEXTLOCAL || k == MG_ENV_PUTNEXT) ? 1 : 3;
    if (cfg->room_size < 4 || cfg->room_size > 8 || (cfg->width - 1) % st || (cfg->height - 1) % st || nc != want_c || nr != want_r)
      return fail(nullptr, MG_ERR_INVALID, "PutNextLocal (1 room), PutNext (1 x 2 rooms), ActionObjDoor / OpenDoor (3 x 3 rooms): room_size 4..8");
    if ((k == MG_ENV_PUTNEXTLOCAL && (cfg->num_dists < 2 || cfg->num_dists > 8)) || (k == MG_ENV_PUTNEXT && (cfg->num_dists < 1 || cfg->num_dists > 4)))
      return fail(nullptr, MG_ERR_INVALID, "PutNextLocal: 2..8 objects; PutNext: 1..4 objects per room");
    if ((unsigned)cfg->num_crossings > 2u) return fail(nullptr, MG_ERR_INVALID, "num_crossings: PutNext start_carrying 0 | 1, OpenDoor select_by 0 | 1 | 2");
  }
  if (cfg->env_kind >= MG_ENV_BABYAI_UNLOCKPICKUP && cfg->env_kind <= MG_ENV_GOTOIMPUNLOCK) {
    const int st = cfg->room_size - 1, k = cfg->env_kind;
    const int nc = st > 0 ? (cfg->width - 1) / st : 0, nr = st > 0 ? (cfg->height - 1) / st : 0;
    const int want_c = (k == MG_ENV_BABYAI_UNLOCKPICKUP || k == MG_ENV_BABYAI_BLOCKEDUNLOCKPICKUP) ? 2 : 3;      // (KeyInBox, 39: 3 x 3)
    const int want_r = (k == MG_ENV_BABYAI_UNLOCKPICKUP || k == MG_ENV_BABYAI_BLOCKEDUNLOCKPICKUP || k == MG_ENV_UNLOCKTOUNLOCK) ? 1 : 3;
    if (cfg->room_size < 4 || cfg->room_size > 8 || (cfg->width - 1) % st || (cfg->height - 1) % st || nc != want_c || nr != want_r ||
        cfg->num_dists < 0 || cfg->num_dists > 8)
      return fail(nullptr, MG_ERR_INVALID, "BabyAI unlock / goto-door / pickup levels: the class's room grid (1 x 2, 1 x 3 or 3 x 3 rooms) of room_size 4..8");
  }
  if (cfg->env_kind >= MG_ENV_BABYAI_GOTO && cfg->env_kind <= MG_ENV_BABYAI_OPEN) {
    const int st = cfg->room_size - 1;
    if (cfg->room_size < 4 || cfg->room_size > 8 || (cfg->width - 1) % st || (cfg->height - 1) % st || (cfg->width - 1) / st < 2 || (cfg->width - 1) / st > 3 ||
        (cfg->height - 1) / st < 2 || (cfg->height - 1) / st > 3 || cfg->num_dists < 1 || cfg->num_dists > 21)
      return fail(nullptr, MG_ERR_INVALID, "BabyAI maze levels: 2..3 x 2..3 rooms of room_size 4..8, 1..21 distractors");
  }
  if (cfg->env_kind == MG_ENV_PUTNEAR && (cfg->width != cfg->height || cfg->width < 5 || cfg->width > 8 || cfg->num_dists < 2 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "PutNear: size 5..8, numObjs 2..8");
  if (cfg->env_kind == MG_ENV_OBSTRUCTEDMAZE && (cfg->room_size != 6 || !((cfg->width == 11 && cfg->height == 6) || (cfg->width == 16 && cfg->height == 16)) ||
      cfg->num_dists < 1 || cfg->num_dists > 4 || (unsigned)cfg->num_crossings > 15u || (unsigned)cfg->agent_start_x > 2u || (unsigned)cfg->agent_start_y > 2u))
    return fail(nullptr, MG_ERR_INVALID, "ObstructedMaze: room_size 6, 1 x 2 (11 x 6) or 3 x 3 (16 x 16) rooms, num_quarters 1..4, agent_room inside the room grid");
  if ((cfg->env_kind == MG_ENV_LOCKEDROOM || cfg->env_kind == MG_ENV_PLAYGROUND) && (cfg->width != cfg->height || cfg->width < 13 || cfg->width > 25))
    return fail(nullptr, MG_ERR_INVALID, "LockedRoom / Playground: square grid of 13..25 cells (the registered size is 19)");
  if ((cfg->env_kind == MG_ENV_PICKUPDIST || cfg->env_kind == MG_ENV_PICKUPDIST_DEBUG || cfg->env_kind == MG_ENV_ONEROOM) &&
      (cfg->width != cfg->height || cfg->width < 5 || cfg->width > 25))
    return fail(nullptr, MG_ERR_INVALID, "PickupDist / OneRoom: one square room of 5..25 cells");
  if (cfg->env_kind == MG_ENV_UNLOCKLOCAL && (cfg->room_size < 5 || cfg->room_size > 9 || cfg->width != 3 * (cfg->room_size - 1) + 1 || cfg->height != cfg->width ||
      cfg->num_dists < 0 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "UnlockLocal is a 3 x 3 RoomGrid: width = height = 3*(room_size-1)+1, room_size 5..9, at most 8 distractors");
  if (cfg->env_kind == MG_ENV_FINDOBJ && (cfg->room_size < 4 || cfg->room_size > 9 || cfg->width != 3 * (cfg->room_size - 1) + 1 || cfg->height != cfg->width))
    return fail(nullptr, MG_ERR_INVALID, "FindObj is a 3 x 3 RoomGrid: width = height = 3*(room_size-1)+1, room_size 4..9");
  if (cfg->env_kind == MG_ENV_OPENREDDOOR && (cfg->room_size < 4 || cfg->room_size > 13 || cfg->width != 2 * (cfg->room_size - 1) + 1 || cfg->height != cfg->room_size))
    return fail(nullptr, MG_ERR_INVALID, "OpenRedDoor is a 1 x 2 RoomGrid: width = 2*(room_size-1)+1, height = room_size in 4..13");
  if (cfg->env_kind == MG_ENV_MULTIROOM && (cfg->width < 6 || cfg->height < 6 || cfg->width > 25 || cfg->height > 25 || cfg->room_size < 4 || cfg->room_size > 15 ||
      cfg->num_crossings < 1 || cfg->num_dists < cfg->num_crossings || cfg->num_dists > 6))
    return fail(nullptr, MG_ERR_INVALID, "MultiRoom: grid up to 25 x 25, maxRoomSize 4..15, 1 <= minNumRooms <= maxNumRooms <= 6 (multiroom.py:89-91)");
  if (cfg->env_kind == MG_ENV_GOTOOBJECT && (cfg->width != cfg->height || cfg->width < 4 || cfg->width > 8 || cfg->num_dists < 1 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "GoToObject: size 4..8, numObjs 1..8");
  if (cfg->env_kind >= MG_ENV_GOTO_REDBALLGREY && cfg->env_kind <= MG_ENV_GOTO_LOCAL && (cfg->width != cfg->height || cfg->width < 4 || cfg->width > 8 || cfg->num_dists < 0 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "BabyAI single-room GoTo levels: room_size 4..8, at most 8 distractors");
  if (cfg->env_kind == MG_ENV_DYNOBS && (cfg->num_dists < 0 || cfg->num_dists > 8 || cfg->width > 16 || cfg->height > 16))
    return fail(nullptr, MG_ERR_INVALID, "DynamicObstacles supports up to 8 obstacles on grids up to 16 x 16");
  if ((cfg->env_kind == MG_ENV_KEYCORRIDOR || cfg->env_kind == MG_ENV_BABYAI_KEYCORRIDOR) && (cfg->room_size < 3 || cfg->width != 3 * (cfg->room_size - 1) + 1 ||
      (cfg->height - 1) % (cfg->room_size - 1) != 0 || (cfg->height - 1) / (cfg->room_size - 1) < 1 || (cfg->height - 1) / (cfg->room_size - 1) > 3 || cfg->width > 16 || cfg->height > 16))
    return fail(nullptr, MG_ERR_INVALID, "KeyCorridor is a 3 x (1..3) RoomGrid with room_size >= 3 and a grid of at most 16 x 16");
  if (cfg->env_kind == MG_ENV_REDBLUEDOORS && (cfg->width != 2 * cfg->height || cfg->height < 4))
    return fail(nullptr, MG_ERR_INVALID, "RedBlueDoors is 2*size wide and size high (redbluedoors.py:70-71)");
  if (cfg->env_kind == MG_ENV_MEMORY && ((cfg->height & 1) == 0 || cfg->height < 7 || cfg->width < 7))
    return fail(nullptr, MG_ERR_INVALID, "Memory needs an odd size >= 7 (memory.py:101 assert)");
  if (cfg->env_kind >= MG_ENV_UNLOCK && cfg->env_kind <= MG_ENV_BLOCKEDUNLOCKPICKUP &&
      (cfg->room_size < 4 || cfg->width != 2 * (cfg->room_size - 1) + 1 || cfg->height != cfg->room_size))
    return fail(nullptr, MG_ERR_INVALID, "Unlock levels are 1 x 2 RoomGrids: width = 2*(room_size-1)+1, height = room_size >= 4");
  if (cfg->env_kind == MG_ENV_FETCH && (cfg->num_dists < 1 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "Fetch supports numObjs in 1..8");
  if (cfg->env_kind == MG_ENV_GOTODOOR && (cfg->width < 5 || cfg->height < 5))
    return fail(nullptr, MG_ERR_INVALID, "GoToDoor needs size >= 5 (gotodoor.py:67 assert)");
  if (cfg->env_kind == MG_ENV_LAVAGAP && (cfg->width < 5 || cfg->height < 5))
    return fail(nullptr, MG_ERR_INVALID, "LavaGap needs width, height >= 5 (lavagap.py:101 assert)");
  if (cfg->env_kind == MG_ENV_DISTSHIFT && (cfg->width < 7 || cfg->strip2_row < 1 || cfg->strip2_row > cfg->height - 2))
    return fail(nullptr, MG_ERR_INVALID, "DistShift needs width >= 7 and the second lava strip inside the grid");
  if (cfg->env_kind == MG_ENV_FOURROOMS && (cfg->width < 7 || cfg->height < 7))
    return fail(nullptr, MG_ERR_INVALID, "FourRooms needs width, height >= 7");
  if (cfg->env_kind == MG_ENV_GOTO_REDBALL && (cfg->width != cfg->height || cfg->width < 4 || cfg->width > 8 || cfg->num_dists > 8))
    return fail(nullptr, MG_ERR_INVALID, "GoToRedBall is a single room of size 4..8 with at most 8 distractors (goto.py:129-131)");
  if (cfg->env_kind == MG_ENV_CROSSING && ((cfg->width & 1) == 0 || (cfg->height & 1) == 0 || cfg->width > 11 || cfg->height > 11))
    return fail(nullptr, MG_ERR_INVALID, "Crossing needs an odd size <= 11 (crossing.py:132 assert)");
  if ((cfg->env_kind == MG_ENV_EMPTY || cfg->env_kind == MG_ENV_DISTSHIFT || cfg->env_kind == MG_ENV_DYNOBS) && cfg->agent_start_x >= 0 &&
      (cfg->agent_start_x >= cfg->width || cfg->agent_start_y < 0 || cfg->agent_start_y >= cfg->height || (unsigned)cfg->agent_start_dir > 3u))
    return fail(nullptr, MG_ERR_INVALID, "agent start outside the grid");
  int ndev = mg_device_count();
  if (ndev < 1) return fail(nullptr, MG_ERR_NO_DEVICE, "no HIP device visible: libminigrid_hip has no CPU fallback");
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= ndev) return fail(nullptr, MG_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);
  // (before anything below asks the runtime about "the" device: the ring cap reads the free memory of the device the handle lives on, not of the
  // calling thread's current one -- ADVICE r5)
  if (hipSetDevice(device) != hipSuccess) return fail(nullptr, MG_ERR_HIP, "hipSetDevice(%d) failed", device);

  mg_env* e = new mg_env();
  e->k = Knobs::from_env();
  e->cfg = *cfg; e->device = device;
  e->N = cfg->num_envs; e->W = cfg->width; e->H = cfg->height; e->cells = e->W * e->H;
  e->CS = (e->cells + 15) & ~15;
  e->GS = e->CS + 4;                                     // odd dword stride: conflict-free same-cell LDS reads
  // empty.py:108-110, distshift.py:118-120: a fixed agent start means _gen_grid draws nothing
  e->static_gen = (cfg->env_kind == MG_ENV_EMPTY || cfg->env_kind == MG_ENV_DISTSHIFT) && cfg->agent_start_x >= 0;
  {
    const bool lane_on = lane_gen_lds_bytes(e->CS, cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN) <= 64 * 1024;
    // REFILL on lanes: the single-room levels, the Unlock family and KeyCorridor (per request segment: what they were tuned on, rounds 4-5).  For the
    // others -- the mazes, MultiRoom, the sentence levels -- lanes lose as a refill in either form (profiles/r5/lane_wide_bench_lines.txt: a few busy
    // lanes per wave; ab_packed_lane_refill.txt: whole waves of busy lanes, requests numbered across the segments -- a wave of 64 diverging maze
    // generators runs 4-7 ms, BabyAI-GoTo 1.70 -> 1.42 G, BossLevel 4.51 -> 1.76): they keep the wavefront-per-episode k_refill.
    // DIRECT generation on lanes: every level with a lane generator -- reset(seed) draws one episode per env and ring slot, every lane is busy, and
    // there lanes win everywhere (ring fill of BabyAI-GoTo x 131 072: 30.4 -> 8.0 ms per slot, BossLevel 25.3 -> 11.8, MultiRoom-N6 2.0 -> 1.8).
    // MG_LANE_PACKED=1: the levels whose refill runs on lanes refill PACKED (A/B; KeyCorridor / UnlockPickup +3 %, GoToRedBall x 32 768 -50 %);
    // MG_LANE_LPW: busy lanes per wavefront of the packed refill (1..64).
    const bool tuned_sparse = lane_gen_kind_base(cfg->env_kind) || lane_gen_kind_product_fn(cfg->env_kind);
    e->lane_gen = lane_on && tuned_sparse;
    // (round 6: the Unlock family and KeyCorridor refill PACKED by default -- long episodes, ~30 requests per segment and batch: whole wavefronts of busy
    // lanes instead of two half-empty ones per segment.  KeyCorridorS3R3 x 131 072: 20.7 -> 22.4 G env-steps/s, profiles/r6/ab_lane_wps.txt; the
    // single-room levels with many resets keep the per-segment form -- GoToRedBall x 32 768: 11.0 per segment, 5.5 packed.  MG_LANE_PACKED=0 / 1: A/B)
    e->lane_packed = e->lane_gen && (e->k.lane_packed >= 0 ? e->k.lane_packed == 1 : lane_gen_kind_product_fn(cfg->env_kind));
    // (from 16 384 envs on: a call of k_generate_lane lasts as long as its slowest lane -- ~5 ms for a maze level whatever the batch --, a call of the
    // cooperative k_generate ~0.8 ms + N / 4.3 M episodes/s: lanes win above ~18 000 envs.  MG_LANE_DIRECT: 0 = never, 1 = at every batch size)
    const int dmode = e->k.lane_direct;
    e->lane_direct = lane_on && lane_gen_kind(cfg->env_kind) && (e->lane_gen || dmode == 1 || (dmode != 0 && cfg->num_envs >= 16384));
    // BURST HYBRID (the levels whose refill stays with k_refill): a batch of at least lane_burst_min requests -- the synchronized truncation burst of a
    // long-episode level, every env at once -- refills on packed lanes (dense: BabyAI-GoTo x 131 072 draws 131 072 episodes in 8 ms on lanes, 30 ms on
    // cooperative wavefronts), everything smaller on k_refill.  The crossover is where k_refill's throughput (~4 M episodes/s) costs more than a lane
    // wave's latency (4-7 ms): ~32 768 requests.  MG_LANE_BURST: the threshold (0 = off).
    // (round 6: the cooperative generator of the RoomGrid mazes is 2.8x faster than it was -- rooms_reach, speculative connect_all, draw budgets: 4.3 -> 12.3 M
    // BabyAI-GoTo episodes/s -- so the crossover moved: 65 536 requests for them.  A de-phased BabyAI-GoTo x 131 072 batch files ~30 000 requests per refill:
    // 5.98 G env-steps/s with the old threshold (some batches went to lanes), 7.71 G without lanes; a synchronized truncation burst (131 072 requests) still
    // refills on lanes: 8.22 G.  MultiRoom (36 M episodes/s on lanes against 15 M) and the sentence levels keep 32 768.  profiles/r6/ab_staged_big_grids_burst_threshold_dephase.txt)
    // (MultiRoom since mg_genmr.h -- the chain search as one flat loop, no grid per lane, 8 KB of LDS per generating wavefront: lanes take every batch
    // of 1 024 requests or more.  MultiRoom-N6 x 65 536, de-phased: 5.9-6.0 G env-steps/s on k_refill, 7.5-7.6 on lanes; synchronized bursts 6.2 / 9.2;
    // 64 lanes per wavefront -- 32: 7.6 / 8.7, 16: 6.8 / 7.5, 8: 6.0 / 6.7.  profiles/r6/bench_lines_multiroom_lanes.txt)
    e->lane_burst_min = (e->lane_direct && !e->lane_gen) ? (cfg->env_kind == MG_ENV_MULTIROOM ? 1024 : (cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN) ? 32768 : 65536) : 0;
    if (e->k.lane_burst >= 0 && e->lane_direct && !e->lane_gen) e->lane_burst_min = e->k.lane_burst;
    e->lane_lpw = e->k.lane_lpw;
  }
  e->sentence = cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN;
  e->live_gen = cfg->env_kind == MG_ENV_DYNOBS;
  {
    // spare ring depth R (power of two).  Levels that draw nothing keep ONE constant spare; DynamicObstacles draws in
    // place (no ring).  cb = R/4 spares per env and batch; a batch is up to 2 cb steps, its refill runs behind it on the generator
    // stream and may lag three batches before a step launch has to wait.  A refill's duration is set by its longest chain of
    // whole-map retries (GoToRedBall: 70-100 us), not by its size, so deeper rings -- fewer, larger refills -- pay for every
    // level.  Measured per step with every consumed episode regenerated inside the timed region (mg_sync closes the open batch;
    // profiles/r2/sweep_ring_sync.txt): GoToRedBall x 32 768: 11.1 us (R = 16), 7.4 (32), 6.4 (64), 7.7 (128);
    // DoorKey-8x8 x 262 144: 19.0 (32), 17.9 (64), 16.8 (128); LavaCrossing FullyObs x 131 072: 18.1 (32), 15.2 (64), 13.5 (128).
    // Default: 128, and 64 for the BabyAI single-room generators (whole-map rejection sampling: their long refill chains do
    // worse with twice the work per refill).  Sized for 288 GB of HBM: 128 spare maps of 64 B are 8 KB per env (2 GB at 262 144
    // envs); capped at min(32 GB, a quarter of the free device memory) of ring (the sentence levels carry a 320 B instruction record per spare).
    int R = 1;
    if (!e->static_gen && !e->live_gen) {
      // (the sentence levels: 64 since their verifier runs inside the fused step loop -- with 16 a refill of ~0.6 ms, the length of its
      // longest LevelGen chain, was due every 8 steps and bounded BossLevel at 76 us per step; 64: 38 us, profiles/r3/bosslevel_ring.txt)
      // (round 4: 128 for the sentence levels as well -- with two generating wavefronts per request segment a refill lasts longer and is due half
      // as often: BossLevel x 131 072 33.4 -> 28.7 us per step, profiles/r4/bosslevel_generator2.txt)
      // (and for the single-room RoomGrid levels that still draw a wavefront per episode -- Unlock, KeyCorridor, ... --: round 2 gave them 64 because
      // their whole-map retries made long refills worse; with today's refill KeyCorridorS3R3 x 131 072 runs 48.1 us per step with 64, 40.8 with 128,
      // profiles/r4/ring_other_levels.txt)
      R = cfg->spare_ring > 0 ? cfg->spare_ring : 128;
      // k_refill_lane (one lane per episode, round 4) does a fifth of k_refill's work per episode but a refill LASTS longer -- a wave runs as
      // long as its unluckiest lane (GoToRedBall: 200-290 us) -- so its levels with many resets take the deepest ring: a refill is then due
      // every 128 steps, not every 32 (GoToRedBall x 32 768: 6.8 us per step with R = 64, 3.9 with 128, 2.9 with 256: profiles/r4/lane_refill_ring.txt)
      // (and the others gain as well -- DoorKey-8x8 x 262 144: 7.93 -> 7.49 us per step, LavaCrossing FullyObs x 131 072: 8.84 -> 8.32)
      if (cfg->spare_ring <= 0 && e->lane_gen) R = 256;
      // (round 5: the big-grid maze levels too -- their wavefront-per-episode refill is latency-bound, ~0.8 ms per episode in a chain of one to three per
      // wave, so a refill of twice the episodes takes about as long and is due half as often: BabyAI-GoTo x 131 072 48.8 -> 41.1 us per step in a window
      // without a truncation burst, 77.6 -> 65.3 with one, profiles/r5/babyai_goto_ring.txt; MultiRoom-N6 and BossLevel: no difference)
      // (large batches only: a small batch's refill is short whatever the ring, and every seeded reset fills the whole ring)
      if (cfg->spare_ring <= 0 && e->cells > 256 && e->N >= 32768 && !(cfg->env_kind >= MG_ENV_OPENTWODOORS && cfg->env_kind <= MG_ENV_LEVELGEN)) R = 256;
      if (e->k.spare_ring >= 4) R = e->k.spare_ring;
      if (R < 4 || R > 256 || (R & (R - 1))) { delete e; return fail(nullptr, MG_ERR_INVALID, "spare_ring must be a power of two in 4..256"); }
      // per ring slot and env: the map, the agent / aux words, the five stream words of the snapshot (+ LevelGen state and the
      // 320-byte instruction record for the sentence levels): everything that scales with R counts against the ring cap (min(32 GB, a quarter of the free memory): below)
      const size_t per_slot_env = (size_t)e->CS + 16 + 40 + (e->sentence ? 4 + INSTR_WORDS * 8 : 0);
      // ... and against a quarter of what the device has free right now (several handles per GPU, or a part with less HBM: ADVICE r4)
      size_t ring_cap = (size_t)32 << 30;                                                  // (32 GB of 288; 18.5 GB for BabyAI-GoTo x 131 072 at R = 256)
      { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) ring_cap = std::min(ring_cap, fr / 4); }
      while (R > 4 && (size_t)R * e->N * per_slot_env > ring_cap) R >>= 1;
    }
    e->R = R; e->cb = std::max(1, R / REFILL_LAG);
  }
  if (const char* bad = configure_obs(e)) { const std::string msg = bad; delete e; return fail(nullptr, MG_ERR_INVALID, "%s", msg.c_str()); }
  // GoToInstr levels: rule_div selects how the described object follows from the mission id (see k_step)
  if (cfg->env_kind == MG_ENV_GOTO_REDBALL || cfg->env_kind == MG_ENV_GOTO_REDBALLGREY) { e->rule = RULE_GOTO; e->rule_cell = (int)CELL_BALL_RED; e->rule_div = 0; }
  if (cfg->env_kind == MG_ENV_GOTO_REDBLUEBALL) { e->rule = RULE_GOTO; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_GOTO_OBJ || cfg->env_kind == MG_ENV_GOTO_LOCAL) { e->rule = RULE_GOTO; e->rule_div = 2; }
  if (cfg->env_kind == MG_ENV_GOTOOBJECT) { e->rule = RULE_GOTOOBJ; e->rule_div = 2; }     // same mission id -> (colour, type) coding as GoToObj
  if (cfg->env_kind == MG_ENV_BABYAI_GOTO) { e->rule = RULE_GOTO_BIG; e->rule_div = 2; }   // (colour, type) from the mission id, like GoToObj
  if (cfg->env_kind == MG_ENV_BABYAI_PICKUP) { e->rule = RULE_PICKUPDESC; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_BABYAI_OPEN) { e->rule = RULE_OPENFRONT; e->rule_div = 6; }
  if (cfg->env_kind == MG_ENV_GOTOIMPUNLOCK) { e->rule = RULE_GOTO_BIG; e->rule_div = 2; }
  if (cfg->env_kind == MG_ENV_BABYAI_GOTODOOR) { e->rule = RULE_GOTO_BIG; e->rule_div = 3; }      // a door by colour, in any state
  if (cfg->env_kind == MG_ENV_GOTOOBJDOOR) { e->rule = RULE_GOTO_BIG; e->rule_div = 4; }          // (colour, key | ball | box | door)
  if (cfg->env_kind == MG_ENV_BABYAI_UNLOCKPICKUP || cfg->env_kind == MG_ENV_BABYAI_BLOCKEDUNLOCKPICKUP || cfg->env_kind == MG_ENV_UNLOCKTOUNLOCK ||
      cfg->env_kind == MG_ENV_UNBLOCKPICKUP || cfg->env_kind == MG_ENV_PICKUPABOVE) { e->rule = RULE_PICKUPDESC; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_BABYAI_UNLOCK) { e->rule = RULE_OPENFRONT; e->rule_div = 6; }
  if (cfg->env_kind == MG_ENV_PUTNEAR) { e->rule = RULE_PUTNEAR; e->rule_div = 2; }        // target (colour, type) = mission id % 18, like GoToObj
  if (e->sentence) e->rule = RULE_SENTENCE;
  if (cfg->env_kind == MG_ENV_PUTNEXTLOCAL || cfg->env_kind == MG_ENV_PUTNEXT) { e->rule = RULE_PUTNEXT; e->rule_div = cfg->env_kind == MG_ENV_PUTNEXT && cfg->num_crossings ? 1 : 0; }
  if (cfg->env_kind == MG_ENV_ACTIONOBJDOOR) { e->rule = RULE_GOTO_BIG; e->rule_div = 5; }        // verb = mission id / 48: go to | pick up | open
  if (cfg->env_kind == MG_ENV_OPENDOOR) { e->rule = RULE_OPENDOOR; e->rule_div = cfg->strip2_row ? 1 : 0; }   // strip2_row = strict (OpenDoorDebug)
  e->goto_kind = e->rule == RULE_GOTO || e->rule == RULE_GOTOOBJ || e->rule == RULE_PUTNEAR || e->rule == RULE_GOTO_BIG || e->rule == RULE_PUTNEXT || e->rule == RULE_OPENDOOR;
  if (cfg->env_kind == MG_ENV_PICKUPDIST || cfg->env_kind == MG_ENV_ONEROOM || cfg->env_kind == MG_ENV_FINDOBJ ||
      cfg->env_kind == MG_ENV_BABYAI_KEYCORRIDOR) { e->rule = RULE_PICKUPDESC; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_PICKUPDIST_DEBUG) { e->rule = RULE_PICKUPDESC; e->rule_div = 2; }      // strict
  if (cfg->env_kind == MG_ENV_OPENREDDOOR || cfg->env_kind == MG_ENV_UNLOCKLOCAL || cfg->env_kind == MG_ENV_KEYINBOX) e->rule = RULE_OPENFRONT;
  if (cfg->env_kind == MG_ENV_FETCH) e->rule = RULE_FETCH;
  if (cfg->env_kind == MG_ENV_GOTODOOR) e->rule = RULE_GOTODOOR;
  if (cfg->env_kind == MG_ENV_DYNOBS) e->rule = RULE_DYNOBS;
  if (cfg->env_kind == MG_ENV_REDBLUEDOORS) e->rule = RULE_REDBLUE;
  if (cfg->env_kind == MG_ENV_MEMORY) e->rule = RULE_MEMORY;
  if (cfg->env_kind == MG_ENV_UNLOCK) { e->rule = RULE_UNLOCK; e->rule_cell = cfg->room_size - 1; }
  if (cfg->env_kind == MG_ENV_UNLOCKPICKUP) { e->rule = RULE_PICKUP; e->rule_cell = (int)T_BOX; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_KEYCORRIDOR) { e->rule = RULE_PICKUP; e->rule_cell = (int)T_BALL; e->rule_div = 1; }
  if (cfg->env_kind == MG_ENV_OBSTRUCTEDMAZE) { e->rule = RULE_PICKUP; e->rule_cell = (int)T_BALL; e->rule_div = 1; }   // THE blue ball: mission id 0 = COLOR_NAMES[0]
  if (cfg->env_kind == MG_ENV_BLOCKEDUNLOCKPICKUP) { e->rule = RULE_PICKUP; e->rule_cell = (int)T_BOX; e->rule_div = 2; }

  // k_step compiles each level rule only into the variant of its rule group
  e->rule_group = (e->rule == RULE_PICKUPDESC || e->rule == RULE_OPENFRONT || e->rule == RULE_GOTO_BIG || e->rule == RULE_PUTNEXT || e->rule == RULE_OPENDOOR) ? GG_ROOMS
                : (e->rule == RULE_GOTO || e->rule == RULE_GOTOOBJ || e->rule == RULE_UNLOCK || e->rule == RULE_PICKUP || e->rule == RULE_PUTNEAR) ? GG_ROOMGRID
                : (e->rule == RULE_DYNOBS || e->rule == RULE_NONE || e->rule == RULE_SENTENCE) ? GG_NONE : GG_LIGHT;

  mg_env* env = e;   // for HIP_TRY
#define TRY_OR_FREE(call) do { hipError_t _e = (call); if (_e != hipSuccess) { int rc = fail(nullptr, MG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_e)); mg_destroy(e); return rc; } } while (0)
  TRY_OR_FREE(hipSetDevice(device));
  if (stream) { e->stream = (hipStream_t)stream; e->own_stream = false; }
  else {
    TRY_OR_FREE(hipStreamCreateWithFlags(&e->stream, cfg->null_stream_sync ? hipStreamDefault : hipStreamNonBlocking));
    e->own_stream = true;
  }
  {
    // the generator stream outranks the step stream: its workgroups are few and short, and whenever step workgroups retire the
    // dispatcher should hand the freed LDS / wave slots to a waiting refill first (a step launch at full residency otherwise
    // starves the refill until the launch drains: measured 161 us per LavaCrossing refill at equal priority)
    int lo = 0, hi = 0;
    TRY_OR_FREE(hipDeviceGetStreamPriorityRange(&lo, &hi));
    TRY_OR_FREE(hipStreamCreateWithPriority(&e->gen_stream, hipStreamNonBlocking, hi));
  }
  TRY_OR_FREE(hipEventCreate(&e->ev0));
  TRY_OR_FREE(hipEventCreate(&e->ev1));
  TRY_OR_FREE(hipEventCreateWithFlags(&e->ev_live, hipEventDisableTiming));
  TRY_OR_FREE(hipEventCreateWithFlags(&e->ev_fill, hipEventDisableTiming));
  for (int i = 0; i < QSETS; i++) {
    TRY_OR_FREE(hipEventCreateWithFlags(&e->ev_step[i], hipEventDisableTiming));
    TRY_OR_FREE(hipEventCreateWithFlags(&e->ev_gen[i], hipEventDisableTiming));
  }
  const size_t N = (size_t)e->N, R = (size_t)e->R;
  TRY_OR_FREE(dalloc(&e->grid, N * e->CS));
  TRY_OR_FREE(dalloc(&e->spare_grid, R * N * e->CS));
  TRY_OR_FREE(dalloc(&e->agent, N));
  TRY_OR_FREE(dalloc(&e->spare_agent, R * N));
  TRY_OR_FREE(dalloc(&e->rng, 5 * N));
  TRY_OR_FREE(dalloc(&e->rng_snap, R * 5 * N));
  TRY_OR_FREE(dalloc(&e->rng_tmp, 5 * N));
  TRY_OR_FREE(dalloc(&e->seeds, N));
  TRY_OR_FREE(dalloc(&e->mask, N));
  TRY_OR_FREE(dalloc(&e->actions, 8 * N));
  TRY_OR_FREE(dalloc(&e->aux, N));
  TRY_OR_FREE(dalloc(&e->spare_aux, R * N));
  TRY_OR_FREE(dalloc(&e->head, N));
  TRY_OR_FREE(dalloc(&e->tail, N));
  TRY_OR_FREE(dalloc(&e->claim, N));
  if (e->sentence) {
    TRY_OR_FREE(dalloc(&e->instr, N * INSTR_WORDS));
    TRY_OR_FREE(dalloc(&e->spare_instr, R * N * INSTR_WORDS));
    TRY_OR_FREE(dalloc(&e->gstate, N));
    TRY_OR_FREE(dalloc(&e->gsnap, R * N));
    TRY_OR_FREE(hipMemsetAsync(e->instr, 0, N * INSTR_WORDS * sizeof(uint64_t), e->stream));
    TRY_OR_FREE(hipMemsetAsync(e->spare_instr, 0, R * N * INSTR_WORDS * sizeof(uint64_t), e->stream));
    TRY_OR_FREE(hipMemsetAsync(e->gstate, 0, N * sizeof(uint32_t), e->stream));
    TRY_OR_FREE(hipMemsetAsync(e->gsnap, 0, R * N * sizeof(uint32_t), e->stream));
  }
  TRY_OR_FREE(hipMemsetAsync(e->aux, 0, N * sizeof(uint64_t), e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->spare_aux, 0, R * N * sizeof(uint64_t), e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->head, 0, N * sizeof(uint32_t), e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->tail, 0, N * sizeof(uint32_t), e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->claim, 0, N * sizeof(uint32_t), e->stream));
  {
    void* h = nullptr;
    TRY_OR_FREE(hipHostMalloc(&h, ERR_WORDS * sizeof(uint32_t), hipHostMallocMapped));
    e->err_host = (volatile uint32_t*)h;
    for (int k = 0; k < ERR_WORDS; k++) e->err_host[k] = 0u;
    TRY_OR_FREE(hipHostGetDevicePointer((void**)&e->err, h, 0));
  }
  TRY_OR_FREE(hipMemsetAsync(e->grid, 0, N * e->CS, e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->spare_grid, 0, R * N * e->CS, e->stream));
  TRY_OR_FREE(hipMemsetAsync(e->agent, 0, N * sizeof(uint64_t), e->stream));
  { int rc = alloc_obs(e); if (rc) { g_create_error = e->last_error; mg_destroy(e); return rc; } }
  if (e->sentence) {
    // 19 KB of buffered draws per generating wave: the direct generator launch (4 waves per workgroup) needs more than 64 KB of LDS
    const int need = (GEN_THREADS / 64) * gen_wave_lds_bytes(e->CS, 4864, true);
    TRY_OR_FREE(gen_max_lds_sentence_pcg(need)); TRY_OR_FREE(gen_max_lds_sentence_philox(need));
  }
#undef TRY_OR_FREE
  (void)env;
  // a usable state from the start: env i seeded with its global index (like reset(seed=env_index_base + i))
  std::vector<uint64_t> seeds(N);
  for (size_t i = 0; i < N; i++) seeds[i] = (uint64_t)(cfg->env_index_base + (long long)i);
  int rc = mg_reset(e, seeds.data(), nullptr);
  if (rc != MG_OK) { g_create_error = e->last_error; mg_destroy(e); return rc; }
  rc = mg_sync(e);
  if (rc != MG_OK) { g_create_error = e->last_error; mg_destroy(e); return rc; }
  if (e->sentence) {
    // the episodes drawn above are a convenience, not part of the env's history: LevelGen's locked_room (generator state carried from
    // episode to episode) starts out None for the caller's first reset(seed=...), as after gym.make
    (void)hipMemsetAsync(e->gstate, 0, N * sizeof(uint32_t), e->stream);
    (void)hipMemsetAsync(e->gsnap, 0, R * N * sizeof(uint32_t), e->stream);
    (void)hipStreamSynchronize(e->stream);
  }
  *out = e;
  return MG_OK;
}

int mg_set_obs_config(mg_env* e, const mg_config* cfg) {
  if (!e || !cfg) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  // everything but the observation part must be what the handle was created with
  mg_config want = e->cfg;
  want.obs_mode = cfg->obs_mode; want.agent_view_size = cfg->agent_view_size; want.tile_size = cfg->tile_size; want.rgb_highlight = cfg->rgb_highlight;
  want.no_death_mask = cfg->no_death_mask; want.death_cost = cfg->death_cost; want.traj_slots = cfg->traj_slots;
  {
    // field by field: a C caller that fills a stack struct member by member leaves the padding bytes undefined, so the
    // comparison must not look at them
    const mg_config& g = *cfg; const mg_config& w = want;
    const bool same = g.abi_version == w.abi_version && g.env_kind == w.env_kind && g.width == w.width && g.height == w.height &&
                      g.max_steps == w.max_steps && g.see_through_walls == w.see_through_walls && g.agent_view_size == w.agent_view_size &&
                      g.obs_mode == w.obs_mode && g.autoreset_mode == w.autoreset_mode && g.rng_mode == w.rng_mode && g.num_envs == w.num_envs &&
                      g.agent_start_x == w.agent_start_x && g.agent_start_y == w.agent_start_y && g.agent_start_dir == w.agent_start_dir &&
                      g.num_crossings == w.num_crossings && g.obstacle_type == w.obstacle_type && g.num_dists == w.num_dists &&
                      /* null_stream_sync: decided by the binding at create time */
                      g.strip2_row == w.strip2_row && g.no_death_mask == w.no_death_mask && g.death_cost == w.death_cost &&
                      g.room_size == w.room_size && g.random_length == w.random_length && g.env_index_base == w.env_index_base &&
                      g.tile_size == w.tile_size && g.rgb_highlight == w.rgb_highlight && g.spare_ring == w.spare_ring && g.traj_slots == w.traj_slots &&
                      g.babyai_done_actions == w.babyai_done_actions;
    if (!same)
      return fail(e, MG_ERR_INVALID, "set_obs_config: only obs_mode, agent_view_size, tile_size, rgb_highlight, no_death_mask, death_cost and traj_slots may change");
  }
  if (const char* bad = validate_obs_cfg(&want)) return fail(e, MG_ERR_INVALID, "%s", bad);
  // quiesce: every refill request served, both streams idle, device errors surfaced
  { int rc = flush_refills(e); if (rc) return rc; }
  { int rc = mg_sync(e); if (rc) return rc; }
  uint64_t c[4];
  { int rc = mg_get_counters(e, c); if (rc) return rc; }
  const mg_config old_cfg = e->cfg;
  e->stat_base[0] = c[1]; e->stat_base[1] = c[2]; e->stat_base[2] = c[3];
  free_obs(e);
  e->cfg = want;
  const char* bad = configure_obs(e);
  int rc = bad ? fail(e, MG_ERR_INVALID, "%s", bad) : alloc_obs(e);
  if (rc != MG_OK) {
    // back to the configuration that worked (its buffers are re-made; the env state was never touched)
    const std::string msg = e->last_error;
    free_obs(e);
    e->cfg = old_cfg;
    if (configure_obs(e) || alloc_obs(e) != MG_OK) return fail(e, MG_ERR_HIP, "set_obs_config failed (%s) and the previous configuration could not be restored", msg.c_str());
    e->last_error = msg;
    return rc;
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->live_gen) { int rc2 = rebuild_live_requests(e, nullptr); if (rc2) return rc2; }
  return MG_OK;
}

int mg_destroy(mg_env* e) {
  if (!e) return MG_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->gen_stream) (void)hipStreamSynchronize(e->gen_stream);
  (void)check_guards(e);
  for (const auto& a : e->allocs) (void)hipFree(a.base);
  e->allocs.clear();
  if (e->err_host) (void)hipHostFree((void*)e->err_host);
  if (e->h_scal) (void)hipHostFree((void*)e->h_scal);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->ev_live) (void)hipEventDestroy(e->ev_live);
  if (e->ev_fill) (void)hipEventDestroy(e->ev_fill);
  for (int i = 0; i < QSETS; i++) {
    if (e->ev_step[i]) (void)hipEventDestroy(e->ev_step[i]);
    if (e->ev_gen[i]) (void)hipEventDestroy(e->ev_gen[i]);
  }
  if (e->gen_stream) (void)hipStreamDestroy(e->gen_stream);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return MG_OK;
}

// draw every ring slot of the selected envs from their current stream position (head = 0, tail = R)
static int refill_whole_ring(mg_env* e, const uint8_t* d_mask, bool restore_gstate = true, hipStream_t st = nullptr) {
  if (e->live_gen) return MG_OK;
  if (!st) st = e->stream;
  const int tb = 256, nb = (e->N + tb - 1) / tb;
  if (uses_ring(e)) {
    hipLaunchKernelGGL(k_ring_restart, dim3(nb), dim3(tb), 0, st, e->head, e->tail, d_mask, (uint32_t)e->R, e->N,
                       (e->sentence && restore_gstate) ? e->gstate : nullptr, e->sentence ? e->gsnap : nullptr);
    HIP_TRY(e, hipGetLastError());
  }
  for (int s = 0; s < e->R; s++) { int rc = launch_generate(e, s, d_mask, st); if (rc) return rc; }
  return MG_OK;
}

int mg_reset(mg_env* e, const uint64_t* seeds, const uint8_t* mask) {
  if (!e) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  const int N = e->N;
  const uint8_t* d_mask = nullptr;
  bool async_fill = false;
  { int rc = await_ring_fill(e); if (rc) return rc; }     // (a ring redraw still running reads e->mask and e->rng)
  e->mask_sparse = false;
  if (mask) {
    HIP_TRY(e, hipMemcpyAsync(e->mask, mask, (size_t)N, hipMemcpyHostToDevice, e->stream));
    d_mask = e->mask;
    size_t set = 0;
    for (int i = 0; i < N; i++) set += mask[i] != 0;
    e->mask_sparse = set * 8 < (size_t)N;                  // fewer than an eighth of the envs: lanes would idle (launch_generate)
  }
  const int tb = 256, nb = (N + tb - 1) / tb;
  if (seeds) {
    // reset(seed=s): reseed, draw this episode, then the ring of spares (episodes 2, 3, ... of the same stream)
    { int rc = flush_refills(e); if (rc) return rc; }
    HIP_TRY(e, hipMemcpyAsync(e->seeds, seeds, (size_t)N * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    if (e->cfg.rng_mode == MG_RNG_PHILOX)
      hipLaunchKernelGGL(k_seed<PhiloxStream>, dim3(nb), dim3(tb), 0, e->stream, e->rng, e->seeds, d_mask, N);
    else
      hipLaunchKernelGGL(k_seed<Pcg64Stream>, dim3(nb), dim3(tb), 0, e->stream, e->rng, e->seeds, d_mask, N);
    HIP_TRY(e, hipGetLastError());
    if (e->sentence) {
      // LevelGen's generator state (locked_room) continues from the LIVE episode, not from the spares drawn ahead of it
      hipLaunchKernelGGL(k_gstate_restore, dim3(nb), dim3(tb), 0, e->stream, e->gstate, e->gsnap, e->head, d_mask, (uint32_t)e->R, N);
      HIP_TRY(e, hipGetLastError());
    }
    int rc = launch_generate(e, -1, d_mask);
    if (rc) return rc;
    if (uses_ring(e) && e->R > 1) async_fill = true;       // the R spare episodes behind it: after the observation (below)
    else {
      rc = refill_whole_ring(e, d_mask, false);
      if (rc) return rc;
    }
  } else if (e->live_gen) {
    // reset(): continue each env's stream from where its last step left it
    int rc = launch_generate(e, -1, d_mask);
    if (rc) return rc;
  } else {
    // reset(): continue each env's own stream == take the next pre-drawn spare
    hipLaunchKernelGGL(k_mark_pending, dim3(nb), dim3(tb), 0, e->stream, e->agent, d_mask, N);
    HIP_TRY(e, hipGetLastError());
  }
  StepParams P;
  fill_step_params(e, P, PHASE_OBSERVE);
  P.obs_mask = d_mask;       // a masked reset() leaves the other envs alone, autoreset-pending ones included
  int rc = launch_step(e, P);
  if (rc != MG_OK || !async_fill) return rc;
  // The observation of a seeded reset needs the live episode only.  The R spare episodes behind it (episodes 2, 3, ... of the same
  // streams: 128 generator launches, tens of milliseconds at 262 144 envs) are drawn on the generator stream AFTER the observation
  // launch -- so that they do not compete with it -- while the caller looks at the observation; the next launch that could take a
  // spare waits for them (await_ring_fill).
  HIP_TRY(e, hipEventRecord(e->ev_live, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->gen_stream, e->ev_live, 0));
  rc = refill_whole_ring(e, d_mask, false, e->gen_stream);
  if (rc) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_fill, e->gen_stream));
  e->fill_pending = true;
  return MG_OK;
}

int mg_step(mg_env* e, const void* actions, int dtype, int on_device) {
  if (!e || !actions) return MG_ERR_INVALID;
  if (dtype < MG_ACT_U8 || dtype > MG_ACT_I64) return fail(e, MG_ERR_INVALID, "bad action dtype");
  HIP_TRY(e, hipSetDevice(e->device));
  StepParams P;
  fill_step_params(e, P, PHASE_STEP);
  P.act_dtype = dtype;
  if (on_device) P.actions = actions;
  else {
    const size_t sz = (size_t)e->N * (dtype == MG_ACT_U8 ? 1 : dtype == MG_ACT_I32 ? 4 : 8);
    HIP_TRY(e, hipMemcpyAsync(e->actions, actions, sz, hipMemcpyHostToDevice, e->stream));
    P.actions = e->actions;
  }
  return launch_step(e, P);
}

int mg_rollout(mg_env* e, int T, uint64_t action_seed, int fused) {
  if (!e || T < 0) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  // step j of the call goes to trajectory slot (T-1-j) mod S: the last step always lands in slot 0 (mg_get_outputs),
  // slot k holds the step k calls before it.  fused: up to max_fused steps per k_step launch, else one launch per step.
  const int chunk = fused ? e->max_fused : 1;
  for (int j = 0; j < T;) {
    const int tc = std::min(chunk, T - j);
    StepParams P;
    fill_step_params(e, P, PHASE_STEP);
    P.act_src = ACT_SRC_PHILOX; P.action_seed = action_seed; P.t0 = e->t; P.T = tc;
    P.slot0 = fused ? (T - 1 - j) % e->S : 0;
    e->t += (uint32_t)tc;
    int rc = launch_step(e, P);
    if (rc) return rc;
    j += tc;
  }
  return MG_OK;
}

int mg_rollout_block(mg_env* e, int T, uint64_t action_seed, int slot0) {
  if (!e) return MG_ERR_INVALID;
  if (T < 1 || T > e->max_fused || slot0 < T - 1 || slot0 >= e->S)
    return fail(e, MG_ERR_INVALID, "rollout_block: 1 <= T <= max_fused_steps (%d) and T - 1 <= slot0 < traj_slots (%d)", e->max_fused, e->S);
  HIP_TRY(e, hipSetDevice(e->device));
  StepParams P;
  fill_step_params(e, P, PHASE_STEP);
  P.act_src = ACT_SRC_PHILOX; P.action_seed = action_seed; P.t0 = e->t; P.T = T; P.slot0 = slot0;
  e->t += (uint32_t)T;
  return launch_step(e, P);
}

int mg_step_many(mg_env* e, const uint8_t* actions, int T, int on_device) {
  if (!e || !actions || T < 0) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  for (int j = 0; j < T;) {
    const int tc = std::min(std::min(e->max_fused, on_device ? e->max_fused : 8), T - j);   // host staging holds 8 steps
    StepParams P;
    fill_step_params(e, P, PHASE_STEP);
    P.act_dtype = MG_ACT_U8; P.T = tc; P.slot0 = (T - 1 - j) % e->S;
    if (on_device) P.actions = actions + (size_t)j * N;
    else {
      HIP_TRY(e, hipMemcpyAsync(e->actions, actions + (size_t)j * N, (size_t)tc * N, hipMemcpyHostToDevice, e->stream));
      P.actions = e->actions;
    }
    int rc = launch_step(e, P);
    if (rc) return rc;
    j += tc;
  }
  return MG_OK;
}

int mg_get_outputs(mg_env* e, mg_outputs* o) {
  if (!e || !o) return MG_ERR_INVALID;
  o->obs = e->out; o->reward = (double*)(e->out + e->off_reward); o->terminated = e->out + e->off_term;
  o->truncated = e->out + e->off_trunc; o->direction = e->out + e->off_dir; o->mission_id = (uint16_t*)(e->out + e->off_mission);
  o->obs_bytes_per_env = e->obs_bytes; o->num_envs = e->N;
  o->action = e->out + e->off_action; o->traj_slots = e->S; o->slot_bytes = (int64_t)e->slot_bytes; o->record_bytes = (int64_t)e->record_bytes;
  o->max_fused_steps = e->max_fused;
  o->sentence = e->sentence ? (uint64_t*)(e->out + e->off_sentence) : nullptr;
  o->scalar_stride = (int64_t)sizeof(mg_step_scalars);
  return MG_OK;
}

int mg_copy_sentence(mg_env* e, int slot, uint64_t* out) {
  if (!e || !out || slot < 0 || slot >= e->S) return MG_ERR_INVALID;
  if (!e->sentence) return fail(e, MG_ERR_INVALID, "mg_copy_sentence: not a sentence level (missions are mission_id table entries)");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpyAsync(out, e->out + (size_t)slot * e->slot_bytes + e->off_sentence, (size_t)e->N * 16, hipMemcpyDeviceToHost, e->stream));
  return check_device_errors(e);
}

int mg_copy_outputs(mg_env* e, uint8_t* obs, double* reward, uint8_t* term, uint8_t* trunc, uint8_t* dir, uint16_t* mission) {
  return mg_copy_slot(e, 0, obs, reward, term, trunc, dir, mission, nullptr);
}

int mg_copy_slot(mg_env* e, int slot, uint8_t* obs, double* reward, uint8_t* term, uint8_t* trunc, uint8_t* dir, uint16_t* mission,
                 uint8_t* action) {
  if (!e || slot < 0 || slot >= e->S) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  const uint8_t* b = e->out + (size_t)slot * e->slot_bytes;
  if (obs) HIP_TRY(e, hipMemcpyAsync(obs, b, N * e->obs_bytes, hipMemcpyDeviceToHost, e->stream));
  const bool scal = reward || term || trunc || dir || mission || action;
  if (scal) {
    // the scalars travel as they lie -- (N) x mg_step_scalars -- into a pinned staging buffer and are dealt out on the host
    if (!e->h_scal) HIP_TRY(e, hipHostMalloc((void**)&e->h_scal, N * sizeof(mg_step_scalars), hipHostMallocDefault));
    HIP_TRY(e, hipMemcpyAsync(e->h_scal, b + e->off_reward, N * sizeof(mg_step_scalars), hipMemcpyDeviceToHost, e->stream));
  }
  { int rc = check_device_errors(e); if (rc) return rc; }        // (waits for the stream)
  if (scal) {
    const mg_step_scalars* sc = e->h_scal;
    for (size_t i = 0; i < N; i++) {
      if (reward) reward[i] = sc[i].reward;
      if (term) term[i] = sc[i].terminated;
      if (trunc) trunc[i] = sc[i].truncated;
      if (dir) dir[i] = sc[i].direction;
      if (mission) mission[i] = sc[i].mission_id;
      if (action) action[i] = sc[i].action;
    }
  }
  return MG_OK;
}

int mg_sync(mg_env* e) {
  if (!e) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  // the generator stream runs ahead-of-need work for the episodes to come: a sync closes the open batch (its refill is launched now)
  // and covers the generator stream too, so that a timed region bracketed by mg_sync pays for every episode consumed inside it --
  // the rings are full again when it returns, whatever their depth
  { int rc = close_batch(e); if (rc) return rc; }
  if (e->gen_stream) HIP_TRY(e, wait_stream(e->gen_stream));
  { int rc = check_device_errors(e); if (rc) return rc; }
  return check_guards(e);
}

// staging buffers of the state exchange, allocated on first use: (N, W, H, 3) u8 + (N, 8) i32
static int state_staging(mg_env* e) {
  if (e->st_grid) return MG_OK;
  const size_t N = (size_t)e->N;
  HIP_TRY(e, env_alloc(e, (void**)&e->st_grid, N * e->cells * 3, "st_grid"));
  HIP_TRY(e, env_alloc(e, (void**)&e->st_agent, N * 8 * sizeof(int32_t), "st_agent"));
  return MG_OK;
}

int mg_get_state(mg_env* e, uint8_t* grid, int32_t* agent) {
  if (!e || !grid || !agent) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  { int rc = state_staging(e); if (rc) return rc; }
  const size_t N = (size_t)e->N, total = N * e->cells;
  const int tb = 256;
  hipLaunchKernelGGL(k_state_encode, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, e->stream, e->grid, e->agent, e->st_grid, e->st_agent,
                     e->N, e->W, e->H, e->CS);
  HIP_TRY(e, hipGetLastError());
  HIP_TRY(e, hipMemcpyAsync(grid, e->st_grid, total * 3, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipMemcpyAsync(agent, e->st_agent, N * 8 * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return MG_OK;
}

int mg_set_state(mg_env* e, const uint8_t* grid, const int32_t* agent) {
  if (!e || !grid || !agent) return MG_ERR_INVALID;
  if (e->sentence) return fail(e, MG_ERR_INVALID, "set_state: the exchanged state does not carry the instruction tree and object identities of a sentence level");
  HIP_TRY(e, hipSetDevice(e->device));
  { int rc = state_staging(e); if (rc) return rc; }
  const size_t N = (size_t)e->N, total = N * e->cells;
  const int tb = 256;
  HIP_TRY(e, hipMemcpyAsync(e->st_grid, grid, total * 3, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->st_agent, agent, N * 8 * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->err_host[4] = 0u;
  hipLaunchKernelGGL(k_state_decode, dim3((unsigned)((total + tb - 1) / tb)), dim3(tb), 0, e->stream, e->st_grid, e->st_agent, e->grid, e->agent,
                     e->claim_bad(), e->N, e->W, e->H, e->CS);
  HIP_TRY(e, hipGetLastError());
  if (e->goto_kind || e->live_gen) {
    // GoToInstr's tracked positions / the obstacle list are re-found from the grid (not part of the exchanged state)
    hipLaunchKernelGGL(k_aux_rebuild, dim3((e->N + tb - 1) / tb), dim3(tb), 0, e->stream, e->grid, e->agent, e->aux, e->N, e->cells, e->CS,
                       e->live_gen ? 2 : ((e->rule == RULE_GOTO_BIG || e->rule == RULE_PUTNEXT) ? 3 : e->rule == RULE_OPENDOOR ? 4 : 1), e->rule_div, e->rule_cell);
    HIP_TRY(e, hipGetLastError());
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->err_host[4]) return fail(e, MG_ERR_INVALID, "an agent record is out of range (position outside the grid, dir > 3, step_count or mission id too large)");
  return MG_OK;
}

// ---- lossless checkpoint: everything a live handle's future depends on ------------------------------------------------
// (mg_get_state / mg_set_state exchange the REFERENCE encoding -- Grid.encode() + the agent tuple -- which, like Grid.decode in the
// reference, cannot carry what a box hides or an instruction tree; this pair carries the library's own state verbatim.)
struct StateHeader { uint32_t magic, version; int32_t N, CS, W, H, env_kind, sentence; uint32_t t; uint32_t pad; uint64_t env_steps; };
constexpr uint32_t STATE_MAGIC = 0x5453474Du;   // "MGST"
static size_t state_bytes(const mg_env* e) {
  const size_t N = (size_t)e->N;
  return sizeof(StateHeader) + N * e->CS + N * 8 + N * 8 + 5 * N * 8 + (e->sentence ? N * INSTR_WORDS * 8 + N * 4 : 0);
}
int mg_state_size(mg_env* e, int64_t* bytes) {
  if (!e || !bytes) return MG_ERR_INVALID;
  *bytes = (int64_t)state_bytes(e);
  return MG_OK;
}
int mg_save_state(mg_env* e, void* buf, int64_t bytes) {
  if (!e || !buf) return MG_ERR_INVALID;
  if (bytes != (int64_t)state_bytes(e)) return fail(e, MG_ERR_INVALID, "save_state: buffer of %lld bytes, mg_state_size says %zu", (long long)bytes, state_bytes(e));
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  { int rc = flush_refills(e); if (rc) return rc; }
  { int rc = mg_sync(e); if (rc) return rc; }
  uint8_t* p = (uint8_t*)buf;
  StateHeader h{ STATE_MAGIC, 1u, e->N, e->CS, e->W, e->H, e->cfg.env_kind, e->sentence ? 1 : 0, e->t, 0u, e->env_steps };
  memcpy(p, &h, sizeof h); p += sizeof h;
  HIP_TRY(e, hipMemcpy(p, e->grid, N * e->CS, hipMemcpyDeviceToHost)); p += N * e->CS;
  HIP_TRY(e, hipMemcpy(p, e->agent, N * 8, hipMemcpyDeviceToHost)); p += N * 8;
  HIP_TRY(e, hipMemcpy(p, e->aux, N * 8, hipMemcpyDeviceToHost)); p += N * 8;
  // every env's stream position "now" = the state before its next unconsumed spare was drawn (as mg_get_rng), SoA [5][N]
  const uint64_t* src = e->rng;
  if (uses_ring(e)) {
    const int tb = 256, nb = (e->N + tb - 1) / tb;
    hipLaunchKernelGGL(k_gather_rng, dim3(nb), dim3(tb), 0, e->stream, e->rng_snap, e->head, (uint32_t)(e->R - 1), e->rng_tmp, e->N);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    src = e->rng_tmp;
  }
  HIP_TRY(e, hipMemcpy(p, src, 5 * N * 8, hipMemcpyDeviceToHost)); p += 5 * N * 8;
  if (e->sentence) {
    HIP_TRY(e, hipMemcpy(p, e->instr, N * INSTR_WORDS * 8, hipMemcpyDeviceToHost)); p += N * INSTR_WORDS * 8;
    // LevelGen's generator state as it was before the next unconsumed spare (like the stream position)
    std::vector<uint32_t> head(N), snap((size_t)e->R * N);
    HIP_TRY(e, hipMemcpy(head.data(), e->head, N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(snap.data(), e->gsnap, (size_t)e->R * N * 4, hipMemcpyDeviceToHost));
    uint32_t* g = (uint32_t*)p;
    for (size_t n = 0; n < N; n++) g[n] = snap[(size_t)(head[n] & (uint32_t)(e->R - 1)) * N + n];
    p += N * 4;
  }
  return MG_OK;
}
int mg_load_state(mg_env* e, const void* buf, int64_t bytes) {
  if (!e || !buf) return MG_ERR_INVALID;
  if (bytes != (int64_t)state_bytes(e)) return fail(e, MG_ERR_INVALID, "load_state: %lld bytes for a handle whose state is %zu bytes", (long long)bytes, state_bytes(e));
  const uint8_t* p = (const uint8_t*)buf;
  StateHeader h;
  memcpy(&h, p, sizeof h); p += sizeof h;
  if (h.magic != STATE_MAGIC || h.version != 1u || h.N != e->N || h.CS != e->CS || h.W != e->W || h.H != e->H || h.env_kind != e->cfg.env_kind ||
      h.sentence != (e->sentence ? 1 : 0))
    return fail(e, MG_ERR_INVALID, "load_state: the checkpoint was taken from a different configuration (level, grid or batch size)");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  { int rc = flush_refills(e); if (rc) return rc; }
  { int rc = mg_sync(e); if (rc) return rc; }
  HIP_TRY(e, hipMemcpy(e->grid, p, N * e->CS, hipMemcpyHostToDevice)); p += N * e->CS;
  HIP_TRY(e, hipMemcpy(e->agent, p, N * 8, hipMemcpyHostToDevice)); p += N * 8;
  HIP_TRY(e, hipMemcpy(e->aux, p, N * 8, hipMemcpyHostToDevice)); p += N * 8;
  HIP_TRY(e, hipMemcpy(e->rng, p, 5 * N * 8, hipMemcpyHostToDevice)); p += 5 * N * 8;
  if (e->sentence) {
    HIP_TRY(e, hipMemcpy(e->instr, p, N * INSTR_WORDS * 8, hipMemcpyHostToDevice)); p += N * INSTR_WORDS * 8;
    HIP_TRY(e, hipMemcpy(e->gstate, p, N * 4, hipMemcpyHostToDevice)); p += N * 4;
  }
  e->t = h.t; e->env_steps = h.env_steps;
  if (e->live_gen) {
    const uint64_t* rec = (const uint64_t*)((const uint8_t*)buf + sizeof(StateHeader) + N * e->CS);
    int rc = rebuild_live_requests(e, rec);
    if (rc) return rc;
  }
  // the spare episodes are not part of a checkpoint: they are re-drawn from the restored stream positions (as mg_set_rng does)
  { int rc = refill_whole_ring(e, nullptr, false); if (rc) return rc; }
  return mg_sync(e);
}

int mg_get_rng(mg_env* e, uint64_t* out) {
  if (!e || !out) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  { int rc = flush_refills(e); if (rc) return rc; }
  // the reference env's stream position "now" is the state BEFORE its next unconsumed spare episode was drawn
  const uint64_t* src = e->rng;
  if (uses_ring(e)) {
    const int tb = 256, nb = (e->N + tb - 1) / tb;
    hipLaunchKernelGGL(k_gather_rng, dim3(nb), dim3(tb), 0, e->stream, e->rng_snap, e->head, (uint32_t)(e->R - 1), e->rng_tmp, e->N);
    HIP_TRY(e, hipGetLastError());
    src = e->rng_tmp;
  }
  std::vector<uint64_t> soa(5 * N);
  HIP_TRY(e, hipMemcpyAsync(soa.data(), src, soa.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  for (size_t n = 0; n < N; n++) for (int k = 0; k < 5; k++) out[n * 5 + k] = soa[k * N + n];
  return MG_OK;
}

int mg_set_rng(mg_env* e, const uint64_t* in) {
  if (!e || !in) return MG_ERR_INVALID;
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t N = (size_t)e->N;
  { int rc = flush_refills(e); if (rc) return rc; }
  std::vector<uint64_t> soa(5 * N);
  for (size_t n = 0; n < N; n++) for (int k = 0; k < 5; k++) soa[k * N + n] = in[n * 5 + k];
  HIP_TRY(e, hipMemcpyAsync(e->rng, soa.data(), soa.size() * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  // re-draw every spare from the injected position
  return refill_whole_ring(e, nullptr);
}

int mg_timer_start(mg_env* e) {
  if (!e) return MG_ERR_INVALID;
  HIP_TRY(e, hipEventRecord(e->ev0, e->stream));
  return MG_OK;
}

int mg_timer_stop(mg_env* e, float* ms) {
  if (!e || !ms) return MG_ERR_INVALID;
  HIP_TRY(e, hipEventRecord(e->ev1, e->stream));
  HIP_TRY(e, wait_event(e->ev1));
  e->burst_bytes = 0;                                      // the step stream is idle
  HIP_TRY(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
  return MG_OK;
}

int mg_ring_depth(mg_env* e) { return !e ? MG_ERR_INVALID : (e->live_gen ? 0 : e->R); }

int mg_get_counters(mg_env* e, uint64_t out[4]) {
  if (!e || !out) return MG_ERR_INVALID;
  std::vector<uint64_t> c(e->ncounters);
  HIP_TRY(e, hipMemcpyAsync(c.data(), e->counters, c.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  const size_t groups = (size_t)e->nwaves, g0 = (size_t)STAT_EPISODES + groups;
  out[0] = e->env_steps; out[1] = out[2] = out[3] = 0;
  for (size_t k = 0; k < groups; k++) out[1] += c[STAT_EPISODES + k];
  for (size_t k = 0; k < STAT_GEN_SLOTS; k++) { out[2] += c[g0 + 2 * k]; out[3] += c[g0 + 2 * k + 1]; }
  out[1] += e->stat_base[0]; out[2] += e->stat_base[1]; out[3] += e->stat_base[2];
  return MG_OK;
}

#if defined(MG_DEBUG_TIMING) || defined(MG_ATTRIBUTION) || defined(MG_GEN_ATTR)
MG_API int mg_debug_stamps(mg_env* e, uint64_t out[12]) {
  HIP_TRY(e, hipMemcpyAsync(out, e->counters + 4, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return MG_OK;
}
#endif

}  // extern "C"
