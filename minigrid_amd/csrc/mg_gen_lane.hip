// The lane-per-episode generator kernels (mg_genlane.h), numpy PCG64 and Philox streams: this unit holds FN 0 (the single-room levels behind one
// run-time switch), FN 2 (the Unlock family) and FN 5 (KeyCorridor) with all three kernels -- the levels whose REFILL runs on lanes --, the request scan
// of the packed refill, and the dispatch over the sibling units mg_gen_lane_{a,b,c,d}.hip (every other generator function: direct generation only --
// reset(seed) and its ring fill draw one episode per env and slot, i.e. every lane is busy, which is where lanes win on every level).
#define MG_LANE_TU_NAME main
#define MG_LANE_TU_REFILL 1
#define MG_LANE_TU_FNS(X) X(0) X(2) X(5)
#include "mg_gen_lane_tu.inc"
#include "mg_genmr.h"

namespace mg {

static bool launch_lane_any(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  return launch_lane_main(which, fn, philox, grid, lds, st, A) || launch_lane_a(which, fn, philox, grid, lds, st, A) ||
         launch_lane_b(which, fn, philox, grid, lds, st, A) || launch_lane_c(which, fn, philox, grid, lds, st, A) ||
         launch_lane_d(which, fn, philox, grid, lds, st, A);
}
// per-segment refill (A.wps wavefronts per request segment)
bool launch_refill_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  return launch_lane_any(0, lane_fn_of_kind(A.gp.kind), philox, grid, lds, st, A);
}
// packed refill: the batch's requests numbered across the segments first (A.seg_off, written here), A.lpw busy lanes per wavefront
bool launch_refill_lane_packed(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  hipLaunchKernelGGL(k_seg_scan<0>, dim3(1), dim3(SEG_SCAN_THREADS), SEG_SCAN_THREADS * sizeof(uint32_t), st, A.seg_count, const_cast<uint32_t*>(A.seg_off), A.nseg);
  if (mr_lanes_ok(A.gp, A.CS)) return launch_lane_mr(1, philox, grid, st, A);         // MultiRoom: no grid per lane (mg_genmr.h)
  return launch_lane_any(1, lane_fn_of_kind(A.gp.kind), philox, grid, lds, st, A);
}
bool launch_generate_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  if (mr_lanes_ok(A.gp, A.CS)) return launch_lane_mr(2, philox, grid, st, A);
  return launch_lane_any(2, lane_fn_of_kind(A.gp.kind), philox, grid, lds, st, A);
}

}  // namespace mg
