// k_refill_lane (mg_genlane.h): one lane per episode, numpy PCG64 and Philox streams
#define MG_GEN_TU_ONLY 1
#include "mg_launch.h"
#include "mg_genlane.h"

namespace mg {

// kernel FN = lane_fn_of_kind(level) (mg_genlane.h): the product build instantiates FN 2 (the Unlock family) and FN 5 (KeyCorridor), the wide
// variant build every generator function
#define MG_LANE_CASE(K, n) case n: if (philox) hipLaunchKernelGGL((K<PhiloxStream, n>), grid, dim3(64), lds, st, A); \
                                   else hipLaunchKernelGGL((K<Pcg64Stream, n>), grid, dim3(64), lds, st, A); return true;
#if MG_LANE_WIDE
#define MG_LANE_CASES(K) MG_LANE_CASE(K, 1) MG_LANE_CASE(K, 2) MG_LANE_CASE(K, 3) MG_LANE_CASE(K, 4) MG_LANE_CASE(K, 5) MG_LANE_CASE(K, 6) \
  MG_LANE_CASE(K, 7) MG_LANE_CASE(K, 8) MG_LANE_CASE(K, 9) MG_LANE_CASE(K, 10) MG_LANE_CASE(K, 11) MG_LANE_CASE(K, 12) MG_LANE_CASE(K, 13) \
  MG_LANE_CASE(K, 14) MG_LANE_CASE(K, 16) MG_LANE_CASE(K, 17) MG_LANE_CASE(K, 18) MG_LANE_CASE(K, 19) \
  MG_LANE_CASE(K, 136) MG_LANE_CASE(K, 137) MG_LANE_CASE(K, 138) MG_LANE_CASE(K, 139) MG_LANE_CASE(K, 140) MG_LANE_CASE(K, 141) MG_LANE_CASE(K, 142) \
  MG_LANE_CASE(K, 143) MG_LANE_CASE(K, 144) MG_LANE_CASE(K, 145)
#else
#define MG_LANE_CASES(K) MG_LANE_CASE(K, 2) MG_LANE_CASE(K, 5)
#endif
static bool launch_refill_lane_fn(int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  switch (fn) { MG_LANE_CASES(k_refill_lane) default: return false; }
}
static bool launch_generate_lane_fn(int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  switch (fn) { MG_LANE_CASES(k_generate_lane) default: return false; }
}

void launch_refill_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  if (launch_refill_lane_fn(lane_fn_of_kind(A.gp.kind), philox, grid, lds, st, A)) return;
  if (philox) hipLaunchKernelGGL((k_refill_lane<PhiloxStream>), grid, dim3(64), lds, st, A);
  else hipLaunchKernelGGL((k_refill_lane<Pcg64Stream>), grid, dim3(64), lds, st, A);
}
void launch_generate_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  if (launch_generate_lane_fn(lane_fn_of_kind(A.gp.kind), philox, grid, lds, st, A)) return;
  if (philox) hipLaunchKernelGGL((k_generate_lane<PhiloxStream>), grid, dim3(64), lds, st, A);
  else hipLaunchKernelGGL((k_generate_lane<Pcg64Stream>), grid, dim3(64), lds, st, A);
}
}  // namespace mg
