// lane-per-episode generator kernels (mg_genlane.h, mg_gen_lane_tu.inc): the generator functions FN of this unit (lane_fn_of_kind)
#define MG_LANE_TU_NAME a
#define MG_LANE_TU_FNS(X) X(1) X(3) X(4) X(6) X(7) X(19)
#include "mg_gen_lane_tu.inc"

// MultiRoom without a grid per lane (mg_genmr.h): which = 1 packed refill | 2 direct generation
#include "mg_genmr.h"
namespace mg {
bool launch_lane_mr(int which, bool philox, dim3 grid, hipStream_t st, const GenArgs& A) {
  const size_t lds = (size_t)mr_lane_lds_bytes(A.gp.H);
  if (which == 1) {
    if (philox) hipLaunchKernelGGL((k_refill_lane_packed_mr<PhiloxStream>), grid, dim3(64), lds, st, A);
    else hipLaunchKernelGGL((k_refill_lane_packed_mr<Pcg64Stream>), grid, dim3(64), lds, st, A);
    return true;
  }
  if (which == 2) {
    if (philox) hipLaunchKernelGGL((k_generate_lane_mr<PhiloxStream>), grid, dim3(64), lds, st, A);
    else hipLaunchKernelGGL((k_generate_lane_mr<Pcg64Stream>), grid, dim3(64), lds, st, A);
    return true;
  }
  return false;
}
}  // namespace mg
