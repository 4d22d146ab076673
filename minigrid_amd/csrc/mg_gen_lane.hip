// k_refill_lane (mg_genlane.h): one lane per episode, numpy PCG64 and Philox streams
#define MG_GEN_TU_ONLY 1
#include "mg_launch.h"
#include "mg_genlane.h"

namespace mg {

void launch_refill_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  if (philox) hipLaunchKernelGGL((k_refill_lane<PhiloxStream>), grid, dim3(64), lds, st, A);
  else hipLaunchKernelGGL((k_refill_lane<Pcg64Stream>), grid, dim3(64), lds, st, A);
}
void launch_generate_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A) {
  if (philox) hipLaunchKernelGGL((k_generate_lane<PhiloxStream>), grid, dim3(64), lds, st, A);
  else hipLaunchKernelGGL((k_generate_lane<Pcg64Stream>), grid, dim3(64), lds, st, A);
}
hipError_t refill_lane_max_lds(int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)k_refill_lane<Pcg64Stream>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)k_refill_lane<PhiloxStream>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace mg
