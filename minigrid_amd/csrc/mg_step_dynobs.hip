// k_roll7 with DynamicObstacles' stream draws (obstacle moves, resets) inside the step loop: rule group GG_DYNOBS, one instantiation per
// stream kind (numpy PCG64 | Philox) and store kind (mg_roll.h, mg_dynobs.h)
#define MG_STEP_TU_ONLY 1
#include "mg_launch.h"
#include "mg_roll.h"

namespace mg {

void launch_roll_dynobs(bool philox, dim3 grid, int nw, size_t lds, hipStream_t st, const StepParams& P) {
  if (philox) {
    if (P.nt) hipLaunchKernelGGL((k_roll7<GG_DYNOBS, false, true, PhiloxStream>), grid, dim3(64 * nw), lds, st, P);
    else hipLaunchKernelGGL((k_roll7<GG_DYNOBS, false, false, PhiloxStream>), grid, dim3(64 * nw), lds, st, P);
  } else {
    if (P.nt) hipLaunchKernelGGL((k_roll7<GG_DYNOBS, false, true, Pcg64Stream>), grid, dim3(64 * nw), lds, st, P);
    else hipLaunchKernelGGL((k_roll7<GG_DYNOBS, false, false, Pcg64Stream>), grid, dim3(64 * nw), lds, st, P);
  }
}
hipError_t roll_max_lds_dynobs(int bytes) {
  const void* fns[] = { (const void*)k_roll7<GG_DYNOBS, false, false, PhiloxStream>, (const void*)k_roll7<GG_DYNOBS, false, true, PhiloxStream>,
                        (const void*)k_roll7<GG_DYNOBS, false, false, Pcg64Stream>, (const void*)k_roll7<GG_DYNOBS, false, true, Pcg64Stream> };
  for (const void* f : fns) { hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); if (e != hipSuccess) return e; }
  return hipSuccess;
}

}  // namespace mg
