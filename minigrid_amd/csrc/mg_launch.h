// mg_launch.h — host-side launch entry points of the kernel translation units.  The k_step instantiations of one rule group and
// the generator kernels of one generator group each live in their own .hip file (mg_step_g*.hip, mg_gen_g*.hip, bodies in
// mg_step_tu.inc / mg_gen_tu.inc) so that the library builds in parallel (minigrid_amd/build.py); mg_api.hip only sees these.
#pragma once
#include <hip/hip_runtime.h>

#ifndef MG_GEN_TU_ONLY
#include "mg_step.h"
#endif
#ifndef MG_STEP_TU_ONLY
#include "mg_genk.h"
#endif

namespace mg {

#ifndef MG_GEN_TU_ONLY
// Launch the k_step<mode, GG, lpe> instantiation of rule group GG; false if the group's TU has no such variant.
#define MG_DECL_STEP_TU(NAME)                                                                                              \
  bool launch_step_##NAME(int mode, int lpe, dim3 grid, size_t lds, hipStream_t st, const StepParams& P);      \
  hipError_t step_max_lds_##NAME(int bytes);                                                                               \
  void launch_roll_##NAME(bool full, dim3 grid, int nw, size_t lds, hipStream_t st, const StepParams& P);                             \
  hipError_t roll_max_lds_##NAME(int bytes);
MG_DECL_STEP_TU(none) MG_DECL_STEP_TU(light) MG_DECL_STEP_TU(roomgrid) MG_DECL_STEP_TU(rooms)
#undef MG_DECL_STEP_TU
// (the sentence levels' k_roll7: the verifier inside the step loop; mg_step_sentence.hip)
void launch_roll_sentence(bool full, dim3 grid, int nw, size_t lds, hipStream_t st, const StepParams& P);
hipError_t roll_max_lds_sentence(int bytes);
// ONE-RULE UNITS (round 6; mg_step_<name>.hip = k_roll7<GG_RULE(group, rule)>, the group's other rules compiled out): name, rule group, rule, has the STAGED split
#define MG_ONE_RULE_UNITS(X) \
  X(goto, GG_ROOMGRID, RULE_GOTO, 0) \
  X(pickup, GG_ROOMGRID, RULE_PICKUP, 0) \
  X(unlock, GG_ROOMGRID, RULE_UNLOCK, 0) \
  X(gotoobj, GG_ROOMGRID, RULE_GOTOOBJ, 0) \
  X(putnear, GG_ROOMGRID, RULE_PUTNEAR, 0) \
  X(gotobig, GG_ROOMS, RULE_GOTO_BIG, 1) \
  X(pickupdesc, GG_ROOMS, RULE_PICKUPDESC, 1) \
  X(openfront, GG_ROOMS, RULE_OPENFRONT, 1) \
  X(putnext, GG_ROOMS, RULE_PUTNEXT, 1) \
  X(opendoor, GG_ROOMS, RULE_OPENDOOR, 1) \
  X(fetch, GG_LIGHT, RULE_FETCH, 0) \
  X(gotodoor, GG_LIGHT, RULE_GOTODOOR, 0) \
  X(redblue, GG_LIGHT, RULE_REDBLUE, 0) \
  X(memory, GG_LIGHT, RULE_MEMORY, 1)
#define MG_UNIT_DECL(NAME, GROUP, RULE, STAGED) \
  void launch_roll_##NAME(bool full, dim3 grid, int nw, size_t lds, hipStream_t st, const StepParams& P); \
  hipError_t roll_max_lds_##NAME(int bytes);
MG_ONE_RULE_UNITS(MG_UNIT_DECL)
#undef MG_UNIT_DECL
// (DynamicObstacles' k_roll7: the stream draws of its step() and reset() inside the step loop; mg_step_dynobs.hip)
void launch_roll_dynobs(bool philox, dim3 grid, int nw, size_t lds, hipStream_t st, const StepParams& P);
hipError_t roll_max_lds_dynobs(int bytes);
#endif

#ifndef MG_STEP_TU_ONLY
// k_generate / k_refill of one generator group and stream kind (numpy PCG64 | Philox): one translation unit per kernel
#define MG_DECL_GEN_TU(NAME)                                                                       \
  void launch_generate_##NAME(dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);            \
  void launch_refill_##NAME(dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);              \
  hipError_t gen_max_lds_##NAME(int bytes);
MG_DECL_GEN_TU(light_pcg) MG_DECL_GEN_TU(roomgrid_pcg) MG_DECL_GEN_TU(rooms_pcg) MG_DECL_GEN_TU(sentence_pcg)
MG_DECL_GEN_TU(light_philox) MG_DECL_GEN_TU(roomgrid_philox) MG_DECL_GEN_TU(rooms_philox) MG_DECL_GEN_TU(sentence_philox)
#undef MG_DECL_GEN_TU
// k_refill_lane (mg_genlane.h): one lane per episode, for the single-room levels (lane_gen_kind)
// (false: the build holds no such kernel for the level)
bool launch_refill_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_refill_lane_packed(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_generate_lane(bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
// the lane kernels' translation units (mg_gen_lane_tu.inc): which = 0 per-segment refill | 1 packed refill | 2 direct generation
bool launch_lane_main(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_lane_a(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_lane_b(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_lane_c(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
bool launch_lane_d(int which, int fn, bool philox, dim3 grid, size_t lds, hipStream_t st, const GenArgs& A);
// MultiRoom's grid-free lane kernels (mg_genmr.h, unit a): which = 1 packed refill | 2 direct generation; the launch sizes its own LDS
bool launch_lane_mr(int which, bool philox, dim3 grid, hipStream_t st, const GenArgs& A);
// dispatch on (generator group, stream kind)
#define MG_GEN_DISPATCH(FN, gg, philox, ...)                                                       \
  do {                                                                                             \
    if (philox) { if (gg == GG_LIGHT) FN##light_philox(__VA_ARGS__); else if (gg == GG_ROOMGRID) FN##roomgrid_philox(__VA_ARGS__);   \
                  else if (gg == GG_ROOMS) FN##rooms_philox(__VA_ARGS__); else FN##sentence_philox(__VA_ARGS__); }                   \
    else { if (gg == GG_LIGHT) FN##light_pcg(__VA_ARGS__); else if (gg == GG_ROOMGRID) FN##roomgrid_pcg(__VA_ARGS__);               \
           else if (gg == GG_ROOMS) FN##rooms_pcg(__VA_ARGS__); else FN##sentence_pcg(__VA_ARGS__); }                                \
  } while (0)
#endif

}  // namespace mg
