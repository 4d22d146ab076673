// lane-per-episode generator kernels (mg_genlane.h, mg_gen_lane_tu.inc): the generator functions FN of this unit (lane_fn_of_kind)
#define MG_LANE_TU_NAME a
#define MG_LANE_TU_FNS(X) X(1) X(3) X(4) X(6) X(7) X(19)
#include "mg_gen_lane_tu.inc"
