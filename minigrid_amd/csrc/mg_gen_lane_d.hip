// lane-per-episode generator kernels (mg_genlane.h, mg_gen_lane_tu.inc): the generator functions FN of this unit (lane_fn_of_kind)
#define MG_LANE_TU_NAME d
#define MG_LANE_TU_FNS(X) X(17) X(18)
#include "mg_gen_lane_tu.inc"
