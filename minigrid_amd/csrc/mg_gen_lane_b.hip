// lane-per-episode generator kernels (mg_genlane.h, mg_gen_lane_tu.inc): the generator functions FN of this unit (lane_fn_of_kind)
#define MG_LANE_TU_NAME b
#define MG_LANE_TU_FNS(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(16)
#include "mg_gen_lane_tu.inc"
