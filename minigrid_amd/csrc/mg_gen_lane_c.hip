// lane-per-episode generator kernels (mg_genlane.h, mg_gen_lane_tu.inc): the generator functions FN of this unit (lane_fn_of_kind)
#define MG_LANE_TU_NAME c
#define MG_LANE_TU_FNS(X) X(136) X(137) X(138) X(139) X(140) X(141) X(142) X(143) X(144) X(145)
#include "mg_gen_lane_tu.inc"
