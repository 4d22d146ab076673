// k_roll7 for ONE rule of rule group GG_ROOMS: RULE_GOTO_BIG by itself (GG_RULE, mg_device.h; MG_RULE, mg_step.h; MG_ONE_RULE_UNITS, mg_launch.h; see mg_step_tu.inc) -- the multi-room BabyAI GoTo levels (22 x 22 grids: the staged split).
// The default 7x7 view and FullyObs of these levels run this unit; their other observation modes keep k_step<., GG_ROOMS>.
#define MG_TU_GG GG_RULE(GG_ROOMS, RULE_GOTO_BIG)
#define MG_TU_NAME gotobig
#define MG_TU_NO_KSTEP 1
#include "mg_step_tu.inc"
