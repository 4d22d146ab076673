// k_roll7 for ONE rule of rule group GG_LIGHT: RULE_GOTODOOR by itself (GG_RULE, mg_device.h; MG_RULE, mg_step.h; MG_ONE_RULE_UNITS, mg_launch.h; see mg_step_tu.inc) -- MiniGrid-GoToDoor-*.
// The default 7x7 view and FullyObs of these levels run this unit; their other observation modes keep k_step<., GG_LIGHT>.
#define MG_TU_GG GG_RULE(GG_LIGHT, RULE_GOTODOOR)
#define MG_TU_NAME gotodoor
#define MG_TU_ROLL_ONLY 1
#include "mg_step_tu.inc"
