// k_roll7 for ONE rule of rule group GG_ROOMGRID: RULE_GOTO by itself (GG_RULE, mg_device.h; MG_RULE, mg_step.h; MG_ONE_RULE_UNITS, mg_launch.h; see mg_step_tu.inc) -- BabyAI-GoToRedBall / -Grey / -RedBlueBall / GoToObj / GoToLocal.
// The default 7x7 view and FullyObs of these levels run this unit; their other observation modes keep k_step<., GG_ROOMGRID>.
#define MG_TU_GG GG_RULE(GG_ROOMGRID, RULE_GOTO)
#define MG_TU_NAME goto
#define MG_TU_ROLL_ONLY 1
#include "mg_step_tu.inc"
