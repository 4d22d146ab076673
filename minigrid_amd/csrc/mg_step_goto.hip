// k_roll7 of the GoTo levels alone (GG_GOTO = rule group GG_ROOMGRID's RULE_GOTO without the group's other rules: mg_device.h; see mg_step_tu.inc).
// The default 7x7 view and FullyObs of BabyAI-GoToRedBall / -Grey / -RedBlueBall / GoToObj / GoToLocal run these; their other observation modes keep k_step<., GG_ROOMGRID>.
#define MG_TU_GG GG_GOTO
#define MG_TU_NAME goto
#define MG_TU_ROLL_ONLY 1
#include "mg_step_tu.inc"
