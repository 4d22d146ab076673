// k_roll7 for ONE rule of rule group GG_ROOMS: RULE_PUTNEXT by itself (GG_RULE, mg_device.h; MG_RULE, mg_step.h; MG_ONE_RULE_UNITS, mg_launch.h; see mg_step_tu.inc) -- the BabyAI PutNext levels.
// The default 7x7 view and FullyObs of these levels run this unit; their other observation modes keep k_step<., GG_ROOMS>.
#define MG_TU_GG GG_RULE(GG_ROOMS, RULE_PUTNEXT)
#define MG_TU_NAME putnext
#define MG_TU_NO_KSTEP 1
#include "mg_step_tu.inc"
