// k_roll7 for ONE rule of rule group GG_ROOMS: RULE_PICKUPDESC by itself (GG_RULE, mg_device.h; MG_RULE, mg_step.h; MG_ONE_RULE_UNITS, mg_launch.h; see mg_step_tu.inc) -- the BabyAI Pickup levels (PickupDist on 8 x 8; Pickup / PickupLoc / PickupAbove ... on 22 x 22: the staged split).
// The default 7x7 view and FullyObs of these levels run this unit; their other observation modes keep k_step<., GG_ROOMS>.
#define MG_TU_GG GG_RULE(GG_ROOMS, RULE_PICKUPDESC)
#define MG_TU_NAME pickupdesc
#define MG_TU_NO_KSTEP 1
#include "mg_step_tu.inc"
