// k_roll7 of the levels whose rule is "the agent carries THE object" alone (GG_PICKUP = rule group GG_ROOMGRID's RULE_PICKUP without the group's other rules:
// mg_device.h; see mg_step_tu.inc): UnlockPickup, BlockedUnlockPickup, KeyCorridor (MiniGrid and BabyAI), ObstructedMaze -- the default 7x7 view and FullyObs.
#define MG_TU_GG GG_PICKUP
#define MG_TU_NAME pickup
#define MG_TU_ROLL_ONLY 1
#include "mg_step_tu.inc"
