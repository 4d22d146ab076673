// k_step instantiations of rule group GG_ROOMS (see mg_step_tu.inc)
#define MG_TU_GG GG_ROOMS
#define MG_TU_NAME rooms
#include "mg_step_tu.inc"
