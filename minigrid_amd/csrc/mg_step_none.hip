// k_step instantiations of rule group GG_NONE (see mg_step_tu.inc)
#define MG_TU_GG GG_NONE
#define MG_TU_NAME none
#include "mg_step_tu.inc"
