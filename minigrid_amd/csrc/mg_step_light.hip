// k_step instantiations of rule group GG_LIGHT (see mg_step_tu.inc)
#define MG_TU_GG GG_LIGHT
#define MG_TU_NAME light
#include "mg_step_tu.inc"
