// k_step instantiations of rule group GG_ROOMGRID (see mg_step_tu.inc)
#define MG_TU_GG GG_ROOMGRID
#define MG_TU_NAME roomgrid
#include "mg_step_tu.inc"
