// refill kernel of generator group GG_ROOMGRID, WavePcg64 streams (see mg_gen_tu.inc)
#define MG_TU_GG GG_ROOMGRID
#define MG_TU_RNG WavePcg64
#define MG_TU_REFILL 1
#define MG_TU_NAME roomgrid_pcg
#include "mg_gen_tu.inc"
