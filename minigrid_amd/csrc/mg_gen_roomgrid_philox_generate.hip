// generate kernel of generator group GG_ROOMGRID, WavePhilox streams (see mg_gen_tu.inc)
#define MG_TU_GG GG_ROOMGRID
#define MG_TU_RNG WavePhilox
#define MG_TU_REFILL 0
#define MG_TU_NAME roomgrid_philox
#include "mg_gen_tu.inc"
