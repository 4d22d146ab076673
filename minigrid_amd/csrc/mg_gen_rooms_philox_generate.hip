// generate kernel of generator group GG_ROOMS, WavePhilox streams (see mg_gen_tu.inc)
#define MG_TU_GG GG_ROOMS
#define MG_TU_RNG WavePhilox
#define MG_TU_REFILL 0
#define MG_TU_NAME rooms_philox
#include "mg_gen_tu.inc"
