// refill kernel of generator group GG_LIGHT, WavePhilox streams (see mg_gen_tu.inc)
#define MG_TU_GG GG_LIGHT
#define MG_TU_RNG WavePhilox
#define MG_TU_REFILL 1
#define MG_TU_NAME light_philox
#include "mg_gen_tu.inc"
