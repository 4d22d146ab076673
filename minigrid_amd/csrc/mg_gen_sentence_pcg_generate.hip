// generate kernel of generator group GG_SENTENCE, WavePcg64 streams (see mg_gen_tu.inc)
#define MG_TU_GG GG_SENTENCE
#define MG_TU_RNG WavePcg64
#define MG_TU_REFILL 0
#define MG_TU_NAME sentence_pcg
#include "mg_gen_tu.inc"
