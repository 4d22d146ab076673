// k_roll7 with the sentence levels' verifier in its step loop (rule group GG_SENTENCE; see mg_step_tu.inc, mg_verify.h)
#define MG_TU_GG GG_SENTENCE
#define MG_TU_NAME sentence
#define MG_TU_ROLL_ONLY 1
#include "mg_step_tu.inc"
