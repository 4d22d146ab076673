// mg_kernels.h — everything: the step kernel (mg_step.h) and the generator kernels (mg_genk.h).
#pragma once
#include "mg_step.h"
#include "mg_genk.h"
