#include "common.h"
extern "C" int psam_gemm_f16_ln(const void*, const void*, const float*, void*, const float*, const float*, int, int, int, int, int, int, int, int, int, int, int, int, void*, int, float*, const float*, const float*, void*) { return PSAM_ERR_ARG; }
