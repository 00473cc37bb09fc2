// Q80 GEMV kernels for group size 32 (see gemv_q80_impl.h)
#define NANO_Q80_GS 32
#define NANO_Q80_ENTRY launch_gemv_q80_gs32
#include "gemv_q80_impl.h"
