// Q80 GEMV kernels for group size 128 (see gemv_q80_impl.h)
#define NANO_Q80_GS 128
#define NANO_Q80_ENTRY launch_gemv_q80_gs128
#include "gemv_q80_impl.h"
