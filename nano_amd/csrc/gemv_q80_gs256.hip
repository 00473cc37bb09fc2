// Q80 GEMV kernels for group size 256 (see gemv_q80_impl.h)
#define NANO_Q80_GS 256
#define NANO_Q80_ENTRY launch_gemv_q80_gs256
#include "gemv_q80_impl.h"
