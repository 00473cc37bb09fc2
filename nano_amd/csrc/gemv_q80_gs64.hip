// Q80 GEMV kernels for group size 64 (see gemv_q80_impl.h)
#define NANO_Q80_GS 64
#define NANO_Q80_ENTRY launch_gemv_q80_gs64
#include "gemv_q80_impl.h"
