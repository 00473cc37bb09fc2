// gemv.hip -- dispatcher of the fused decode GEMVs (out[d] = W[d,n] . act[n] for up to 8 sequences) by weight format:
//   FP32 matmul (reference infer/infer.c:637-651)            -> gemv_f32.hip
//   Q80  matmul_quant + quantize (infer.c:654-679, tensor.c:21-46) -> gemv_q80_impl.h (one translation unit per group size)
//   Q4K  matmul_q4k (tensor.c:438-471)                        -> gemv_q4k.hip
#include "kernels.h"

namespace nano {

hipError_t launch_gemv(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    (void)max_wg;
    if (quant == 0x80u) return launch_gemv_q80(a, st);
    return launch_gemv_f32(a, st);
}

// number of (max, row) arg-max partials launch_gemv() will write per sequence for these arguments (sizes tile_max;
// 0 = none written, the arg-max kernel scans the logits)
uint32_t gemv_tiles(uint32_t quant, const GemvArgs &a) {
    if (quant == 0x80u) return gemv_q80_partials(a);
    if (quant == 0x42u) return gemv_q4k_partials(a);
    return 0;
}

}  // namespace nano
