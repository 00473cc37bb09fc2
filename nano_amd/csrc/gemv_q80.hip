// gemv_q80.hip -- dispatcher of the Q80 (W8A8) decode GEMVs; the kernels live in gemv_q80_impl.h, built once per
// quantization group size (gemv_q80_gs32/64/128/256.hip).
#include "gemv_q80_host.h"

namespace nano {

hipEvent_t g_q80_probe_start = nullptr, g_q80_probe_stop = nullptr;

hipError_t launch_gemv_q80_gs32(const GemvArgs &a, hipStream_t st);
hipError_t launch_gemv_q80_gs64(const GemvArgs &a, hipStream_t st);
hipError_t launch_gemv_q80_gs128(const GemvArgs &a, hipStream_t st);
hipError_t launch_gemv_q80_gs256(const GemvArgs &a, hipStream_t st);

// number of (max,row) arg-max partials a STORE launch with tile_max will write per sequence (0: none -- scan the logits)
uint32_t gemv_q80_partials(const GemvArgs &a) { return use_stream(a) ? STREAM_WGS * 4 : 0; }

hipError_t launch_gemv_q80(const GemvArgs &a, hipStream_t st) {
    if (a.nb == 0 || a.nb > 8 || a.gs == 0 || a.n % a.gs || a.n % 16 || a.nseg == 0 || a.nseg > 3) return hipErrorInvalidValue;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s < a.nseg; s++) if (a.seg[s].rows % 4) return hipErrorInvalidValue;
    switch (a.gs) {
    case 32: return launch_gemv_q80_gs32(a, st);
    case 64: return launch_gemv_q80_gs64(a, st);
    case 128: return launch_gemv_q80_gs128(a, st);
    case 256: return launch_gemv_q80_gs256(a, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace nano
