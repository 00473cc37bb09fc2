// gemm_q80_g6_p5.hip -- G6 MODE P instantiations for rows of up to 10240 values (NV = 5 float4 items per thread: Qwen3-4B's hidden
// size 9728, the W2 launch of one or two sequences).  The kernel is in gemm_q80_g6_impl.h.
#include "gemm_q80_g6_impl.h"

namespace nano {

template <int NBC>
static hipError_t p5_go(const G6Dev &d, size_t lds, uint32_t rounds, bool ms, hipStream_t st) {
    if (ms) { if (rounds == 4u) return g6_launch_t<G6_P, false, NBC, 5, 4, true>(d, lds, st); return hipErrorInvalidValue; }
    if (rounds == 3u) return g6_launch_t<G6_P, false, NBC, 5, 3, false>(d, lds, st);
    if (rounds == 4u) return g6_launch_t<G6_P, false, NBC, 5, 4, false>(d, lds, st);
    return hipErrorInvalidValue;
}

hipError_t g6p_launch_nv5(const void *dv, size_t lds, uint32_t nbc, uint32_t rounds, bool ms, hipStream_t st) {
    const G6Dev &d = *static_cast<const G6Dev *>(dv);
    if (nbc == 1u) return p5_go<1>(d, lds, rounds, ms, st);
    if (nbc == 2u) return p5_go<2>(d, lds, rounds, ms, st);
    return hipErrorInvalidValue;
}

}  // namespace nano
