// gemm_q80_g6_p2.hip -- G6 MODE P instantiations for rows of up to 4096 values (NV = 2 float4 items per thread: Qwen3-4B's n_embd
// 2560 and q_dim 4096): sequence capacities 1 / 2 (more sequences quantize their activations once, in a launch of their own: the prologue of 8 x 2560 values per workgroup measured 9.7 us), the round counts of its launches (QKV: 2 rounds over 3 segments, W1|W3: 4,
// Wo: 1) and the split-attention combine of one sequence.  The kernel is in gemm_q80_g6_impl.h.
#include "gemm_q80_g6_impl.h"

namespace nano {

// (rounds, multi-segment) pairs instantiated per capacity
static bool p2_has(uint32_t rounds, bool ms) { return ms ? (rounds == 2u || rounds == 4u) : (rounds == 1u || rounds == 2u || rounds == 4u); }

template <int NBC>
static hipError_t p2_go(const G6Dev &d, size_t lds, uint32_t rounds, bool ms, hipStream_t st) {
    if (ms) {
        if (rounds == 2u) return g6_launch_t<G6_P, false, NBC, 2, 2, true>(d, lds, st);
        if (rounds == 4u) return g6_launch_t<G6_P, false, NBC, 2, 4, true>(d, lds, st);
    } else {
        if (rounds == 1u) return g6_launch_t<G6_P, false, NBC, 2, 1, false>(d, lds, st);
        if (rounds == 2u) return g6_launch_t<G6_P, false, NBC, 2, 2, false>(d, lds, st);
        if (rounds == 4u) return g6_launch_t<G6_P, false, NBC, 2, 4, false>(d, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t g6p_launch_nv2(const void *dv, size_t lds, uint32_t nbc, uint32_t rounds, bool ms, bool comb, hipStream_t st) {
    const G6Dev &d = *static_cast<const G6Dev *>(dv);                  // (the argument block's type lives in an unnamed namespace: passed opaquely between translation units)
    if (comb) {
        if (nbc != 1u || ms) return hipErrorInvalidValue;
        if (rounds == 1u) return g6_launch_t<G6_P, true, 1, 2, 1, false>(d, lds, st);
        if (rounds == 2u) return g6_launch_t<G6_P, true, 1, 2, 2, false>(d, lds, st);
        return hipErrorInvalidValue;
    }
    switch (nbc) {
    case 1: return p2_go<1>(d, lds, rounds, ms, st);
    case 2: return p2_go<2>(d, lds, rounds, ms, st);
    default: return hipErrorInvalidValue;
    }
}

bool g6p_has(uint32_t nv, uint32_t nbc, uint32_t rounds, bool ms, bool comb) {
    if (nv == 2u) return comb ? (nbc == 1u && !ms && (rounds == 1u || rounds == 2u)) : ((nbc == 1u || nbc == 2u) && p2_has(rounds, ms));
    if (nv == 5u) return !comb && (nbc == 1u || nbc == 2u) && (ms ? rounds == 4u : (rounds == 3u || rounds == 4u));
    return false;
}

}  // namespace nano
