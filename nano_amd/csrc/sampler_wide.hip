// sampler_wide.hip — the device sampler's second phase for WIDE nuclei (round 5; SURVEY 8f-2, reference infer/infer.c:1062-1109).
//
// sampler.hip sorts a superset of the nucleus in LDS: at most NANO_SAMPLE_MAX_CANDIDATES (8192) tokens.  Near-uniform distributions
// (random-init models at temperature 1: ~137 k of Qwen3's 151 936 tokens inside top_p = 0.9) do not fit, and rounds 2-4 handed those
// steps back to the host loops (D2H of V logits + softmax + qsort of 152 k entries: 76 tokens/s where the device path runs 1600).
// This phase runs when the first one reports NANO_SAMPLE_FALLBACK, on the numerators and the exact denominator it left on the device:
//   W1  every candidate (p = e / sum >= cutoff, infer.c:1064-1072) becomes a 64-bit key (probability bits, ~index) in a buffer of
//       V entries (the rest stays 0); arrival order is irrelevant, the key is a total order;
//   W2  the keys are sorted in descending order = the reference's qsort (probability descending; glibc's merge sort is stable, so
//       equal probabilities stay in index order = ~index descending) — rocPRIM's device radix sort, the one library call of this
//       path (a plain library sort, like a plain library GEMM);
//   W3-W5 (sampler.hip) the reference's sequential float sum over the sorted list (infer.c:1078-1084: the cut is the first running sum
//       above top_p; infer.c:1096-1108: the draw is the first running sum above r = coin * cumulative) through the chunk functions of
//       exact_math.h: exact sums at every 256-entry boundary by one wave, only the two chunks that hold the cut and the draw added
//       element by element.
// Results (token, nucleus size, the six most probable tokens) are the reference's bit for bit: tests/test_gpu_sampler.py holds them to
// the compiled reference's goldens at V = 151 936 and to the oracle on ties (all-equal logits: 136 743 equal probabilities).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace nano {
namespace {

__global__ __launch_bounds__(256) void samp_wide_filter_kernel(const SampleArgs a) {
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4, lane = threadIdx.x & 63;
    const float sum = a.sum[0];
    const float4 e = reinterpret_cast<const float4 *>(a.e)[i0 / 4];
    const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k;
        const float p = ev[k] / sum;                                   // the reference's division (infer.c:631)
        const bool cand = i < a.V && p >= a.cutoff;
        const unsigned long long mask = __ballot(cand);
        if (mask == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.ncand, (uint32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (cand && slot < a.wide_cap) a.wide_in[slot] = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned long long)(0xffffffffu - i);
    }
}

}  // namespace

size_t sample_wide_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    unsigned long long *none = nullptr;
    if (rocprim::radix_sort_keys_desc(nullptr, bytes, none, none, (size_t)n, 0u, 64u, (hipStream_t)0) != hipSuccess) return 0;
    return bytes ? bytes : 256;
}

hipError_t launch_sample_wide(const SampleArgs &a, void *temp, size_t temp_bytes, hipStream_t st) {
    if (!a.wide_in || !a.wide_out || !a.wide_p || !temp || a.wide_cap < a.V) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.wide_in, 0, (size_t)a.wide_cap * 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(samp_wide_filter_kernel, dim3(a.nch * SAMPLE_CHUNK / 1024), dim3(256), 0, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    size_t bytes = temp_bytes;
    if ((e = rocprim::radix_sort_keys_desc(temp, bytes, a.wide_in, a.wide_out, (size_t)a.wide_cap, 0u, 64u, st)) != hipSuccess) return e;
    return launch_sample_wide_cut(a, st);
}

}  // namespace nano
