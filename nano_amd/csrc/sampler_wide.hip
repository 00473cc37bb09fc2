// sampler_wide.hip — the device sampler's second phase for WIDE nuclei (round 5; SURVEY 8f-2, reference infer/infer.c:1062-1109).
//
// sampler.hip sorts a superset of the nucleus in LDS: at most NANO_SAMPLE_MAX_CANDIDATES (8192) tokens.  Near-uniform distributions
// (random-init models at temperature 1: ~137 k of Qwen3's 151 936 tokens inside top_p = 0.9) do not fit, and rounds 2-4 handed those
// steps back to the host loops (D2H of V logits + softmax + qsort of 152 k entries: 76 tokens/s where the device path runs 1600).
// This phase runs when the first one reports NANO_SAMPLE_FALLBACK, on the numerators and the exact denominator it left on the device:
//   W1  entry i of a V-entry buffer becomes the 64-bit word (probability bits << 32 | ~i) when token i is a candidate (p = e / sum >=
//       cutoff, infer.c:1064-1072), 0 otherwise -- in INDEX order, no compaction;
//   W2  the words are sorted by their probability bits, descending and STABLE: equal probabilities stay in index order, which is what the
//       reference's qsort gives (glibc's merge sort is stable; infer.c:1074-1076), and the zeros end up behind the last candidate.
//       Round 6: a hand-written least-significant-digit radix sort (round 5 called rocPRIM's 64-bit radix sort here -- the one
//       library routine of the path): four passes over the 8-bit digits of the probability bits, ONE kernel each --
//         * the digit totals of all four passes are order-independent: the filter kernel (W1) counts them while it builds the list, and
//           the per-tile counts of pass 0's input with them;
//         * a pass's scatter kernel: a tile (1024 words) finds its first output slot per digit (exclusive scan of the pass's totals +
//           the counts of the tiles in front of it), ranks its words among the equal digits in front of them IN TILE ORDER -- the lanes
//           of a wave whose words carry the same digit are found with ballots over the digit's bits, a word's rank is the population
//           count below it -- moves them, and adds each moved word to the NEXT pass's count of the tile it lands in (a global atomic);
//       half the passes of a 64-bit sort (the index half of the word is carried, not sorted: stability does its work), five launches
//       where a histogram / scan / scatter triple per pass takes thirteen;
//   W3-W5 (sampler.hip) the reference's sequential float sum over the sorted list (infer.c:1078-1084: the cut is the first running sum
//       above top_p; infer.c:1096-1108: the draw is the first running sum above r = coin * cumulative) through the chunk functions of
//       exact_math.h: exact sums at every 256-entry boundary by one wave, only the two chunks that hold the cut and the draw added
//       element by element.
// Results (token, nucleus size, the six most probable tokens) are the reference's bit for bit: tests/test_gpu_sampler.py holds them to
// the compiled reference's goldens at V = 151 936 and to the oracle on ties (all-equal logits: 136 743 equal probabilities).
#include <cstring>
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace nano {
namespace {

constexpr uint32_t WT = 1024;                                         // entries per tile (a workgroup of 256 threads x 4)

// digit of a word in pass `sh` (bits sh .. sh + 7 of the word), reversed: ascending over it = descending over the probability
__device__ __forceinline__ uint32_t wdigit(unsigned long long w, uint32_t sh) { return 255u - (uint32_t)((w >> sh) & 255ull); }

// Scratch of the sort (uint32): tot[4][256] -- entries per digit of each pass over the WHOLE list (order-independent: the filter kernel counts
// all four while it builds the list; left zero by the last scatter for the next call) -- and hist[4][ntile][256], the per-tile counts of each
// pass's INPUT arrangement: pass 0's by the filter kernel, pass p + 1's by pass p's scatter (one global atomic per entry it moves).
struct WideSort { uint32_t *tot, *hist; uint32_t ntile; };

__global__ __launch_bounds__(256) void samp_wide_filter_kernel(const SampleArgs a, const WideSort ws) {
    __shared__ uint32_t h[4][256];
    const uint32_t tid = threadIdx.x, i0 = (blockIdx.x * 256 + tid) * 4, lane = tid & 63;
    for (int p = 0; p < 4; p++) h[p][tid] = 0;
    for (uint32_t p = 1; p < 4; p++) ws.hist[((size_t)p * ws.ntile + blockIdx.x) * 256 + tid] = 0;        // the later passes' tile counts start at zero
    __syncthreads();
    const float sum = a.sum[0];
    const float4 e = reinterpret_cast<const float4 *>(a.e)[i0 / 4];
    const float ev[4] = {e.x, e.y, e.z, e.w};
    unsigned long long w[4];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k;
        const float p = ev[k] / sum;                                   // the reference's division (infer.c:631)
        const bool cand = i < a.V && p >= a.cutoff;
        w[k] = cand ? ((unsigned long long)__float_as_uint(p) << 32) | (unsigned long long)(0xffffffffu - i) : 0ull;
        n += (uint32_t)__popcll(__ballot(cand));
#pragma unroll
        for (int ps = 0; ps < 4; ps++) atomicAdd(&h[ps][wdigit(w[k], 32u + 8u * (uint32_t)ps)], 1u);
    }
    // tile order of the sort: entry k * 256 + tid of a tile -- store this thread's four consecutive words transposed into that order?  No: the
    // sort's tile order IS the storage order; a thread's four words are entries 4 tid .. 4 tid + 3 and the scatter reads them the same way.
    { ulonglong2 *o = reinterpret_cast<ulonglong2 *>(a.wide_in + i0); o[0] = make_ulonglong2(w[0], w[1]); o[1] = make_ulonglong2(w[2], w[3]); }
    if (lane == 0 && n) atomicAdd(a.ncand, n);
    __syncthreads();
    ws.hist[(size_t)blockIdx.x * 256 + tid] = h[0][tid];               // pass 0's tile counts
    for (int p = 0; p < 4; p++) if (h[p][tid]) atomicAdd(&ws.tot[p * 256 + tid], h[p][tid]);
}

// One pass: tile b moves its 1024 words to their sorted slots, stable.  First slot of digit d for this tile = sum of tot[pass][d' < d] + sum of
// hist[pass][b' < b][d]; the words of a tile are ranked in storage order: thread t holds words 4 t .. 4 t + 3, a wave 256 consecutive words.
__global__ __launch_bounds__(256) void samp_wide_scatter_kernel(const unsigned long long *in, unsigned long long *out, const WideSort ws, uint32_t pass) {
    __shared__ uint32_t cnt[4][256];                                   // [wave][digit]: the wave's words of that digit, then their first slot
    __shared__ uint32_t base[256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, sh = 32u + 8u * pass, b = blockIdx.x;
    const uint32_t *tot = ws.tot + pass * 256, *hist = ws.hist + (size_t)pass * ws.ntile * 256;
    uint32_t *hnext = pass < 3u ? ws.hist + (size_t)(pass + 1u) * ws.ntile * 256 : nullptr;
#pragma unroll
    for (int g = 0; g < 4; g++) cnt[g][tid] = 0;
    const ulonglong2 *t = reinterpret_cast<const ulonglong2 *>(in + (size_t)b * WT + tid * 4);
    const ulonglong2 w01 = t[0], w23 = t[1];
    // digit `tid`: the words of the tiles in front of this one (independent loads, eight in flight)
    uint32_t before = 0;
    for (uint32_t t0 = 0; t0 < b; t0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) v[q] = t0 + q < b ? hist[(size_t)(t0 + q) * 256 + tid] : 0u;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) before += v[q];
    }
    const uint32_t mytot = tot[tid];
    base[tid] = mytot;
    const unsigned long long w[4] = {w01.x, w01.y, w23.x, w23.y};
    uint32_t dg[4], rk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) dg[k] = wdigit(w[k], sh);
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {                           // inclusive scan of the digit totals (Hillis-Steele through LDS)
        const uint32_t v = tid >= o ? base[tid - o] : 0u;
        __syncthreads();
        base[tid] += v;
        __syncthreads();
    }
    const uint32_t first = base[tid] - mytot + before;                 // first slot of digit `tid` for this tile
    // ranks inside the wave's 256 words (order: lane, then k).  bits[k2][bit] = the lanes whose word k2 has that digit bit set
    unsigned long long bits[4][8];
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
        for (int bit = 0; bit < 8; bit++) bits[k2][bit] = __ballot((dg[k2] >> bit) & 1u);
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t r = 0;
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            unsigned long long same = ~0ull;                           // lanes whose word k2 carries this word's digit
#pragma unroll
            for (int bit = 0; bit < 8; bit++) same &= ((dg[k] >> bit) & 1u) ? bits[k2][bit] : ~bits[k2][bit];
            r += (uint32_t)__popcll(same & below);                     // ... in lower lanes
            if (k2 < k && dg[k2] == dg[k]) r += 1u;                    // ... in this lane, in front of word k
        }
        rk[k] = r;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) atomicMax(&cnt[wv][dg[k]], rk[k] + 1u);    // the last word of a digit knows the wave's count
    __syncthreads();
    {                                                                  // thread d: the four waves' counts of digit d -> their first slots
        uint32_t run = first;
#pragma unroll
        for (int g = 0; g < 4; g++) { const uint32_t c = cnt[g][tid]; cnt[g][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pos = cnt[wv][dg[k]] + rk[k];
        out[pos] = w[k];
        if (hnext) atomicAdd(&hnext[(size_t)(pos / WT) * 256 + wdigit(w[k], sh + 8u)], 1u);      // the next pass's count of the tile this word lands in
    }
}

}  // namespace

// scratch of the sort of n words (uint32): tot[4][256] + hist[4][tiles][256]
size_t sample_wide_temp_bytes(uint32_t n) { return (size_t)(4 * 256 + 4 * (size_t)((n + WT - 1) / WT) * 256) * 4 + 256; }

hipError_t launch_sample_wide(const SampleArgs &a, void *temp, size_t temp_bytes, hipStream_t st) {
    if (!a.wide_in || !a.wide_out || !a.wide_p || !temp || a.wide_cap < a.V || a.wide_cap % WT || a.wide_cap != a.nch * SAMPLE_CHUNK) return hipErrorInvalidValue;
    if (temp_bytes < sample_wide_temp_bytes(a.wide_cap)) return hipErrorInvalidValue;
    WideSort ws{};
    ws.tot = reinterpret_cast<uint32_t *>(temp); ws.hist = ws.tot + 4 * 256; ws.ntile = a.wide_cap / WT;
    hipError_t e = hipMemsetAsync(ws.tot, 0, 4 * 256 * 4, st);        // (the totals are accumulated with atomics)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(samp_wide_filter_kernel, dim3(ws.ntile), dim3(256), 0, st, a, ws);
    unsigned long long *src = a.wide_in, *dst = a.wide_out;
    for (uint32_t pass = 0; pass < 4; pass++) {                        // the buffers swap roles: an even number of passes ends in wide_in
        hipLaunchKernelGGL(samp_wide_scatter_kernel, dim3(ws.ntile), dim3(256), 0, st, src, dst, ws, pass);
        unsigned long long *t = src; src = dst; dst = t;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    SampleArgs w = a;
    w.wide_out = src;                                                  // the cut kernels read the sorted words under this name
    return launch_sample_wide_cut(w, st);
}

}  // namespace nano
