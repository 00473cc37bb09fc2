// sampler.hip — the reference's sampler on the device, token-for-token identical to the host code.
//
// Reference: generate_next_token (infer/infer.c:1156-1189), softmax (:616-634), sample_top_p (:1062-1109): repetition penalty
// (divide the logits of every token seen so far), divide by the temperature, softmax (first-max, libm expf, index-order
// float sum, divide), keep p >= (1-top_p)/(V-1), qsort by probability (glibc's merge sort: stable, so equal
// probabilities stay in index order), cut where the running sum passes top_p, draw with one xorshift64* coin.
//
// Every step whose result depends on evaluation order is evaluated in the reference's order:
//   * expf        exact_expf_nonpos (exact_math.h): the same double-precision operation sequence as the pinned libm;
//   * the sum     chunk functions (exact_math.h): integer mantissa increments per 256-element chunk, folded as a tree;
//                 one wave then scans 64 chunk functions per step, jumps the running sum over the longest prefix that
//                 applies and adds element by element only the chunk in which the sum changes binade (about ten of the
//                 594 chunks of Qwen3's vocabulary);
//   * the sort    a total order (probability desc, index asc) in LDS, so any network gives the stable result; when the
//                 candidates are many only a superset of the nucleus is sorted: a 256-bin histogram of the numerators
//                 (counts + fixed-point masses) tells at which bin the mass passes top_p, tokens in later bins are
//                 dropped, and the cut is accepted only if it fell strictly above every dropped token;
//   * the cut and the draw   one thread, sequential, over the sorted nucleus.
// Six dependent kernels, ~V*4 B each way through L2; nothing but a 52-byte result crosses PCIe.
#include "kernels.h"
#include "exact_math.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>

namespace nano {
using namespace nano_exact;

static constexpr int CH = SAMPLE_CHUNK;           // elements per chunk = one wave x float4
static_assert(SAMPLE_BINS == 256, "the histogram is zeroed / flushed by 256-thread workgroups and scanned 4 bins per lane");

// histogram bin of a softmax numerator e in [0, 1]: 8 bins per binade, bin 7 = {1.0}, larger e -> smaller bin,
// everything below 2^-31 (and 0) in the last bin
__device__ __forceinline__ uint32_t exp_bin(float e) {
    const uint32_t u = __float_as_uint(e), f = u >> 23;
    const uint32_t b = (127u - f) * 8u + (7u - ((u >> 20) & 7u));
    return f > 127u ? 0u : (b > SAMPLE_BINS - 1 ? SAMPLE_BINS - 1 : b);
}

__global__ __launch_bounds__(256) void seen_set_kernel(const uint32_t *ids, uint32_t n, uint8_t *seen) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) seen[ids[i]] = 1;
}

// K1: y = (seen ? l / penalty : l) / temperature over the padded range (padding = -inf -> numerator 0), one max per workgroup
__global__ __launch_bounds__(256) void samp_prep_kernel(const SampleArgs a) {
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k;
        float l = -INFINITY;
        if (i < a.V) {
            l = a.logits[i];
            if (a.seen && a.seen[i]) l /= a.penalty;
            if (a.temperature != 0.0f) l /= a.temperature;
        }
        v[k] = l; mx = fmaxf(mx, l);
    }
    *reinterpret_cast<float4 *>(a.y + i0) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.temperature == 0.0f) return;            // penalised arg-max only: no softmax
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) a.pmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// K2: numerators e = expf(y - max) and one approximate float sum per chunk (used only to guess each chunk's binade)
__global__ __launch_bounds__(256) void samp_exp_kernel(const SampleArgs a) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= a.nch) return;
    float m = -INFINITY;                                           // max over the nch/4 workgroup maxima of K1 (<= 256)
    for (uint32_t j = lane; j < a.nch / 4; j += 64) m = fmaxf(m, a.pmax[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float4 y = reinterpret_cast<const float4 *>(a.y)[c * 64 + lane];
    float4 e;
    e.x = exact_expf_nonpos(y.x - m, kExp2Tab); e.y = exact_expf_nonpos(y.y - m, kExp2Tab);
    e.z = exact_expf_nonpos(y.z - m, kExp2Tab); e.w = exact_expf_nonpos(y.w - m, kExp2Tab);
    reinterpret_cast<float4 *>(a.e)[c * 64 + lane] = e;
    float s = (e.x + e.y) + (e.z + e.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) a.approx[c] = s;
    // per-bin count and mass of the numerators; the mass in 2^-40 fixed point so that the totals do not depend on the
    // order of the atomics (truncation < 2^-40 per token: 1.4e-7 over the whole vocabulary)
    __shared__ uint32_t hcnt[SAMPLE_BINS];
    __shared__ unsigned long long hmass[SAMPLE_BINS];
    hcnt[threadIdx.x] = 0; hmass[threadIdx.x] = 0;
    __syncthreads();
    const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (ev[k] != 0.0f) {
            const uint32_t b = exp_bin(ev[k]);
            atomicAdd(&hcnt[b], 1u);
            atomicAdd(&hmass[b], (unsigned long long)((double)ev[k] * 0x1p40));
        }
    __syncthreads();
    if (hcnt[threadIdx.x]) { atomicAdd(&a.bin_cnt[threadIdx.x], hcnt[threadIdx.x]); atomicAdd(&a.bin_mass[threadIdx.x], hmass[threadIdx.x]); }
}

// K3: the chunk function of every chunk, for the binade its approximate prefix falls in
__global__ __launch_bounds__(256) void samp_chunkfn_kernel(const SampleArgs a) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= a.nch) return;
    float pre = 0.0f;
    for (uint32_t j = lane; j < c; j += 64) pre += a.approx[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
    const uint32_t Es = sum_exp(__float_as_uint(pre));
    const float4 e = reinterpret_cast<const float4 *>(a.e)[c * 64 + lane];
    ChunkFn f{0u, 0u};
    chunk_push(f, __float_as_uint(e.x), Es); chunk_push(f, __float_as_uint(e.y), Es);
    chunk_push(f, __float_as_uint(e.z), Es); chunk_push(f, __float_as_uint(e.w), Es);
#pragma unroll
    for (int st = 1; st < 64; st <<= 1) {
        ChunkFn g;
        g.dE = __shfl_down(f.dE, st, 64); g.dO = __shfl_down(f.dO, st, 64);
        if ((lane & (2 * st - 1)) == 0) f = chunk_then(f, g);
    }
    if (lane == 0) { a.fn[c] = make_uint2(f.dE, f.dO); a.spec[c] = Es; }
}

// K4: one wave carries the exact running sum through the chunk functions; where one does not apply (the sum is in
// another binade than guessed, or leaves it inside the chunk) the chunk's 256 numerators are added one by one.
__device__ __attribute__((noinline)) uint32_t samp_walk_chunk(const float *e, uint32_t c, uint32_t lane, uint32_t sb) {
    const float4 v = reinterpret_cast<const float4 *>(e)[c * 64 + lane];
    float s = __uint_as_float(sb);
#pragma unroll
    for (int k = 0; k < 64; k++) {                                  // elements 4k..4k+3 of the chunk live in lane k
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), k));
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), k));
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), k));
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), k));
    }
    return __builtin_amdgcn_readfirstlane(__float_as_uint(s));
}

// Inclusive scan of chunk functions over the 64 lanes (lane i <- chunk cur .. cur + i composed in order), VALU only: row shifts
// inside the 16-lane rows, then row_bcast 15 / 31 across them.  Lanes without a source keep the identity {0, 0} (apply nothing,
// valid).  (Round 3: the __shfl_up form -- three ds_bpermute per step, 18 per scan -- was ~1 us of each of the ~19 scans.)
template <int CTRL, int ROWMASK> __device__ __forceinline__ uint32_t dpp_or_id(uint32_t idv, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)idv, (int)v, CTRL, ROWMASK, 0xF, false);
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ void scan_step(ChunkFn &f, uint32_t &valid) {
    ChunkFn g;
    g.dE = dpp_or_id<CTRL, ROWMASK>(0u, f.dE); g.dO = dpp_or_id<CTRL, ROWMASK>(0u, f.dO);
    const uint32_t gv = dpp_or_id<CTRL, ROWMASK>(1u, valid);
    f = chunk_then(g, f); valid &= gv;
}
__device__ __forceinline__ void wave_scan_fn(ChunkFn &f, uint32_t &valid) {
    scan_step<0x111, 0xF>(f, valid);          // row_shr:1
    scan_step<0x112, 0xF>(f, valid);          // row_shr:2
    scan_step<0x114, 0xF>(f, valid);          // row_shr:4
    scan_step<0x118, 0xF>(f, valid);          // row_shr:8
    scan_step<0x142, 0xA>(f, valid);          // row_bcast:15 -> rows 1, 3
    scan_step<0x143, 0xC>(f, valid);          // row_bcast:31 -> rows 2, 3
}

__global__ __launch_bounds__(64) void samp_propagate_kernel(const SampleArgs a) {
    __shared__ uint32_t s_dE[SAMPLE_MAX_CHUNKS + 64], s_dO[SAMPLE_MAX_CHUNKS + 64], s_spec[SAMPLE_MAX_CHUNKS + 64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t c = lane; c < a.nch + 64; c += 64) {
        const bool in = c < a.nch;
        const uint2 f = in ? a.fn[c] : make_uint2(0u, 0u);
        s_dE[c] = f.x; s_dO[c] = f.y; s_spec[c] = in ? a.spec[c] : 0xffffffffu;      // past the end: applies to no sum
    }
    __syncthreads();
    // Each step the 64 lanes look at the next 64 chunk functions: an inclusive scan composes them, the running sum jumps
    // over the longest prefix that applies (same binade as guessed, mantissa stays below 2^24), and the first chunk
    // that does not apply is added element by element.  ~15 steps for Qwen3's 594 chunks.
    uint32_t sb = 0, walks = 0, cur = 0;
    while (cur < a.nch) {
        const uint32_t E = sum_exp(sb), M = sum_man(sb);
        ChunkFn f{s_dE[cur + lane], s_dO[cur + lane]};
        uint32_t valid = s_spec[cur + lane] == E ? 1u : 0u;
        wave_scan_fn(f, valid);
        const uint32_t tot = M + ((M & 1u) ? f.dO : f.dE);
        const unsigned long long ok = __ballot(valid != 0u && tot < (1u << 24));
        const uint32_t n = ok == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~ok);         // ok is a prefix of the lanes
        if (n) { sb = ((E - 1u) << 23) + (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)n - 1); cur += n; }
        if (n < 64 && cur < a.nch) { sb = samp_walk_chunk(a.e, cur, lane, sb); walks++; cur++; }
        sb = __builtin_amdgcn_readfirstlane(sb); cur = __builtin_amdgcn_readfirstlane(cur);
    }
    const float sum = __uint_as_float(sb);
    // Which tokens have to be sorted?  All candidates if they fit; otherwise the bins down to the first one at which
    // the mass of the numerators passes top_p (with head-room for the rounding of the float running sum) — the cut then
    // falls inside them.  Lane l owns bins 4l..4l+3; cumulative counts / masses by a wave scan.
    uint32_t c[4]; unsigned long long ms[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { c[k] = a.bin_cnt[lane * 4 + k]; ms[k] = a.bin_mass[lane * 4 + k]; }
#pragma unroll
    for (int k = 1; k < 4; k++) { c[k] += c[k - 1]; ms[k] += ms[k - 1]; }
    uint32_t ct = c[3]; unsigned long long mt = ms[3];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t oc = __shfl_up(ct, o, 64); const unsigned long long om = __shfl_up(mt, o, 64);
        if (lane >= (uint32_t)o) { ct += oc; mt += om; }
    }
    const uint32_t cbase = ct - c[3]; const unsigned long long mbase = mt - ms[3];
    const double need = (double)a.top_p * 1.002 * (double)sum * 0x1p40;
    uint32_t bm = SAMPLE_BINS - 1, b6 = SAMPLE_BINS - 1;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
        if ((double)(mbase + ms[k]) > need) bm = lane * 4 + k;
        if (cbase + c[k] >= 6u) b6 = lane * 4 + k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bm = min(bm, (uint32_t)__shfl_xor(bm, o, 64)); b6 = min(b6, (uint32_t)__shfl_xor(b6, o, 64)); }
    const uint32_t bc = exp_bin(a.cutoff * sum * 0.999f);            // every candidate (p >= cutoff) lies in bins <= bc
    uint32_t cc = bc % 4 == 0 ? c[0] : bc % 4 == 1 ? c[1] : bc % 4 == 2 ? c[2] : c[3];
    cc = __shfl(cbase + cc, bc / 4, 64);
    if (lane == 0) {
        a.sum[0] = sum; a.res->sum_bits = sb; a.res->walked_chunks = walks;
        *a.bstar = cc <= a.cap ? SAMPLE_BINS - 1 : max(bm, b6);
    }
}

// K5: p = e / sum; candidates p >= cutoff appended as keys (probability bits, ~index): one u64 compare = the
// reference's order (probability descending, then index ascending — glibc's qsort is a stable merge sort).
// Candidates in bins after `bstar` (chosen by the propagate kernel) are only counted, and their largest probability
// recorded for the check in the pick kernel.
__global__ __launch_bounds__(256) void samp_filter_kernel(const SampleArgs a) {
    const uint32_t i0 = (blockIdx.x * 256 + threadIdx.x) * 4, lane = threadIdx.x & 63;
    const float sum = a.sum[0];
    const uint32_t bstar = *a.bstar;
    const float4 e = reinterpret_cast<const float4 *>(a.e)[i0 / 4];
    const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t i = i0 + k;
        const float p = ev[k] / sum;
        const bool cand = i < a.V && p >= a.cutoff;
        const bool keep = cand && exp_bin(ev[k]) <= bstar;
        const bool drop = cand && !keep;
        const unsigned long long mask = __ballot(keep), dmask = __ballot(drop);
        if (dmask != 0) {
            float pm = drop ? p : 0.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) pm = fmaxf(pm, __shfl_xor(pm, o, 64));
            if (lane == 0) { atomicAdd(a.ndrop, (uint32_t)__popcll(dmask)); atomicMax(a.dropmax, __float_as_uint(pm)); }
        }
        if (mask == 0) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.ncand, (uint32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (keep && slot < a.cap) a.cand[slot] = ((unsigned long long)__float_as_uint(p) << 32) | (unsigned long long)(0xffffffffu - i);
    }
}

// K6: sort the candidates, cut the nucleus, draw.  Also re-arms the two cells the next call's K1/K5 accumulate into.
__global__ __launch_bounds__(1024) void samp_pick_kernel(const SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long pick_lds[];     // keys [SAMPLE_MAX_CANDIDATES] + four scalars
    unsigned long long *key = pick_lds;
    float *s_rp = reinterpret_cast<float *>(pick_lds + SAMPLE_MAX_CANDIDATES);
    uint32_t *s_u = reinterpret_cast<uint32_t *>(s_rp + 1);
    const uint32_t tid = threadIdx.x;
    const uint32_t n0 = *a.ncand, ndrop = *a.ndrop;
    const float dropmax = __uint_as_float(*a.dropmax);
    __syncthreads();
    if (tid == 0) { *a.ncand = 0; *a.ndrop = 0; *a.dropmax = 0; a.res->n_candidates = n0 + ndrop; a.res->n_sorted = n0; }
    if (tid < SAMPLE_BINS) { a.bin_cnt[tid] = 0; a.bin_mass[tid] = 0; }
    if (n0 == 0 || n0 > a.cap || (ndrop && n0 < 6)) { if (tid == 0) { a.res->status = NANO_SAMPLE_FALLBACK; a.res->token = 0; } return; }
    uint32_t n = 2;
    while (n < n0) n <<= 1;
    for (uint32_t i = tid; i < n; i += 1024) key[i] = i < n0 ? a.cand[i] : 0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= n; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < n / 2; t += 1024) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const unsigned long long x = key[i], y = key[p];
                const bool desc = (i & k) == 0;
                if (desc ? (x < y) : (x > y)) { key[i] = y; key[p] = x; }
            }
            __syncthreads();
        }
    // The two sequential loops of sample_top_p (infer.c:1078-1106).  The cut: one thread adds the sorted probabilities in
    // order (they are >= 0, the running sum never decreases: sixteen additions per test) and leaves every running sum in place
    // of the probability it consumed.  The draw adds the SAME numbers in the same order, so its running sums are those bits:
    // instead of a second sequential pass all threads look for the first stored sum above r (round 3: the second pass was a
    // third of this kernel at a nucleus of thousands).
    float &s_r = s_rp[0];
    uint32_t &s_last = s_u[0], &s_pick = s_u[1], &s_ok = s_u[2];
    uint32_t *kw = reinterpret_cast<uint32_t *>(key);                  // probability bits = high word of key i
    if (tid == 0) {
        auto prob = [&](uint32_t i) { return __uint_as_float(kw[2 * i + 1]); };
        float probe5 = n0 > 5 ? prob(5) : 0.0f;                       // (index 5 may be overwritten below)
        float cum = 0.0f, plast = 0.0f;
        uint32_t last = n0 - 1;
        bool cut = false;
        uint32_t i = 0;
        for (; i + 16 <= n0 && !cut; i += 16) {
            float pv[16], cs[16];
#pragma unroll
            for (int k = 0; k < 16; k++) pv[k] = prob(i + k);
            cs[0] = cum + pv[0];
#pragma unroll
            for (int k = 1; k < 16; k++) cs[k] = cs[k - 1] + pv[k];
            if (cs[15] > a.top_p) {                                    // cumulative_prob > top_p (infer.c:1078-1084)
#pragma unroll
                for (int k = 15; k >= 0; k--) if (cs[k] > a.top_p) { last = i + k; cum = cs[k]; plast = pv[k]; }
                cut = true;
            } else cum = cs[15];
#pragma unroll
            for (int k = 0; k < 16; k++) kw[2 * (i + k) + 1] = __float_as_uint(cs[k]);      // (sums past `last` are never read)
        }
        for (; i < n0 && !cut; i++) {
            const float pv = prob(i);
            cum += pv; kw[2 * i + 1] = __float_as_uint(cum);
            if (cum > a.top_p) { last = i; plast = pv; cut = true; }
        }
        if (!cut) plast = 0.0f;                                        // (only read when cut)
        // tokens were dropped: the sorted list is the reference's only down to the largest dropped probability
        // (and so are the six most probable tokens reported to the observation hook)
        const float pdeep = last > 5 ? plast : probe5;                 // probability of entry max(last, 5); n0 >= 6 whenever ndrop != 0
        const bool ok = !(ndrop && !(cut && pdeep > dropmax));
        s_ok = ok ? 1u : 0u; s_last = last; s_pick = last;             // r >= every running sum: probindex[last_idx] (infer.c:1108)
        s_r = a.coin * cum;
        if (!ok) { a.res->status = NANO_SAMPLE_FALLBACK; a.res->token = 0; }
    }
    __syncthreads();
    if (!s_ok) return;
    {
        const float r = s_r;
        const uint32_t last = s_last;
        for (uint32_t i = tid; i <= last; i += 1024) {                  // r < cdf (infer.c:1100-1106): the first running sum above r
            const float c = __uint_as_float(kw[2 * i + 1]);
            const float before = i ? __uint_as_float(kw[2 * i - 1]) : -1.0f;
            if (c > r && !(before > r)) s_pick = i;                    // (nondecreasing sums: exactly one such i, or none)
        }
    }
    __syncthreads();
    if (tid != 0) return;
    a.res->token = 0xffffffffu - (uint32_t)key[s_pick];
    a.res->status = NANO_SAMPLE_OK;
    a.res->nucleus = s_last + 1;
    for (uint32_t i = 0; i < 6; i++) a.res->top[i] = i < n0 ? 0xffffffffu - (uint32_t)key[i] : 0u;
}

// ---- wide nuclei (second phase, sampler_wide.hip): the cut and the draw over ALL candidates, sorted by a device radix sort ---------------
// The reference's two sequential loops (infer.c:1078-1084, 1096-1108) walk the sorted probabilities with one float running sum.  That sum
// is the same kind of object as the softmax denominator above: inside a binade it is integer arithmetic, so the chunk functions of
// exact_math.h apply unchanged -- 256 sorted probabilities per chunk, one wave scans 64 chunk functions per step and records the EXACT
// running sum at every chunk boundary; only the chunk in which the sum passes top_p (and the one in which it passes r = coin * sum) is
// added element by element.  (First build of the round: one thread adding all ~137 k probabilities of a flat Qwen3 distribution in
// order, 1.3 ms per token -- 546 tokens/s at temperature 1 against 77 through the host loops; this form: see profiles/r05_sample_decode_probe.txt.)

// W3: probabilities of the sorted keys (zeros behind the last candidate) + one approximate float sum per chunk
__global__ __launch_bounds__(256) void samp_wide_unpack_kernel(const SampleArgs a) {
    const uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= a.nch) return;
    const ulonglong2 *k = reinterpret_cast<const ulonglong2 *>(a.wide_out + (size_t)c * CH + lane * 4);
    const ulonglong2 k0 = k[0], k1 = k[1];
    float4 p;
    p.x = __uint_as_float((uint32_t)(k0.x >> 32)); p.y = __uint_as_float((uint32_t)(k0.y >> 32));
    p.z = __uint_as_float((uint32_t)(k1.x >> 32)); p.w = __uint_as_float((uint32_t)(k1.y >> 32));
    reinterpret_cast<float4 *>(a.wide_p)[c * 64 + lane] = p;
    float s = (p.x + p.y) + (p.z + p.w);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) a.approx[c] = s;
}

// one chunk element by element, every running sum kept: lane k gets the sums after its four elements (4k .. 4k+3)
__device__ __attribute__((noinline)) uint32_t samp_walk_chunk_sums(const float *e, uint32_t c, uint32_t lane, uint32_t sb, float &r0, float &r1, float &r2, float &r3) {
    const float4 v = reinterpret_cast<const float4 *>(e)[c * 64 + lane];
    float s = __uint_as_float(sb);
    r0 = r1 = r2 = r3 = 0.0f;
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const bool mine = lane == (uint32_t)k;
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), k)); r0 = mine ? s : r0;
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), k)); r1 = mine ? s : r1;
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), k)); r2 = mine ? s : r2;
        s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), k)); r3 = mine ? s : r3;
    }
    return __builtin_amdgcn_readfirstlane(__float_as_uint(s));
}
// index (0..255) of the first running sum of the chunk above `thr`, or 256; *at = that sum
__device__ __forceinline__ uint32_t first_above(float r0, float r1, float r2, float r3, float thr, uint32_t lane, float *at) {
    const uint32_t j = r0 > thr ? 0u : r1 > thr ? 1u : r2 > thr ? 2u : r3 > thr ? 3u : 4u;
    const unsigned long long m = __ballot(j < 4u);
    if (m == 0) return 256u;
    const int L = __builtin_ctzll(m);
    const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)j, L);
    const float v = jl == 0 ? r0 : jl == 1 ? r1 : jl == 2 ? r2 : r3;
    *at = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L));
    return (uint32_t)L * 4u + jl;
}

// W5: one wave: exact running sums at the chunk boundaries of the sorted list, the cut, the draw
__global__ __launch_bounds__(64) void samp_wide_cut_kernel(const SampleArgs a) {
    __shared__ uint32_t s_dE[SAMPLE_MAX_CHUNKS + 64], s_dO[SAMPLE_MAX_CHUNKS + 64], s_spec[SAMPLE_MAX_CHUNKS + 64], s_bound[SAMPLE_MAX_CHUNKS + 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t n0 = *a.ncand;
    for (uint32_t c = lane; c < a.nch + 64; c += 64) {
        const bool in = c < a.nch;
        const uint2 f = in ? a.fn[c] : make_uint2(0u, 0u);
        s_dE[c] = f.x; s_dO[c] = f.y; s_spec[c] = in ? a.spec[c] : 0xffffffffu;
    }
    if (lane == 0) { *a.ncand = 0; a.res->n_candidates = n0; a.res->n_sorted = n0; }
    if (n0 == 0 || n0 > a.wide_cap) { if (lane == 0) { a.res->status = NANO_SAMPLE_FALLBACK; a.res->token = 0; } return; }
    __syncthreads();
    const uint32_t nchw = (n0 + CH - 1) / CH;                        // chunks that hold candidates (the rest are zeros)
    const float *p = a.wide_p;
    uint32_t sb = 0, cur = 0;
    while (cur < nchw && !(__uint_as_float(sb) > a.top_p)) {
        const uint32_t E = sum_exp(sb), M = sum_man(sb);
        ChunkFn f{s_dE[cur + lane], s_dO[cur + lane]};
        uint32_t valid = s_spec[cur + lane] == E ? 1u : 0u;
        wave_scan_fn(f, valid);
        const uint32_t tot = M + ((M & 1u) ? f.dO : f.dE);
        const unsigned long long ok = __ballot(valid != 0u && tot < (1u << 24));
        const uint32_t n = ok == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~ok);
        if (lane < n) s_bound[cur + lane] = ((E - 1u) << 23) + tot;  // the exact sum behind chunk cur + lane
        if (n) { sb = ((E - 1u) << 23) + (uint32_t)__builtin_amdgcn_readlane((int)tot, (int)n - 1); cur += n; }
        if (n < 64 && cur < nchw && !(__uint_as_float(sb) > a.top_p)) {
            sb = samp_walk_chunk(p, cur, lane, sb);
            if (lane == 0) s_bound[cur] = sb;
            cur++;
        }
        sb = __builtin_amdgcn_readfirstlane(sb); cur = __builtin_amdgcn_readfirstlane(cur);
    }
    __syncthreads();
    const uint32_t ncov = cur < nchw ? cur : nchw;                   // s_bound[0 .. ncov) are exact
    if (ncov == 0u) {
        // top_p below zero: the loop above never ran (0 > top_p) and there is no boundary to read (round-5 advice: s_bound[ncov - 1] was an
        // out-of-bounds LDS read).  The reference's cut (infer.c:1078-1081) stops at the FIRST entry -- its probability already exceeds
        // top_p -- and the draw over that one entry returns it (infer.c:1096-1108).
        if (lane == 0) {
            a.res->token = 0xffffffffu - (uint32_t)a.wide_out[0];
            a.res->status = NANO_SAMPLE_OK;
            a.res->nucleus = 1u;
            for (uint32_t i = 0; i < 6; i++) a.res->top[i] = i < n0 ? 0xffffffffu - (uint32_t)a.wide_out[i] : 0u;
        }
        return;
    }
    auto first_chunk_above = [&](float thr, uint32_t upto) {         // first c < upto with the sum behind chunk c above thr, or upto
        for (uint32_t c0 = 0; c0 < upto; c0 += 64) {
            const uint32_t c = c0 + lane;
            const unsigned long long m = __ballot(c < upto && __uint_as_float(s_bound[c]) > thr);
            if (m) return c0 + (uint32_t)__builtin_ctzll(m);
        }
        return upto;
    };
    // the cut (infer.c:1078-1084): the first running sum above top_p; without one the whole list and its total
    uint32_t last = n0 - 1u;
    float cum = __uint_as_float(s_bound[ncov - 1u]);
    float r0, r1, r2, r3;
    const uint32_t cstar = first_chunk_above(a.top_p, ncov);
    if (cstar < ncov) {
        (void)samp_walk_chunk_sums(p, cstar, lane, cstar ? s_bound[cstar - 1u] : 0u, r0, r1, r2, r3);
        float at = 0.0f;
        const uint32_t k = first_above(r0, r1, r2, r3, a.top_p, lane, &at);
        if (k < 256u) { last = cstar * CH + k; cum = at; }
    }
    // the draw (infer.c:1096-1108): the first running sum above r among entries 0 .. last, else entry `last`
    const float r = a.coin * cum;
    const uint32_t clast = last / CH;
    uint32_t pick = last;
    uint32_t cr = first_chunk_above(r, clast);                       // whole chunks in front of the one that holds `last`
    {
        if (cr != cstar || cstar >= ncov) (void)samp_walk_chunk_sums(p, cr, lane, cr ? s_bound[cr - 1u] : 0u, r0, r1, r2, r3);
        float at = 0.0f;
        const uint32_t k = first_above(r0, r1, r2, r3, r, lane, &at);
        if (k < 256u && cr * CH + k <= last) pick = cr * CH + k;
    }
    if (lane != 0) return;
    a.res->token = 0xffffffffu - (uint32_t)a.wide_out[pick];
    a.res->status = NANO_SAMPLE_OK;
    a.res->nucleus = last + 1u;
    for (uint32_t i = 0; i < 6; i++) a.res->top[i] = i < n0 ? 0xffffffffu - (uint32_t)a.wide_out[i] : 0u;
}

hipError_t launch_sample_wide_cut(const SampleArgs &a, hipStream_t st) {
    SampleArgs w = a;
    w.e = a.wide_p;                                                  // the chunk-function kernel reads its addends from `e`
    hipLaunchKernelGGL(samp_wide_unpack_kernel, dim3(a.nch / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(samp_chunkfn_kernel, dim3(a.nch / 4), dim3(256), 0, st, w);
    hipLaunchKernelGGL(samp_wide_cut_kernel, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_seen_set(const uint32_t *ids, uint32_t n, uint8_t *seen, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(seen_set_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ids, n, seen);
    return hipGetLastError();
}

hipError_t launch_sample_prep(const SampleArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(samp_prep_kernel, dim3(a.nch * CH / 1024), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_sample(const SampleArgs &a, hipStream_t st) {
    const uint32_t wgs = a.nch * CH / 1024;       // nch is a multiple of 4
    hipLaunchKernelGGL(samp_prep_kernel, dim3(wgs), dim3(256), 0, st, a);
    hipLaunchKernelGGL(samp_exp_kernel, dim3(a.nch / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(samp_chunkfn_kernel, dim3(a.nch / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(samp_propagate_kernel, dim3(1), dim3(64), 0, st, a);
    hipLaunchKernelGGL(samp_filter_kernel, dim3(wgs), dim3(256), 0, st, a);
    constexpr size_t pick_lds_bytes = (size_t)SAMPLE_MAX_CANDIDATES * 8 + 64;
    static std::atomic<unsigned long long> armed{0};                  // bit per device: the attribute call is a host round trip
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !((armed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(samp_pick_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pick_lds_bytes);
        if (dev >= 0 && dev < 64) armed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(samp_pick_kernel, dim3(1), dim3(1024), pick_lds_bytes, st, a);
    return hipGetLastError();
}

}  // namespace nano
