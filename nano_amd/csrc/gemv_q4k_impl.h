// gemv_q4k_impl.h -- what the Q4K kernels share (gemv_q4k.hip: the (row, group)-item kernel for 1..8 sequences;
// gemv_q4k_chunk.hip: the 16-byte-chunk kernel for one sequence): the staged activation group, the block quantizers
// (workgroup-cooperative, from registers, from LDS), the activation staging and the operator-test unpacker.
#pragma once
#include <float.h>

#include "gemv_common.h"

namespace nano {

typedef unsigned int u32x2_q __attribute__((ext_vector_type(2)));

struct XGroup {            // 32 bytes per (sequence, group), 16-byte aligned
    uint32_t pk[4];        // the group's 32 nibbles, packed like value[16 g .. 16 g + 15] of a block
    float sq, bq;          // (float)s6 * s_scale, (float)b6 * s_bias
    int sumq;
    int _pad;
};

// One 256-value block quantized by the 256 threads of a workgroup (thread t <-> element t).
// Returns the thread's 4-bit code; group-level results go to `grp_out[g]` (if non-null, written by
// the group's first lane) and the raw block header fields to hdr (thread 0 .. as needed).
struct Q4kBlockHdr { float s_scale, s_bias; uint32_t sb[3]; };

__device__ __forceinline__ uint32_t q4k_quantize_block_coop(float v, bool valid, float *tmp /* >= 16 floats LDS */,
                                                            Q4kBlockHdr &hdr, float &sq_f, float &bq_f) {
    const int t = threadIdx.x, g = t >> 5;
    // reference: min starts at FLT_MAX, max at FLT_TRUE_MIN, updated with strict comparisons
    float lo = valid ? v : FLT_MAX;
    float hi = valid ? v : FLT_TRUE_MIN;
    lo = (lo < FLT_MAX) ? lo : FLT_MAX;            // NaN -> ignored like the reference's comparisons
    hi = (hi > FLT_TRUE_MIN) ? hi : FLT_TRUE_MIN;
    for (int o = 16; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    const float gsc = (lo <= 0.0f) ? div_const<15>(hi - lo) : div_const<15>(hi);
    const float gbi = (lo <= 0.0f) ? (-lo) : 0.0f;
    uint32_t nib = 0;
    if (valid && gsc != 0.0f) nib = (uint32_t)(nearest_int_magic((v + gbi) / gsc) & 0x0f);
    __syncthreads();
    if ((t & 31) == 0) { tmp[g] = gsc; tmp[8 + g] = gbi; }
    __syncthreads();
    float smax = FLT_TRUE_MIN, bmax = FLT_TRUE_MIN;
#pragma unroll
    for (int k = 0; k < 8; k++) { if (tmp[k] > smax) smax = tmp[k]; if (tmp[8 + k] > bmax) bmax = tmp[8 + k]; }
    const float s_scale = div_const<63>(smax), s_bias = div_const<63>(bmax);
    uint32_t s6[8], b6[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s6[k] = (s_scale == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(tmp[k] / s_scale) & 0x3f);
        b6[k] = (s_bias == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(tmp[8 + k] / s_bias) & 0x3f);
    }
    hdr.s_scale = s_scale; hdr.s_bias = s_bias;
    hdr.sb[0] = hdr.sb[1] = hdr.sb[2] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        hdr.sb[0] |= ((((s6[4 + i] & 0x30) << 2) | (s6[i] & 0x3f)) & 0xffu) << (8 * i);
        hdr.sb[1] |= ((((b6[4 + i] & 0x30) << 2) | (b6[i] & 0x3f)) & 0xffu) << (8 * i);
        hdr.sb[2] |= ((((b6[4 + i] & 0x0f) << 4) | (s6[4 + i] & 0x0f)) & 0xffu) << (8 * i);
    }
    sq_f = (float)s6[g] * s_scale;     // what get_group_scale_and_bias() will read back (tensor.c:137-140)
    bq_f = (float)b6[g] * s_bias;
    return nib;
}

namespace {

// keep (NV > 0 only): the normalised values stay in r.x for quantize_q4k_regs() instead of going to xn
template <int ROLE, int B, int NV>
__device__ __forceinline__ void stage_xn(const GemvDev &a, Staged<B, NV> &r, float *xn, float *red, uint32_t n4, bool keep) {
    // rmsnorm / split-attention combine of the activation into xn[B][n4] (same code path as the FP32 GEMV's staging)
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n;
    const uint32_t lane = tid & 63u, wid = tid >> 6, NW = nthr >> 6;
    const bool norm = has_flag<ROLE>(a, F_NORM), comb = has_flag<ROLE>(a, F_COMBINE);
    float *wgt = red + B * 16;
    if constexpr (NV == 0) {
        if (comb) combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
        for (uint32_t b = 0; b < a.nb; b++) {
            const float *x = a.xin + (size_t)b * a.xin_bstride;
            float ss = 1.0f;
            if (norm) {
                float acc = 0.0f;
                for (uint32_t i = tid * 4u; i < n; i += nthr * 4u) {
                    const float4 v = comb ? combine4(a, b, i, wgt) : *reinterpret_cast<const float4 *>(x + i);
                    acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
                }
                acc = dpp_wave_sum(acc);
                __syncthreads();
                if (lane == 0) red[wid] = acc;
                __syncthreads();
                float t = 0.0f;
                for (uint32_t w = 0; w < NW; w++) t += red[w];
                t /= (float)n; t += 1e-5f;
                ss = 1.0f / sqrtf(t);
            }
            for (uint32_t i = tid * 4u; i < n; i += nthr * 4u) {
                float4 v = comb ? combine4(a, b, i, wgt) : *reinterpret_cast<const float4 *>(x + i);
                if (norm) {
                    const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                    v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
                }
                *reinterpret_cast<float4 *>(xn + b * n4 + i) = v;
            }
        }
        __syncthreads();
    } else {
        if (comb) {
            if constexpr (B == 1) {
                const bool pre_ml = a.attn_n_head * 8u <= nthr;
                if (pre_ml) combine_weights<B, true>(a, wgt, r.ml_m, r.ml_l); else combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
                    const float *wg = wgt + (size_t)((i < n ? i : 0u) / a.attn_hd) * 8u;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int sp = 0; sp < 8; sp++) {
                        const float w = wg[sp];
                        acc.x += r.pv[j][sp].x * w; acc.y += r.pv[j][sp].y * w; acc.z += r.pv[j][sp].z * w; acc.w += r.pv[j][sp].w * w;
                    }
                    r.x[0][j] = acc;
                }
            } else {
                combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
                for (int b = 0; b < B; b++)
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
                        r.x[b][j] = (i < n && b < (int)a.nb) ? combine4(a, b, i, wgt) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
        }
        float ss[B];
#pragma unroll
        for (int b = 0; b < B; b++) ss[b] = 1.0f;
        if (norm) {
#pragma unroll
            for (int b = 0; b < B; b++) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc += r.x[b][j].x * r.x[b][j].x; acc += r.x[b][j].y * r.x[b][j].y;
                    acc += r.x[b][j].z * r.x[b][j].z; acc += r.x[b][j].w * r.x[b][j].w;
                }
                acc = dpp_wave_sum(acc);
                if (lane == 0) red[b * 16 + wid] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < B; b++) {
                float t = 0.0f;
                for (uint32_t w = 0; w < NW; w++) t += red[b * 16 + w];
                t /= (float)n; t += 1e-5f;
                ss[b] = 1.0f / sqrtf(t);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
#pragma unroll
            for (int b = 0; b < B; b++) {
                float4 v = r.x[b][j];
                if (norm) {
                    v.x = r.nw[j].x * (ss[b] * v.x); v.y = r.nw[j].y * (ss[b] * v.y);
                    v.z = r.nw[j].z * (ss[b] * v.z); v.w = r.nw[j].w * (ss[b] * v.w);
                }
                if (keep) r.x[b][j] = v;
                else if (i < n) *reinterpret_cast<float4 *>(xn + b * n4 + i) = v;
            }
        }
        if (!keep) __syncthreads();
    }
}

// The block quantizer on values that are still in registers (whole blocks only: n % 256 == 0).  Thread t of a launch
// holds elements 4 (t + j nthr) .. +3, so a 32-element group is 8 consecutive lanes and a 256-element block is exactly one
// wave: group min / max / nibble sum by three DPP steps, the block's maximum scale and bias by three cross-lane steps more
// -- no LDS round trip and no barrier between the phases (quantize_q4k_wg needs two).  Same values, same comparisons
// (reference tensor.c:144-242).  Ends with a barrier (SYNC; without it the caller's next barrier completes the staged groups).
template <int B, int NV, bool SYNC = true>
__device__ __forceinline__ void quantize_q4k_regs(const GemvDev &a, const Staged<B, NV> &r, XGroup *xg) {
    if constexpr (NV == 0) { (void)a; (void)r; (void)xg; } else {
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n, GT = (n >> 8) * 8u;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
        const bool valid = i < n;                                     // whole waves: n % 256 == 0
        // a wave past the end of the row (512 threads on a 1024-value row: half of them) skips the ~170 instructions: it would
        // share its SIMD's issue slots with a wave that has a block to quantize
        if (!valid) continue;
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < (int)a.nb) {
                const float4 v = r.x[b][j];
                float lo = FLT_MAX, hi = FLT_TRUE_MIN;                // reference: strict comparisons from these start values (NaN ignored)
                lo = (v.x < lo) ? v.x : lo; lo = (v.y < lo) ? v.y : lo; lo = (v.z < lo) ? v.z : lo; lo = (v.w < lo) ? v.w : lo;
                hi = (v.x > hi) ? v.x : hi; hi = (v.y > hi) ? v.y : hi; hi = (v.z > hi) ? v.z : hi; hi = (v.w > hi) ? v.w : hi;
                lo = fminf(lo, DPP_F(lo, 0xB1)); hi = fmaxf(hi, DPP_F(hi, 0xB1));
                lo = fminf(lo, DPP_F(lo, 0x4E)); hi = fmaxf(hi, DPP_F(hi, 0x4E));
                lo = fminf(lo, DPP_F(lo, 0x141)); hi = fmaxf(hi, DPP_F(hi, 0x141));
                const float gsc = (lo <= 0.0f) ? div_const<15>(hi - lo) : div_const<15>(hi);
                const float gbi = (lo <= 0.0f) ? (-lo) : 0.0f;
                uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
                if (gsc != 0.0f) {
                    n0 = (uint32_t)(nearest_int_magic((v.x + gbi) / gsc) & 0x0f); n1 = (uint32_t)(nearest_int_magic((v.y + gbi) / gsc) & 0x0f);
                    n2 = (uint32_t)(nearest_int_magic((v.z + gbi) / gsc) & 0x0f); n3 = (uint32_t)(nearest_int_magic((v.w + gbi) / gsc) & 0x0f);
                }
                const int sum = dpp_group_sum<8>((int)(n0 + n1 + n2 + n3));
                // the block's 8 groups are the 8 lane-octets of this wave
                float smax = (gsc > FLT_TRUE_MIN) ? gsc : FLT_TRUE_MIN, bmax = (gbi > FLT_TRUE_MIN) ? gbi : FLT_TRUE_MIN;   // the reference's strict comparisons
                // across the wave's eight lane-octets: lane ^ 8 by a DPP row rotate, lane ^ 16 / ^ 32 by v_permlane16/32_swap (VALU only;
                // round 3: the three ds_bpermute pairs this replaces were ~0.25 us of every Q4K launch's prologue)
                {
                    const float so = DPP_F(smax, 0x128), bo = DPP_F(bmax, 0x128);
                    smax = (so > smax) ? so : smax; bmax = (bo > bmax) ? bo : bmax;
                }
                {
                    const u32x2_q rs_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(smax), __float_as_uint(smax), false, false);
                    const u32x2_q rb_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(bmax), __float_as_uint(bmax), false, false);
                    const float s0 = __uint_as_float(rs_[0]), s1 = __uint_as_float(rs_[1]), b0 = __uint_as_float(rb_[0]), b1 = __uint_as_float(rb_[1]);
                    smax = (s1 > s0) ? s1 : s0; bmax = (b1 > b0) ? b1 : b0;
                }
                {
                    const u32x2_q rs_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(smax), __float_as_uint(smax), false, false);
                    const u32x2_q rb_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(bmax), __float_as_uint(bmax), false, false);
                    const float s0 = __uint_as_float(rs_[0]), s1 = __uint_as_float(rs_[1]), b0 = __uint_as_float(rb_[0]), b1 = __uint_as_float(rb_[1]);
                    smax = (s1 > s0) ? s1 : s0; bmax = (b1 > b0) ? b1 : b0;
                }
                const float s_scale = div_const<63>(smax), s_bias = div_const<63>(bmax);
                const uint32_t s6 = (s_scale == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(gsc / s_scale) & 0x3f);
                const uint32_t b6 = (s_bias == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(gbi / s_bias) & 0x3f);
                if (valid) {
                    const uint32_t tg = tid & 7u;
                    XGroup *o = xg + (size_t)b * GT + (i >> 5);
                    // this thread's four elements 4 tg .. 4 tg + 3 are two bytes of the packed group
                    uint8_t *ob = reinterpret_cast<uint8_t *>(o) + tg * 2;
                    *reinterpret_cast<uint16_t *>(ob) = (uint16_t)(n0 | (n1 << 4) | (n2 << 8) | (n3 << 12));
                    if (tg == 0) { o->sq = (float)s6 * s_scale; o->bq = (float)b6 * s_bias; o->sumq = sum; o->_pad = 0; }   // sq / bq: what get_group_scale_and_bias() reads back (tensor.c:137-140)
                }
            }
        }
    }
    if (SYNC) __syncthreads();
    }
}

// Block-quantize xn[B][n4] into the staged groups xg[B][GT] (reference quantize_tensor_q4k_in_situ on a 1-D tensor,
// tensor.c:281-310 + 144-242), all blocks at once: phase 1 = one thread per element, phase 2 = one thread per group.
// tmp: [B][bpl][16] floats (group scales, group biases).  Ends with a barrier.
__device__ __forceinline__ void quantize_q4k_wg(const GemvDev &a, const float *xn, XGroup *xg, float *tmp, uint32_t n4, int nbq) {
    const int n = (int)a.n, tid = threadIdx.x, nthr = (int)a.nthr;
    const int bpl = (n + 255) / 256, GT = bpl * 8;
    // phase 1: a thread owns FOUR consecutive elements (one 16-byte LDS read; the four divisions are independent), a
    // 32-element group = 8 consecutive lanes (three DPP steps for min / max / nibble sum)
    for (int idx = tid; idx < nbq * bpl * 64; idx += nthr) {
        const int t4 = idx & 63, j = (idx >> 6) % bpl, b = (idx >> 6) / bpl;
        const int d = (n >= (j + 1) * 256) ? 256 : (n - j * 256);
        const int e0 = 4 * t4;                                            // first of this thread's elements inside the block (d % 4 == 0)
        const bool valid = e0 < d;
        const float4 v = valid ? *reinterpret_cast<const float4 *>(xn + (size_t)b * n4 + (size_t)j * d + e0) : make_float4(0.f, 0.f, 0.f, 0.f);   // sic: j*d (reference tensor.c:307)
        // reference: min starts at FLT_MAX, max at FLT_TRUE_MIN, strict comparisons (NaN ignored)
        float lo = FLT_MAX, hi = FLT_TRUE_MIN;
        if (valid) {
            lo = (v.x < lo) ? v.x : lo; lo = (v.y < lo) ? v.y : lo; lo = (v.z < lo) ? v.z : lo; lo = (v.w < lo) ? v.w : lo;
            hi = (v.x > hi) ? v.x : hi; hi = (v.y > hi) ? v.y : hi; hi = (v.z > hi) ? v.z : hi; hi = (v.w > hi) ? v.w : hi;
        }
        lo = fminf(lo, DPP_F(lo, 0xB1)); hi = fmaxf(hi, DPP_F(hi, 0xB1));
        lo = fminf(lo, DPP_F(lo, 0x4E)); hi = fmaxf(hi, DPP_F(hi, 0x4E));
        lo = fminf(lo, DPP_F(lo, 0x141)); hi = fmaxf(hi, DPP_F(hi, 0x141));
        const float gsc = (lo <= 0.0f) ? div_const<15>(hi - lo) : div_const<15>(hi);
        const float gbi = (lo <= 0.0f) ? (-lo) : 0.0f;
        uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        if (valid && gsc != 0.0f) {
            n0 = (uint32_t)(nearest_int_magic((v.x + gbi) / gsc) & 0x0f); n1 = (uint32_t)(nearest_int_magic((v.y + gbi) / gsc) & 0x0f);
            n2 = (uint32_t)(nearest_int_magic((v.z + gbi) / gsc) & 0x0f); n3 = (uint32_t)(nearest_int_magic((v.w + gbi) / gsc) & 0x0f);
        }
        const int g = t4 >> 3, tg = t4 & 7;                               // group of the block, thread inside the group
        XGroup *o = xg + (size_t)b * GT + j * 8 + g;
        // this thread's four elements 4 tg .. 4 tg + 3 are two bytes of the packed group
        uint8_t *ob = reinterpret_cast<uint8_t *>(o) + tg * 2;
        *reinterpret_cast<uint16_t *>(ob) = (uint16_t)(n0 | (n1 << 4) | (n2 << 8) | (n3 << 12));
        const int sum = dpp_group_sum<8>((int)(n0 + n1 + n2 + n3));
        if (tg == 0) { o->sumq = sum; o->_pad = 0; float *tp = tmp + ((size_t)b * bpl + j) * 16; tp[g] = gsc; tp[8 + g] = gbi; }
    }
    __syncthreads();
    for (int idx = tid; idx < nbq * GT; idx += nthr) {
        const int gg = idx % GT, b = idx / GT, j = gg >> 3, g = gg & 7;
        const float *tp = tmp + ((size_t)b * bpl + j) * 16;
        const float4 s0 = *reinterpret_cast<const float4 *>(tp), s1 = *reinterpret_cast<const float4 *>(tp + 4);      // the block's 8 group scales
        const float4 c0 = *reinterpret_cast<const float4 *>(tp + 8), c1 = *reinterpret_cast<const float4 *>(tp + 12);  // ... and 8 group biases
        const float sv[8] = { s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w }, bv[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
        float smax = FLT_TRUE_MIN, bmax = FLT_TRUE_MIN, sg = sv[0], bg = bv[0];
#pragma unroll
        for (int k = 0; k < 8; k++) { if (sv[k] > smax) smax = sv[k]; if (bv[k] > bmax) bmax = bv[k]; sg = (k == g) ? sv[k] : sg; bg = (k == g) ? bv[k] : bg; }
        const float s_scale = div_const<63>(smax), s_bias = div_const<63>(bmax);
        const uint32_t s6 = (s_scale == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(sg / s_scale) & 0x3f);
        const uint32_t b6 = (s_bias == 0.0f) ? 0u : (uint32_t)(nearest_int_magic(bg / s_bias) & 0x3f);
        XGroup *o = xg + (size_t)b * GT + gg;
        o->sq = (float)s6 * s_scale;       // what get_group_scale_and_bias() reads back (tensor.c:137-140)
        o->bq = (float)b6 * s_bias;
    }
    __syncthreads();
}

// operator-test path: unpack caller-supplied activation blocks (one sequence) into the staged groups
__device__ __forceinline__ void unpack_q4k_wg(const GemvDev &a, XGroup *xg) {
    const int n = (int)a.n, GT = ((n + 255) / 256) * 8;
    const uint8_t *x4 = reinterpret_cast<const uint8_t *>(a.xq_in);
    for (int gg = threadIdx.x; gg < GT; gg += (int)a.nthr) {
        const uint8_t *blk = x4 + (size_t)(gg >> 3) * 160;
        const int g = gg & 7;
        const float s_scale = *reinterpret_cast<const float *>(blk + 12), s_bias = *reinterpret_cast<const float *>(blk + 16);
        uint32_t s6, b6;
        q4k_unpack6(*reinterpret_cast<const uint32_t *>(blk + 20), *reinterpret_cast<const uint32_t *>(blk + 24),
                    *reinterpret_cast<const uint32_t *>(blk + 28), g, s6, b6);
        XGroup o; int sum = 0;
        for (int m = 0; m < 4; m++) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(blk + 32 + g * 16 + m * 4);
            o.pk[m] = w;
            sum += (int)__builtin_amdgcn_udot8(w, 0x11111111u, 0u, false);
        }
        o.sq = (float)s6 * s_scale; o.bq = (float)b6 * s_bias; o.sumq = sum; o._pad = 0;
        xg[gg] = o;
    }
    __syncthreads();
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bload_u4(__amdgpu_buffer_rsrc_t r, uint32_t off, bool nt) {
    const i32x4 v = nt ? __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
}

}  // namespace
}  // namespace nano
