// gemv_q4k.hip -- Q4K (W4A4) fused GEMV and the run-time activation block quantizer for gfx950.
//
// Restates, for the device:
//   * quantize_one_block_q4k_in_situ / quantize_tensor_q4k_in_situ (reference infer/tensor.c:144-242,281-310):
//     per 32-value group min/max -> 4-bit asymmetric codes, then the 8 group scales and 8 biases are
//     themselves quantized to 6 bits against the block maxima and packed into 12 bytes.  All of it is
//     order-free (min/max, elementwise divides, magic-number rounding) -> BIT-EXACT for equal inputs.
//     The reference's partial-block source offset j*d (tensor.c:307) is kept.
//   * dot_two_blocks_q4k / matmul_q4k (reference infer/tensor.c:359-434,438-471): three integer sums
//     per group (v_dot8_u32_u4 on the packed nibbles), the four-term float combine in the reference's
//     operation order, groups summed in order inside a block and blocks summed in order along the
//     row -> bit-identical fp32 results for equal quantized inputs.
//
// Block layout in HBM (160 B, reference infer/tensor.h:116-135), blocks 16-byte aligned after upload:
//   +0 u32 0x42 | +4 u32 length | +8 u32 meta | +12 f32 s_scale | +16 f32 s_bias | +20 u8 sb[12] | +32 u8 value[128]
//
// LDS staging of the activation (per sequence, per group): the 32 nibbles packed exactly like a weight group's 16 value
// bytes (element 2i in the low nibble of byte i), so that ONE v_dot8_u32_u4 per dword gives the eight products of
// matching elements (round 3; rounds 1-2 split both operands into even / odd byte lanes for v_dot4_u32_u8: 24 instead of
// 8 instructions per group and sequence); the dequantized 6-bit scale / bias floats and the nibble sum.
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

// ---- stand-alone activation quantizer (operator tests): x[n] -> ceil(n/256) blocks of 160 B ----------
__global__ __launch_bounds__(256) void quantize_q4k_kernel(const float *x, uint32_t n, uint8_t *blocks) {
    __shared__ float tmp[16];
    __shared__ uint8_t nibs[256];
    const int t = threadIdx.x;
    const uint32_t bpl = (n + 255) / 256;
    for (uint32_t j = 0; j < bpl; j++) {
        const uint32_t d = (n >= (j + 1) * 256) ? 256 : (n - j * 256);
        const bool valid = (uint32_t)t < d;
        const float v = valid ? x[(size_t)j * d + t] : 0.0f;     // sic: j*d (reference tensor.c:307)
        Q4kBlockHdr hdr; float sq, bq;
        const uint32_t nib = q4k_quantize_block_coop(v, valid, tmp, hdr, sq, bq);
        nibs[t] = (uint8_t)nib;
        __syncthreads();
        uint8_t *blk = blocks + (size_t)j * 160;
        if (t < 128) blk[32 + t] = (uint8_t)((nibs[2 * t] & 0x0f) | (nibs[2 * t + 1] << 4));
        if (t == 0) {
            uint32_t *w = reinterpret_cast<uint32_t *>(blk);
            w[0] = 0x42u; w[1] = d; w[2] = 0u;
            w[3] = __float_as_uint(hdr.s_scale); w[4] = __float_as_uint(hdr.s_bias);
            w[5] = hdr.sb[0]; w[6] = hdr.sb[1]; w[7] = hdr.sb[2];
        }
        __syncthreads();
    }
}
hipError_t launch_quantize_q4k(const float *x, uint32_t n, uint8_t *blocks, hipStream_t st) {
    hipLaunchKernelGGL(quantize_q4k_kernel, dim3(1), dim3(256), 0, st, x, n, blocks);
    return hipGetLastError();
}

// ---- fused GEMV (SLAB structure, see gemv_q80_impl.h) -------------------------------------------------------------
// A workgroup owns `rw` consecutive rows; its items (row, 32-weight group) are dealt to its threads, every thread issues
// the activation loads and then ALL its weight loads (16 nibble bytes + the 32-byte block header, buffer descriptors)
// at kernel entry; the activation is normalised from registers into LDS, block-quantized by the whole workgroup in two
// barrier-separated phases (group min/max + nibbles, then the 6-bit scale/bias quantization against the block maxima),
// and the per-group results land in an LDS table that one thread per (row, sequence) folds in the reference's order.
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
// (reference tensor.c:144-242).  Ends with a barrier: the staged groups are complete.
template <int B, int NV>
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
    __syncthreads();
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

template <int ROLE, int B, int NV, int IPT>
__global__ __launch_bounds__(1024) void gemv_q4k_slab_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nthr = (int)a.nthr;
    const uint32_t n = a.n, n4 = (n + 3) & ~3u;
    const uint32_t bpl = (n + 255) / 256, GT = bpl * 8, GTP = GT + 4;        // row pitch of the product table: 16-byte aligned rows
    const uint32_t RW = a.rw;
    const uint32_t epi = role_epi<ROLE>(a);
    const bool swiglu = epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = swiglu ? 2 : 1;
    // LDS: xg[B][GT] | xn[B][n4] | tmp[B][bpl][16] | red[B*16 (+ combine weights)] | P[B][nmat][RW][GTP]
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *xn = reinterpret_cast<float *>(smem + (size_t)B * GT * sizeof(XGroup));
    float *tmp = xn + B * n4;
    float *red = tmp + B * bpl * 16;
    float *P = red + B * 16 + (has_flag<ROLE>(a, F_COMBINE) ? B * a.attn_n_head * 8 : 0);

    // late-read arguments are fetched with the first ones (karg_touch, gemv_common.h)
    karg_touch(a.out[0]); karg_touch(a.out_bstride[0]); karg_touch(a.out_pstride[0]); karg_touch(a.nb); karg_touch(a.magic_nchunk); karg_touch(a.log2_tiles); karg_touch(a.tile_max); karg_touch(a.ntiles);
    if (!swiglu) { karg_touch(a.out[1]); karg_touch(a.out[2]); karg_touch(a.out_bstride[1]); karg_touch(a.out_bstride[2]); karg_touch(a.out_pstride[1]); karg_touch(a.out_pstride[2]); }
    karg_touch(a.pos);
    if (ROLE == R_GENERIC || ROLE == R_RESID || ROLE == R_RESID_COMBINE) { karg_touch(a.resid_add); karg_touch(a.resid_add_bstride); }
    NANO_STAMP(a.stamps, 0, tid);
    Staged<B, NV> sx;
    stage_issue<ROLE, B, NV>(a, sx);

    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const uint8_t *w0 = reinterpret_cast<const uint8_t *>(sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2]);
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0, rows0 * bpl * 160u);
    const __amdgpu_buffer_rsrc_t rw1 = mkrsrc(swiglu ? a.w[1] : nullptr, swiglu ? rows0 * bpl * 160u : 0u);

    // item it -> (matrix, local row, group); group fastest so that consecutive lanes read consecutive nibble runs
    const uint32_t items = RW * GT * nmat;
    uint4 nibv[IPT], h0v[IPT], h1v[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t it = (uint32_t)tid + (uint32_t)k * nthr;
        const uint32_t rr = __umulhi(it, a.magic_nchunk), gg = it - rr * GT;   // it / GT, it % GT (magic_nchunk = ceil(2^32 / GT) here); rr = mat * RW + local row
        // Which matrix: only the SwiGLU launch has two, and when a matrix's items fill whole waves (RW x GT a multiple of 64:
        // the launcher gives the R_NORM_SWIGLU role to such plans only, others run R_GENERIC) the choice is wave-uniform.  Said to the compiler (readfirstlane), the descriptor
        // is picked with scalar selects; left per-lane it becomes a waterfall loop around each of the three loads -- and the
        // register reuse between them put a full s_waitcnt vmcnt(0) in the middle of the issue phase (round 3: ~1 us per launch).
        auto issue_item = [&](const uint32_t mat) __attribute__((always_inline)) {
            const uint32_t rl = rr - mat * RW;
            const uint32_t boff = (it < items) ? ((lrow0 + rl) * bpl + (gg >> 3)) * 160u : OOB;   // rows beyond the segment: out of range -> 0
            const bool m1 = mat != 0;
            nibv[k] = m1 ? bload_u4(rw1, boff == OOB ? OOB : boff + 32u + (gg & 7u) * 16u, true) : bload_u4(rw0, boff == OOB ? OOB : boff + 32u + (gg & 7u) * 16u, true);
            h0v[k] = m1 ? bload_u4(rw1, boff, false) : bload_u4(rw0, boff, false);
            h1v[k] = m1 ? bload_u4(rw1, boff == OOB ? OOB : boff + 16u, false) : bload_u4(rw0, boff == OOB ? OOB : boff + 16u, false);
        };
        if (!swiglu) issue_item(0u);
        else if (ROLE == R_NORM_SWIGLU) issue_item((uint32_t)__builtin_amdgcn_readfirstlane((int)(rr >= RW ? 1u : 0u)));   // the launcher checked: whole waves per matrix
        else issue_item(rr >= RW ? 1u : 0u);
    }
    const uint32_t lrw = a.log2_tiles;                                // log2(RW) here
    const int fb = tid >> lrw, frl = tid & ((int)RW - 1);
    const bool fold_live = tid < (int)(RW * B) && fb < (int)a.nb && lrow0 + frl < rows0;
    // the position of a pos-indexed output (v-cache row) is fetched now and used only by the final store: no wait
    // here (a wait on it would also wait for every weight load issued above -- vmcnt counts in order)
    uint32_t opos = 0;
    if (ops && fold_live) opos = a.pos[fb];
    float oldv = 0.0f;
    if (epi == GEMV_EPI_RESID && fold_live) oldv = out0[(size_t)fb * obs + lrow0 + frl];      // residual stream: never pos-indexed
    float addv = 0.0f;                                              // LoRA o-branch: x += (W.act + addv), reference order
    const bool has_add = epi == GEMV_EPI_RESID && a.resid_add != nullptr;
    if (has_add && fold_live) addv = a.resid_add[(size_t)fb * a.resid_add_bstride + lrow0 + frl];

    NANO_STAMP(a.stamps, 1, tid);                                   // every load issued
    if (has_flag<ROLE>(a, F_PRE)) unpack_q4k_wg(a, xg);
    else {
        const bool regq = NV > 0 && (n & 255u) == 0u;                // whole blocks: quantize from registers, wave-local
        stage_xn<ROLE, B, NV>(a, sx, xn, red, n4, regq);
        NANO_STAMP(a.stamps, 2, red[0]);                            // the activation arrived and is normalised
        if (regq) quantize_q4k_regs<B, NV>(a, sx, xg);
        else quantize_q4k_wg(a, xn, xg, tmp, n4, (int)a.nb);
    }
    NANO_STAMP(a.stamps, 3, xg[0].sq);                              // block-quantized activation staged in LDS

#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint32_t it = (uint32_t)tid + (uint32_t)k * nthr;
        if (it < items) {
            const uint32_t rr = __umulhi(it, a.magic_nchunk), gg = it - rr * GT, g = gg & 7u;
            const uint4 nib = nibv[k];
            // the header words nobody reads stay "live" up to here: declared dead at the load, their registers were handed to the
            // next load's address -- which then had to wait (s_waitcnt vmcnt(0)) for the load in flight to write them
            asm volatile("" :: "v"(h0v[k].x), "v"(h0v[k].z));
            const float s_scale = __uint_as_float(h0v[k].w);
            const int len = (int)h0v[k].y;
            uint32_t s6, b6;
            q4k_unpack6(h1v[k].y, h1v[k].z, h1v[k].w, (int)g, s6, b6);
            const float sp = (float)s6 * s_scale, bp = (float)b6 * __uint_as_float(h1v[k].x);
            const int glen = (len >= (int)(g + 1) * 32) ? 32 : (len - 32 * (int)g);
            const uint32_t wn[4] = { nib.x, nib.y, nib.z, nib.w };
            uint32_t sump = 0;                                           // sum of the weight nibbles: v_dot8_u32_u4 against eight ones
#pragma unroll
            for (int m = 0; m < 4; m++) sump = __builtin_amdgcn_udot8(wn[m], 0x11111111u, sump, false);
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (b < (int)a.nb) {
                    const XGroup &xq = xg[(size_t)b * GT + gg];
                    uint32_t spq = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) spq = __builtin_amdgcn_udot8(wn[m], xq.pk[m], spq, false);      // nibble k of both dwords = element 8 m + k
                    const float sq = xq.sq, bq = xq.bq;
                    // reference tensor.c:425-428, same association
                    const float grp = sp * sq * (float)(int)spq - sp * bq * (float)(int)sump - sq * bp * (float)xq.sumq + glen * bp * bq;
                    P[((size_t)b * nmat * RW + rr) * GTP + gg] = grp;
                }
            }
        }
    }
    NANO_STAMP(a.stamps, 4, (float)nibv[IPT - 1].x);                // this thread's weights arrived, its products are in the table
    __syncthreads();
    NANO_STAMP(a.stamps, 5, P[0]);

    // ordered fold (groups inside a block, then blocks along the row; reference tensor.c:359-434, 438-471).  The eight group
    // values of a block are two 16-byte LDS reads; the reads of four blocks go out together and the block sums (independent
    // chains) overlap -- only the sum over the blocks is serial.
    if (tid < (int)(RW * B)) {
        float res[2] = {0.0f, 0.0f};
        const uint32_t nfull = n >> 8;                                   // whole 256-value blocks
        for (uint32_t mat = 0; mat < nmat; mat++) {
            const float *f = P + ((size_t)fb * nmat * RW + mat * RW + frl) * GTP;
            float line = 0.0f;
            uint32_t blk = 0;
            for (; blk + 4 <= nfull; blk += 4) {
                float4 v[4][2];
#pragma unroll
                for (int k = 0; k < 4; k++) { v[k][0] = *reinterpret_cast<const float4 *>(f + (blk + k) * 8); v[k][1] = *reinterpret_cast<const float4 *>(f + (blk + k) * 8 + 4); }
                float ds[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float d = 0.0f;
                    d += v[k][0].x; d += v[k][0].y; d += v[k][0].z; d += v[k][0].w;
                    d += v[k][1].x; d += v[k][1].y; d += v[k][1].z; d += v[k][1].w;
                    ds[k] = d;
                }
                line += ds[0]; line += ds[1]; line += ds[2]; line += ds[3];
            }
            for (; blk < bpl; blk++) {
                const int d = ((int)n >= (int)(blk + 1) * 256) ? 256 : ((int)n - (int)blk * 256);
                const int gv = (d + 31) >> 5;
                float ds = 0.0f;
                for (int g = 0; g < gv; g++) ds += f[blk * 8 + g];
                line += ds;
            }
            res[mat] = line;
        }
        // write-through (sc1) store, see gemv_q80_impl.h: nothing is left for the write-back at the end of the kernel
        const float val = finish_epi(epi, has_add ? res[0] + addv : res[0], res[1], oldv);
        if (fold_live) __hip_atomic_store(out0 + (size_t)fb * obs + (size_t)opos * ops + lrow0 + frl, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // classifier launches (one STORE segment): this workgroup's (max, first row) arg-max partial per sequence, so that the
        // arg-max kernel scans gridDim.x pairs instead of every logit (the Q80 STREAM kernel's tile_max, gemv_q80_impl.h)
        if (a.tile_max) {
            float bv = fold_live ? val : -INFINITY;
            uint32_t bi = fold_live ? lrow0 + (uint32_t)frl : 0xffffffffu;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                if (o < (int)RW) {                                       // the RW fold threads of a sequence are RW consecutive lanes
                    const float ov = __shfl_xor(bv, o, 64);
                    const uint32_t oi = __shfl_xor(bi, o, 64);
                    if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
            }
            if (frl == 0 && fb < (int)a.nb) { float *tm = a.tile_max + ((size_t)fb * a.ntiles + blockIdx.x) * 2; tm[0] = bv; tm[1] = __uint_as_float(bi); }
        }
    }
    NANO_STAMP_END(a.stamps, 6);
}

struct Q4kPlan { uint32_t rw, nthr, ipt, nv; };
static Q4kPlan plan_q4k(const GemvArgs &a, int B) {
    const uint32_t GT = ((a.n + 255) / 256) * 8, nmat = a.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    uint32_t align = 0;
    if (nseg > 1) for (uint32_t s = 0; s < nseg; s++) align |= a.seg[s].rows;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    // ~512 items (8 KB of nibbles) per workgroup, >= 128 workgroups; tall matrices: up to 2048 items
    uint32_t rw = 4;
    // Every workgroup quantizes the whole activation before its first product, so large matrices (Qwen3-4B's layers, >= 8 M
    // weights) take up to 4096 items per workgroup: swept on the device with tools/q4k_wide.sh (QKV 17.0 -> 9.0 us, W1|W3
    // 28.0 -> 19.9, Wo 10.4 -> 7.5); Qwen3-0.6B's are fastest at 512 (1024: -3 %).  NANO_Q4K_ITEMS overrides the former.
    const bool large = (uint64_t)rows * a.n >= (8u << 20);
    uint32_t cap = large ? 4096u : rows >= 16384 ? 2048u : 512u;
    static const char *cap_env = getenv("NANO_Q4K_ITEMS");           // measurement knobs: items per workgroup (large matrices),
    if (cap_env && large) cap = (uint32_t)strtoul(cap_env, nullptr, 0);
    static const uint32_t cap_small = getenv("NANO_Q4K_ITEMS_SMALL") ? (uint32_t)strtoul(getenv("NANO_Q4K_ITEMS_SMALL"), nullptr, 0) : 0u;   // ... (per-layer matrices of small models),
    static const uint32_t nthr_max = getenv("NANO_Q4K_NTHR") ? (uint32_t)strtoul(getenv("NANO_Q4K_NTHR"), nullptr, 0) : 512u;                 // ... threads per workgroup
    if (cap_small && !large && rows < 16384) cap = cap_small;
    static const uint32_t cap_swiglu = getenv("NANO_Q4K_ITEMS_SWIGLU") ? (uint32_t)strtoul(getenv("NANO_Q4K_ITEMS_SWIGLU"), nullptr, 0) : 0u;   // ... of the two-matrix launch alone
    if (!large && nmat == 2 && rows < 16384) cap = cap_swiglu ? cap_swiglu : 1024u;     // measured (Qwen3-0.6B W1|W3): 384 workgroups of 512 items 1542 tok/s, 192 of 1024: 1595
    while (rw < 64 && (align % (rw * 2)) == 0 && (rw * 2) * GT * nmat <= cap && rows / (rw * 2) >= 128) rw *= 2;
    for (;; rw /= 2) {
        const uint32_t items = rw * GT * nmat;
        uint32_t nthr = ((items + 63) / 64) * 64;
        if (nthr > nthr_max) nthr = nthr_max;
        if (nthr < 256) nthr = 256;
        uint32_t want = ((a.n / 4 + 63) / 64) * 64;        // the block quantizer is one thread per element: ~4 elements per thread
        if (want > 1024) want = 1024;
        if (nthr < want) nthr = want;
        if (nthr < rw * (uint32_t)B) nthr = ((rw * (uint32_t)B + 63) / 64) * 64;
        const uint32_t ipt = (items + nthr - 1) / nthr;
        if (ipt <= 4 || rw <= 4) return Q4kPlan{rw, nthr, ipt, (a.n + 4 * nthr - 1) / (4 * nthr)};   // the kernel is instantiated for <= 4 items per thread
    }
}

// dynamic LDS of a launch with capacity B: quantized groups, the activations, block / sequence scratch, combine weights, products
static size_t q4k_lds_bytes(uint32_t n, uint32_t epi, bool combine, uint32_t attn_n_head, uint32_t rw, uint32_t B) {
    const uint32_t nmat = epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n4 = (n + 3) & ~3u, bpl = (n + 255) / 256, GT = bpl * 8;
    return (size_t)B * GT * sizeof(XGroup) + (B * n4 + B * bpl * 16 + B * 16 + (combine ? (size_t)B * attn_n_head * 8 : 0) + (size_t)B * nmat * rw * (GT + 4)) * 4 + 16;
}

template <int ROLE, int B, int NV, int IPT>
static hipError_t launch_q4k_t(const GemvDev &d, const Q4kPlan &p, uint32_t rows, hipStream_t st) {
    const size_t lds = q4k_lds_bytes(d.n, d.epi, (d.flags & F_COMBINE) != 0, d.attn_n_head, p.rw, B);
    if (lds > 160 * 1024) return hipErrorInvalidValue;                 // gemv_q4k_fit_batch() tells the caller how many sequences fit
    auto kern = &gemv_q4k_slab_kernel<ROLE, B, NV, IPT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemvDev dd = d; dd.nthr = p.nthr;
    { const uint32_t GT = ((d.n + 255) / 256) * 8; dd.magic_nchunk = (uint32_t)(((1ull << 32) + GT - 1) / GT); dd.log2_tiles = 0; while ((1u << dd.log2_tiles) < p.rw) dd.log2_tiles++;
      dd.units = ((p.rw * GT) % 64u == 0u && p.nthr % 64u == 0u) ? 1u : 0u; }   // kernel: item -> (row, group), thread -> (sequence, row)
    hipLaunchKernelGGL(kern, dim3((rows + p.rw - 1) / p.rw), dim3(p.nthr), lds, st, dd);
    return hipGetLastError();
}
template <int ROLE, int B>
static hipError_t launch_q4k_r(const GemvDev &d, const Q4kPlan &p, uint32_t rows, hipStream_t st) {
    if (p.ipt > 4) return hipErrorInvalidValue;
#define Q4K_GO(NV_, IPT_) do { if constexpr (B * NV_ <= 8) return launch_q4k_t<ROLE, B, NV_, IPT_>(d, p, rows, st); } while (0)
    int nv = p.nv <= 1 ? 1 : p.nv <= 2 ? 2 : p.nv <= 4 ? 4 : 0;
    const int ipt = p.ipt <= 1 ? 1 : p.ipt <= 2 ? 2 : 4;
    if (B * nv > 8) nv = 0;
    if (nv == 1) { if (ipt == 1) Q4K_GO(1, 1); if (ipt == 2) Q4K_GO(1, 2); Q4K_GO(1, 4); }
    if (nv == 2) { if (ipt == 1) Q4K_GO(2, 1); if (ipt == 2) Q4K_GO(2, 2); Q4K_GO(2, 4); }
    if (nv == 4) { if (ipt == 1) Q4K_GO(4, 1); if (ipt == 2) Q4K_GO(4, 2); Q4K_GO(4, 4); }
    if (ipt == 1) Q4K_GO(0, 1);
    if (ipt == 2) Q4K_GO(0, 2);
    Q4K_GO(0, 4);
    return hipErrorInvalidValue;
#undef Q4K_GO
}
}  // namespace
// (max, row) arg-max partials a STORE launch with tile_max writes per sequence: one per workgroup of a one-segment launch
// (the classifier); 0 = none, the arg-max kernel scans the logits
uint32_t gemv_q4k_partials(const GemvArgs &a) {
    if (!a.tile_max || a.epi != GEMV_EPI_STORE || a.nseg != 1 || a.nb == 0 || a.nb > 8 || a.seg[0].out_pstride) return 0;
    const int B = a.nb <= 1 ? 1 : a.nb <= 2 ? 2 : a.nb <= 4 ? 4 : 8;
    const Q4kPlan p = plan_q4k(a, B);
    return (a.seg[0].rows + p.rw - 1) / p.rw;
}
namespace {
template <int B>
static hipError_t launch_q4k_b(const GemvArgs &a, hipStream_t st) {
    GemvDev d = to_dev(a);
    if (a.x4_in) { d.flags |= F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(a.x4_in); }
    const Q4kPlan p = plan_q4k(a, B);
    d.rw = p.rw;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    d.ntiles = gemv_q4k_partials(a);                                // arg-max partials: one per workgroup (0: none)
    if (!d.ntiles || d.ntiles != (rows + p.rw - 1) / p.rw) { d.tile_max = nullptr; d.ntiles = 0; }
    if constexpr (B == 1) {
        const uint32_t f = d.flags;
        if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_q4k_r<R_NORM_STORE, B>(d, p, rows, st);
        if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_q4k_r<R_RESID, B>(d, p, rows, st);
        if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_q4k_r<R_RESID_COMBINE, B>(d, p, rows, st);
        const uint32_t GT = ((d.n + 255) / 256) * 8;
        if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU && (p.rw * GT) % 64u == 0u && p.nthr % 64u == 0u) return launch_q4k_r<R_NORM_SWIGLU, B>(d, p, rows, st);
    }
    return launch_q4k_r<R_GENERIC, B>(d, p, rows, st);
}

}  // namespace

// Sequences per launch that fit the 160 KB of LDS (every workgroup holds the whole quantized activation of each sequence):
// 8 for Qwen3-0.6B's row lengths, 2 for Qwen3-4B's hidden size 9728.  The caller slices larger steps (backend.hip gemv()).
uint32_t gemv_q4k_fit_batch(const GemvArgs &a) {
    for (uint32_t c = 8; c > 1; c >>= 1)
        if (q4k_lds_bytes(a.n, a.epi, a.attn_part != nullptr, a.attn_n_head, plan_q4k(a, (int)c).rw, c) <= 160 * 1024) return c;
    return 1;
}

hipError_t launch_gemv_q4k(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    (void)max_wg;
    if (a.nb == 0 || a.nb > 8 || a.n % 4 || a.nseg == 0 || a.nseg > 3) return hipErrorInvalidValue;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s < a.nseg; s++) if (a.seg[s].rows % 4) return hipErrorInvalidValue;
    if (a.nb <= 1) return launch_q4k_b<1>(a, st);
    if (a.nb <= 2) return launch_q4k_b<2>(a, st);
    if (a.nb <= 4) return launch_q4k_b<4>(a, st);
    return launch_q4k_b<8>(a, st);
}

}  // namespace nano
