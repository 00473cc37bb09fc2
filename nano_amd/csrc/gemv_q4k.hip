// gemv_q4k.hip -- Q4K (W4A4) fused GEMV and the run-time activation block quantizer for gfx950.
//
// Restates, for the device:
//   * quantize_one_block_q4k_in_situ / quantize_tensor_q4k_in_situ (reference infer/tensor.c:144-242,281-310):
//     per 32-value group min/max -> 4-bit asymmetric codes, then the 8 group scales and 8 biases are
//     themselves quantized to 6 bits against the block maxima and packed into 12 bytes.  All of it is
//     order-free (min/max, elementwise divides, magic-number rounding) -> BIT-EXACT for equal inputs.
//     The reference's partial-block source offset j*d (tensor.c:307) is kept.
//   * dot_two_blocks_q4k / matmul_q4k (reference infer/tensor.c:359-434,438-471): three integer sums
//     per group (v_dot4_u32_u8 on split nibbles), the four-term float combine in the reference's
//     operation order, groups summed in order inside a block and blocks summed in order along the
//     row -> bit-identical fp32 results for equal quantized inputs.
//
// Block layout in HBM (160 B, reference infer/tensor.h:116-135), blocks 16-byte aligned after upload:
//   +0 u32 0x42 | +4 u32 length | +8 u32 meta | +12 f32 s_scale | +16 f32 s_bias | +20 u8 sb[12] | +32 u8 value[128]
//
// LDS staging of the activation (per sequence, per group): the 32 nibbles pre-split into
// "even elements" / "odd elements" byte lanes (the same split `& 0x0F0F0F0F`, `>> 4 & 0x0F0F0F0F`
// applied to a weight dword yields), the dequantized 6-bit scale/bias floats and the nibble sum.
#include <float.h>

#include "device_common.h"
#include "kernels.h"

namespace nano {

struct XGroup {            // 48 bytes per (sequence, group), 16-byte aligned
    uint32_t lo[4];        // even-index nibbles of dword m as 4 bytes
    uint32_t hi[4];        // odd-index nibbles
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
    const float gsc = (lo <= 0.0f) ? ((hi - lo) / 15.0f) : (hi / 15.0f);
    const float gbi = (lo <= 0.0f) ? (-lo) : 0.0f;
    uint32_t nib = 0;
    if (valid && gsc != 0.0f) nib = (uint32_t)(nearest_int_magic((v + gbi) / gsc) & 0x0f);
    __syncthreads();
    if ((t & 31) == 0) { tmp[g] = gsc; tmp[8 + g] = gbi; }
    __syncthreads();
    float smax = FLT_TRUE_MIN, bmax = FLT_TRUE_MIN;
#pragma unroll
    for (int k = 0; k < 8; k++) { if (tmp[k] > smax) smax = tmp[k]; if (tmp[8 + k] > bmax) bmax = tmp[8 + k]; }
    const float s_scale = smax / 63.0f, s_bias = bmax / 63.0f;
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

// ---- fused GEMV ---------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ void prologue_q4k(const GemvArgs &a, XGroup *xg, float *xn, float *red) {
    const int t = threadIdx.x;
    const int n = (int)a.n;
    const int bpl = (n + 255) / 256, GT = bpl * 8;
    uint8_t *xgb = reinterpret_cast<uint8_t *>(xg);
    if (a.x4_in) {          // operator-test path: unpack caller-supplied blocks (one sequence)
        for (int gg = t; gg < GT; gg += blockDim.x) {
            const uint8_t *blk = a.x4_in + (size_t)(gg >> 3) * 160;
            const int g = gg & 7;
            const float s_scale = *reinterpret_cast<const float *>(blk + 12), s_bias = *reinterpret_cast<const float *>(blk + 16);
            uint32_t s6, b6;
            q4k_unpack6(*reinterpret_cast<const uint32_t *>(blk + 20), *reinterpret_cast<const uint32_t *>(blk + 24),
                        *reinterpret_cast<const uint32_t *>(blk + 28), g, s6, b6);
            XGroup o; int sum = 0;
            for (int m = 0; m < 4; m++) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(blk + 32 + g * 16 + m * 4);
                o.lo[m] = w & 0x0f0f0f0fu; o.hi[m] = (w >> 4) & 0x0f0f0f0fu;
                sum += (int)__builtin_amdgcn_udot4(o.lo[m], 0x01010101u, 0u, false) + (int)__builtin_amdgcn_udot4(o.hi[m], 0x01010101u, 0u, false);
            }
            o.sq = (float)s6 * s_scale; o.bq = (float)b6 * s_bias; o.sumq = sum; o._pad = 0;
            xg[gg] = o;
        }
        __syncthreads();
        return;
    }
    for (int b = 0; b < B; b++) {
        if (b >= (int)a.nb) break;
        const float *x = a.xin + (size_t)b * a.xin_bstride;
        float ss = 1.0f;
        if (a.norm_w) {
            float acc = 0.0f;
            for (int i = t * 4; i < n; i += blockDim.x * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + i);
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            }
            ss = block_sum(acc, red);
            ss /= (float)n; ss += 1e-5f; ss = 1.0f / sqrtf(ss);
        }
        __syncthreads();
        if (a.attn_part) {      // input = combination of the split attention partials (attn.hip), as in gemv.hip
            const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
            float *wgt = red + 32;
            for (uint32_t h = t; h < nh; h += blockDim.x) {
                const float *ml = a.attn_ml + ((size_t)b * nh + h) * ns * 2;
                float M = -INFINITY;
                for (uint32_t s = 0; s < ns; s++) if (ml[2 * s + 1] > 0.0f) M = fmaxf(M, ml[2 * s]);
                float L = 0.0f;
                for (uint32_t s = 0; s < ns; s++) {
                    const float e = (ml[2 * s + 1] > 0.0f) ? expf(ml[2 * s] - M) : 0.0f;
                    wgt[h * ns + s] = e;
                    L += ml[2 * s + 1] * e;
                }
                for (uint32_t s = 0; s < ns; s++) wgt[h * ns + s] = wgt[h * ns + s] / L;
            }
            __syncthreads();
            const float *part = a.attn_part + (size_t)b * ns * n;
            for (int i = t; i < n; i += blockDim.x) {
                const int h = i / (int)a.attn_hd;
                float acc = 0.0f;
                for (uint32_t s = 0; s < ns; s++) acc += part[(size_t)s * n + i] * wgt[h * ns + s];
                xn[i] = acc;
            }
        } else {
            for (int i = t; i < n; i += blockDim.x) xn[i] = a.norm_w ? a.norm_w[i] * (ss * x[i]) : x[i];
        }
        __syncthreads();
        for (int j = 0; j < bpl; j++) {
            const int d = (n >= (j + 1) * 256) ? 256 : (n - j * 256);
            const bool valid = t < d;
            const float v = valid ? xn[(size_t)j * d + t] : 0.0f;       // sic: j*d
            Q4kBlockHdr hdr; float sq, bq;
            const uint32_t nib = q4k_quantize_block_coop(v, valid, red, hdr, sq, bq);
            const int g = t >> 5, e = t & 31;
            XGroup *o = xg + (size_t)b * GT + j * 8 + g;
            // byte lane of element e inside the split-nibble dwords
            uint8_t *ob = reinterpret_cast<uint8_t *>(o);
            ob[((e & 1) ? 16 : 0) + (e >> 3) * 4 + ((e & 7) >> 1)] = (uint8_t)nib;
            const int sum = group_sum_i((int)nib, 32);
            if (e == 0) { o->sq = sq; o->bq = bq; o->sumq = sum; o->_pad = 0; }
            __syncthreads();
        }
    }
    (void)xgb;
    __syncthreads();
}

template <int B, int RB>
__global__ __launch_bounds__(256) void gemv_q4k_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = (int)a.n;
    const int bpl = (n + 255) / 256, GT = bpl * 8;
    const int pitch = GT + 1;
    // LDS carve: XGroup[B*GT] | xn[n] | red[32] | fold[4][RB*B*pitch]
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *xn = reinterpret_cast<float *>(smem + (size_t)B * GT * sizeof(XGroup));
    float *red = xn + ((n + 3) & ~3);
    float *foldbase = red + 32;

    prologue_q4k<B>(a, xg, xn, red);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float *fold = foldbase + (size_t)wid * (RB * B) * pitch;
    const int nb = (int)a.nb;
    const int items = RB * GT;
    const int pr = lane / B, pb = lane % B;

    for (uint32_t tile = blockIdx.x * 4 + wid; tile < a.tiles; tile += gridDim.x * 4) {
        uint32_t sidx, row0;
        if (a.epi == GEMV_EPI_SWIGLU) { sidx = 0; row0 = tile * RB; if (row0 >= a.seg[0].rows) continue; }
        else {
            uint32_t tt = tile; bool found = false;
            for (uint32_t s = 0; s < a.nseg; s++) {
                const uint32_t cnt = (a.seg[s].rows + RB - 1) / RB;
                if (tt < cnt) { sidx = s; row0 = tt * RB; found = true; break; }
                tt -= cnt;
            }
            if (!found) continue;
        }
        float res[2] = {0.0f, 0.0f};
        const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
        for (int pass = 0; pass < npass; pass++) {
            const GemvSeg &sg = a.seg[sidx + pass];
            const uint8_t *W = reinterpret_cast<const uint8_t *>(sg.w);
            for (int it = lane; it < items; it += 64) {
                const int r = it / GT, gg = it % GT, blk = gg >> 3, g = gg & 7;
                const uint32_t row = row0 + r;
                if (row < sg.rows) {
                    const uint8_t *wb = W + ((size_t)row * bpl + blk) * 160;
                    const uint4 nib = *reinterpret_cast<const uint4 *>(wb + 32 + g * 16);
                    const uint4 hq = *reinterpret_cast<const uint4 *>(wb + 16);        // s_bias, sb[0..11]
                    const float s_scale = *reinterpret_cast<const float *>(wb + 12);
                    const int len = *reinterpret_cast<const int *>(wb + 4);
                    uint32_t s6, b6;
                    q4k_unpack6(hq.y, hq.z, hq.w, g, s6, b6);
                    const float sp = (float)s6 * s_scale, bp = (float)b6 * __uint_as_float(hq.x);
                    const int glen = (len >= (g + 1) * 32) ? 32 : (len - 32 * g);
                    const uint32_t wl[4] = { nib.x & 0x0f0f0f0fu, nib.y & 0x0f0f0f0fu, nib.z & 0x0f0f0f0fu, nib.w & 0x0f0f0f0fu };
                    const uint32_t wh[4] = { (nib.x >> 4) & 0x0f0f0f0fu, (nib.y >> 4) & 0x0f0f0f0fu,
                                             (nib.z >> 4) & 0x0f0f0f0fu, (nib.w >> 4) & 0x0f0f0f0fu };
                    uint32_t sump = 0;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        sump = __builtin_amdgcn_udot4(wl[m], 0x01010101u, sump, false);
                        sump = __builtin_amdgcn_udot4(wh[m], 0x01010101u, sump, false);
                    }
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        if (b < nb) {
                            const XGroup &xq = xg[(size_t)b * GT + gg];
                            uint32_t spq = 0;
#pragma unroll
                            for (int m = 0; m < 4; m++) {
                                spq = __builtin_amdgcn_udot4(wl[m], xq.lo[m], spq, false);
                                spq = __builtin_amdgcn_udot4(wh[m], xq.hi[m], spq, false);
                            }
                            const float sq = xq.sq, bq = xq.bq;
                            // reference tensor.c:425-428, same association
                            const float grp = sp * sq * (float)(int)spq - sp * bq * (float)(int)sump - sq * bp * (float)xq.sumq + glen * bp * bq;
                            fold[(r * B + b) * pitch + gg] = grp;
                        }
                    }
                }
            }
            // ordered fold (groups inside a block, then blocks along the row)
            float line = 0.0f;
            if (lane < RB * B && pb < nb) {
                const float *f = fold + lane * pitch;
                for (int blk = 0; blk < bpl; blk++) {
                    const int d = (n >= (blk + 1) * 256) ? 256 : (n - blk * 256);
                    const int gv = (d + 31) >> 5;
                    float ds = 0.0f;
                    for (int g = 0; g < gv; g++) ds += f[blk * 8 + g];
                    line += ds;
                }
            }
            res[pass] = line;
        }
        if (lane < RB * B && pb < nb) {
            const uint32_t row = row0 + pr;
            if (row < a.seg[sidx].rows) {
                const GemvSeg &s = a.seg[sidx];
                size_t off = (size_t)pb * s.out_bstride;
                if (s.out_pstride) off += (size_t)a.pos[pb] * s.out_pstride;
                float *o = s.out + off + row;
                if (a.epi == GEMV_EPI_STORE) *o = res[0];
                else if (a.epi == GEMV_EPI_RESID) *o = *o + res[0];
                else { float h = res[0]; h *= (1.0f / (1.0f + expf(-h))); h *= res[1]; *o = h; }
            }
        }
    }
}

static size_t q4k_lds_bytes(uint32_t n, int B, int RB) {
    const size_t bpl = (n + 255) / 256, GT = bpl * 8;
    return (size_t)B * GT * sizeof(XGroup) + (((size_t)n + 3) & ~(size_t)3) * 4 + 32 * 4 + (size_t)4 * RB * B * (GT + 1) * 4;
}

template <int B>
static hipError_t launch_q4k_b(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    constexpr int RB = GEMV_RB;
    uint32_t tiles = 0;
    if (a.epi == GEMV_EPI_SWIGLU) tiles = (a.seg[0].rows + RB - 1) / RB;
    else for (uint32_t s = 0; s < a.nseg; s++) tiles += (a.seg[s].rows + RB - 1) / RB;
    a.tiles = tiles;
    uint32_t wgs = (tiles + 3) / 4;
    if (wgs > max_wg) wgs = max_wg;
    if (!wgs) return hipSuccess;
    const size_t lds = q4k_lds_bytes(a.n, B, RB);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemv_q4k_kernel<B, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemv_q4k_kernel<B, RB>), dim3(wgs), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_gemv_q4k(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    if (a.nb <= 1) return launch_q4k_b<1>(a, max_wg, st);
    if (a.nb <= 2) return launch_q4k_b<2>(a, max_wg, st);
    if (a.nb <= 4) return launch_q4k_b<4>(a, max_wg, st);
    return launch_q4k_b<8>(a, max_wg, st);
}

}  // namespace nano
