// gemv.hip -- fused decode GEMVs for gfx950 (MI355X): out[d] = W[d,n] . act[n] for up to 8 sequences.
//
// One kernel family covers every weight-streaming step of the reference's forward:
//   * matmul        (FP32,      reference infer/infer.c:637-651)
//   * matmul_quant  (Q80 W8A8,  reference infer/infer.c:654-679) preceded by quantize (infer/tensor.c:21-46)
//   * matmul_q4k    (Q4K W4A4,  reference infer/tensor.c:438-471) preceded by quantize_tensor_q4k_in_situ
//                   (infer/tensor.c:281-310) -- see gemv_q4k.hip
// with the surrounding elementwise work fused in:
//   prologue : optional rmsnorm (infer.c:601-614) of the input vector + activation re-quantization,
//              recomputed by every workgroup from the (L2-resident) fp32 vector -> no extra launch;
//   epilogue : plain store (q / raw k / v-cache row / logits), residual add (infer.c:906-908,963-965),
//              or SwiGLU of the (W1,W3) row pair (infer.c:937-944).
//
// Mapping (HBM-bound byte work, no MFMA: one token has 2 flop/byte):
//   * a workgroup is 4 waves; a wave owns a tile of RB consecutive rows and walks each row in
//     1 KiB chunks, lane l loading bytes [16l,16l+16) of the chunk with one global_load_dwordx4
//     (fully coalesced, row-major weight blocks exactly as they sit in the model file);
//   * Q80: v_dot4_i32_i8 on the 16 int8 of a lane, integer reduction over the gs/16 lanes of a
//     quantization group, per-group float combine ((float)ival * ws) * xs, and the per-row sum over
//     groups folded IN THE REFERENCE'S GROUP ORDER by one lane per (row, sequence) through a small LDS
//     table -- so given identical int8 inputs the fp32 result is bit-identical to the reference;
//   * FP32: per-lane partial sums over the lane's float4 slices, wave tree reduction.
#include "device_common.h"
#include "kernels.h"

namespace nano {

// ------------------------------------------------------------------------------------------------
// prologues (executed redundantly by each workgroup; B = compile-time capacity, nb = live sequences)
// ------------------------------------------------------------------------------------------------

// optional rmsnorm, result (fp32) into LDS xf[b*n + i]
template <int B>
__device__ __forceinline__ void prologue_f32(const GemvArgs &a, float *xf, float *red) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = (int)a.n;
    for (int b = 0; b < B; b++) {
        if (b >= (int)a.nb) break;
        const float *x = a.xin + (size_t)b * a.xin_bstride;
        float ss = 1.0f;
        if (a.norm_w) {
            float acc = 0.0f;
            for (int i = tid * 4; i < n; i += nthr * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + i);
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            }
            ss = block_sum(acc, red);
            ss /= (float)n;
            ss += 1e-5f;
            ss = 1.0f / sqrtf(ss);
        }
        for (int i = tid * 4; i < n; i += nthr * 4) {
            float4 v = *reinterpret_cast<const float4 *>(x + i);
            if (a.norm_w) {
                const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
            }
            *reinterpret_cast<float4 *>(xf + (size_t)b * n + i) = v;
        }
    }
    __syncthreads();
}

// optional rmsnorm + Q80 quantization: int8 into xq[b*n + i], group scales into xs[b*(n/gs) + g]
template <int B>
__device__ __forceinline__ void prologue_q80(const GemvArgs &a, int8_t *xq, float *xs, float *red) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = (int)a.n, gs = (int)a.gs, ng = n / gs;
    const int tpg = gs / 4;                        // threads per quantization group (4 elements each)
    const int iters = (n + nthr * 4 - 1) / (nthr * 4);
    if (a.xq_in) {      // operator-test path: caller supplied the quantized activation (one sequence)
        for (int i = tid; i < n; i += nthr) xq[i] = a.xq_in[i];
        for (int i = tid; i < ng; i += nthr) xs[i] = a.xs_in[i];
        __syncthreads();
        return;
    }
    for (int b = 0; b < B; b++) {
        if (b >= (int)a.nb) break;
        const float *x = a.xin + (size_t)b * a.xin_bstride;
        float ss = 1.0f;
        if (a.norm_w) {
            float acc = 0.0f;
            for (int i = tid * 4; i < n; i += nthr * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + i);
                acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
            }
            ss = block_sum(acc, red);
            ss /= (float)n;
            ss += 1e-5f;
            ss = 1.0f / sqrtf(ss);
        }
        for (int it = 0; it < iters; it++) {
            const int i = (it * nthr + tid) * 4;
            const bool act = i < n;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) {
                v = *reinterpret_cast<const float4 *>(x + i);
                if (a.norm_w) {
                    const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                    v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
                }
            }
            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            m = group_max(m, tpg);
            const float scale = m / 127.0f;
            if (act) {
                const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale);
                const int q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
                const uint32_t packed = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) |
                                        ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                reinterpret_cast<uint32_t *>(xq + (size_t)b * n)[i >> 2] = packed;
                if ((tid % tpg) == 0) xs[(size_t)b * ng + i / gs] = scale;
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// tile bookkeeping
// ------------------------------------------------------------------------------------------------
template <int RB>
__device__ __forceinline__ bool locate_tile(const GemvArgs &a, uint32_t tile, uint32_t &seg, uint32_t &row0) {
    if (a.epi == GEMV_EPI_SWIGLU) { seg = 0; row0 = tile * RB; return row0 < a.seg[0].rows; }
    for (uint32_t s = 0; s < a.nseg; s++) {
        const uint32_t t = (a.seg[s].rows + RB - 1) / RB;
        if (tile < t) { seg = s; row0 = tile * RB; return true; }
        tile -= t;
    }
    return false;
}

__device__ __forceinline__ float *out_ptr(const GemvArgs &a, const GemvSeg &s, int b) {
    size_t off = (size_t)b * s.out_bstride;
    if (s.out_pstride) off += (size_t)a.pos[b] * s.out_pstride;
    return s.out + off;
}

__device__ __forceinline__ void emit(const GemvArgs &a, const GemvSeg &s, int b, uint32_t row, float v, float v2) {
    float *o = out_ptr(a, s, b) + row;
    if (a.epi == GEMV_EPI_STORE) *o = v;
    else if (a.epi == GEMV_EPI_RESID) *o = *o + v;             // x[i] += xb2[i]
    else {                                                      // SwiGLU: silu(w1 x) * (w3 x)
        float h = v;
        h *= (1.0f / (1.0f + expf(-h)));
        h *= v2;
        *o = h;
    }
}

// ------------------------------------------------------------------------------------------------
// Q80
// ------------------------------------------------------------------------------------------------
template <int B, int RB>
__global__ __launch_bounds__(256) void gemv_q80_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = (int)a.n, gs = (int)a.gs, ng = n / gs;
    const int lpg = gs / 16;                 // lanes per quantization group
    const int GC = 1024 / gs;                // groups per 1 KiB chunk
    const int pitch = GC + 1;
    // LDS carve: xq [B*n] | xs [B*ng] | red[32] | fold[4][RB*B*pitch]
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    const size_t xq_bytes = ((size_t)B * n + 15) & ~(size_t)15;
    float *xs = reinterpret_cast<float *>(smem + xq_bytes);
    float *red = xs + (((size_t)B * ng + 3) & ~(size_t)3);
    float *foldbase = red + 32;

    prologue_q80<B>(a, xq, xs, red);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float *fold = foldbase + (size_t)wid * (RB * B) * pitch;
    const int nchunk = (n + 1023) >> 10;
    const int gl = lane / lpg;               // group index within the chunk this lane belongs to
    const bool leader = (lane % lpg) == 0;
    const int nb = (int)a.nb;
    const int pr = lane / B, pb = lane % B;  // (row, sequence) pair folded by this lane

    for (uint32_t tile = blockIdx.x * 4 + wid; tile < a.tiles; tile += gridDim.x * 4) {
        uint32_t sidx, row0;
        if (!locate_tile<RB>(a, tile, sidx, row0)) continue;
        float res[2] = {0.0f, 0.0f};
        const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
        for (int pass = 0; pass < npass; pass++) {
            const GemvSeg &sg = a.seg[sidx + pass];
            const int8_t *W = reinterpret_cast<const int8_t *>(sg.w);
            const float *WS = sg.ws;
            float val = 0.0f;
            for (int c = 0; c < nchunk; c++) {
                const int col = (c << 10) + lane * 16;
                const bool act = col < n;
                int4 wv[RB];
                float wsc[RB];
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    const uint32_t row = row0 + r;
                    const bool ok = act && row < sg.rows;
                    wv[r] = ok ? *reinterpret_cast<const int4 *>(W + (size_t)row * n + col) : make_int4(0, 0, 0, 0);
                    wsc[r] = (ok && leader) ? WS[(size_t)row * ng + c * GC + gl] : 0.0f;
                }
#pragma unroll
                for (int b = 0; b < B; b++) {
                    if (b < nb) {
                        const int4 xv = act ? *reinterpret_cast<const int4 *>(xq + (size_t)b * n + col) : make_int4(0, 0, 0, 0);
                        const float xsc = act ? xs[(size_t)b * ng + c * GC + gl] : 0.0f;
#pragma unroll
                        for (int r = 0; r < RB; r++) {
                            int iv = __builtin_amdgcn_sdot4(wv[r].x, xv.x, 0, false);
                            iv = __builtin_amdgcn_sdot4(wv[r].y, xv.y, iv, false);
                            iv = __builtin_amdgcn_sdot4(wv[r].z, xv.z, iv, false);
                            iv = __builtin_amdgcn_sdot4(wv[r].w, xv.w, iv, false);
                            iv = group_sum_i(iv, lpg);
                            const float p = ((float)iv * wsc[r]) * xsc;
                            if (leader) fold[(r * B + b) * pitch + gl] = p;
                        }
                    }
                }
                // ordered fold: val += p[g] for g ascending (reference infer.c:668-674)
                const int gvalid = min(GC, ng - c * GC);
                if (lane < RB * B && pb < nb) {
                    const float *f = fold + lane * pitch;
                    for (int g = 0; g < gvalid; g++) val += f[g];
                }
            }
            res[pass] = val;
        }
        if (lane < RB * B && pb < nb) {
            const uint32_t row = row0 + pr;
            if (row < a.seg[sidx].rows) emit(a, a.seg[sidx], pb, row, res[0], res[1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// FP32
// ------------------------------------------------------------------------------------------------
template <int B, int RB>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = (int)a.n;
    float *xf = reinterpret_cast<float *>(smem);
    float *red = xf + (((size_t)B * n + 3) & ~(size_t)3);

    prologue_f32<B>(a, xf, red);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = (n + 255) >> 8;
    const int nb = (int)a.nb;

    for (uint32_t tile = blockIdx.x * 4 + wid; tile < a.tiles; tile += gridDim.x * 4) {
        uint32_t sidx, row0;
        if (!locate_tile<RB>(a, tile, sidx, row0)) continue;
        float res[2][RB][B];
        const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int b = 0; b < B; b++) res[pass][r][b] = 0.0f;
            if (pass < npass) {
                const GemvSeg &sg = a.seg[sidx + pass];
                const float *W = reinterpret_cast<const float *>(sg.w);
                for (int c = 0; c < nchunk; c++) {
                    const int col = (c << 8) + lane * 4;
                    const bool act = col < n;
                    float4 wv[RB];
#pragma unroll
                    for (int r = 0; r < RB; r++) {
                        const uint32_t row = row0 + r;
                        wv[r] = (act && row < sg.rows) ? *reinterpret_cast<const float4 *>(W + (size_t)row * n + col)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        if (b < nb) {
                            const float4 xv = act ? *reinterpret_cast<const float4 *>(xf + (size_t)b * n + col)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                            for (int r = 0; r < RB; r++) {
                                float acc = res[pass][r][b];
                                acc += wv[r].x * xv.x; acc += wv[r].y * xv.y; acc += wv[r].z * xv.z; acc += wv[r].w * xv.w;
                                res[pass][r][b] = acc;
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < RB; r++)
#pragma unroll
                    for (int b = 0; b < B; b++) res[pass][r][b] = wave_sum(res[pass][r][b]);
            }
        }
        // lane (r*B+b) emits its pair
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (lane == r * B + b && b < nb && row0 + r < a.seg[sidx].rows)
                    emit(a, a.seg[sidx], b, row0 + r, res[0][r][b], res[1][r][b]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t count_tiles(const GemvArgs &a, int RB) {
    if (a.epi == GEMV_EPI_SWIGLU) return (a.seg[0].rows + RB - 1) / RB;
    uint32_t t = 0;
    for (uint32_t s = 0; s < a.nseg; s++) t += (a.seg[s].rows + RB - 1) / RB;
    return t;
}

size_t gemv_lds_bytes(uint32_t quant, uint32_t n, uint32_t gs, int B) {
    if (quant == 0x80u) {
        const int RB = GEMV_RB;
        size_t xq = ((size_t)B * n + 15) & ~(size_t)15;
        size_t xs = (((size_t)B * (n / gs) + 3) & ~(size_t)3) * 4;
        size_t fold = (size_t)4 * RB * B * (1024 / gs + 1) * 4;
        return xq + xs + 32 * 4 + fold;
    }
    return ((((size_t)B * n + 3) & ~(size_t)3) + 32) * 4;
}

template <int B>
static hipError_t launch_b(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    constexpr int RB = GEMV_RB;
    a.tiles = count_tiles(a, RB);
    uint32_t wgs = (a.tiles + 3) / 4;
    if (wgs > max_wg) wgs = max_wg;
    if (wgs == 0) return hipSuccess;
    const size_t lds = gemv_lds_bytes(quant, a.n, a.gs, B);
    if (quant == 0x80u) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemv_q80_kernel<B, RB>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemv_q80_kernel<B, RB>), dim3(wgs), dim3(256), lds, st, a);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemv_f32_kernel<B, RB>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemv_f32_kernel<B, RB>), dim3(wgs), dim3(256), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemv(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    if (a.nb <= 1) return launch_b<1>(quant, a, max_wg, st);
    if (a.nb <= 2) return launch_b<2>(quant, a, max_wg, st);
    if (a.nb <= 4) return launch_b<4>(quant, a, max_wg, st);
    return launch_b<8>(quant, a, max_wg, st);
}

}  // namespace nano
