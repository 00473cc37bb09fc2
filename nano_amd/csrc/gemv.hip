// gemv.hip -- fused decode GEMVs for gfx950 (MI355X): out[d] = W[d,n] . act[n] for up to 8 sequences.
//
// One kernel family covers every weight-streaming step of the reference's forward:
//   * matmul        (FP32,      reference infer/infer.c:637-651)
//   * matmul_quant  (Q80 W8A8,  reference infer/infer.c:654-679) preceded by quantize (infer/tensor.c:21-46)
//   * matmul_q4k    (Q4K W4A4,  reference infer/tensor.c:438-471) -- see gemv_q4k.hip
// with the surrounding elementwise work fused in:
//   prologue : optional combine of the split attention partials (attn.hip), optional rmsnorm
//              (infer.c:601-614) of the input vector, activation re-quantization -- recomputed from the
//              (L2-resident) fp32 vector instead of costing a launch.  For <= 2 sequences every WAVE stages
//              the activation for itself (wave-private LDS, DPP reductions, no workgroup barrier at all);
//              for 4..8 sequences the four waves split the sequences and meet at one barrier;
//   epilogue : plain store (q / raw k / v-cache row / logits [+ per-tile arg-max partial]), residual add
//              (infer.c:906-908,963-965), or SwiGLU of the (W1,W3) row pair (infer.c:937-944).
//
// Mapping (HBM-bound byte work, 2 flop/byte: no MFMA):
//   * a workgroup is 4 independent waves; a wave owns a tile of TR consecutive rows and streams them in
//     batches of up to 8 rows x 1 KiB: lane l loads bytes [16l,16l+16) of a row chunk with one
//     global_load_dwordx4 (fully coalesced; the row-major weight blocks stay exactly as in the model
//     file).  The next batch (weights + the group leaders' weight scales) is issued before the current one
//     is consumed, and the very first batch is issued BEFORE the prologue so its HBM latency overlaps the
//     activation staging.  TR is 1/2/4 for the small per-layer GEMVs (so that >= ~1024 waves exist) and
//     16/8 for the classifier.
//   * Q80: v_dot4_i32_i8 on the 16 int8 of a lane, DPP integer reduction over the gs/16 lanes of a
//     quantization group, the group's leader lane forms ((float)ival * ws) * xs and parks it in a small
//     per-wave LDS table; one lane per (row, sequence) then adds the groups IN THE REFERENCE'S ORDER
//     (infer.c:668-674) -- given identical int8 inputs the fp32 result is bit-identical to the reference.
//   * FP32: per-lane partial sums over the lane's float4 slices, DPP wave reduction (tolerance 1e-5).
#include "device_common.h"
#include "kernels.h"

namespace nano {

// ------------------------------------------------------------------------------------------------
// DPP cross-lane helpers (no LDS traffic, unlike __shfl)
// ------------------------------------------------------------------------------------------------
#define DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))

template <int W>   // sum over aligned groups of W lanes (W = 1,2,4,8,16); every lane gets the group sum
__device__ __forceinline__ int dpp_group_sum(int v) {
    if (W >= 2) v += DPP_I(v, 0xB1);     // quad_perm [1,0,3,2]
    if (W >= 4) v += DPP_I(v, 0x4E);     // quad_perm [2,3,0,1]
    if (W >= 8) v += DPP_I(v, 0x141);    // row_half_mirror
    if (W >= 16) v += DPP_I(v, 0x140);   // row_mirror
    return v;
}
template <int W>
__device__ __forceinline__ float dpp_group_max(float v) {
    if (W >= 2) v = fmaxf(v, DPP_F(v, 0xB1));
    if (W >= 4) v = fmaxf(v, DPP_F(v, 0x4E));
    if (W >= 8) v = fmaxf(v, DPP_F(v, 0x141));
    if (W >= 16) v = fmaxf(v, DPP_F(v, 0x140));
    if (W >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (W >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v += DPP_F(v, 0xB1); v += DPP_F(v, 0x4E); v += DPP_F(v, 0x141); v += DPP_F(v, 0x140);   // row (16-lane) totals
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ------------------------------------------------------------------------------------------------
// per-wave activation staging
// ------------------------------------------------------------------------------------------------

// combine weights of the attention splits (attn.hip) for sequence b: wgt[h*ns + s] = e^{m_s-M} / L, by one wave
__device__ __forceinline__ void attn_weights_wave(const GemvArgs &a, int b, float *wgt, int lane) {
    const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
    for (uint32_t h = lane; h < nh; h += 64) {
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
}

// four consecutive input elements i..i+3 of sequence b (plain vector or attention combine)
__device__ __forceinline__ float4 input4(const GemvArgs &a, int b, int i, const float *wgt) {
    if (!a.attn_part) return *reinterpret_cast<const float4 *>(a.xin + (size_t)b * a.xin_bstride + i);
    const uint32_t ns = a.attn_nsplit;
    const float *part = a.attn_part + (size_t)b * ns * a.n;
    const int h = i / (int)a.attn_hd;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t s = 0; s < ns; s++) {
        const float w = wgt[h * ns + s];
        const float4 o = *reinterpret_cast<const float4 *>(part + (size_t)s * a.n + i);
        acc.x += o.x * w; acc.y += o.y * w; acc.z += o.z * w; acc.w += o.w * w;
    }
    return acc;
}

// rmsnorm scale of sequence b computed by ONE wave (reference infer.c:603-609; tree order, tol 1e-5)
__device__ __forceinline__ float rms_scale_wave(const GemvArgs &a, int b, int lane) {
    const int n = (int)a.n;
    const float *x = a.xin + (size_t)b * a.xin_bstride;
    float acc = 0.0f;
    for (int i = lane * 4; i < n; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
    }
    float ss = dpp_wave_sum(acc);
    ss /= (float)n;
    ss += 1e-5f;
    return 1.0f / sqrtf(ss);
}

// One wave stages sequence b as fp32 (optionally normalised) into xf[n]
__device__ __forceinline__ void stage_f32_wave(const GemvArgs &a, int b, float *xf, const float *wgt, int lane) {
    const int n = (int)a.n;
    const float ss = a.norm_w ? rms_scale_wave(a, b, lane) : 1.0f;
    for (int i = lane * 4; i < n; i += 256) {
        float4 t = input4(a, b, i, wgt);
        if (a.norm_w) {
            const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
            t.x = w.x * (ss * t.x); t.y = w.y * (ss * t.y); t.z = w.z * (ss * t.z); t.w = w.w * (ss * t.w);
        }
        *reinterpret_cast<float4 *>(xf + i) = t;
    }
}

// ------------------------------------------------------------------------------------------------
// tile cursor: flat iteration over (tile, pass, chunk, row batch) so that loads can run one batch ahead
// ------------------------------------------------------------------------------------------------
struct Cursor {
    uint32_t tile, sidx, row0;
    int pass, c, r0;
    bool valid;
};

template <int TR>
__device__ __forceinline__ bool locate_tile(const GemvArgs &a, uint32_t tile, uint32_t &seg, uint32_t &row0) {
    if (tile >= a.tiles) return false;
    if (a.epi == GEMV_EPI_SWIGLU) { seg = 0; row0 = tile * TR; return row0 < a.seg[0].rows; }
    for (uint32_t s = 0; s < a.nseg; s++) {
        const uint32_t t = (a.seg[s].rows + TR - 1) / TR;
        if (tile < t) { seg = s; row0 = tile * TR; return true; }
        tile -= t;
    }
    return false;
}

template <int TR, int RBL>
__device__ __forceinline__ void cursor_advance(const GemvArgs &a, Cursor &cu, int nchunk, int npass, uint32_t stride) {
    cu.r0 += RBL;
    if (cu.r0 < TR) return;
    cu.r0 = 0;
    if (++cu.c < nchunk) return;
    cu.c = 0;
    if (++cu.pass < npass) return;
    cu.pass = 0;
    cu.tile += stride;
    cu.valid = locate_tile<TR>(a, cu.tile, cu.sidx, cu.row0);
}

__device__ __forceinline__ float *out_ptr(const GemvArgs &a, const GemvSeg &s, int b) {
    size_t off = (size_t)b * s.out_bstride;
    if (s.out_pstride) off += (size_t)a.pos[b] * s.out_pstride;
    return s.out + off;
}

__device__ __forceinline__ float finish(const GemvArgs &a, float v, float v2, float old) {
    if (a.epi == GEMV_EPI_STORE) return v;
    if (a.epi == GEMV_EPI_RESID) return old + v;               // x[i] += xb2[i]
    float h = v;                                                // SwiGLU: silu(w1 x) * (w3 x)
    h *= (1.0f / (1.0f + expf(-h)));
    h *= v2;
    return h;
}

// ------------------------------------------------------------------------------------------------
// FP32
// ------------------------------------------------------------------------------------------------
template <int B, int TR>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RBL = TR;
    constexpr bool PRIV = (B <= 2);
    const int n = (int)a.n;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
    const size_t wgt_f = a.attn_part ? (((size_t)a.attn_n_head * a.attn_nsplit + 3) & ~(size_t)3) : 0;
    float *xf, *wgt;
    if (PRIV) { xf = reinterpret_cast<float *>(smem) + (size_t)wid * (B * n4 + wgt_f); wgt = xf + B * n4; }
    else { xf = reinterpret_cast<float *>(smem); wgt = xf + B * n4 + (size_t)wid * wgt_f; }

    const int nchunk = (n + 255) >> 8;
    const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
    const uint32_t stride = gridDim.x * 4;
    const int nb = (int)a.nb;

    Cursor cu{ blockIdx.x * 4 + wid, 0, 0, 0, 0, 0, false };
    cu.valid = locate_tile<TR>(a, cu.tile, cu.sidx, cu.row0);

    float4 wa[RBL], wb[RBL];
    auto issue = [&](const Cursor &c, float4 (&w)[RBL]) {
        const GemvSeg &sg = a.seg[c.sidx + c.pass];
        const float *W = reinterpret_cast<const float *>(sg.w);
        const int col = (c.c << 8) + lane * 4;
#pragma unroll
        for (int r = 0; r < RBL; r++) {
            const uint32_t row = c.row0 + r;
            w[r] = (col < n && row < sg.rows) ? ld_stream_f4(W + (size_t)row * n + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (cu.valid) issue(cu, wa);

    if (PRIV) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < nb) {
                if (a.attn_part) attn_weights_wave(a, b, wgt, lane);
                stage_f32_wave(a, b, xf + (size_t)b * n4, wgt, lane);
            }
        }
    } else {
        for (int b = wid; b < nb; b += 4) {
            if (a.attn_part) attn_weights_wave(a, b, wgt, lane);
            stage_f32_wave(a, b, xf + (size_t)b * n4, wgt, lane);
        }
        __syncthreads();
    }

    float acc[TR][B], res0[TR][B];
#pragma unroll
    for (int r = 0; r < TR; r++)
#pragma unroll
        for (int b = 0; b < B; b++) { acc[r][b] = 0.0f; res0[r][b] = 0.0f; }

    while (cu.valid) {
        Cursor nx = cu;
        cursor_advance<TR, RBL>(a, nx, nchunk, npass, stride);
        if (nx.valid) issue(nx, wb);

        const int col = (cu.c << 8) + lane * 4;
        const bool act = col < n;
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < nb) {
                const float4 xv = act ? *reinterpret_cast<const float4 *>(xf + (size_t)b * n4 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < TR; r++) {
                    float t = acc[r][b];
                    t += wa[r].x * xv.x; t += wa[r].y * xv.y; t += wa[r].z * xv.z; t += wa[r].w * xv.w;
                    acc[r][b] = t;
                }
            }
        }
        if (cu.c + 1 == nchunk) {
#pragma unroll
            for (int r = 0; r < TR; r++)
#pragma unroll
                for (int b = 0; b < B; b++) acc[r][b] = dpp_wave_sum(acc[r][b]);
            if (cu.pass + 1 == npass) {
#pragma unroll
                for (int r = 0; r < TR; r++)
#pragma unroll
                    for (int b = 0; b < B; b++)
                        if (lane == r * B + b && b < nb && cu.row0 + r < a.seg[cu.sidx].rows) {
                            float *o = out_ptr(a, a.seg[cu.sidx], b) + cu.row0 + r;
                            const float old = (a.epi == GEMV_EPI_RESID) ? *o : 0.0f;
                            *o = finish(a, (npass == 2) ? res0[r][b] : acc[r][b], acc[r][b], old);
                        }
                if (a.tile_max && lane < B && lane < nb) {
                    float best = -INFINITY; uint32_t bi = 0xffffffffu;
#pragma unroll
                    for (int r = 0; r < TR; r++)
#pragma unroll
                        for (int b = 0; b < B; b++)
                            if (b == lane && cu.row0 + r < a.seg[cu.sidx].rows && (bi == 0xffffffffu || acc[r][b] > best)) { best = acc[r][b]; bi = cu.row0 + r; }
                    float *tm = a.tile_max + ((size_t)lane * a.tiles + cu.tile) * 2;
                    tm[0] = best; tm[1] = __uint_as_float(bi);
                }
            } else {
#pragma unroll
                for (int r = 0; r < TR; r++)
#pragma unroll
                    for (int b = 0; b < B; b++) res0[r][b] = acc[r][b];
            }
#pragma unroll
            for (int r = 0; r < TR; r++)
#pragma unroll
                for (int b = 0; b < B; b++) acc[r][b] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < RBL; r++) wa[r] = wb[r];
        cu = nx;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t count_tiles(const GemvArgs &a, int TR) {
    if (a.epi == GEMV_EPI_SWIGLU) return (a.seg[0].rows + TR - 1) / TR;
    uint32_t t = 0;
    for (uint32_t s = 0; s < a.nseg; s++) t += (a.seg[s].rows + TR - 1) / TR;
    return t;
}
static inline uint32_t total_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}
static inline size_t attn_wgt_floats(const GemvArgs &a) { return a.attn_part ? ((size_t)a.attn_n_head * a.attn_nsplit + 3) & ~(size_t)3 : 0; }

template <int B, int TR>
static hipError_t launch_f32_tr(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    a.tiles = count_tiles(a, TR);
    uint32_t wgs = (a.tiles + 3) / 4;
    if (wgs > max_wg) wgs = max_wg;
    if (!wgs) return hipSuccess;
    const size_t n4 = ((size_t)a.n + 3) & ~(size_t)3;
    const size_t lds = ((B <= 2) ? 4 * (B * n4 + attn_wgt_floats(a)) : B * n4 + 4 * attn_wgt_floats(a)) * 4;
    auto kern = &gemv_f32_kernel<B, TR>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int B>
static hipError_t launch_f32(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    const uint32_t rows = total_rows(a);
    if (B >= 8 || rows < 2048) return launch_f32_tr<B, (B >= 8) ? 2 : 1>(a, max_wg, st);
    if (rows < 4096) return launch_f32_tr<B, 2>(a, max_wg, st);
    return launch_f32_tr<B, (B >= 8) ? 2 : 4>(a, max_wg, st);
}

hipError_t launch_gemv(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    if (quant == 0x80u) return launch_gemv_q80(a, st);
    if (a.nb <= 1) return launch_f32<1>(a, max_wg, st);
    if (a.nb <= 2) return launch_f32<2>(a, max_wg, st);
    if (a.nb <= 4) return launch_f32<4>(a, max_wg, st);
    return launch_f32<8>(a, max_wg, st);
}

// number of (max, row) arg-max partials launch_gemv() will write per sequence for these arguments (sizes tile_max;
// 0 = none written, scan the logits)
uint32_t gemv_tiles(uint32_t quant, const GemvArgs &a) {
    if (quant == 0x80u) return gemv_q80_partials(a);
    if (quant == 0x00u) {
        const int B = a.nb <= 1 ? 1 : a.nb <= 2 ? 2 : a.nb <= 4 ? 4 : 8;
        const uint32_t rows = total_rows(a);
        const int tr = (B >= 8) ? 2 : (rows < 2048 ? 1 : rows < 4096 ? 2 : 4);
        return count_tiles(a, tr);
    }
    return count_tiles(a, GEMV_RB);
}

}  // namespace nano
