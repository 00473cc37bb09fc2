// gemv.hip -- fused decode GEMVs for gfx950 (MI355X): out[d] = W[d,n] . act[n] for up to 8 sequences.
//
// One kernel family covers every weight-streaming step of the reference's forward:
//   * matmul        (FP32,      reference infer/infer.c:637-651)
//   * matmul_quant  (Q80 W8A8,  reference infer/infer.c:654-679) preceded by quantize (infer/tensor.c:21-46)
//   * matmul_q4k    (Q4K W4A4,  reference infer/tensor.c:438-471) -- see gemv_q4k.hip
// with the surrounding elementwise work fused in:
//   prologue : optional combine of the split attention partials (attn.hip), optional rmsnorm
//              (infer.c:601-614) of the input vector, activation re-quantization -- recomputed by every
//              workgroup from the (L2-resident) fp32 vector, so no extra launch and no extra sync;
//   epilogue : plain store (q / raw k / v-cache row / logits), residual add (infer.c:906-908,963-965),
//              or SwiGLU of the (W1,W3) row pair (infer.c:937-944).
//
// Mapping (HBM-bound byte work, 2 flop/byte: no MFMA):
//   * a workgroup is 4 waves; a wave owns a tile of TR consecutive rows and streams them in batches
//     of up to 8 rows x 1 KiB: lane l loads bytes [16l,16l+16) of a row chunk with one
//     global_load_dwordx4 (fully coalesced; the row-major weight blocks stay exactly as they sit in the
//     model file).  The next batch is issued before the current one is consumed, and the very first
//     batch is issued BEFORE the prologue so its HBM latency overlaps the activation staging.
//   * Q80: v_dot4_i32_i8 on the 16 int8 of a lane, DPP integer reduction over the gs/16 lanes of a
//     quantization group, then through a small per-wave LDS table: (A) all 64 lanes apply
//     ((float)ival * ws) * xs to the (row, group) entries in parallel (the weight scales of a tile are one
//     coalesced load), (B) one lane per (row, sequence) adds the groups IN THE REFERENCE'S ORDER
//     (infer.c:668-674) -- given identical int8 inputs the fp32 result is bit-identical to the reference.
//     Big GEMVs (the classifier) use 64-row tiles so that all 64 lanes fold; small ones use 4-row tiles
//     so that >=1024 waves exist.
//   * FP32: per-lane partial sums over the lane's float4 slices, wave tree reduction (tolerance 1e-5).
#include "device_common.h"
#include "kernels.h"

namespace nano {

// ------------------------------------------------------------------------------------------------
// cross-lane integer reduction over aligned groups of W lanes (W = 2,4,8,16) using DPP only
// ------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ int dpp_group_sum(int v) {
    if (W >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    if (W >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    if (W >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    if (W >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
    return v;
}

// ------------------------------------------------------------------------------------------------
// shared prologue pieces
// ------------------------------------------------------------------------------------------------

// Input element i of sequence b: either the plain fp32 vector or the combination of the split
// attention partials (flash-decoding style, see attn.hip): xba[i] = sum_s o_s[i] * w[h][s].
struct InputView {
    const float *x;          // plain vector (nullptr when combining)
    const float *part;       // [nsplit][q_dim] partial outputs of this sequence
    const float *wgt;        // LDS: [n_head][nsplit] combine weights
    uint32_t nsplit, hd, q_dim;
    __device__ __forceinline__ float4 load4(int i) const {
        if (x) return *reinterpret_cast<const float4 *>(x + i);
        const int h = i / (int)hd;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t s = 0; s < nsplit; s++) {
            const float w = wgt[h * nsplit + s];
            const float4 o = *reinterpret_cast<const float4 *>(part + (size_t)s * q_dim + i);
            acc.x += o.x * w; acc.y += o.y * w; acc.z += o.z * w; acc.w += o.w * w;
        }
        return acc;
    }
};

// combine weights of the attention splits for sequence b into LDS wgt[n_head][nsplit]
__device__ __forceinline__ void attn_combine_weights(const GemvArgs &a, int b, float *wgt) {
    const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
    for (uint32_t h = threadIdx.x; h < nh; h += blockDim.x) {
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
}

template <int B>
__device__ __forceinline__ void block_sum_multi(float (&v)[B], float *red /* >= 16*B floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int b = 0; b < B; b++) v[b] = wave_sum(v[b]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < B; b++) red[wid * B + b] = v[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; b++) {
        float t = 0.0f;
        for (int w = 0; w < nw; w++) t += red[w * B + b];
        v[b] = t;
    }
}

// rmsnorm scale factors for all live sequences at once (one pair of barriers)
template <int B>
__device__ __forceinline__ void rms_scales(const GemvArgs &a, float (&ss)[B], float *red) {
    const int tid = threadIdx.x, nthr = blockDim.x, n = (int)a.n;
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; b++) {
        acc[b] = 0.0f;
        if (b < (int)a.nb) {
            const float *x = a.xin + (size_t)b * a.xin_bstride;
            for (int i = tid * 4; i < n; i += nthr * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + i);
                acc[b] += v.x * v.x; acc[b] += v.y * v.y; acc[b] += v.z * v.z; acc[b] += v.w * v.w;
            }
        }
    }
    block_sum_multi<B>(acc, red);
#pragma unroll
    for (int b = 0; b < B; b++) {
        float s = acc[b];
        s /= (float)n;
        s += 1e-5f;
        ss[b] = 1.0f / sqrtf(s);
    }
}

// fp32 activations (optionally normalised) into LDS xf[b*n + i]
template <int B>
__device__ __forceinline__ void prologue_f32(const GemvArgs &a, float *xf, float *red, float *wgt) {
    const int tid = threadIdx.x, nthr = blockDim.x, n = (int)a.n;
    float ss[B];
    if (a.norm_w) rms_scales<B>(a, ss, red);
#pragma unroll
    for (int b = 0; b < B; b++) {
        if (b < (int)a.nb) {
            InputView in{ a.attn_part ? nullptr : a.xin + (size_t)b * a.xin_bstride,
                          a.attn_part ? a.attn_part + (size_t)b * a.attn_nsplit * a.n : nullptr, wgt, a.attn_nsplit, a.attn_hd, a.n };
            if (a.attn_part) attn_combine_weights(a, b, wgt);
            for (int i = tid * 4; i < n; i += nthr * 4) {
                float4 v = in.load4(i);
                if (a.norm_w) {
                    const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                    v.x = w.x * (ss[b] * v.x); v.y = w.y * (ss[b] * v.y); v.z = w.z * (ss[b] * v.z); v.w = w.w * (ss[b] * v.w);
                }
                *reinterpret_cast<float4 *>(xf + (size_t)b * n + i) = v;
            }
            if (a.attn_part) __syncthreads();
        }
    }
    __syncthreads();
}

// Q80: optional rmsnorm + quantization: int8 into xq[b*n + i], group scales into xs[b*(n/gs) + g]
template <int B, int GS>
__device__ __forceinline__ void prologue_q80(const GemvArgs &a, int8_t *xq, float *xs, float *red, float *wgt) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = (int)a.n, ng = n / GS;
    constexpr int tpg = GS / 4;                    // threads per quantization group (4 elements each)
    const int iters = (n + nthr * 4 - 1) / (nthr * 4);
    if (a.xq_in) {      // operator-test path: caller supplied the quantized activation (one sequence)
        for (int i = tid; i < n; i += nthr) xq[i] = a.xq_in[i];
        for (int i = tid; i < ng; i += nthr) xs[i] = a.xs_in[i];
        __syncthreads();
        return;
    }
    float ss[B];
    if (a.norm_w) rms_scales<B>(a, ss, red);
#pragma unroll
    for (int b = 0; b < B; b++) {
        if (b < (int)a.nb) {
            InputView in{ a.attn_part ? nullptr : a.xin + (size_t)b * a.xin_bstride,
                          a.attn_part ? a.attn_part + (size_t)b * a.attn_nsplit * a.n : nullptr, wgt, a.attn_nsplit, a.attn_hd, a.n };
            if (a.attn_part) attn_combine_weights(a, b, wgt);
            for (int it = 0; it < iters; it++) {
                const int i = (it * nthr + tid) * 4;
                const bool act = i < n;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (act) {
                    v = in.load4(i);
                    if (a.norm_w) {
                        const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                        v.x = w.x * (ss[b] * v.x); v.y = w.y * (ss[b] * v.y); v.z = w.z * (ss[b] * v.z); v.w = w.w * (ss[b] * v.w);
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
                    if ((tid % tpg) == 0) xs[(size_t)b * ng + i / GS] = scale;
                }
            }
            if (a.attn_part) __syncthreads();
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// tile cursor: flat iteration over (tile, pass, chunk, row batch) so that loads can run one batch ahead
// ------------------------------------------------------------------------------------------------
template <int TR, int RBL>
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
__device__ __forceinline__ void cursor_advance(const GemvArgs &a, Cursor<TR, RBL> &cu, int nchunk, int npass, uint32_t stride) {
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
template <int B, int TR, int GS>
__global__ __launch_bounds__(256) void gemv_q80_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RBL = (TR < 8) ? TR : 8;           // rows per load batch
    constexpr int LPG = GS / 16;                     // lanes per quantization group
    constexpr int GC = 1024 / GS;                    // groups per 1 KiB chunk
    constexpr int PITCH = GC + 1;
    constexpr int PAIRS = TR * B;                    // (row, sequence) pairs per tile, <= 64
    constexpr int NWS = (TR * GC + 63) / 64;         // weight-scale dwords per lane per (tile, chunk)
    static_assert(PAIRS <= 64, "tile too large");
    const int n = (int)a.n, ng = n / GS;
    // LDS carve: xq [B*n] | xs [B*ng] | red[16*B] | wgt[attn] | tab[4][PAIRS*PITCH]
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    const size_t xq_bytes = ((size_t)B * n + 15) & ~(size_t)15;
    float *xs = reinterpret_cast<float *>(smem + xq_bytes);
    float *red = xs + (((size_t)B * ng + 3) & ~(size_t)3);
    float *wgt = red + 16 * B;
    float *tabbase = wgt + ((a.attn_part ? a.attn_n_head * a.attn_nsplit + 3 : 0) & ~3u);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float *tab = tabbase + (size_t)wid * PAIRS * PITCH;     // ints travel as float bit patterns (one LDS type)
    const int nchunk = (n + 1023) >> 10;
    const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
    const uint32_t stride = gridDim.x * 4;
    const int gl = lane / LPG;
    const bool leader = (lane % LPG) == 0;
    const int nb = (int)a.nb;

    using Cur = Cursor<TR, RBL>;
    Cur cu{ blockIdx.x * 4 + wid, 0, 0, 0, 0, 0, false };
    cu.valid = locate_tile<TR>(a, cu.tile, cu.sidx, cu.row0);

    int4 wa[RBL], wb[RBL];
    float wsr[NWS], wsn[NWS];

    auto issue = [&](const Cur &c, int4 (&w)[RBL]) {
        const GemvSeg &sg = a.seg[c.sidx + c.pass];
        const int8_t *W = reinterpret_cast<const int8_t *>(sg.w);
        const int col = (c.c << 10) + lane * 16;
#pragma unroll
        for (int r = 0; r < RBL; r++) {
            const uint32_t row = c.row0 + c.r0 + r;
            w[r] = (col < n && row < sg.rows) ? *reinterpret_cast<const int4 *>(W + (size_t)row * n + col) : make_int4(0, 0, 0, 0);
        }
    };
    auto issue_scales = [&](const Cur &c, float (&wsr)[NWS]) {   // the tile's weight scales for chunk c.c: (r, g) = idx / GC, idx % GC
        const GemvSeg &sg = a.seg[c.sidx + c.pass];
#pragma unroll
        for (int k = 0; k < NWS; k++) {
            const int idx = k * 64 + lane;
            const int r = idx / GC, g = idx % GC;
            const uint32_t row = c.row0 + r;
            const int gg = c.c * GC + g;
            wsr[k] = (idx < TR * GC && row < sg.rows && gg < ng) ? sg.ws[(size_t)row * ng + gg] : 0.0f;
        }
    };

    if (cu.valid) { issue(cu, wa); issue_scales(cu, wsr); }   // in flight across the prologue

    prologue_q80<B, GS>(a, xq, xs, red, wgt);

    float val = 0.0f, res0 = 0.0f;
    while (cu.valid) {
        Cur nx = cu;
        cursor_advance<TR, RBL>(a, nx, nchunk, npass, stride);
        if (nx.valid) {
            issue(nx, wb);
            if (nx.r0 == 0) issue_scales(nx, wsn);     // nx opens a new (tile|pass|chunk)
        }

        // ---- consume batch `cu` ------------------------------------------------------------------
        const int col = (cu.c << 10) + lane * 16;
        const bool act = col < n;
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < nb) {
                const int4 xv = act ? *reinterpret_cast<const int4 *>(xq + (size_t)b * n + col) : make_int4(0, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RBL; r++) {
                    int iv = __builtin_amdgcn_sdot4(wa[r].x, xv.x, 0, false);
                    iv = __builtin_amdgcn_sdot4(wa[r].y, xv.y, iv, false);
                    iv = __builtin_amdgcn_sdot4(wa[r].z, xv.z, iv, false);
                    iv = __builtin_amdgcn_sdot4(wa[r].w, xv.w, iv, false);
                    iv = dpp_group_sum<LPG>(iv);
                    if (leader) tab[((cu.r0 + r) * B + b) * PITCH + gl] = __int_as_float(iv);
                }
            }
        }

        if (cu.r0 + RBL >= TR) {                       // last batch of this (tile, pass, chunk)
            // (A) parallel: p = ((float)ival * ws) * xs            reference infer.c:672
#pragma unroll
            for (int k = 0; k < NWS; k++) {
                const int idx = k * 64 + lane;
                if (idx < TR * GC) {
                    const int r = idx / GC, g = idx % GC;
                    const int gg = cu.c * GC + g;
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        if (b < nb) {
                            const int e = (r * B + b) * PITCH + g;
                            const float xsc = (gg < ng) ? xs[(size_t)b * ng + gg] : 0.0f;
                            tab[e] = ((float)__float_as_int(tab[e]) * wsr[k]) * xsc;
                        }
                    }
                }
            }
            // (B) ordered: val += p[g], g ascending            reference infer.c:668-674
            const int gvalid = min(GC, ng - cu.c * GC);
            if (lane < PAIRS) {
                const float *f = tab + lane * PITCH;
                for (int g = 0; g < gvalid; g++) val += f[g];
            }
            if (cu.c + 1 == nchunk) {                   // row finished for this pass
                if (cu.pass + 1 == npass) {
                    if (lane < PAIRS && (lane % B) < nb) {
                        const uint32_t row = cu.row0 + lane / B;
                        if (row < a.seg[cu.sidx].rows)
                            emit(a, a.seg[cu.sidx], lane % B, row, (npass == 2) ? res0 : val, val);
                    }
                } else {
                    res0 = val;
                }
                val = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < NWS; k++) wsr[k] = wsn[k];
        }
#pragma unroll
        for (int r = 0; r < RBL; r++) wa[r] = wb[r];
        cu = nx;
    }
}

// ------------------------------------------------------------------------------------------------
// FP32
// ------------------------------------------------------------------------------------------------
template <int B, int TR>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RBL = TR;
    const int n = (int)a.n;
    float *xf = reinterpret_cast<float *>(smem);
    float *red = xf + (((size_t)B * n + 3) & ~(size_t)3);
    float *wgt = red + 16 * B;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nchunk = (n + 255) >> 8;
    const int npass = (a.epi == GEMV_EPI_SWIGLU) ? 2 : 1;
    const uint32_t stride = gridDim.x * 4;
    const int nb = (int)a.nb;

    using Cur = Cursor<TR, RBL>;
    Cur cu{ blockIdx.x * 4 + wid, 0, 0, 0, 0, 0, false };
    cu.valid = locate_tile<TR>(a, cu.tile, cu.sidx, cu.row0);

    float4 wa[RBL], wb[RBL];
    auto issue = [&](const Cur &c, float4 (&w)[RBL]) {
        const GemvSeg &sg = a.seg[c.sidx + c.pass];
        const float *W = reinterpret_cast<const float *>(sg.w);
        const int col = (c.c << 8) + lane * 4;
#pragma unroll
        for (int r = 0; r < RBL; r++) {
            const uint32_t row = c.row0 + r;
            w[r] = (col < n && row < sg.rows) ? *reinterpret_cast<const float4 *>(W + (size_t)row * n + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (cu.valid) issue(cu, wa);

    prologue_f32<B>(a, xf, red, wgt);

    float acc[TR][B], res0[TR][B];
#pragma unroll
    for (int r = 0; r < TR; r++)
#pragma unroll
        for (int b = 0; b < B; b++) { acc[r][b] = 0.0f; res0[r][b] = 0.0f; }

    while (cu.valid) {
        Cur nx = cu;
        cursor_advance<TR, RBL>(a, nx, nchunk, npass, stride);
        if (nx.valid) issue(nx, wb);

        const int col = (cu.c << 8) + lane * 4;
        const bool act = col < n;
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < nb) {
                const float4 xv = act ? *reinterpret_cast<const float4 *>(xf + (size_t)b * n + col) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                for (int b = 0; b < B; b++) acc[r][b] = wave_sum(acc[r][b]);
            if (cu.pass + 1 == npass) {
#pragma unroll
                for (int r = 0; r < TR; r++)
#pragma unroll
                    for (int b = 0; b < B; b++)
                        if (lane == r * B + b && b < nb && cu.row0 + r < a.seg[cu.sidx].rows)
                            emit(a, a.seg[cu.sidx], b, cu.row0 + r, (npass == 2) ? res0[r][b] : acc[r][b], acc[r][b]);
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

static inline size_t attn_wgt_floats(const GemvArgs &a) { return a.attn_part ? ((size_t)a.attn_n_head * a.attn_nsplit + 3) & ~(size_t)3 : 0; }

template <int B, int TR, int GS>
static hipError_t launch_q80(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    a.tiles = count_tiles(a, TR);
    uint32_t wgs = (a.tiles + 3) / 4;
    if (wgs > max_wg) wgs = max_wg;
    if (!wgs) return hipSuccess;
    const size_t xq = ((size_t)B * a.n + 15) & ~(size_t)15;
    const size_t xs = (((size_t)B * (a.n / GS) + 3) & ~(size_t)3) * 4;
    const size_t lds = xq + xs + (16 * B + attn_wgt_floats(a)) * 4 + (size_t)4 * TR * B * (1024 / GS + 1) * 4;
    auto kern = &gemv_q80_kernel<B, TR, GS>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, st, a);
    return hipGetLastError();
}

template <int B, int GS>
static hipError_t launch_q80_tr(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    // big GEMVs: 64/B-row tiles (every lane folds) once that still leaves >= 2 tiles per wave slot
    uint32_t rows = 0;
    for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    constexpr int TRBIG = (B <= 2) ? 16 : 8;
    if (a.epi != GEMV_EPI_SWIGLU && rows / TRBIG >= 4096) return launch_q80<B, TRBIG, GS>(a, max_wg, st);
    return launch_q80<B, 4, GS>(a, max_wg, st);
}

template <int B>
static hipError_t launch_q80_gs(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    switch (a.gs) {
    case 32: return launch_q80_tr<B, 32>(a, max_wg, st);
    case 64: return launch_q80_tr<B, 64>(a, max_wg, st);
    case 128: return launch_q80_tr<B, 128>(a, max_wg, st);
    case 256: return launch_q80_tr<B, 256>(a, max_wg, st);
    default: return hipErrorInvalidValue;
    }
}

template <int B>
static hipError_t launch_f32(GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    constexpr int TR = (B >= 8) ? 2 : 4;
    a.tiles = count_tiles(a, TR);
    uint32_t wgs = (a.tiles + 3) / 4;
    if (wgs > max_wg) wgs = max_wg;
    if (!wgs) return hipSuccess;
    const size_t lds = ((((size_t)B * a.n + 3) & ~(size_t)3) + 16 * B + attn_wgt_floats(a)) * 4;
    auto kern = &gemv_f32_kernel<B, TR>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_gemv(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    if (quant == 0x80u) {
        if (a.nb <= 1) return launch_q80_gs<1>(a, max_wg, st);
        if (a.nb <= 2) return launch_q80_gs<2>(a, max_wg, st);
        if (a.nb <= 4) return launch_q80_gs<4>(a, max_wg, st);
        return launch_q80_gs<8>(a, max_wg, st);
    }
    if (a.nb <= 1) return launch_f32<1>(a, max_wg, st);
    if (a.nb <= 2) return launch_f32<2>(a, max_wg, st);
    if (a.nb <= 4) return launch_f32<4>(a, max_wg, st);
    return launch_f32<8>(a, max_wg, st);
}

}  // namespace nano
