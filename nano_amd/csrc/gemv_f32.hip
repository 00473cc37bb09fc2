// gemv_f32.hip -- FP32 decode GEMV for gfx950 (reference matmul, infer/infer.c:637-651), same latency-oriented SLAB
// structure as the Q80 kernels (gemv_q80_impl.h): a workgroup owns `rw` consecutive rows, its work units (4 rows x one
// 256-float column chunk, x2 matrices for SwiGLU) are dealt to its waves, every wave issues the activation loads and
// then ALL its weight loads at kernel entry through buffer descriptors (one memory round trip), rmsnorm / the
// split-attention combine run from registers while the weights are in flight.
// The reference adds the n products of a row sequentially; here a lane accumulates its float4 slices with fused
// multiply-adds, a DPP tree sums the 64 lanes and one thread adds the chunk partials in order -- the summation ORDER
// differs, so results match the reference to rounding (stated tolerance 1e-5 relative, DESIGN.md "Parity").
// HBM-bound byte work (2 flop / 4 bytes): no MFMA.
#include "gemv_common.h"

namespace nano {

namespace {

template <int ROLE, int B, int NV>
__device__ __forceinline__ void stage_finish_f32(const GemvDev &a, Staged<B, NV> &r, float *xf, float *red, uint32_t n4) {
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
                *reinterpret_cast<float4 *>(xf + b * n4 + i) = v;
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
        if (norm) {                     // rmsnorm scale (infer.c:603-609); tree order
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
                if (i < n) *reinterpret_cast<float4 *>(xf + b * n4 + i) = v;
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float4 bload_wf(__amdgpu_buffer_rsrc_t r, uint32_t off) {          // streamed once: non-temporal
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

template <int ROLE, int B, int NV, int UPW>
__global__ __launch_bounds__(1024) void gemv_f32_slab_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (int)(a.nthr >> 6);
    const uint32_t n = a.n, n4 = (n + 3) & ~3u;
    const uint32_t nchunk = a.nchunk, PC = (nchunk + 3) & ~3u;      // a.nchunk: 256-float chunks per row
    const uint32_t RW = a.rw;
    const uint32_t epi = role_epi<ROLE>(a);
    const bool swiglu = epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = swiglu ? 2 : 1;
    float *xf = reinterpret_cast<float *>(smem);                   // [B][n4]
    float *red = xf + B * n4;                                      // [B][16] (+ combine weights [B][n_head][8])
    float *P = red + B * 16 + (has_flag<ROLE>(a, F_COMBINE) ? B * a.attn_n_head * 8 : 0);   // [B][nmat][RW][PC]

    Staged<B, NV> sx;
    stage_issue<ROLE, B, NV>(a, sx);

    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const float *w0 = reinterpret_cast<const float *>(sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2]);
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const uint32_t tmask = (1u << a.log2_tiles) - 1u;

    float4 wv[UPW][TR];
#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const uint32_t u = (uint32_t)wid + (uint32_t)k * NW;
        const uint32_t t = (u * a.magic_nchunk) >> 16;                 // u / nchunk
        const uint32_t c = u - t * nchunk;
        const uint32_t tl = t & tmask, mat = t >> a.log2_tiles;
        const bool live = u < a.units;
        const __amdgpu_buffer_rsrc_t rw_ = mkrsrc(mat ? reinterpret_cast<const float *>(a.w[1]) : w0, live ? rows0 * n * 4u : 0u);
        const uint32_t lrow = lrow0 + tl * TR;
        const uint32_t col = (c << 8) + (uint32_t)lane * 4u;
        const uint32_t base = (col < n) ? (lrow * n + col) * 4u : OOB;
#pragma unroll
        for (int r = 0; r < TR; r++) wv[k][r] = bload_wf(rw_, base + (uint32_t)r * n * 4u);
    }
    const int lrw = (int)a.log2_tiles + 2;
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

    stage_finish_f32<ROLE, B, NV>(a, sx, xf, red, n4);

#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const uint32_t u = (uint32_t)wid + (uint32_t)k * NW;
        if (u < a.units) {
            const uint32_t t = (u * a.magic_nchunk) >> 16;
            const uint32_t c = u - t * nchunk;
            const uint32_t tl = t & tmask, mat = t >> a.log2_tiles;
            const uint32_t col = (c << 8) + (uint32_t)lane * 4u;
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (b < (int)a.nb) {
                    const float4 xv = (col < n) ? *reinterpret_cast<const float4 *>(xf + b * n4 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int r = 0; r < TR; r++) {
                        float p = wv[k][r].x * xv.x;
                        p = __builtin_fmaf(wv[k][r].y, xv.y, p); p = __builtin_fmaf(wv[k][r].z, xv.z, p); p = __builtin_fmaf(wv[k][r].w, xv.w, p);
                        p = dpp_wave_sum(p);
                        if (lane == 0) P[(((size_t)b * nmat + mat) * RW + tl * TR + r) * PC + c] = p;
                    }
                }
            }
        }
    }
    __syncthreads();

    if (tid < (int)(RW * B)) {
        float v0 = 0.0f, v1 = 0.0f;
        const float *p0 = P + (((size_t)fb * nmat) * RW + frl) * PC;
        const float *p1 = p0 + (size_t)RW * PC;
        for (uint32_t c = 0; c < nchunk; c++) { v0 += p0[c]; if (swiglu) v1 += p1[c]; }
        // write-through (sc1) store, see gemv_q80_impl.h: nothing is left for the write-back at the end of the kernel
        if (fold_live) __hip_atomic_store(out0 + (size_t)fb * obs + (size_t)opos * ops + lrow0 + frl, finish_epi(epi, has_add ? v0 + addv : v0, v1, oldv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct F32Plan { uint32_t rw, nw, upw, nv; };
static F32Plan plan_f32(const GemvArgs &a, int B) {
    const uint32_t nchunk = (a.n + 255) / 256, nmat = a.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    uint32_t align = 0;
    if (nseg > 1) for (uint32_t s = 0; s < nseg; s++) align |= a.seg[s].rows;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    uint32_t rw = 4;                 // 4 KiB of weights per unit: a few units per workgroup, >= 256 workgroups
    while (rw < 32 && (align % (rw * 2)) == 0 && (rw * 2 / 4) * nchunk <= 8 && rows / (rw * 2) >= 256) rw *= 2;
    const uint32_t units = (rw / 4) * nchunk * nmat;
    uint32_t nw = units < 8 ? units : 8;
    uint32_t want = (a.n * (uint32_t)(B > 2 ? B / 2 : 1) + 1023) / 1024;
    if (want > 16) want = 16;
    if (nw < want) nw = want;
    if (nw * 64 < rw * (uint32_t)B) nw = (rw * (uint32_t)B + 63) / 64;
    if (nw < 2) nw = 2;
    uint32_t upw = (units + nw - 1) / nw;
    while (upw > 4 && nw < 16) { nw++; upw = (units + nw - 1) / nw; }
    return F32Plan{rw, nw, upw, (a.n + 256 * nw - 1) / (256 * nw)};
}

template <int ROLE, int B, int NV, int UPW>
static hipError_t launch_f32_t(const GemvDev &d, const F32Plan &p, uint32_t rows, hipStream_t st) {
    const uint32_t nmat = d.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n4 = (d.n + 3) & ~3u, pc = (d.nchunk + 3) & ~3u;
    const size_t lds = (B * n4 + B * 16 + ((d.flags & F_COMBINE) ? (size_t)B * d.attn_n_head * 8 : 0) + (size_t)B * nmat * p.rw * pc) * 4;
    auto kern = &gemv_f32_slab_kernel<ROLE, B, NV, UPW>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemvDev dd = d; dd.nthr = 64 * p.nw;
    hipLaunchKernelGGL(kern, dim3((rows + p.rw - 1) / p.rw), dim3(64 * p.nw), lds, st, dd);
    return hipGetLastError();
}
template <int ROLE, int B>
static hipError_t launch_f32_r(const GemvDev &d, const F32Plan &p, uint32_t rows, hipStream_t st) {
    if (p.upw > 4) return hipErrorInvalidValue;
#define F32_GO(NV_, UPW_) do { if constexpr (B * NV_ <= 8) return launch_f32_t<ROLE, B, NV_, UPW_>(d, p, rows, st); } while (0)
    int nv = p.nv <= 1 ? 1 : p.nv <= 2 ? 2 : p.nv <= 4 ? 4 : 0;
    const int upw = p.upw <= 1 ? 1 : p.upw <= 2 ? 2 : 4;
    if (B * nv > 8) nv = 0;
    if (nv == 1) { if (upw == 1) F32_GO(1, 1); if (upw == 2) F32_GO(1, 2); F32_GO(1, 4); }
    if (nv == 2) { if (upw == 1) F32_GO(2, 1); if (upw == 2) F32_GO(2, 2); F32_GO(2, 4); }
    if (nv == 4) { if (upw == 1) F32_GO(4, 1); if (upw == 2) F32_GO(4, 2); F32_GO(4, 4); }
    if (upw == 1) F32_GO(0, 1);
    if (upw == 2) F32_GO(0, 2);
    F32_GO(0, 4);
    return hipErrorInvalidValue;
#undef F32_GO
}
template <int B>
static hipError_t launch_f32_b(const GemvArgs &a, hipStream_t st) {
    GemvDev d = to_dev(a);
    d.tile_max = nullptr;
    const F32Plan p = plan_f32(a, B);
    d.nchunk = (a.n + 255) / 256;
    d.magic_nchunk = (65536 + d.nchunk - 1) / d.nchunk;
    d.rw = p.rw;
    uint32_t l2 = 0; while ((1u << l2) < p.rw / 4) l2++;
    d.log2_tiles = l2;
    d.units = (p.rw / 4) * d.nchunk * (d.epi == GEMV_EPI_SWIGLU ? 2 : 1);
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    if constexpr (B == 1) {
        const uint32_t f = d.flags;
        if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_f32_r<R_NORM_STORE, B>(d, p, rows, st);
        if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_f32_r<R_RESID, B>(d, p, rows, st);
        if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_f32_r<R_RESID_COMBINE, B>(d, p, rows, st);
        if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU) return launch_f32_r<R_NORM_SWIGLU, B>(d, p, rows, st);
    }
    return launch_f32_r<R_GENERIC, B>(d, p, rows, st);
}

}  // namespace

hipError_t launch_gemv_f32(const GemvArgs &a, hipStream_t st) {
    if (a.nb == 0 || a.nb > 8 || a.n % 4 || a.nseg == 0 || a.nseg > 3 || a.xq_in) return hipErrorInvalidValue;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s < a.nseg; s++) if (a.seg[s].rows % 4) return hipErrorInvalidValue;
    if (a.nb <= 1) return launch_f32_b<1>(a, st);
    if (a.nb <= 2) return launch_f32_b<2>(a, st);
    if (a.nb <= 4) return launch_f32_b<4>(a, st);
    return launch_f32_b<8>(a, st);
}

}  // namespace nano
