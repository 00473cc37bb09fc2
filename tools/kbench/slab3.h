// slab3.h -- per-layer Q80 GEMV ("slab") with buffer loads and branch-free address math (prototype).
#pragma once
#include "q80k.h"

namespace k {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

enum : uint32_t { F_NORM = 1u, F_PRE = 2u };

struct Slab {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3];
    uint32_t n, ng, rw, log2_tiles, nchunk, magic_nchunk, units, epi, flags, nb;
    const float *xin; const float *norm_w;
    uint32_t xin_bstride, out_bstride[3];
    const int8_t *xq_in; const float *xs_in;
    unsigned long long *dbg;
};
#define TS3(i) do { if (TS && a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int4 bload16(__amdgpu_buffer_rsrc_t r, uint32_t off, bool nt) {
    const i32x4_t v = nt ? __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 bload16f(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float bload4f(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

constexpr uint32_t OOB = 0x7ffffff0u;

// ABL (ablation, measurement only): 1 no quantization math, 2 no dots, 4 no ordered fold, 8 no weight loads, 16 no activation loads, 32 no prologue barrier work at all
template <int GS, int B, int NV, int UPW, bool TS, int ABL = 0, int ROLE = 0>
__global__ __launch_bounds__(1024) void q80_slab3(const Slab a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 4;
    constexpr int LPG = GS / 16, GC = 1024 / GS;
    constexpr int NS = (TR + LPG - 1) / LPG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthr = blockDim.x, NW = nthr >> 6;
    const uint32_t n = a.n, ng = a.ng;
    const uint32_t n16 = (n + 15) & ~15u, ng4 = (ng + 3) & ~3u;
    const uint32_t PITCH = (GC == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    const uint32_t RW = a.rw;
    const uint32_t epi = ROLE == 0 ? a.epi : ROLE == 1 ? (uint32_t)EPI_STORE : ROLE == 2 ? (uint32_t)EPI_RESID : (uint32_t)EPI_SWIGLU;
    const bool swiglu = epi == EPI_SWIGLU;
    const uint32_t nmat = swiglu ? 2 : 1;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);                 // [B][n16]
    float *xs = reinterpret_cast<float *>(smem + B * n16);         // [B][ng4]
    float *red = xs + B * ng4;                                     // [B][16]
    float *P = red + B * 16;                                       // [B][nmat][RW][PITCH]
    TS3(0);

    // ---- activation loads first ---------------------------------------------------------------------------
    float4 xv[B][NV], nwv[NV];
    const bool pre = ROLE == 0 ? (a.flags & F_PRE) != 0 : false, norm = ROLE == 0 ? (a.flags & F_NORM) != 0 : (ROLE == 1 || ROLE == 3);
    {
        const __amdgpu_buffer_rsrc_t rx = mkrsrc(a.xin, pre ? 0u : ((a.nb - 1) * a.xin_bstride + n) * 4u);
        const __amdgpu_buffer_rsrc_t rn = mkrsrc(a.norm_w, norm ? n * 4u : 0u);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (uint32_t)(tid + j * nthr) * 4u;
            const uint32_t off = (i < n) ? i * 4u : OOB;
#pragma unroll
            for (int b = 0; b < B; b++) xv[b][j] = (ABL & 16) ? make_float4(1.f, 2.f, 3.f, (float)tid) : bload16f(rx, (b < (int)a.nb) ? off + (uint32_t)b * a.xin_bstride * 4u : OOB);
            nwv[j] = (ABL & 16) ? make_float4(1.f, 1.f, 1.f, 1.f) : bload16f(rn, off);
        }
    }

    // ---- this workgroup's rows: one segment ------------------------------------------------------------------
    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
    const float *ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const uint32_t tmask = (1u << a.log2_tiles) - 1u;

    int4 wv[UPW][TR];
    float sv[UPW][NS];
#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const uint32_t u = (uint32_t)wid + (uint32_t)k * NW;
        const uint32_t t = (u * a.magic_nchunk) >> 16;             // u / nchunk
        const uint32_t c = u - t * a.nchunk;
        const uint32_t tl = t & tmask, mat = t >> a.log2_tiles;
        const bool live = u < a.units;
        const __amdgpu_buffer_rsrc_t rw_ = mkrsrc(mat ? a.w[1] : w0, live ? rows0 * n : 0u);
        const __amdgpu_buffer_rsrc_t rs_ = mkrsrc(mat ? a.ws[1] : ws0, live ? rows0 * ng * 4u : 0u);
        const uint32_t lrow = lrow0 + tl * TR;
        const uint32_t col = (c << 10) + (uint32_t)lane * 16u;
        const uint32_t base = (col < n) ? lrow * n + col : OOB;
#pragma unroll
        for (int r = 0; r < TR; r++) wv[k][r] = (ABL & 8) ? make_int4(r, lane, 3, 4) : bload16(rw_, base + (uint32_t)r * n, true);
        const uint32_t g = c * GC + (uint32_t)lane / LPG;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t r = ((uint32_t)lane % LPG) + s * LPG;
            sv[k][s] = (ABL & 8) ? 1.0f : bload4f(rs_, (r < TR && g < ng) ? ((lrow + r) * ng + g) * 4u : OOB);
        }
    }
    const int lrw = (int)a.log2_tiles + 2;                          // log2(rows per workgroup)
    const int fb = tid >> lrw, frl = tid & ((int)RW - 1);           // fold thread -> (sequence, local row)
    const bool fold_live = tid < (int)(RW * B) && fb < (int)a.nb && lrow0 + frl < rows0;
    float oldv = 0.0f;
    if (epi == EPI_RESID && fold_live) oldv = out0[(size_t)fb * obs + lrow0 + frl];
    TS3(1);

    // ---- rmsnorm + Q80 quantization from registers ---------------------------------------------------------
    if (pre) {
        for (uint32_t i = tid * 16; i < n; i += nthr * 16) *reinterpret_cast<int4 *>(xq + i) = *reinterpret_cast<const int4 *>(a.xq_in + i);
        for (uint32_t i = tid; i < ng; i += nthr) xs[i] = a.xs_in[i];
    } else {
        float ss[B];
#pragma unroll
        for (int b = 0; b < B; b++) ss[b] = 1.0f;
        if (norm) {
#pragma unroll
            for (int b = 0; b < B; b++) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NV; j++) { acc += xv[b][j].x * xv[b][j].x; acc += xv[b][j].y * xv[b][j].y; acc += xv[b][j].z * xv[b][j].z; acc += xv[b][j].w * xv[b][j].w; }
                acc = dpp_wave_sum(acc);
                if (lane == 0) red[b * 16 + wid] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < B; b++) {
                float t = 0.0f;
                for (int w = 0; w < NW; w++) t += red[b * 16 + w];
                t /= (float)n; t += 1e-5f;
                ss[b] = 1.0f / sqrtf(t);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (uint32_t)(tid + j * nthr) * 4u;
#pragma unroll
            for (int b = 0; b < B; b++) {
                float4 v = xv[b][j];
                if (norm) { v.x = nwv[j].x * (ss[b] * v.x); v.y = nwv[j].y * (ss[b] * v.y); v.z = nwv[j].z * (ss[b] * v.z); v.w = nwv[j].w * (ss[b] * v.w); }
                float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                m = dpp_group_max<GS / 4>(m);
                const float scale = m / 127.0f;
                if (i < n) {
                    const int q0 = (ABL & 1) ? (int)v.x : q80_quant1(v.x, scale), q1 = (ABL & 1) ? (int)v.y : q80_quant1(v.y, scale), q2 = (ABL & 1) ? (int)v.z : q80_quant1(v.z, scale), q3 = (ABL & 1) ? (int)v.w : q80_quant1(v.w, scale);
                    *reinterpret_cast<uint32_t *>(xq + b * n16 + i) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                    if ((tid % (GS / 4)) == 0) xs[b * ng4 + i / GS] = scale;
                }
            }
        }
    }
    __syncthreads();
    TS3(2);

#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const uint32_t u = (uint32_t)wid + (uint32_t)k * NW;
        if (u < a.units) {
            const uint32_t t = (u * a.magic_nchunk) >> 16;
            const uint32_t c = u - t * a.nchunk;
            const uint32_t tl = t & tmask, mat = t >> a.log2_tiles;
            const uint32_t col = (c << 10) + (uint32_t)lane * 16u;
            const uint32_t g = c * GC + (uint32_t)lane / LPG;
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int4 xvq = (col < n) ? *reinterpret_cast<const int4 *>(xq + b * n16 + col) : make_int4(0, 0, 0, 0);
                int iv[TR];
#pragma unroll
                for (int r = 0; r < TR; r++) {
                    if (ABL & 2) { iv[r] = wv[k][r].x + xvq.x; continue; }
                    int t2 = __builtin_amdgcn_sdot4(wv[k][r].x, xvq.x, 0, false);
                    t2 = __builtin_amdgcn_sdot4(wv[k][r].y, xvq.y, t2, false);
                    t2 = __builtin_amdgcn_sdot4(wv[k][r].z, xvq.z, t2, false);
                    t2 = __builtin_amdgcn_sdot4(wv[k][r].w, xvq.w, t2, false);
                    iv[r] = dpp_group_sum<LPG>(t2);
                }
                const float xsc = (g < ng) ? xs[b * ng4 + g] : 0.0f;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const uint32_t r = ((uint32_t)lane % LPG) + s * LPG;
                    int v = iv[0];
#pragma unroll
                    for (int q = 1; q < TR; q++) v = (q == (int)r) ? iv[q] : v;
                    if (r < TR && g < ng) P[(((size_t)b * nmat + mat) * RW + tl * TR + r) * PITCH + g] = ((float)v * sv[k][s]) * xsc;
                }
            }
        }
    }
    TS3(3);
    __syncthreads();
    TS3(4);
    if (tid < (int)(RW * B)) {
        const int b = fb, rl = frl;
        // ordered fold (reference infer.c:668-674): all LDS reads of a 16-group batch are issued before the
        // dependent add chain; groups beyond ng add +0.0f (exact: the running value is never -0.0f)
        float v0 = 0.0f, v1 = 0.0f;
        const float *p0 = P + (((size_t)b * nmat) * RW + rl) * PITCH;
        const float *p1 = p0 + (size_t)RW * PITCH;
        for (uint32_t g0 = 0; g0 < ((ABL & 4) ? 1u : ng); g0 += 16) {
            float4 t[4], u[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                t[q] = (g0 + 4 * q < ng4) ? *reinterpret_cast<const float4 *>(p0 + g0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (swiglu) u[q] = (g0 + 4 * q < ng4) ? *reinterpret_cast<const float4 *>(p1 + g0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t g = g0 + 4 * q;
                v0 += (g < ng) ? t[q].x : 0.0f; v0 += (g + 1 < ng) ? t[q].y : 0.0f; v0 += (g + 2 < ng) ? t[q].z : 0.0f; v0 += (g + 3 < ng) ? t[q].w : 0.0f;
            }
            if (swiglu) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t g = g0 + 4 * q;
                    v1 += (g < ng) ? u[q].x : 0.0f; v1 += (g + 1 < ng) ? u[q].y : 0.0f; v1 += (g + 2 < ng) ? u[q].z : 0.0f; v1 += (g + 3 < ng) ? u[q].w : 0.0f;
                }
            }
        }
        TS3(5);
        if (fold_live) {
            float o = v0;
            if (swiglu) { float h = v0; h *= (1.0f / (1.0f + expf(-h))); h *= v1; o = h; }
            else if (epi == EPI_RESID) o = oldv + v0;
            out0[(size_t)b * obs + lrow0 + rl] = o;
        }
    }
    TS3(6);
}

static inline size_t slab3_lds(uint32_t n, uint32_t gs, uint32_t rw, uint32_t nmat, uint32_t B) {
    const size_t n16 = (n + 15) & ~15u, ng = n / gs, ng4 = (ng + 3) & ~3u;
    const size_t pitch = (1024 / gs == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    return B * n16 + B * ng4 * 4 + B * 64 + (size_t)B * nmat * rw * pitch * 4;
}

}  // namespace k
