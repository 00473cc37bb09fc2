// q80k.h -- prototype Q80 GEMV kernels (stream = tall matrices / classifier, slab = per-layer matrices).
// Development copy used by tools/kbench; the production versions live in nano_amd/csrc/gemv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace k {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ int4 ld16(const void *p) {
    if (NT) { const i32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(p)); return make_int4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const int4 *>(p);
}

#define DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))

template <int W> __device__ __forceinline__ int dpp_group_sum(int v) {
    if (W >= 2) v += DPP_I(v, 0xB1);
    if (W >= 4) v += DPP_I(v, 0x4E);
    if (W >= 8) v += DPP_I(v, 0x141);
    if (W >= 16) v += DPP_I(v, 0x140);
    return v;
}
template <int W> __device__ __forceinline__ float dpp_group_max(float v) {
    if (W >= 2) v = fmaxf(v, DPP_F(v, 0xB1));
    if (W >= 4) v = fmaxf(v, DPP_F(v, 0x4E));
    if (W >= 8) v = fmaxf(v, DPP_F(v, 0x141));
    if (W >= 16) v = fmaxf(v, DPP_F(v, 0x140));
    if (W >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (W >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v += DPP_F(v, 0xB1); v += DPP_F(v, 0x4E); v += DPP_F(v, 0x141); v += DPP_F(v, 0x140);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int q80_quant1(float x, float scale) {
    float qv = x / scale;
    float r = roundf(qv);
    return (r != r) ? 0 : (int)r;
}

enum : uint32_t { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

struct Seg { const int8_t *w; const float *ws; float *out; uint32_t rows; };
struct Args {
    Seg seg[3];
    uint32_t nseg, n, epi, rows_per_wg;
    const float *xin; const float *norm_w;
    const int8_t *xq_in; const float *xs_in;   // test path: pre-quantized activation
    unsigned long long *dbg;                   // optional timestamps [wg][8] (100 MHz realtime counter)
};
#define TSTAMP(i) do { if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ---------------------------------------------------------------------------------------------------
// workgroup-cooperative activation staging: rmsnorm (optional) + Q80 quantization of x[n] into LDS
// thread t owns elements [4t,4t+4) of every 4*nthr stride; a group is GS/4 consecutive threads
// ---------------------------------------------------------------------------------------------------
template <int GS>
__device__ __forceinline__ void stage_q80_wg(const Args &a, int8_t *xq, float *xs, float *red) {
    const int n = (int)a.n, tid = threadIdx.x, nthr = blockDim.x;
    if (a.xq_in) {
        for (int i = tid * 16; i < n; i += nthr * 16) *reinterpret_cast<int4 *>(xq + i) = *reinterpret_cast<const int4 *>(a.xq_in + i);
        for (int i = tid; i < n / GS; i += nthr) xs[i] = a.xs_in[i];
        __syncthreads();
        return;
    }
    float ss = 1.0f;
    if (a.norm_w) {
        float acc = 0.0f;
        for (int i = tid * 4; i < n; i += nthr * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(a.xin + i);
            acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
        }
        acc = dpp_wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        float t = 0.0f;
        for (int w = 0; w < (nthr >> 6); w++) t += red[w];
        t /= (float)n; t += 1e-5f;
        ss = 1.0f / sqrtf(t);
    }
    for (int i = tid * 4; i < n; i += nthr * 4) {     // n % (4*GS/4 ...) : n multiple of GS, nthr*4 multiple of GS
        float4 v = *reinterpret_cast<const float4 *>(a.xin + i);
        if (a.norm_w) {
            const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
            v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
        }
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = dpp_group_max<GS / 4>(m);
        const float scale = m / 127.0f;
        const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
        *reinterpret_cast<uint32_t *>(xq + i) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
        if ((tid % (GS / 4)) == 0) xs[i / GS] = scale;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// STREAM kernel: tall matrix (classifier).  A wave owns 16-row tiles, tile t = wave + k*nwaves (the chip
// sweeps memory linearly), next tile's 16 KiB in flight while the current one is consumed.
//   lane l loads bytes [16l,16l+16) of each row chunk; int group sums by DPP; leaders park them in a
//   wave-private LDS table; lane l then owns (row l/4, groups (l%4)*F..+F) : the scales of a whole tile
//   arrive as ONE coalesced load, products ((float)iv*ws)*xs are formed by all 64 lanes, and the
//   reference's ascending group order is kept by a 4-stage quad chain.
// ---------------------------------------------------------------------------------------------------
template <int GS, bool NT>
__global__ __launch_bounds__(256) void q80_stream(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LPG = GS / 16, GC = 1024 / GS, F = GC / 4;
    static_assert(F >= 1, "GS too large");
    const int n = (int)a.n, ng = n / GS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = (n + 15) & ~15;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + n16);
    float *red = xs + ((ng + 3) & ~3);
    int *tab = reinterpret_cast<int *>(red + 8) + wid * 16 * GC;

    const Seg sg = a.seg[0];
    const uint32_t rows = sg.rows;
    const int nchunk = (n + 1023) >> 10;
    const uint32_t ntiles = (rows + 15) >> 4;
    const uint32_t nwaves = gridDim.x * 4, wave_g = blockIdx.x * 4 + wid;
    const uint32_t nunits = (wave_g < ntiles) ? ((ntiles - wave_g + nwaves - 1) / nwaves) * nchunk : 0;

    int4 wA[16], wB[16];
    float sA[F], sB[F];
    auto issue = [&](uint32_t u, int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + (u / nchunk) * nwaves;
        const int c = (int)(u % nchunk);
        const uint32_t row0 = tile << 4;
        const int col = (c << 10) + lane * 16;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = min(row0 + r, rows - 1);
            w[r] = (col < n) ? ld16<NT>(sg.w + (size_t)row * n + col) : make_int4(0, 0, 0, 0);
        }
        const uint32_t rr = min(row0 + (lane >> 2), rows - 1);
        const int gb = c * GC + (lane & 3) * F;
#pragma unroll
        for (int f = 0; f < F; f++) s[f] = (gb + f < ng) ? sg.ws[(size_t)rr * ng + gb + f] : 0.0f;
    };
    if (nunits) issue(0, wA, sA);

    stage_q80_wg<GS>(a, xq, xs, red);

    float val = 0.0f;
    auto consume = [&](uint32_t u, int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + (u / nchunk) * nwaves;
        const int c = (int)(u % nchunk);
        const int col = (c << 10) + lane * 16;
        const int4 xv = (col < n) ? *reinterpret_cast<const int4 *>(xq + col) : make_int4(0, 0, 0, 0);
        int iv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int t = __builtin_amdgcn_sdot4(w[r].x, xv.x, 0, false);
            t = __builtin_amdgcn_sdot4(w[r].y, xv.y, t, false);
            t = __builtin_amdgcn_sdot4(w[r].z, xv.z, t, false);
            t = __builtin_amdgcn_sdot4(w[r].w, xv.w, t, false);
            iv[r] = dpp_group_sum<LPG>(t);
        }
        if ((lane % LPG) == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) tab[r * GC + lane / LPG] = iv[r];
        }
        const int gb = c * GC + (lane & 3) * F;
        float p[F];
#pragma unroll
        for (int f = 0; f < F; f++) {
            const int v = tab[(lane >> 2) * GC + (lane & 3) * F + f];
            const float xsc = (gb + f < ng) ? xs[gb + f] : 0.0f;
            p[f] = ((float)v * s[f]) * xsc;
        }
        // ordered fold over the 4 lanes of a row: stage k adds lane k's products onto the running value
        float cur = val;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            float t = cur;
#pragma unroll
            for (int f = 0; f < F; f++) t += p[f];      // groups beyond ng contribute +0.0f exactly
            cur = (st == 0) ? DPP_F(t, 0x00) : (st == 1) ? DPP_F(t, 0x55) : (st == 2) ? DPP_F(t, 0xAA) : DPP_F(t, 0xFF);
        }
        val = cur;
        if (c + 1 == nchunk) {
            const uint32_t row = (tile << 4) + (lane >> 2);
            if ((lane & 3) == 0 && row < rows) sg.out[row] = val;
            val = 0.0f;
        }
    };
    for (uint32_t u = 0; u < nunits; u += 2) {
        if (u + 1 < nunits) issue(u + 1, wB, sB);
        consume(u, wA, sA);
        if (u + 1 < nunits) {
            if (u + 2 < nunits) issue(u + 2, wA, sA);
            consume(u + 1, wB, sB);
        }
    }
}

template <int GS, bool NT>
__global__ __launch_bounds__(256) void q80_stream_sb(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LPG = GS / 16, GC = 1024 / GS, F = GC / 4;
    static_assert(F >= 1, "GS too large");
    const int n = (int)a.n, ng = n / GS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n16 = (n + 15) & ~15;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + n16);
    float *red = xs + ((ng + 3) & ~3);
    int *tab = reinterpret_cast<int *>(red + 8) + wid * 16 * GC;

    const Seg sg = a.seg[0];
    const uint32_t rows = sg.rows;
    const int nchunk = (n + 1023) >> 10;
    const uint32_t ntiles = (rows + 15) >> 4;
    const uint32_t nwaves = gridDim.x * 4, wave_g = blockIdx.x * 4 + wid;
    const uint32_t nunits = (wave_g < ntiles) ? ((ntiles - wave_g + nwaves - 1) / nwaves) * nchunk : 0;

    int4 wA[16];
    float sA[F];
    auto issue = [&](uint32_t u, int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + (u / nchunk) * nwaves;
        const int c = (int)(u % nchunk);
        const uint32_t row0 = tile << 4;
        const int col = (c << 10) + lane * 16;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t row = min(row0 + r, rows - 1);
            w[r] = (col < n) ? ld16<NT>(sg.w + (size_t)row * n + col) : make_int4(0, 0, 0, 0);
        }
        const uint32_t rr = min(row0 + (lane >> 2), rows - 1);
        const int gb = c * GC + (lane & 3) * F;
#pragma unroll
        for (int f = 0; f < F; f++) s[f] = (gb + f < ng) ? sg.ws[(size_t)rr * ng + gb + f] : 0.0f;
    };
    if (nunits) issue(0, wA, sA);

    stage_q80_wg<GS>(a, xq, xs, red);

    float val = 0.0f;
    auto consume = [&](uint32_t u, int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + (u / nchunk) * nwaves;
        const int c = (int)(u % nchunk);
        const int col = (c << 10) + lane * 16;
        const int4 xv = (col < n) ? *reinterpret_cast<const int4 *>(xq + col) : make_int4(0, 0, 0, 0);
        int iv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int t = __builtin_amdgcn_sdot4(w[r].x, xv.x, 0, false);
            t = __builtin_amdgcn_sdot4(w[r].y, xv.y, t, false);
            t = __builtin_amdgcn_sdot4(w[r].z, xv.z, t, false);
            t = __builtin_amdgcn_sdot4(w[r].w, xv.w, t, false);
            iv[r] = dpp_group_sum<LPG>(t);
        }
        if ((lane % LPG) == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) tab[r * GC + lane / LPG] = iv[r];
        }
        const int gb = c * GC + (lane & 3) * F;
        float p[F];
#pragma unroll
        for (int f = 0; f < F; f++) {
            const int v = tab[(lane >> 2) * GC + (lane & 3) * F + f];
            const float xsc = (gb + f < ng) ? xs[gb + f] : 0.0f;
            p[f] = ((float)v * s[f]) * xsc;
        }
        // ordered fold over the 4 lanes of a row: stage k adds lane k's products onto the running value
        float cur = val;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            float t = cur;
#pragma unroll
            for (int f = 0; f < F; f++) t += p[f];      // groups beyond ng contribute +0.0f exactly
            cur = (st == 0) ? DPP_F(t, 0x00) : (st == 1) ? DPP_F(t, 0x55) : (st == 2) ? DPP_F(t, 0xAA) : DPP_F(t, 0xFF);
        }
        val = cur;
        if (c + 1 == nchunk) {
            const uint32_t row = (tile << 4) + (lane >> 2);
            if ((lane & 3) == 0 && row < rows) sg.out[row] = val;
            val = 0.0f;
        }
    };
    for (uint32_t u = 0; u < nunits; u++) {          // single buffer: occupancy (4+ waves per SIMD) hides the latency
        if (u) issue(u, wA, sA);
        consume(u, wA, sA);
    }
}

static inline size_t stream_lds(uint32_t n, uint32_t gs) {
    const size_t n16 = (n + 15) & ~15u, ng4 = ((n / gs) + 3) & ~3u;
    return n16 + ng4 * 4 + 32 + 4 * 16 * (1024 / gs) * 4;
}

// ---------------------------------------------------------------------------------------------------
// SLAB kernel: per-layer matrices (a few MB).  A workgroup owns rows_per_wg consecutive rows x the whole
// row length; its work units (4 rows x one 1 KiB chunk [x 2 matrices for SwiGLU]) are dealt to its waves,
// every wave issues ALL its loads before anything else (one memory round trip per kernel), the group
// products ((float)iv*ws)*xs land in a workgroup LDS table and one thread per row folds them in the
// reference's ascending group order.
// ---------------------------------------------------------------------------------------------------
template <int GS, int UPW, bool NT>
__global__ __launch_bounds__(256) void q80_slab(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 4;
    constexpr int LPG = GS / 16, GC = 1024 / GS;
    constexpr int NS = (TR + LPG - 1) / LPG;           // rows a lane owns after the group reduction
    const int n = (int)a.n, ng = n / GS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, NW = blockDim.x >> 6;
    const int n16 = (n + 15) & ~15, ng4 = (ng + 3) & ~3;
    const int PITCH = (GC == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    const int RW = (int)a.rows_per_wg;
    const int nmat = (a.epi == EPI_SWIGLU) ? 2 : 1;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + n16);
    float *red = xs + ng4;
    float *P = red + 8;                                // [nmat][RW][PITCH]

    const int nchunk = (n + 1023) >> 10;
    const int units = (RW / TR) * nchunk * nmat;       // unit u -> (mat, tile, chunk), chunk fastest
    const uint32_t grow0 = blockIdx.x * RW;            // first global row of this workgroup
    uint32_t total_rows = 0;
    if (a.epi == EPI_SWIGLU) total_rows = a.seg[0].rows;
    else for (uint32_t s = 0; s < a.nseg; s++) total_rows += a.seg[s].rows;

    // global row -> (segment, local row); segments' row counts are multiples of TR
    auto locate = [&](uint32_t grow, int mat, const int8_t *&w, const float *&ws, uint32_t &lrow) {
        if (a.epi == EPI_SWIGLU) { w = a.seg[mat].w; ws = a.seg[mat].ws; lrow = grow; return; }
        uint32_t s = 0, r = grow;
        while (s + 1 < a.nseg && r >= a.seg[s].rows) { r -= a.seg[s].rows; s++; }
        w = a.seg[s].w; ws = a.seg[s].ws; lrow = r;
    };

    TSTAMP(0);
    int4 wv[UPW][TR];
    float sv[UPW][NS];
#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % (RW / TR), mat = u / (nchunk * (RW / TR));
            const uint32_t grow = grow0 + tl * TR;
            const int8_t *w; const float *ws; uint32_t lrow;
            locate(min(grow, total_rows - TR), mat, w, ws, lrow);
            const int col = (c << 10) + lane * 16;
#pragma unroll
            for (int r = 0; r < TR; r++) wv[k][r] = (col < n) ? ld16<NT>(w + (size_t)(lrow + r) * n + col) : make_int4(0, 0, 0, 0);
            const int g = c * GC + lane / LPG;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                sv[k][s] = (r < TR && g < ng) ? ws[(size_t)(lrow + r) * ng + g] : 0.0f;
            }
        }
    }
    // residual stream: old value, issued early
    float oldv = 0.0f;
    if (a.epi == EPI_RESID && tid < RW && grow0 + tid < total_rows) oldv = a.seg[0].out[grow0 + tid];

    TSTAMP(1);
    stage_q80_wg<GS>(a, xq, xs, red);
    TSTAMP(2);

#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % (RW / TR), mat = u / (nchunk * (RW / TR));
            const int col = (c << 10) + lane * 16;
            const int4 xv = (col < n) ? *reinterpret_cast<const int4 *>(xq + col) : make_int4(0, 0, 0, 0);
            int iv[TR];
#pragma unroll
            for (int r = 0; r < TR; r++) {
                int t = __builtin_amdgcn_sdot4(wv[k][r].x, xv.x, 0, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].y, xv.y, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].z, xv.z, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].w, xv.w, t, false);
                iv[r] = dpp_group_sum<LPG>(t);
            }
            const int g = c * GC + lane / LPG;
            const float xsc = (g < ng) ? xs[g] : 0.0f;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                int v = iv[0];
#pragma unroll
                for (int q = 1; q < TR; q++) v = (q == r) ? iv[q] : v;
                if (r < TR && g < ng) P[((size_t)mat * RW + tl * TR + r) * PITCH + g] = ((float)v * sv[k][s]) * xsc;
            }
        }
    }
    TSTAMP(3);
    __syncthreads();
    TSTAMP(4);
    if (tid < RW && grow0 + tid < total_rows) {
        float v0 = 0.0f, v1 = 0.0f;
        const float *p0 = P + (size_t)tid * PITCH;
        for (int g = 0; g < ng; g += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p0 + g);
            v0 += t.x; if (g + 1 < ng) v0 += t.y; if (g + 2 < ng) v0 += t.z; if (g + 3 < ng) v0 += t.w;
        }
        if (nmat == 2) {
            const float *p1 = P + ((size_t)RW + tid) * PITCH;
            for (int g = 0; g < ng; g += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p1 + g);
                v1 += t.x; if (g + 1 < ng) v1 += t.y; if (g + 2 < ng) v1 += t.z; if (g + 3 < ng) v1 += t.w;
            }
        }
        const uint32_t grow = grow0 + tid;
        if (a.epi == EPI_SWIGLU) {
            float h = v0; h *= (1.0f / (1.0f + expf(-h))); h *= v1;
            a.seg[0].out[grow] = h;
        } else if (a.epi == EPI_RESID) {
            a.seg[0].out[grow] = oldv + v0;
        } else {
            uint32_t s = 0, r = grow;
            while (s + 1 < a.nseg && r >= a.seg[s].rows) { r -= a.seg[s].rows; s++; }
            a.seg[s].out[r] = v0;
        }
    }
    TSTAMP(5);
}


// ---------------------------------------------------------------------------------------------------
// SLAB v2: same mapping as q80_slab, latency-restructured:
//   * every kernarg-derived address is computed branch-free (no dependent scalar-load chains);
//   * the activation (and norm weight) loads are issued first, then ALL weight / scale loads, then the
//     residual's old value -- one memory round trip for everything;
//   * rmsnorm + quantization work on registers (x is read from memory once).
// ---------------------------------------------------------------------------------------------------
template <int GS, int UPW, int NV, bool NT>
__global__ __launch_bounds__(256) void q80_slab2(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 4;
    constexpr int LPG = GS / 16, GC = 1024 / GS;
    constexpr int NS = (TR + LPG - 1) / LPG;
    const int n = (int)a.n, ng = n / GS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
    const int n16 = (n + 15) & ~15, ng4 = (ng + 3) & ~3;
    const int PITCH = (GC == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    const int RW = (int)a.rows_per_wg;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int nmat = swiglu ? 2 : 1;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + n16);
    float *red = xs + ng4;
    float *P = red + 8;
    TSTAMP(0);

    // ---- activation loads first (critical path) -----------------------------------------------------
    float4 xv[NV], nwv[NV];
    const bool pre = a.xq_in != nullptr;
    if (!pre) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int i = (tid + j * nthr) * 4;
            xv[j] = (i < n) ? *reinterpret_cast<const float4 *>(a.xin + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.norm_w) nwv[j] = (i < n) ? *reinterpret_cast<const float4 *>(a.norm_w + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- this workgroup's rows: one segment (segment row counts are multiples of rows_per_wg) ----------
    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.seg[0].rows, b1 = b0 + a.seg[1].rows;
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.seg[0].w : sel == 1 ? a.seg[1].w : a.seg[2].w;
    const float *ws0 = sel == 0 ? a.seg[0].ws : sel == 1 ? a.seg[1].ws : a.seg[2].ws;
    float *out0 = sel == 0 ? a.seg[0].out : sel == 1 ? a.seg[1].out : a.seg[2].out;
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const int nchunk = (n + 1023) >> 10;
    const int tiles = RW / TR;
    const int units = tiles * nchunk * nmat;

    int4 wv[UPW][TR];
    float sv[UPW][NS];
#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % tiles, mat = u / (nchunk * tiles);
            const int8_t *w = mat ? a.seg[1].w : w0;
            const float *ws = mat ? a.seg[1].ws : ws0;
            const uint32_t lrow = lrow0 + tl * TR;
            const int col = (c << 10) + lane * 16;
#pragma unroll
            for (int r = 0; r < TR; r++) wv[k][r] = (col < n) ? ld16<NT>(w + (size_t)(lrow + r) * n + col) : make_int4(0, 0, 0, 0);
            const int g = c * GC + lane / LPG;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                sv[k][s] = (r < TR && g < ng) ? ws[(size_t)(lrow + r) * ng + g] : 0.0f;
            }
        }
    }
    float oldv = 0.0f;
    if (a.epi == EPI_RESID && tid < RW) oldv = out0[lrow0 + tid];
    TSTAMP(1);

    // ---- rmsnorm + Q80 quantization from registers ---------------------------------------------------------
    if (pre) {
        for (int i = tid * 16; i < n; i += nthr * 16) *reinterpret_cast<int4 *>(xq + i) = *reinterpret_cast<const int4 *>(a.xq_in + i);
        for (int i = tid; i < ng; i += nthr) xs[i] = a.xs_in[i];
    } else {
        float ss = 1.0f;
        if (a.norm_w) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NV; j++) { acc += xv[j].x * xv[j].x; acc += xv[j].y * xv[j].y; acc += xv[j].z * xv[j].z; acc += xv[j].w * xv[j].w; }
            acc = dpp_wave_sum(acc);
            if (lane == 0) red[wid] = acc;
            __syncthreads();
            float t = 0.0f;
            for (int w = 0; w < NW; w++) t += red[w];
            t /= (float)n; t += 1e-5f;
            ss = 1.0f / sqrtf(t);
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int i = (tid + j * nthr) * 4;
            float4 v = xv[j];
            if (a.norm_w) { v.x = nwv[j].x * (ss * v.x); v.y = nwv[j].y * (ss * v.y); v.z = nwv[j].z * (ss * v.z); v.w = nwv[j].w * (ss * v.w); }
            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            m = dpp_group_max<GS / 4>(m);
            const float scale = m / 127.0f;
            if (i < n) {
                const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
                *reinterpret_cast<uint32_t *>(xq + i) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                if ((tid % (GS / 4)) == 0) xs[i / GS] = scale;
            }
        }
    }
    __syncthreads();
    TSTAMP(2);

#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % tiles, mat = u / (nchunk * tiles);
            const int col = (c << 10) + lane * 16;
            const int4 xvq = (col < n) ? *reinterpret_cast<const int4 *>(xq + col) : make_int4(0, 0, 0, 0);
            int iv[TR];
#pragma unroll
            for (int r = 0; r < TR; r++) {
                int t = __builtin_amdgcn_sdot4(wv[k][r].x, xvq.x, 0, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].y, xvq.y, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].z, xvq.z, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].w, xvq.w, t, false);
                iv[r] = dpp_group_sum<LPG>(t);
            }
            const int g = c * GC + lane / LPG;
            const float xsc = (g < ng) ? xs[g] : 0.0f;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                int v = iv[0];
#pragma unroll
                for (int q = 1; q < TR; q++) v = (q == r) ? iv[q] : v;
                if (r < TR && g < ng) P[((size_t)mat * RW + tl * TR + r) * PITCH + g] = ((float)v * sv[k][s]) * xsc;
            }
        }
    }
    TSTAMP(3);
    __syncthreads();
    TSTAMP(4);
    if (tid < RW) {
        float v0 = 0.0f, v1 = 0.0f;
        const float *p0 = P + (size_t)tid * PITCH;
        for (int g = 0; g < ng; g += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p0 + g);
            v0 += t.x; if (g + 1 < ng) v0 += t.y; if (g + 2 < ng) v0 += t.z; if (g + 3 < ng) v0 += t.w;
        }
        if (nmat == 2) {
            const float *p1 = P + ((size_t)RW + tid) * PITCH;
            for (int g = 0; g < ng; g += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p1 + g);
                v1 += t.x; if (g + 1 < ng) v1 += t.y; if (g + 2 < ng) v1 += t.z; if (g + 3 < ng) v1 += t.w;
            }
        }
        TSTAMP(5);
        float o = v0;
        if (swiglu) { float h = v0; h *= (1.0f / (1.0f + expf(-h))); h *= v1; o = h; }
        else if (a.epi == EPI_RESID) o = oldv + v0;
        out0[lrow0 + tid] = o;
    }
    TSTAMP(6);
}

template <int GS, int UPW, int NV, bool NT>
__global__ __launch_bounds__(256) void q80_slab2rep(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TR = 4;
    constexpr int LPG = GS / 16, GC = 1024 / GS;
    constexpr int NS = (TR + LPG - 1) / LPG;
    const int n = (int)a.n, ng = n / GS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
    const int n16 = (n + 15) & ~15, ng4 = (ng + 3) & ~3;
    const int PITCH = (GC == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    const int RW = (int)a.rows_per_wg;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int nmat = swiglu ? 2 : 1;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + n16);
    float *red = xs + ng4;
    float *P = red + 8;
    if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + 7] = __builtin_readcyclecounter();
    for (int rep = 0; rep < 2; rep++) {
    TSTAMP(rep * 8 + 0);

    // ---- activation loads first (critical path) -----------------------------------------------------
    float4 xv[NV], nwv[NV];
    const bool pre = a.xq_in != nullptr;
    if (!pre) {
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int i = (tid + j * nthr) * 4;
            xv[j] = (i < n) ? *reinterpret_cast<const float4 *>(a.xin + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.norm_w) nwv[j] = (i < n) ? *reinterpret_cast<const float4 *>(a.norm_w + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- this workgroup's rows: one segment (segment row counts are multiples of rows_per_wg) ----------
    const uint32_t grow0 = blockIdx.x * RW;
    const uint32_t b0 = a.seg[0].rows, b1 = b0 + a.seg[1].rows;
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.seg[0].w : sel == 1 ? a.seg[1].w : a.seg[2].w;
    const float *ws0 = sel == 0 ? a.seg[0].ws : sel == 1 ? a.seg[1].ws : a.seg[2].ws;
    float *out0 = sel == 0 ? a.seg[0].out : sel == 1 ? a.seg[1].out : a.seg[2].out;
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    const int nchunk = (n + 1023) >> 10;
    const int tiles = RW / TR;
    const int units = tiles * nchunk * nmat;

    int4 wv[UPW][TR];
    float sv[UPW][NS];
#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % tiles, mat = u / (nchunk * tiles);
            const int8_t *w = mat ? a.seg[1].w : w0;
            const float *ws = mat ? a.seg[1].ws : ws0;
            const uint32_t lrow = lrow0 + tl * TR;
            const int col = (c << 10) + lane * 16;
#pragma unroll
            for (int r = 0; r < TR; r++) wv[k][r] = (col < n) ? ld16<NT>(w + (size_t)(lrow + r) * n + col) : make_int4(0, 0, 0, 0);
            const int g = c * GC + lane / LPG;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                sv[k][s] = (r < TR && g < ng) ? ws[(size_t)(lrow + r) * ng + g] : 0.0f;
            }
        }
    }
    float oldv = 0.0f;
    if (a.epi == EPI_RESID && tid < RW) oldv = out0[lrow0 + tid];
    TSTAMP(rep * 8 + 1);

    // ---- rmsnorm + Q80 quantization from registers ---------------------------------------------------------
    if (pre) {
        for (int i = tid * 16; i < n; i += nthr * 16) *reinterpret_cast<int4 *>(xq + i) = *reinterpret_cast<const int4 *>(a.xq_in + i);
        for (int i = tid; i < ng; i += nthr) xs[i] = a.xs_in[i];
    } else {
        float ss = 1.0f;
        if (a.norm_w) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < NV; j++) { acc += xv[j].x * xv[j].x; acc += xv[j].y * xv[j].y; acc += xv[j].z * xv[j].z; acc += xv[j].w * xv[j].w; }
            acc = dpp_wave_sum(acc);
            if (lane == 0) red[wid] = acc;
            __syncthreads();
            float t = 0.0f;
            for (int w = 0; w < NW; w++) t += red[w];
            t /= (float)n; t += 1e-5f;
            ss = 1.0f / sqrtf(t);
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int i = (tid + j * nthr) * 4;
            float4 v = xv[j];
            if (a.norm_w) { v.x = nwv[j].x * (ss * v.x); v.y = nwv[j].y * (ss * v.y); v.z = nwv[j].z * (ss * v.z); v.w = nwv[j].w * (ss * v.w); }
            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            m = dpp_group_max<GS / 4>(m);
            const float scale = m / 127.0f;
            if (i < n) {
                const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
                *reinterpret_cast<uint32_t *>(xq + i) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                if ((tid % (GS / 4)) == 0) xs[i / GS] = scale;
            }
        }
    }
    __syncthreads();
    TSTAMP(rep * 8 + 2);

#pragma unroll
    for (int k = 0; k < UPW; k++) {
        const int u = wid + k * NW;
        if (u < units) {
            const int c = u % nchunk, tl = (u / nchunk) % tiles, mat = u / (nchunk * tiles);
            const int col = (c << 10) + lane * 16;
            const int4 xvq = (col < n) ? *reinterpret_cast<const int4 *>(xq + col) : make_int4(0, 0, 0, 0);
            int iv[TR];
#pragma unroll
            for (int r = 0; r < TR; r++) {
                int t = __builtin_amdgcn_sdot4(wv[k][r].x, xvq.x, 0, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].y, xvq.y, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].z, xvq.z, t, false);
                t = __builtin_amdgcn_sdot4(wv[k][r].w, xvq.w, t, false);
                iv[r] = dpp_group_sum<LPG>(t);
            }
            const int g = c * GC + lane / LPG;
            const float xsc = (g < ng) ? xs[g] : 0.0f;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int r = (lane % LPG) + s * LPG;
                int v = iv[0];
#pragma unroll
                for (int q = 1; q < TR; q++) v = (q == r) ? iv[q] : v;
                if (r < TR && g < ng) P[((size_t)mat * RW + tl * TR + r) * PITCH + g] = ((float)v * sv[k][s]) * xsc;
            }
        }
    }
    TSTAMP(rep * 8 + 3);
    __syncthreads();
    TSTAMP(rep * 8 + 4);
    if (tid < RW) {
        float v0 = 0.0f, v1 = 0.0f;
        const float *p0 = P + (size_t)tid * PITCH;
        for (int g = 0; g < ng; g += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(p0 + g);
            v0 += t.x; if (g + 1 < ng) v0 += t.y; if (g + 2 < ng) v0 += t.z; if (g + 3 < ng) v0 += t.w;
        }
        if (nmat == 2) {
            const float *p1 = P + ((size_t)RW + tid) * PITCH;
            for (int g = 0; g < ng; g += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(p1 + g);
                v1 += t.x; if (g + 1 < ng) v1 += t.y; if (g + 2 < ng) v1 += t.z; if (g + 3 < ng) v1 += t.w;
            }
        }
        TSTAMP(rep * 8 + 5);
        float o = v0;
        if (swiglu) { float h = v0; h *= (1.0f / (1.0f + expf(-h))); h *= v1; o = h; }
        else if (a.epi == EPI_RESID) o = oldv + v0;
        out0[lrow0 + tid] = o;
    }
    TSTAMP(rep * 8 + 6);
    __syncthreads();
    }
    if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + 15] = __builtin_readcyclecounter();
}


static inline size_t slab_lds(uint32_t n, uint32_t gs, uint32_t rw, uint32_t nmat) {
    const size_t n16 = (n + 15) & ~15u, ng = n / gs, ng4 = (ng + 3) & ~3u;
    const size_t pitch = (1024 / gs == 16) ? (((ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    return n16 + ng4 * 4 + 32 + (size_t)nmat * rw * pitch * 4;
}

}  // namespace k
