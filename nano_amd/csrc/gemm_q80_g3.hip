// gemm_q80_g3.hip -- G3: the batched Q80 (W8A8) GEMM for matrices with MANY rows (classifier, W1|W3, QKV), 8..64 tokens per
// weight read.  Same arithmetic as every other Q80 kernel of this library (exact int32 group sums on the matrix cores,
// one v_mfma_i32_16x16x64_i8 per 64 bytes of a quantization group; products ((float)ival * ws) * xs; ascending-group
// accumulation: bit-identical to the GEMV path and to the reference's matmul_quant, infer/infer.c:654-679) -- what
// changes against gemm_q80.hip's G2 is who owns what:
//   * PERSISTENT workgroups: each walks tiles t = blockIdx, blockIdx + grid, ... and its load queue runs AHEAD ACROSS
//     tile boundaries (two 64 KB steps in flight per workgroup), so the start-up round trip is paid once per workgroup,
//     not once per 40 KB tile (a Qwen3-4B W1|W3 tile has only 5 passes: G2 spent most of a tile's life waiting for its
//     first bytes);
//   * a tile = RT row tiles of 16 rows (x 2 matrices for SwiGLU) x ALL tokens; every step (512 bytes of row length) the
//     eight waves bring the tile's weight pieces, the step's activation fragments (already in MFMA B order,
//     quant_rows_frag_kernel) and the step's scales into LDS -- the activations ONCE per workgroup and step instead of
//     once per 16-row tile: L2 traffic per weight byte drops from TT (G2) to TT / RT;
//   * a wave owns (row tile, token tile) pairs for the whole row length and keeps their running values in REGISTERS:
//     no product table, no fold stage, one barrier per step; an A fragment read from LDS feeds up to two token tiles
//     (or a B fragment both SwiGLU matrices).
// LDS: weight stage 2 x WR x 528 B (row pitch 528: conflict-free ds_read_b128 of the A fragments), activation stage
// 2 x TT x 8 KiB (fragment order: linear, conflict-free), scales 2 x (WR x GPP + TT x GPP x 16) floats.
#include <type_traits>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct G3Dev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, epi, nb, npass, ntiles;
    const int8_t *xf; const float *xsf; const uint32_t *pos;
};

constexpr uint32_t G3_PK = 512, G3_PITCH = 528;

template <int GS, bool SW, int RT, int TT>
__global__ __launch_bounds__(512) void gemm_q80_g3_kernel(const G3Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FR = GS / 64;                                        // MFMAs per quantization group
    constexpr uint32_t GPP = G3_PK / (uint32_t)GS;                     // groups per step
    constexpr uint32_t nmat = SW ? 2u : 1u, TR = 16u * RT;             // rows of the tile (per matrix)
    constexpr uint32_t WR = TR * nmat;                                 // weight rows staged per step
    constexpr uint32_t NWL = WR / 16u;                                 // 1 KiB weight pieces per wave and step (8 waves)
    constexpr uint32_t WPR = 8u / RT;                                  // waves per row tile
    constexpr uint32_t NBT = (TT + WPR - 1u) / WPR;                    // token tiles per wave
    static_assert(WR >= 16 && WR <= 64 && (8 % RT) == 0 && GS % 64 == 0, "tile shape");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n = a.n, ng = a.ng, npass = a.npass;
    const uint32_t m = lane & 15u, kq = lane >> 4;

    // LDS
    constexpr uint32_t WST = WR * G3_PITCH, XST = TT * 8192u, WSC = WR * GPP, XSC = TT * GPP * 16u;
    int8_t *wst = reinterpret_cast<int8_t *>(smem);                    // [2][WR][528]
    int8_t *xst = wst + 2u * WST;                                      // [2][TT][8][1024]
    float *wsl = reinterpret_cast<float *>(xst + 2u * XST);            // [2][WR][GPP]
    float *xsl = wsl + 2u * WSC;                                       // [2][TT][GPP][16]

    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(a.xf, (uint32_t)TT * ng * (uint32_t)FR * 1024u);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(a.xsf, (uint32_t)TT * ng * 64u);
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];

    // tile -> segment (a tile lies inside one segment: checked on the host)
    auto seg_of = [&](uint32_t grow0, const int8_t *&w0, const float *&ws0, float *&out0, uint32_t &rows0, uint32_t &obs, uint32_t &ops, uint32_t &lrow0) {
        const int sel = SW ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
        w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
        ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
        out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
        rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
        obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
        ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
        lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);
    };

    // ---- the load queue: step s = (tile li of this workgroup, pass lp); two steps in flight in registers --------------------
    const uint32_t mytiles = a.ntiles > blockIdx.x ? (a.ntiles - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
    const uint32_t total = mytiles * npass;
    int4 rw[2][NWL], rx[2][TT];
    float rws[2], rxs_[2];
    uint32_t li = 0, lp = 0;                                           // load cursor
    auto issue = [&](int slot) {
        const bool on = li < mytiles;
        const uint32_t grow0 = (blockIdx.x + li * gridDim.x) * TR;
        const int8_t *w0; const float *ws0; float *o0; uint32_t rows0, obs, ops, lrow0;
        seg_of(on ? grow0 : 0u, w0, ws0, o0, rows0, obs, ops, lrow0);
        const __amdgpu_buffer_rsrc_t r0 = mkrsrc(w0, on ? rows0 * n : 0u), r1 = mkrsrc(SW ? a.w[1] : nullptr, (SW && on) ? rows0 * n : 0u);
        const __amdgpu_buffer_rsrc_t s0 = mkrsrc(ws0, on ? rows0 * ng * 4u : 0u), s1 = mkrsrc(SW ? a.ws[1] : nullptr, (SW && on) ? rows0 * ng * 4u : 0u);
        const uint32_t col = lp * G3_PK + (lane & 31u) * 16u;
#pragma unroll
        for (uint32_t j = 0; j < NWL; j++) {                           // piece q = rows 2q, 2q+1 of the staged block
            const uint32_t rw0 = (wid * NWL + j) * 2u, mt = rw0 / TR, lr = rw0 % TR + (lane >> 5);     // mt: wave-uniform (TR is even)
            const uint32_t off = col < n ? (lrow0 + lr) * n + col : OOB;
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128((SW && mt) ? r1 : r0, (int)off, 0, 2);
            rw[slot][j] = make_int4(v.x, v.y, v.z, v.w);
        }
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)TT; j++) {                  // activation fragment blocks of the step: (token tile, 1 KiB block)
            const uint32_t q = wid * (uint32_t)TT + j, tt = q >> 3, blk = q & 7u, g = lp * GPP + blk / (uint32_t)FR;
            const uint32_t off = (on && g < ng) ? ((tt * ng + g) * (uint32_t)FR + blk % (uint32_t)FR) * 1024u + lane * 16u : OOB;
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)off, 0, 0);
            rx[slot][j] = make_int4(v.x, v.y, v.z, v.w);
        }
        {   // scales of the step: weight scales [WR][GPP] and activation scales [TT][GPP][16], one float per thread each
            const uint32_t rwi = tid / GPP, gl = tid % GPP, lr = rwi % TR, g = lp * GPP + gl;
            const uint32_t mt = (wid * (64u / GPP)) / TR;                                               // wave-uniform: a wave's 64 / GPP rows lie in one matrix
            rws[slot] = bload_f((SW && mt) ? s1 : s0, (tid < WSC && g < ng) ? ((lrow0 + lr) * ng + g) * 4u : OOB);
            const uint32_t tt = tid / (GPP * 16u), rem = tid % (GPP * 16u), gx = lp * GPP + rem / 16u;
            rxs_[slot] = bload_f(rxs, (on && tid < XSC && gx < ng) ? ((tt * ng + gx) * 16u + (rem & 15u)) * 4u : OOB);
        }
        if (++lp == npass) { lp = 0; li++; }
    };
    auto stage_write = [&](int slot, int buf) {
#pragma unroll
        for (uint32_t j = 0; j < NWL; j++) {
            const uint32_t rwi = (wid * NWL + j) * 2u + (lane >> 5);
            *reinterpret_cast<int4 *>(wst + (size_t)buf * WST + (size_t)rwi * G3_PITCH + (lane & 31u) * 16u) = rw[slot][j];
        }
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)TT; j++) {
            const uint32_t q = wid * (uint32_t)TT + j;
            *reinterpret_cast<int4 *>(xst + (size_t)buf * XST + (size_t)q * 1024u + lane * 16u) = rx[slot][j];
        }
        if (tid < WSC) wsl[buf * WSC + tid] = rws[slot];
        if (tid < XSC) xsl[buf * XSC + tid] = rxs_[slot];
    };

    // ---- this wave's (row tile, token tile) pairs ---------------------------------------------------------------------------
    const uint32_t rt = wid / WPR, sub = wid % WPR;
    float acc[SW ? 2 : 1][NBT][4], oldv[NBT][4];
#pragma unroll
    for (uint32_t bt = 0; bt < NBT; bt++)
#pragma unroll
        for (int i = 0; i < 4; i++) { acc[0][bt][i] = 0.0f; if (SW) acc[SW ? 1 : 0][bt][i] = 0.0f; oldv[bt][i] = 0.0f; }

    uint32_t ci = 0, cp = 0;                                           // compute cursor
    auto compute = [&](int buf) {
        const int8_t *wb = wst + (size_t)buf * WST;
        const int8_t *xb = xst + (size_t)buf * XST;
        const float *wsb = wsl + buf * WSC, *xsb = xsl + buf * XSC;
        const uint32_t gcnt = ng - cp * GPP < GPP ? ng - cp * GPP : GPP;
#pragma unroll 2
        for (uint32_t gl = 0; gl < GPP; gl++) {
            i32x4 fa[SW ? 2 : 1][FR];
#pragma unroll
            for (int mt = 0; mt < (int)nmat; mt++)
#pragma unroll
                for (int ks = 0; ks < FR; ks++)
                    fa[mt][ks] = *reinterpret_cast<const i32x4 *>(wb + ((size_t)mt * TR + rt * 16u + m) * G3_PITCH + gl * (uint32_t)GS + (uint32_t)ks * 64u + kq * 16u);
            float wsv[SW ? 2 : 1][4];
#pragma unroll
            for (int mt = 0; mt < (int)nmat; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++) wsv[mt][i] = wsb[((uint32_t)mt * TR + rt * 16u + kq * 4u + i) * GPP + gl];
#pragma unroll
            for (uint32_t bt = 0; bt < NBT; bt++) {
                const uint32_t tt = sub + bt * WPR;
                if (TT % WPR == 0 || tt < (uint32_t)TT) {                              // wave-uniform
                    const float xsc = xsb[(tt * GPP + gl) * 16u + m];
#pragma unroll
                    for (int mt = 0; mt < (int)nmat; mt++) {
                        v4i c = {0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < FR; ks++) {
                            const i32x4 fbv = *reinterpret_cast<const i32x4 *>(xb + ((size_t)tt * 8u + gl * (uint32_t)FR + (uint32_t)ks) * 1024u + lane * 16u);
                            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[mt][ks], fbv, c, 0, 0, 0);
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {                                                         // infer.c:672, ascending groups
                            const float p = ((float)c[i] * wsv[mt][i]) * xsc;
                            acc[mt][bt][i] = gl < gcnt ? acc[mt][bt][i] + p : acc[mt][bt][i];
                        }
                    }
                }
            }
        }
    };
    auto tile_begin = [&]() {                                          // residual epilogue: the old values go into the load queue early
        if (a.epi == GEMV_EPI_RESID) {
            const uint32_t grow0 = (blockIdx.x + ci * gridDim.x) * TR;
            const int8_t *w0; const float *ws0; float *o0; uint32_t rows0, obs, ops, lrow0;
            seg_of(grow0, w0, ws0, o0, rows0, obs, ops, lrow0);
#pragma unroll
            for (uint32_t bt = 0; bt < NBT; bt++) {
                const uint32_t tt = sub + bt * WPR, t = tt * 16u + m;
                if (tt < (uint32_t)TT && t < a.nb) {
                    const float *o = o0 + (size_t)t * obs + (ops ? (size_t)a.pos[t] * ops : 0) + lrow0 + rt * 16u + kq * 4u;
#pragma unroll
                    for (int i = 0; i < 4; i++) if (lrow0 + rt * 16u + kq * 4u + i < rows0) oldv[bt][i] = o[i];
                }
            }
        }
    };
    auto tile_end = [&]() {
        const uint32_t grow0 = (blockIdx.x + ci * gridDim.x) * TR;
        const int8_t *w0; const float *ws0; float *o0; uint32_t rows0, obs, ops, lrow0;
        seg_of(grow0, w0, ws0, o0, rows0, obs, ops, lrow0);
#pragma unroll
        for (uint32_t bt = 0; bt < NBT; bt++) {
            const uint32_t tt = sub + bt * WPR, t = tt * 16u + m;
            if (tt < (uint32_t)TT && t < a.nb) {
                float *o = o0 + (size_t)t * obs + (ops ? (size_t)a.pos[t] * ops : 0) + lrow0 + rt * 16u + kq * 4u;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (lrow0 + rt * 16u + kq * 4u + i < rows0) o[i] = finish_epi(a.epi, acc[0][bt][i], acc[SW ? 1 : 0][bt][i], oldv[bt][i]);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) { acc[0][bt][i] = 0.0f; if (SW) acc[SW ? 1 : 0][bt][i] = 0.0f; }
        }
    };

    if (total == 0) return;
    issue(0); issue(1);
    stage_write(0, 0); issue(0);                                       // step 0 -> stage 0; slot 0 now carries step 2
    __syncthreads();
    for (uint32_t s = 0;; s += 2) {
        // even step s: compute from stage 0; step s+1 (slot 1) -> stage 1; slot 1 re-issued with step s+3
        if (cp == 0) tile_begin();
        stage_write(1, 1); issue(1);
        compute(0);
        if (++cp == npass) { tile_end(); cp = 0; ci++; }
        if (s + 1 >= total) break;
        __syncthreads();
        // odd step s+1: compute from stage 1; step s+2 (slot 0) -> stage 0; slot 0 re-issued with step s+4
        if (cp == 0) tile_begin();
        stage_write(0, 0); issue(0);
        compute(1);
        if (++cp == npass) { tile_end(); cp = 0; ci++; }
        if (s + 2 >= total) break;
        __syncthreads();
    }
}

template <int GS, bool SW, int RT, int TT>
static hipError_t launch_t(const G3Dev &d, uint32_t max_wg, hipStream_t st) {
    constexpr uint32_t GPP = G3_PK / (uint32_t)GS, WR = 16u * RT * (SW ? 2u : 1u);
    const size_t lds = 2 * (size_t)WR * G3_PITCH + 2 * (size_t)TT * 8192 + 2 * ((size_t)WR * GPP + (size_t)TT * GPP * 16) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = &gemm_q80_g3_kernel<GS, SW, RT, TT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const uint32_t per_cu = lds > 80 * 1024 ? 1u : 2u;               // co-resident workgroups per CU (LDS; 512 threads each)
    uint32_t grid = max_wg / 8u * per_cu;                             // max_wg = 8 x CUs (backend.hip)
    if (grid > d.ntiles) grid = d.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, d);
    return hipGetLastError();
}

static uint32_t total_rows3(const GemvArgs &a) {
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    return rows;
}
// rows of a tile: the largest of 64 / 32 / 16 (32 / 16 for SwiGLU) that keeps >= 192 tiles and divides every segment
static uint32_t pick_rt(const GemvArgs &a) {
    const uint32_t rows = total_rows3(a);
    for (uint32_t rt = a.epi == GEMV_EPI_SWIGLU ? 2u : 4u; rt >= 1u; rt >>= 1) {
        bool ok = rows % (16u * rt) == 0 && rows / (16u * rt) >= 192u;
        if (a.epi != GEMV_EPI_SWIGLU) for (uint32_t s = 0; s < a.nseg; s++) ok = ok && a.seg[s].rows % (16u * rt) == 0;
        if (ok) return rt;
    }
    return 0;
}

template <int GS>
static hipError_t launch_gs(const GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    G3Dev d{};
    for (int i = 0; i < 3; i++) {
        const bool live = i < (int)a.nseg;
        d.w[i] = live ? reinterpret_cast<const int8_t *>(a.seg[i].w) : nullptr;
        d.ws[i] = live ? a.seg[i].ws : nullptr;
        d.out[i] = live ? a.seg[i].out : nullptr;
        d.rows[i] = live ? a.seg[i].rows : 0;
        d.out_bstride[i] = live ? a.seg[i].out_bstride : 0;
        d.out_pstride[i] = live ? a.seg[i].out_pstride : 0;
    }
    if (a.epi == GEMV_EPI_SWIGLU) { d.rows[1] = 0; d.rows[2] = 0; }
    d.n = a.n; d.ng = a.n / a.gs; d.epi = a.epi; d.nb = a.nb; d.npass = (a.n + G3_PK - 1) / G3_PK;
    d.xf = a.xq_in; d.xsf = a.xs_in; d.pos = a.pos;
    const uint32_t rt = pick_rt(a), tt = (a.nb + 15) / 16;
    if (!rt) return hipErrorInvalidValue;
    d.ntiles = total_rows3(a) / (16u * rt);
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
#define G3_GO(RT_, TT_) do { if (sw) { if constexpr (RT_ <= 2) return launch_t<GS, true, RT_, TT_>(d, max_wg, st); } else return launch_t<GS, false, RT_, TT_>(d, max_wg, st); } while (0)
#define G3_TT(RT_) do { if (tt <= 1) G3_GO(RT_, 1); else if (tt <= 2) G3_GO(RT_, 2); else G3_GO(RT_, 4); } while (0)
    if (rt == 4) G3_TT(4);
    if (rt == 2) G3_TT(2);
    G3_TT(1);
#undef G3_TT
#undef G3_GO
    return hipErrorInvalidValue;
}

}  // namespace

// does G3 take this launch?  (many rows; group size 64 or 128; what G2 takes otherwise)
bool gemm_q80_g3_supports(const GemvArgs &a) {
    if (!gemm_q80_g2_supports(a) || !(a.gs == 64 || a.gs == 128)) return false;
    return pick_rt(a) != 0;
}
// a.xq_in / a.xs_in: the activations in fragment order (launch_quant_rows_frag); max_wg = 8 x CUs
hipError_t launch_gemm_q80_g3(const GemvArgs &a, uint32_t max_wg, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g3_supports(a)) return hipErrorInvalidValue;
    return a.gs == 64 ? launch_gs<64>(a, max_wg, st) : launch_gs<128>(a, max_wg, st);
}

}  // namespace nano
