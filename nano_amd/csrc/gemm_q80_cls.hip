// gemm_q80_cls.hip -- GC: batched Q80 (W8A8) GEMM for TALL matrices with short rows (the classifier: vocab x n_embd), 2..64
// tokens per weight read, built like the STREAM GEMV of gemv_q80_impl.h: persistent waves sweep the matrix linearly, a wave owns
// whole 16-row tiles (the full row length), so the reference's ascending group order (infer/infer.c:668-674) is a running value
// in its own registers -- no chain between waves, no LDS product table, no barrier after the prologue.
//
//   * the activations (quant_rows_frag_kernel's MFMA B-fragment order, gemm_q80.hip) of the first LT token tiles are staged in
//     LDS ONCE per workgroup and read from there by every wave for every row tile (round 3's GEMM gave each wave its own row tile and lets
//     it fetch the fragments from L2 again: for a 151 936-row classifier that is as many L2 bytes as weight bytes per token
//     tile -- 131 us at 8 tokens and 205..290 us at 64 where the weights alone stream in ~65 us); token tiles that do not fit
//     LDS come from L2 as before;
//   * a wave keeps the next half chunk's 16 x 512 B of weights in flight (registers) while it multiplies the current one out of
//     its transposition buffer (wave-private LDS, pitch 528: conflict-free ds_read_b128 of the MFMA A fragments);
//   * one v_mfma_i32_16x16x64_i8 per (group, token tile) -> exact int32 group sums -> ((float)ival * ws) * xs (infer.c:672) added
//     to the running value of (row, token) in ascending group order: bit-identical to the GEMV path and to the reference.
// Takes: group size 64, one STORE segment of >= 16384 rows, group count a multiple of 8 or 4.  (backend.hip routes the classifier
// of batched steps here.)
#include <atomic>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct GCDev {
    const int8_t *w; const float *ws; float *out;
    uint32_t rows, out_bstride, n, ng, nb, nhc, ntiles, tt, lt, nwaves;     // lt: token tiles staged in LDS
    const int8_t *xf; const float *xsf;
};

constexpr uint32_t GC_PITCH = 528, GC_WBUF = 16 * GC_PITCH;             // transposition buffer of one wave: 16 rows x 512 B
constexpr uint32_t GC_LDS_WAVE = GC_WBUF + 512;                         // + weight scales [8 groups][16 rows]

template <int TT>
__global__ __launch_bounds__(512, 2) void gemm_q80_cls_kernel(const GCDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    karg_touch(a.out); karg_touch(a.out_bstride); karg_touch(a.rows); karg_touch(a.xf); karg_touch(a.xsf);      // (late-read arguments with the first batch)
    const uint32_t nw = blockDim.x >> 6;
    const uint32_t n = a.n, ng = a.ng, nhc = a.nhc, tt = a.tt, lt = a.lt;
    const uint32_t m = lane & 15u, kq = lane >> 4;

    // LDS: xfl[lt][ng][1024] int8 | xsl[lt][ng][16] float | per wave: wbuf[16][528] int8, wsl[8][16] float
    int8_t *xfl = reinterpret_cast<int8_t *>(smem);
    float *xsl = reinterpret_cast<float *>(smem + (size_t)lt * ng * 1024u);
    int8_t *wbuf = reinterpret_cast<int8_t *>(smem) + (size_t)lt * ng * (1024u + 64u) + (size_t)wid * GC_LDS_WAVE;
    float *wsl = reinterpret_cast<float *>(wbuf + GC_WBUF);

    const __amdgpu_buffer_rsrc_t rw = mkrsrc(a.w, a.rows * n);
    const __amdgpu_buffer_rsrc_t rs = mkrsrc(a.ws, a.rows * ng * 4u);
    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(a.xf, tt * ng * 1024u);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(a.xsf, tt * ng * 64u);

    // this wave's tiles: wave_g, wave_g + nwaves, ... -- the chip sweeps the matrix linearly
    const uint32_t wave_g = blockIdx.x * nw + wid;
    const uint32_t my_tiles = wave_g < a.ntiles ? (a.ntiles - wave_g + a.nwaves - 1u) / a.nwaves : 0u;
    const uint32_t units = my_tiles * nhc;                              // (tile, half chunk), half chunk fastest

    // ---- weight pieces of a unit: 16 rows x 512 B, two rows per load instruction (lane l: row 2r + l/32) -----------------------
    int4 wA[8];
    const uint32_t wrow = lane >> 5, wcol = (lane & 31u) * 16u;
    uint32_t it = 0, ih = 0;                                            // cursor of the NEXT unit to issue
    auto issue_w = [&]() {
        const uint32_t row0 = (wave_g + it * a.nwaves) * 16u;
        const uint32_t col = ih * 512u + wcol;
        const uint32_t base = (it < my_tiles && col < n) ? (row0 + wrow) * n + col : OOB;       // rows beyond the matrix: out of range -> 0
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)(base == OOB ? OOB : base + (uint32_t)(2 * r) * n), 0, 2);
            wA[r] = make_int4(v.x, v.y, v.z, v.w);
        }
        if (++ih == nhc) { ih = 0; it++; }
    };
    issue_w();

    // ---- prologue: the first lt token tiles' fragments and scales into LDS (every wave reads them for every tile) ---------------
    {
        const uint32_t nvec = lt * ng * 64u;                            // 16-byte pieces
        for (uint32_t i = tid; i < nvec; i += blockDim.x) {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(i * 16u), 0, 0);
            *reinterpret_cast<i32x4 *>(xfl + (size_t)i * 16u) = v;
        }
        const uint32_t nsc = lt * ng * 4u;                              // float4 pieces of the scales
        for (uint32_t i = tid; i < nsc; i += blockDim.x) *reinterpret_cast<float4 *>(xsl + (size_t)i * 4u) = bload_f4(rxs, i * 16u);
    }
    __syncthreads();                                                   // the only barrier

    float acc[TT][4];
    uint32_t ct = 0, ch = 0;                                            // cursor of the unit being consumed
    for (uint32_t u = 0; u < units; u++) {
        const uint32_t row0 = (wave_g + ct * a.nwaves) * 16u, g0 = ch * 8u;
        if (ch == 0) {
#pragma unroll
            for (int t = 0; t < TT; t++) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
        }
        // 1. the unit's weight pieces: registers -> the transposition buffer; its scales (lanes 0..31: row l/2, groups g0 + 4 (l%2) .. +3)
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<int4 *>(wbuf + (size_t)(2 * r + wrow) * GC_PITCH + wcol) = wA[r];
        const uint32_t sg = g0 + (lane & 1u) * 4u;
        const float4 wsv = bload_f4(rs, (lane < 32u && sg < ng) ? ((row0 + (lane >> 1)) * ng + sg) * 4u : OOB);     // rows beyond the matrix: 0
        // 2. fragments of the token tiles that are not staged (tiles >= lt), all of this half chunk at once
        i32x4 fbg[TT > 2 ? 2 : 1][8]; float xsg[TT > 2 ? 2 : 1][8];
        if constexpr (TT > 2) if (lt < tt) {                           // (uniform: nothing to fetch when every tile is staged)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const uint32_t t = lt + (uint32_t)e;
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    fbg[e][j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)((g0 + j < ng && t < tt) ? lane * 16u : OOB), (int)((t * ng + g0 + j) * 1024u), 0);
                    xsg[e][j] = bload_f(rxs, (g0 + j < ng && t < tt) ? ((t * ng + g0 + j) * 16u + m) * 4u : OOB);
                }
            }
        }
        issue_w();                                                      // the next unit's weights (behind the fragments in the load queue)
        if (lane < 32u) {
            const uint32_t r = lane >> 1, gq = (lane & 1u) * 4u;
            wsl[(gq + 0u) * 16u + r] = wsv.x; wsl[(gq + 1u) * 16u + r] = wsv.y; wsl[(gq + 2u) * 16u + r] = wsv.z; wsl[(gq + 3u) * 16u + r] = wsv.w;
        }
        // 3. 8 groups x token tiles: A fragment from the transposition buffer, B fragment from LDS (staged) or registers, MFMA,
        //    product, running value += in ascending group order (infer.c:668-674)
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (g0 + j < ng) {                                          // (uniform; the last half chunk may hold 4 groups)
                const i32x4 fa = *reinterpret_cast<const i32x4 *>(wbuf + (size_t)m * GC_PITCH + j * 64u + kq * 16u);
                const float4 wv = *reinterpret_cast<const float4 *>(wsl + j * 16u + kq * 4u);
#pragma unroll
                for (int t = 0; t < TT; t++) {
                    if ((uint32_t)t < tt) {
                        i32x4 fb; float xsc;
                        if ((uint32_t)t < lt) {
                            fb = *reinterpret_cast<const i32x4 *>(xfl + ((size_t)((uint32_t)t * ng + g0 + j) * 64u + lane) * 16u);
                            xsc = xsl[((uint32_t)t * ng + g0 + j) * 16u + m];
                        } else if constexpr (TT > 2) {
                            fb = fbg[(t - 2) & 1][j]; xsc = xsg[(t - 2) & 1][j];       // (lt == 2 whenever tiles stay unstaged)
                        } else { fb = i32x4{0, 0, 0, 0}; xsc = 0.0f; }
                        const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, v4i{0, 0, 0, 0}, 0, 0, 0);
                        acc[t][0] += ((float)cv[0] * wv.x) * xsc; acc[t][1] += ((float)cv[1] * wv.y) * xsc;       // infer.c:672
                        acc[t][2] += ((float)cv[2] * wv.z) * xsc; acc[t][3] += ((float)cv[3] * wv.w) * xsc;
                    }
                }
            }
        }
        // 4. after the tile's last half chunk: store the 16 rows x 16 tokens of every token tile
        if (ch + 1u == nhc) {
#pragma unroll
            for (int t = 0; t < TT; t++) {
                const uint32_t tok = (uint32_t)t * 16u + m, row = row0 + kq * 4u;
                if ((uint32_t)t < tt && tok < a.nb) {
                    float *o = a.out + (size_t)tok * a.out_bstride + row;
                    if (row + 3u < a.rows) *reinterpret_cast<float4 *>(o) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; i++) if (row + i < a.rows) o[i] = acc[t][i];
                    }
                }
            }
        }
        if (++ch == nhc) { ch = 0; ct++; }
    }
}

template <int TT>
static void launch_gc_tt(const GCDev &d, uint32_t nwg, uint32_t waves, size_t lds, hipStream_t st) {
    auto kern = &gemm_q80_cls_kernel<TT>;
    static std::atomic<unsigned long long> armed{0};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !((armed.load(std::memory_order_acquire) >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) armed.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(waves * 64), lds, st, d);
}

}  // namespace

// host predicate: the launches GC takes (the classifier of a batched step)
bool gemm_q80_cls_supports(const GemvArgs &a) {
    if (a.gs != 64 || a.nseg != 1 || a.epi != GEMV_EPI_STORE || a.seg[0].out_pstride != 0 || a.attn_part || a.resid_add) return false;
    if (a.nb < 2 || a.nb > 64 || a.n % 64 || (a.n / 64) % 4 != 0 || a.n > 8192) return false;
    if (a.seg[0].rows < 16384 || (a.seg[0].out_bstride % 4) != 0) return false;          // tall matrices; 16-byte output stores
    if ((uint64_t)a.seg[0].rows * a.n >= (1ull << 32) - (1u << 20)) return false;         // 32-bit buffer offsets
    const uint32_t ng = a.n / 64, tt = (a.nb + 15) / 16;
    // LDS: at least min(tt, 2) token tiles staged next to 4 waves' buffers
    const size_t need = (size_t)(tt < 2 ? tt : 2) * ng * 1088u + 4u * GC_LDS_WAVE;
    return need <= 160u * 1024u;
}

// a.xq_in / a.xs_in: the activations in fragment order (launch_quant_rows_frag)
hipError_t launch_gemm_q80_cls(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_cls_supports(a)) return hipErrorInvalidValue;
    GCDev d{};
    d.w = reinterpret_cast<const int8_t *>(a.seg[0].w); d.ws = a.seg[0].ws; d.out = a.seg[0].out;
    d.rows = a.seg[0].rows; d.out_bstride = a.seg[0].out_bstride;
    d.n = a.n; d.ng = a.n / 64; d.nb = a.nb; d.nhc = (d.ng + 7) / 8;
    d.ntiles = (d.rows + 15) / 16; d.tt = (a.nb + 15) / 16;
    d.xf = a.xq_in; d.xsf = a.xs_in;
    const uint32_t TTc = d.tt <= 1 ? 1u : d.tt == 2 ? 2u : 4u;
    // waves per workgroup (one workgroup per CU): as many as leave room to stage every token tile, or at least two of them (the
    // kernel keeps at most two unstaged tiles in registers)
    const size_t tile_lds = (size_t)d.ng * 1088u;
    uint32_t waves = 8, lt = 0;
    for (; waves >= 4; waves -= 2) {
        const size_t room = 160u * 1024u - (size_t)waves * GC_LDS_WAVE - 256u;
        lt = (uint32_t)(room / tile_lds);
        if (lt > d.tt) lt = d.tt;
        if (lt == d.tt || (lt >= 2 && d.tt - lt <= 2)) break;
    }
    if (waves < 4) return hipErrorInvalidValue;
    if (lt < d.tt && lt > 2) lt = 2;                                    // unstaged tiles are tiles 2 and 3 (kernel)
    if (lt < d.tt && (lt != 2 || d.tt > 4)) return hipErrorInvalidValue;
    d.lt = lt;
    const uint32_t cus = a.cus ? a.cus : 256u;
    const uint32_t nwg = cus;                                           // one persistent workgroup per CU
    d.nwaves = nwg * waves;
    const size_t lds = (size_t)lt * tile_lds + (size_t)waves * GC_LDS_WAVE + 64u;
    if (TTc == 1) launch_gc_tt<1>(d, nwg, waves, lds, st);
    else if (TTc == 2) launch_gc_tt<2>(d, nwg, waves, lds, st);
    else launch_gc_tt<4>(d, nwg, waves, lds, st);
    return hipGetLastError();
}

}  // namespace nano
