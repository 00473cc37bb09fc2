// gemm_q80_g4.hip -- G4: batched Q80 (W8A8) GEMM, 2..64 tokens per weight read, built like the classifier's STREAM GEMV:
// every WAVE is on its own.  A wave owns one 16-row tile (the W1 and W3 tiles of the same rows for SwiGLU) and one tile
// of 16 tokens for the WHOLE row length, walks the row in 1 KiB chunks and keeps the running values in registers -- no
// workgroup barrier, no product table, no cross-wave fold: while one wave waits for memory the other seven on its CU
// compute.  Arithmetic as everywhere else: exact int32 group sums on the matrix cores (one v_mfma_i32_16x16x64_i8 per
// 64-byte quantization group), products ((float)ival * ws) * xs, groups added in ascending order -- bit-identical to the
// GEMV path and to the reference's matmul_quant (infer/infer.c:654-679).
//
// Per chunk (16 rows x 1 KiB = 16 groups):
//   * the 16 coalesced 1 KiB row pieces were issued one chunk AHEAD (64 VGPRs in flight per wave, 128 KB per CU) and
//     pass through a wave-private LDS buffer (row pitch 1040 B) that turns them into MFMA A fragments (ds_read_b128,
//     conflict free); a wave's LDS operations complete in order, so no barrier is involved;
//   * the chunk's 16 activation fragments arrive in MFMA B order straight from L2 (quant_rows_frag_kernel wrote them
//     that way): one coalesced 1 KiB load each, issued BEFORE the next chunk's weight pieces -- loads complete in issue
//     order, so waiting for the fragments leaves the prefetch in flight;
//   * weight scales (one 64-byte run per row) and activation scales (one 1 KiB run) are staged through LDS into the
//     accumulator layout.
// More than 16 tokens: the token tiles of a row tile are separate waves of the SAME workgroup; they run the same
// instruction stream over the same weight bytes, so HBM sees the bytes once and the CU's L1 serves the repeats.
// Takes: group size 64, group count a multiple of 4 (float4 scale runs), interior segments multiples of 16 rows.
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct G4Dev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, epi, nb, nchunk, ntiles, tt;
    const int8_t *xf; const float *xsf; const uint32_t *pos;
};

constexpr uint32_t G4_PITCH = 1040, G4_WBUF = 16 * G4_PITCH;          // transposition buffer of one wave
constexpr uint32_t G4_LDS_WAVE = G4_WBUF + 2 * 1024 + 1024;           // + weight scales (two matrices) + activation scales

template <bool SW>
__global__ __launch_bounds__(256, 2) void gemm_q80_g4_kernel(const G4Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * 4u + wid;                          // global wave -> (row tile, token tile)
    const uint32_t tile = gw / a.tt, tt = gw % a.tt;
    if (tile >= a.ntiles) return;                                      // wave-uniform; no barrier anywhere below
    const uint32_t n = a.n, ng = a.ng, nchunk = a.nchunk;
    const uint32_t m = lane & 15u, kq = lane >> 4;

    int8_t *wbuf = reinterpret_cast<int8_t *>(smem) + (size_t)wid * G4_LDS_WAVE;
    float *wsl = reinterpret_cast<float *>(wbuf + G4_WBUF);            // [nmat][16 groups][16 rows]
    float *xsl = wsl + 512;                                            // [16 groups][16 tokens]

    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const uint32_t grow0 = tile * 16u;
    const int sel = SW ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
    const float *ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);

    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0, rows0 * n), rw1 = mkrsrc(SW ? a.w[1] : nullptr, SW ? rows0 * n : 0u);
    const __amdgpu_buffer_rsrc_t rs0 = mkrsrc(ws0, rows0 * ng * 4u), rs1 = mkrsrc(SW ? a.ws[1] : nullptr, SW ? rows0 * ng * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(a.xf, a.tt * ng * 1024u);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(a.xsf, a.tt * ng * 64u);

    // ---- first in the load queue: the old values of a residual epilogue ---------------------------------------------------
    const uint32_t tok = tt * 16u + m;
    const bool tok_live = tok < a.nb;
    float *orow = out0 + (size_t)tok * obs + lrow0 + kq * 4u;
    float oldv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.epi == GEMV_EPI_RESID && tok_live) {
#pragma unroll
        for (int i = 0; i < 4; i++) if (lrow0 + kq * 4u + i < rows0) oldv[i] = orow[i];      // the residual stream is never position indexed
    }
    uint32_t opos = 0;
    if (ops && tok_live) opos = a.pos[tok];

    // ---- weight pieces: 16 rows x 1 KiB of one matrix, coalesced (lane l: bytes [16 l, 16 l + 16) of the row chunk) -------
    int4 wA[16];
    auto issue_w = [&](uint32_t c, int mt) {
        const uint32_t col = (c << 10) + lane * 16u;
        const uint32_t base = (c < nchunk && col < n) ? lrow0 * n + col : OOB;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128((SW && mt) ? rw1 : rw0, (int)(base == OOB ? OOB : base + (uint32_t)r * n), 0, 2);   // rows beyond the segment: out of range -> 0
            wA[r] = make_int4(v.x, v.y, v.z, v.w);
        }
    };
    issue_w(0, 0);

    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t c = 0; c < nchunk; c++) {
        // 1. the chunk's weight pieces (first matrix): registers -> the transposition buffer
#pragma unroll
        for (int r = 0; r < 16; r++) *reinterpret_cast<int4 *>(wbuf + (size_t)r * G4_PITCH + lane * 16u) = wA[r];
        // 2. what this chunk needs NOW: activation fragments + scales (issued before the prefetch below)
        i32x4 fb[16];
        const uint32_t g0 = c * 16u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++)
            fb[j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(g0 + j < ng ? (tt * ng + g0 + j) * 1024u + lane * 16u : OOB), 0, 0);
        const uint32_t sg = g0 + (lane & 3u) * 4u;                      // lane l: row l/4, groups g0 + 4 (l%4) .. +3
        const uint32_t so = sg < ng ? ((lrow0 + (lane >> 2)) * ng + sg) * 4u : OOB;
        const float4 wsv0 = bload_f4(rs0, so), wsv1 = SW ? bload_f4(rs1, so) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t xg = g0 + (lane >> 2);                           // lane l: group g0 + l/4, tokens 4 (l%4) .. +3
        const float4 xsv = bload_f4(rxs, xg < ng ? ((tt * ng + xg) * 16u + (lane & 3u) * 4u) * 4u : OOB);
        // 3. prefetch: the next unit's weight pieces (SwiGLU: this chunk's second matrix; else the next chunk)
        issue_w(SW ? c : c + 1u, SW ? 1 : 0);
        // 4. scales -> LDS in accumulator order: wsl[g][row], xsl[g][token]
        {
            const uint32_t r = lane >> 2, gq = (lane & 3u) * 4u;
            wsl[(gq + 0u) * 16u + r] = wsv0.x; wsl[(gq + 1u) * 16u + r] = wsv0.y; wsl[(gq + 2u) * 16u + r] = wsv0.z; wsl[(gq + 3u) * 16u + r] = wsv0.w;
            if (SW) { wsl[256u + (gq + 0u) * 16u + r] = wsv1.x; wsl[256u + (gq + 1u) * 16u + r] = wsv1.y; wsl[256u + (gq + 2u) * 16u + r] = wsv1.z; wsl[256u + (gq + 3u) * 16u + r] = wsv1.w; }
            *reinterpret_cast<float4 *>(xsl + lane * 4u) = xsv;
        }
        // 5. 16 groups: A fragment from LDS, MFMA, products, ascending accumulation (infer.c:668-674).  Groups beyond the
        //    row multiply zeros by zero scales: + 0.0f, exact (a running value is never -0.0f: scales are >= 0).
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
            const i32x4 fa = *reinterpret_cast<const i32x4 *>(wbuf + (size_t)m * G4_PITCH + j * 64u + kq * 16u);
            const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb[j], v4i{0, 0, 0, 0}, 0, 0, 0);
            const float4 wv = *reinterpret_cast<const float4 *>(wsl + j * 16u + kq * 4u);
            const float xsc = xsl[j * 16u + m];
            acc0[0] += ((float)cv[0] * wv.x) * xsc; acc0[1] += ((float)cv[1] * wv.y) * xsc;                 // infer.c:672
            acc0[2] += ((float)cv[2] * wv.z) * xsc; acc0[3] += ((float)cv[3] * wv.w) * xsc;
        }
        if (SW) {
            // second matrix of the chunk: same activation fragments and activation scales
#pragma unroll
            for (int r = 0; r < 16; r++) *reinterpret_cast<int4 *>(wbuf + (size_t)r * G4_PITCH + lane * 16u) = wA[r];
            issue_w(c + 1u, 0);
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) {
                const i32x4 fa = *reinterpret_cast<const i32x4 *>(wbuf + (size_t)m * G4_PITCH + j * 64u + kq * 16u);
                const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb[j], v4i{0, 0, 0, 0}, 0, 0, 0);
                const float4 wv = *reinterpret_cast<const float4 *>(wsl + 256u + j * 16u + kq * 4u);
                const float xsc = xsl[j * 16u + m];
                acc1[0] += ((float)cv[0] * wv.x) * xsc; acc1[1] += ((float)cv[1] * wv.y) * xsc;
                acc1[2] += ((float)cv[2] * wv.z) * xsc; acc1[3] += ((float)cv[3] * wv.w) * xsc;
            }
        }
    }
    if (tok_live) {
        float *o = orow + (size_t)opos * ops;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (lrow0 + kq * 4u + i < rows0) o[i] = finish_epi(a.epi, i == 0 ? acc0[0] : i == 1 ? acc0[1] : i == 2 ? acc0[2] : acc0[3],
                                                               i == 0 ? acc1[0] : i == 1 ? acc1[1] : i == 2 ? acc1[2] : acc1[3], oldv[i]);
    }
}

static uint32_t total_rows4(const GemvArgs &a) {
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    return rows;
}

}  // namespace

bool gemm_q80_g4_supports(const GemvArgs &a) {
    if (!gemm_q80_g2_supports(a) || a.gs != 64) return false;
    if ((a.n / a.gs) % 4 != 0) return false;                           // float4 runs of scales
    if ((uint64_t)total_rows4(a) * a.n >= (1ull << 31)) {              // 32-bit buffer offsets: per segment
        for (uint32_t s = 0; s < a.nseg; s++) if ((uint64_t)a.seg[s].rows * a.n >= (1ull << 32) - (1u << 20)) return false;
    }
    return true;
}

// a.xq_in / a.xs_in: the activations in fragment order (launch_quant_rows_frag)
hipError_t launch_gemm_q80_g4(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g4_supports(a)) return hipErrorInvalidValue;
    G4Dev d{};
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
    d.n = a.n; d.ng = a.n / a.gs; d.epi = a.epi; d.nb = a.nb; d.nchunk = (a.n + 1023) / 1024;
    d.ntiles = (total_rows4(a) + 15) / 16;
    d.tt = (a.nb + 15) / 16; if (d.tt == 3) d.tt = 4;                   // waves of a tile stay inside one 4-wave workgroup
    d.xf = a.xq_in; d.xsf = a.xs_in; d.pos = a.pos;
    const size_t lds = 4 * (size_t)G4_LDS_WAVE;
    const uint32_t waves = d.ntiles * d.tt;
    if (a.epi == GEMV_EPI_SWIGLU) {
        auto kern = &gemm_q80_g4_kernel<true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((waves + 3) / 4), dim3(256), lds, st, d);
    } else {
        auto kern = &gemm_q80_g4_kernel<false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((waves + 3) / 4), dim3(256), lds, st, d);
    }
    return hipGetLastError();
}

}  // namespace nano
