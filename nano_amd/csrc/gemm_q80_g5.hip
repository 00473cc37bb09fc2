// gemm_q80_g5.hip -- G5: batched Q80 (W8A8) GEMM, 2..64 tokens per weight read; the row length of a 16-row tile is
// split over a TEAM of waves and the reference's ascending group order is kept by handing the running values from wave to
// wave (a chain through LDS), so a matrix with few rows still has thousands of waves with all their loads in flight.
//
//   * unit of work: (16-row tile) x (half chunk = 8 quantization groups = 512 B of every row).  Wave kw of a team of nkw
//     waves owns the half chunks kw, kw + nkw, kw + 2 nkw, ...; its first one's 16 x 512 B are issued at kernel entry
//     (8 coalesced loads, two rows each), the next one is prefetched while the current one is consumed.
//   * the pieces pass through a wave-private LDS buffer (row pitch 528 B) that turns them into MFMA A fragments
//     (ds_read_b128); the activations arrive in B-fragment order (quant_rows_frag_kernel); one v_mfma_i32_16x16x64_i8 per
//     (group, token tile) gives the exact int32 group sums; products ((float)ival * ws) * xs (infer/infer.c:672).
//   * ALL token tiles (up to 4 x 16 tokens) are taken by the same wave, so a weight byte is read once from HBM and once
//     from LDS per token tile, never again from L2.
//   * the chain: per token tile a slot (256 running values) and a counter (half chunks folded so far) in LDS.  The owner
//     of half chunk h forms its 8 x 4 products, waits until the counter says h, adds them to the slot's values in
//     ascending group order (infer.c:668-674), stores the values back and sets the counter to h + 1: one short link per
//     half chunk, the products of later half chunks and token tiles are formed while earlier links are still travelling.
//     The owner of the last half chunk runs the epilogue.  SwiGLU: the W1 team and the W3 team of a row tile sit in the
//     same workgroup; W3's last owner publishes, W1's last owner combines.
//   * SwiGLU launches can hand the W2 GEMM its quantized input: with four row-tile pairs per workgroup the four finishing
//     waves hold one 64-row quantization group of hb for 16 tokens; they exchange their maxima through LDS and write the
//     group in fragment order themselves (one activation-quantizer launch less per layer; same bits).
// Bit-identical to the GEMV path and to the reference's matmul_quant (infer/infer.c:654-679).
// Takes: group size 64, group count a multiple of 4, interior segments multiples of 16 rows.
// Round 3: BALANCED tiles -- a tile is trw <= 16 rows fitted to the CU count (launch_gemm_q80_g5), the matrix-core tile
// stays 16 x 16 with its unused rows zero.
#include <atomic>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct G5Dev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, epi, nb, nhc, ntiles, tt, nkw, cpw, teams, nmat;
    uint32_t trw, tc0, tc1;             // rows per tile (even, <= 16: balanced tiles), tiles up to the end of segment 0 / 1
    uint32_t t_off;                     // canonical fold without the chain: byte offset in LDS of the unit-sum table [team][TT][nhc][256] (0: none -- chained)
    uint32_t canon;                     // 1: the fast path's canonical fold (kernels.h q80_canonical()): the half chunk's 8 products are summed on
                                        // their own (= the unit sum S_u), then added to the running value; 0 (strict mode): group by group
    const int8_t *xf; const float *xsf; const uint32_t *pos;
    // SwiGLU launches, optional (4 row-tile pairs per workgroup): the outputs also leave as Q80 groups of 64 in fragment order,
    // i.e. the next GEMM's activation operand (what quant_rows_frag_kernel would make of them); ng2 = rows / 64
    int8_t *xf2; float *xsf2; uint32_t ng2;
    uint32_t stage_x;                   // 1 (nkw == 1, TT == 1): the tile's activation fragments [ng][1 KiB] are copied to LDS once per workgroup; every wave reads them there
    uint32_t hoist_ws;                  // 1: every wave fetches the weight scales of its first half chunk at kernel entry (TT == 1 launches)
    unsigned long long *stamps;         // measurement builds only (NANO_STAMPS): [workgroup][8] stamps of the workgroup's first wave, or nullptr
};

constexpr uint32_t G5_PITCH = 528, G5_WBUF = 16 * G5_PITCH;           // transposition buffer of one wave: 16 rows x 512 B
constexpr uint32_t G5_LDS_WAVE = G5_WBUF + 512 + 512;                 // + weight scales [8][16] + activation scales [8][16]
constexpr uint32_t G5_MAX_WAVES = 12;

__device__ __forceinline__ uint32_t lds_load_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_rel(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// TAB: the fast path's table fold (t_off != 0) instead of the chain -- a template parameter: the two forms do not share registers
template <int TT, bool TAB>
__global__ __launch_bounds__(G5_MAX_WAVES * 64, 3) void gemm_q80_g5_kernel(const G5Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the arguments first read inside the tile loop (activation fragments, positions, the fragment outputs) come with the first batch
    karg_touch(a.xf); karg_touch(a.xsf); karg_touch(a.pos); karg_touch(a.xf2); karg_touch(a.xsf2); karg_touch(a.ng2);
    karg_touch(a.out[0]); karg_touch(a.out[1]); karg_touch(a.out[2]);
    const uint32_t nkw = a.nkw, nmat = a.nmat;
    const uint32_t team = wid / nkw, kw = wid % nkw;
    const uint32_t unit = blockIdx.x * a.teams + team;                  // (row tile, matrix)
    const uint32_t tile = unit / nmat, mat = unit % nmat;
    const uint32_t n = a.n, ng = a.ng, nhc = a.nhc, tt = a.tt;
    const uint32_t m = lane & 15u, kq = lane >> 4;

    const uint32_t nwaves = a.teams * nkw;
    int8_t *wbuf = reinterpret_cast<int8_t *>(smem) + (size_t)wid * G5_LDS_WAVE;
    float *wsl = reinterpret_cast<float *>(wbuf + G5_WBUF);            // [8 groups][16 rows]
    float *xsl = wsl + 128;                                            // [8 groups][16 tokens]
    float *slots = reinterpret_cast<float *>(smem + (size_t)nwaves * G5_LDS_WAVE);        // [team][TT][256]
    uint32_t *flags = reinterpret_cast<uint32_t *>(slots + (size_t)a.teams * TT * 256u);  // [team][TT]
    float *slot = slots + (size_t)team * TT * 256u;
    uint32_t *flag = flags + team * TT;
    float *gmax = reinterpret_cast<float *>(flags + ((a.teams * TT + 3u) & ~3u));          // [TT][4 row tiles][16 tokens] (fused group quantizer)
    uint32_t *gcnt = reinterpret_cast<uint32_t *>(gmax + TT * 64);                         // [TT] finishers arrived
    unsigned char *xstage = reinterpret_cast<unsigned char *>(gcnt) + 256;                 // [ng][1024] (stage_x launches; 16-byte aligned: everything before it is)

    // ---- which segment (q | k | v share a launch; SwiGLU: matrix 0 = W1, matrix 1 = W3 over the same rows) -----------------
    // A tile is trw <= 16 rows of ONE segment (balanced tiles: trw is fitted so that the tiles spread evenly over the CUs; the
    // matrix-core tile stays 16 x 16, rows trw..15 of it are zero and never stored)
    const uint32_t trw = a.trw;
    const int sel = nmat == 2 ? (int)mat : (int)(tile >= a.tc0) + (int)(tile >= a.tc1);
    const int8_t *w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
    const float *ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
    const int osel = nmat == 2 ? 0 : sel;
    float *out0 = osel == 0 ? a.out[0] : osel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = nmat == 2 ? a.rows[0] : sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = osel == 0 ? a.out_bstride[0] : osel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = osel == 0 ? a.out_pstride[0] : osel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = (tile - (nmat == 2 ? 0u : sel == 0 ? 0u : sel == 1 ? a.tc0 : a.tc1)) * trw;
    const bool live = tile < a.ntiles;

    const __amdgpu_buffer_rsrc_t rw = mkrsrc(w0, live ? rows0 * n : 0u);
    const __amdgpu_buffer_rsrc_t rs = mkrsrc(ws0, live ? rows0 * ng * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(a.xf, tt * ng * 1024u);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(a.xsf, tt * ng * 64u);

    const uint32_t hlast = nhc - 1u;                                   // its owner in the W1 / only team runs the epilogue

    // ---- weight pieces of a half chunk: 16 rows x 512 B, two rows per load instruction (lane l: row 2r + l/32) -------------
    int4 wA[8];
    const uint32_t wrow = lane >> 5, wcol = (lane & 31u) * 16u;
    auto issue_w = [&](uint32_t h) {
        const uint32_t col = h * 512u + wcol;
        const uint32_t base = (h < nhc && col < n) ? (lrow0 + wrow) * n + col : OOB;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            // rows beyond the segment: out of range -> 0 (the scalar offset takes part in the range check of a raw buffer only
            // through the address, so the row step stays in the lane offset where the segment's last rows need the check)
            // (trw is even: rows 2r, 2r + 1 of the tile are live or dead together -- a wave-uniform select)
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)((base == OOB || (uint32_t)(2 * r) >= trw) ? OOB : base + (uint32_t)(2 * r) * n), 0, 2);
            wA[r] = make_int4(v.x, v.y, v.z, v.w);
        }
    };
    NANO_STAMP(a.stamps, 0, lane);
    issue_w(kw);
    // The weight scales of the first half chunk come from HBM like the weights; asked for only when the weights have landed
    // (the loop below) they cost a second memory round trip before the first matrix instruction.  Four registers per lane:
    // the TT == 1 instantiation has the room (140 of 168), the wider ones spill already.
    float4 ws_first = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool hoist = TT == 1 && a.hoist_ws != 0u;
    if (hoist) {
        const uint32_t sg = kw * 8u + (lane & 1u) * 4u;
        ws_first = bload_f4(rs, (lane < 32u && (lane >> 1) < trw && sg < ng && kw < nhc) ? ((lrow0 + (lane >> 1)) * ng + sg) * 4u : OOB);
    }
    if (kw == 0u && lane < (uint32_t)TT) flag[lane] = 0u;
    if (wid == 0u && lane < (uint32_t)TT) gcnt[lane] = 0u;
    // Workgroups whose waves all walk the whole row (nkw == 1: the W1|W3 launch with fused output quantizer, eight waves) read
    // the SAME fragments eight times from L2 -- as many bytes through the CU's vector memory path as the weights themselves.
    // Copied to LDS once here (ng KiB, 40 for Qwen3-4B's hidden size), the loop reads them with ds_read_b128: + 1 ... 3 % at 16
    // sequences, +-1 % at 8 (the loop stays issue bound, ~2.2 us per half chunk and wave: DESIGN.md section 3, G5 phase stamps).
    if constexpr (TT == 1) {
        if (a.stage_x) {
            const uint32_t units = ng * 64u, nthr = nwaves * 64u;     // 16-byte units
            for (uint32_t i = threadIdx.x; i < units; i += nthr) {
                const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(i * 16u), 0, 0);
                *reinterpret_cast<i32x4 *>(xstage + (size_t)i * 16u) = v;
            }
        }
    }
    __syncthreads();                                                   // the only workgroup barrier: the counters are armed (and the fragments staged)
    if (!live) return;

    // (Round 3 tried issuing the first half chunk's weight scales and activation fragments at kernel entry, ahead of the weights'
    // arrival, and the next half chunk's at the end of each turn: the fragments held across the chain link push the kernel past
    // the 168 registers of three waves per SIMD -- 10 to 12 spilled registers, Qwen3-4B 8 sequences 2.16 -> 3.79 ms.  Dropped.
    // Second try with the registers solved -- first call peeled, an 8-wave instantiation with 256 registers for the prefetch --:
    // no spills, bit-identical, and SLOWER, 8 sequences 2.103 -> 2.192 ms, 16: 2.159 -> 2.250: every wave of the chip asking for
    // the same fragment lines at the same moment delays the weights behind them (first weights land after 3.4-3.8 us instead of
    // 2.3, profiles/r03_g5_stamps_fragments_at_entry_dropped.txt).  Only the weight SCALES are fetched early: + 1.1 ... 1.5 %.
    // Third: two weight buffers per wave (two half chunks in flight) in that 8-wave instantiation: 2.083 -> 2.096 ... 2.130 ms --
    // the loop of a wave that owns five half chunks (W1|W3) is not short of bytes in flight
    // (profiles/r03_g5_stamps_two_weight_buffers_dropped.txt).)
    auto load_ws = [&](float4 &wsv, uint32_t h) {                       // lanes 0..31: row l/2, groups 8h + 4 (l%2) .. +3
        const uint32_t sg = h * 8u + (lane & 1u) * 4u;
        wsv = bload_f4(rs, (lane < 32u && (lane >> 1) < trw && sg < ng && h < nhc) ? ((lrow0 + (lane >> 1)) * ng + sg) * 4u : OOB);
    };
    auto load_fb = [&](i32x4 (&fb)[8], float4 &xsv, uint32_t g0, uint32_t t) {
        bool staged = false;
        if constexpr (TT == 1) staged = a.stage_x != 0u;
        if (staged) {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                fb[j] = (g0 + j < ng) ? *reinterpret_cast<const i32x4 *>(xstage + (size_t)(g0 + j) * 1024u + lane * 16u) : i32x4{0, 0, 0, 0};
        } else {
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            fb[j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)((g0 + j < ng && t < tt) ? lane * 16u : OOB), (int)((t * ng + g0 + j) * 1024u), 0);   // uniform part in the scalar offset; the range check is on the lane part
        }
        const uint32_t xg = g0 + (lane >> 2);                           // lanes 0..31: group g0 + l/4, tokens 4 (l%4) .. +3
        xsv = bload_f4(rxs, (lane < 32u && xg < ng && t < tt) ? ((t * ng + xg) * 16u + (lane & 3u) * 4u) * 4u : OOB);
    };

    for (uint32_t h = kw; h < nhc; h += nkw) {
        const uint32_t g0 = h * 8u;
        // 1. the half chunk's weight pieces: registers -> the transposition buffer
#pragma unroll
        for (int r = 0; r < 8; r++) *reinterpret_cast<int4 *>(wbuf + (size_t)(2 * r + wrow) * G5_PITCH + wcol) = wA[r];
        if (h == kw) NANO_STAMP(a.stamps, 1, (float)wA[7].x);          // the first half chunk's weights arrived
        // 2. what this half chunk needs now: weight scales (lanes 0..31: row l/2, groups g0 + 4 (l%2) .. +3), first fragments
        float4 wsv;
        if (hoist) wsv = ws_first; else load_ws(wsv, h);               // hoisted: fetched at entry / behind the previous half chunk's prefetch
        i32x4 fb[8]; float4 xsv;
        load_fb(fb, xsv, g0, 0u);
        issue_w(h + nkw);                                               // prefetch; behind the fragments in the load queue (loads return in issue order)
        if (hoist) load_ws(ws_first, h + nkw);                          // ... and the next half chunk's weight scales behind its weights
        if (lane < 32u) {
            const uint32_t r = lane >> 1, gq = (lane & 1u) * 4u;
            wsl[(gq + 0u) * 16u + r] = wsv.x; wsl[(gq + 1u) * 16u + r] = wsv.y; wsl[(gq + 2u) * 16u + r] = wsv.z; wsl[(gq + 3u) * 16u + r] = wsv.w;
        }
#pragma unroll
        for (uint32_t t = 0; t < (uint32_t)TT; t++) {
            if (t < tt) {
                if (lane < 32u) *reinterpret_cast<float4 *>(xsl + lane * 4u) = xsv;
                // 3. 8 groups: A fragment from LDS, MFMA, products
                float p[8][4];
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) {
                    const i32x4 fa = *reinterpret_cast<const i32x4 *>(wbuf + (size_t)m * G5_PITCH + j * 64u + kq * 16u);
                    const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb[j], v4i{0, 0, 0, 0}, 0, 0, 0);
                    const float4 wv = *reinterpret_cast<const float4 *>(wsl + j * 16u + kq * 4u);
                    const float xsc = xsl[j * 16u + m];
                    p[j][0] = ((float)cv[0] * wv.x) * xsc; p[j][1] = ((float)cv[1] * wv.y) * xsc;                 // infer.c:672
                    p[j][2] = ((float)cv[2] * wv.z) * xsc; p[j][3] = ((float)cv[3] * wv.w) * xsc;
                }
                if (h == kw && t == 0u) NANO_STAMP(a.stamps, 2, p[7][3]);      // scales + fragments arrived, products of the first half chunk done
                // 4. the next token tile's fragments, once this tile's are consumed (none left: out-of-range addresses)
                if (t + 1u < (uint32_t)TT) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_fb(fb, xsv, g0, t + 1u);
                }
                // 5. the chain link of (half chunk h, token tile t): the running values so far, this half chunk's groups in
                //    ascending order, on to the owner of h + 1 -- or out through the epilogue
                const uint32_t tok = t * 16u + m;
                const bool fin = h == hlast && mat == 0u;
                float *orow = out0 + (size_t)tok * obs + lrow0 + kq * 4u;
                float oldv[4] = {0.f, 0.f, 0.f, 0.f};
                if (fin && a.epi == GEMV_EPI_RESID && tok < a.nb) {      // issued before the wait: the residual stream is never position indexed
#pragma unroll
                    for (int i = 0; i < 4; i++) if (kq * 4u + i < trw && lrow0 + kq * 4u + i < rows0) oldv[i] = orow[i];
                }
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                // (round 4) fast path: unit sums to a table, ONE wait per (tile, token tile) instead of a link per half chunk
                if constexpr (TAB) {
                    float su[4] = {p[0][0], p[0][1], p[0][2], p[0][3]};
#pragma unroll
                    for (uint32_t j = 1; j < 8; j++) if (g0 + j < ng) { su[0] += p[j][0]; su[1] += p[j][1]; su[2] += p[j][2]; su[3] += p[j][3]; }
                    float *tb = reinterpret_cast<float *>(smem + a.t_off) + ((size_t)(team * TT + t) * nhc) * 256u;
                    if (h != hlast) {
                        *reinterpret_cast<float4 *>(tb + (size_t)h * 256u + lane * 4u) = make_float4(su[0], su[1], su[2], su[3]);
                        if (lane == 0u) __hip_atomic_fetch_add(flag + t, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        continue;                                       // (the W3 team's last owner and every last owner fall through to the fold below)
                    }
                    for (uint32_t spin = 0; lds_load_acq(flag + t) != hlast && spin < (1u << 24); spin++) __builtin_amdgcn_s_sleep(1);
                    if (hlast == 0u) { acc[0] = su[0]; acc[1] = su[1]; acc[2] = su[2]; acc[3] = su[3]; }
                    else {
                        float4 r = *reinterpret_cast<const float4 *>(tb + lane * 4u);
                        for (uint32_t u = 1; u < hlast; u++) { const float4 q = *reinterpret_cast<const float4 *>(tb + (size_t)u * 256u + lane * 4u); r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w; }
                        acc[0] = r.x + su[0]; acc[1] = r.y + su[1]; acc[2] = r.z + su[2]; acc[3] = r.w + su[3];
                    }
                } else {
                if (h != 0u) {
                    while (lds_load_acq(flag + t) != h) __builtin_amdgcn_s_sleep(1);
                    const float4 in = *reinterpret_cast<const float4 *>(slot + t * 256 + lane * 4u);
                    acc[0] = in.x; acc[1] = in.y; acc[2] = in.z; acc[3] = in.w;
                }
                // groups beyond ng (the last half chunk when ng % 8 == 4) contribute products +0.0f: exact, because a running
                // value that started at +0.0f (here as in the reference, infer.c:668) is never -0.0f -- x + (-0.0f) and
                // x + (+0.0f) only differ for x == -0.0f
                if (a.canon) {                                           // S_u = ((p_0 + p_1) + ...) + p_7, then running + S_u (gemm_q80_g6.hip's shape)
                    float su[4] = {p[0][0], p[0][1], p[0][2], p[0][3]};
#pragma unroll
                    for (uint32_t j = 1; j < 8; j++) if (g0 + j < ng) { su[0] += p[j][0]; su[1] += p[j][1]; su[2] += p[j][2]; su[3] += p[j][3]; }
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[i] = h != 0u ? acc[i] + su[i] : su[i];
                } else {
#pragma unroll
                for (uint32_t j = 0; j < 8; j++) { acc[0] += p[j][0]; acc[1] += p[j][1]; acc[2] += p[j][2]; acc[3] += p[j][3]; }
                }
                }
                if (h == kw && t == 0u) NANO_STAMP(a.stamps, 3, acc[0]);       // the first wave's first chain link folded (h = 0: no wait before it)
                if (!fin) {
                    *reinterpret_cast<float4 *>(slot + t * 256 + lane * 4u) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    lds_store_rel(flag + t, h + 1u);
                } else {
                    // ---- epilogue: store | residual add | SwiGLU with the W3 team's values ------------------------------------
                    float v1[4] = {0.f, 0.f, 0.f, 0.f};
                    if (nmat == 2u) {
                        const uint32_t *f3 = flag + TT;                // the W3 team is the next team of the workgroup
                        while (lds_load_acq(f3 + t) != nhc) __builtin_amdgcn_s_sleep(1);
                        const float4 in = *reinterpret_cast<const float4 *>(slot + TT * 256 + t * 256 + lane * 4u);
                        v1[0] = in.x; v1[1] = in.y; v1[2] = in.z; v1[3] = in.w;
                    }
                    float val[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) val[i] = finish_epi(a.epi, acc[i], v1[i], oldv[i]);
                    if (tok < a.nb) {
                        const uint32_t opos = ops ? a.pos[tok] : 0u;
                        float *o = orow + (size_t)opos * ops;
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (kq * 4u + i < trw && lrow0 + kq * 4u + i < rows0) o[i] = val[i];      // (write-through stores measured slower here: 2.19 -> 2.22 ms at 8 sequences)
                    }
                    if constexpr (TT == 1) if (a.xf2) {               // (one token tile only: measured slower with more, and its registers would cost the other instantiations)
                        // ---- the 64-row Q80 group of these outputs (infer/tensor.c:21-46): this wave holds rows 16 rt .. +15 of
                        //      it for 16 tokens, the three other finishing waves of the workgroup the rest; they exchange their
                        //      maxima through LDS and each writes its quarter of the group in fragment order
                        const uint32_t rt = (team >> 1) & 3u;
                        float mx = fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3])));
                        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                        if (kq == 0u) gmax[(t * 4u + rt) * 16u + m] = mx;
                        if (lane == 0u) __hip_atomic_fetch_add(gcnt + t, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        while (lds_load_acq(gcnt + t) != 4u) __builtin_amdgcn_s_sleep(1);
                        const float g4 = fmaxf(fmaxf(gmax[(t * 4u + 0u) * 16u + m], gmax[(t * 4u + 1u) * 16u + m]),
                                               fmaxf(gmax[(t * 4u + 2u) * 16u + m], gmax[(t * 4u + 3u) * 16u + m]));
                        const float scale = div_const<127>(g4);
                        const uint32_t packed = (uint32_t)(q80_quant1(val[0], scale) & 0xff) | ((uint32_t)(q80_quant1(val[1], scale) & 0xff) << 8) |
                                                ((uint32_t)(q80_quant1(val[2], scale) & 0xff) << 16) | ((uint32_t)(q80_quant1(val[3], scale) & 0xff) << 24);
                        if (tok < a.nb) {
                            const size_t gb = (size_t)t * a.ng2 + (tile >> 2);
                            *reinterpret_cast<uint32_t *>(a.xf2 + gb * 1024u + (size_t)(rt * 16u + m) * 16u + kq * 4u) = packed;
                            if (rt == 0u && kq == 0u) a.xsf2[gb * 16u + m] = scale;
                        }
                    }
                }
            }
        }
    }
    NANO_STAMP_END(a.stamps, 6);                                       // the workgroup's last wave (the finisher of the chain) ends
}

static uint32_t total_rows5(const GemvArgs &a) {
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    return rows;
}

template <int TT, bool TAB>
static void launch_tt(const G5Dev &d, uint32_t nwg, uint32_t waves, size_t lds, hipStream_t st) {
    auto kern = &gemm_q80_g5_kernel<TT, TAB>;
    static std::atomic<bool> armed[64];                                // once per instantiation and device: a host call per launch costs microseconds
    int dev = 0; (void)hipGetDevice(&dev);                             // (two threads arming the same device twice is harmless)
    if (dev < 0 || dev >= 64 || !armed[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) armed[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(waves * 64), lds, st, d);
}

}  // namespace

bool gemm_q80_g5_supports(const GemvArgs &a) {
    if (!gemm_q80_g2_supports(a) || a.gs != 64) return false;
    if ((a.n / a.gs) % 4 != 0) return false;                           // float4 runs of scales
    if (a.nb > 64) return false;
    for (uint32_t s = 0; s < a.nseg; s++) if ((uint64_t)a.seg[s].rows * a.n >= (1ull << 32) - (1u << 20)) return false;   // 32-bit buffer offsets per segment
    return true;
}

// a.xq_in / a.xs_in: the activations in fragment order (launch_quant_rows_frag)
// xf2 / xsf2 (SwiGLU launches whose row count is a multiple of 64, or nullptr): the outputs also as the next GEMM's operand
// Up to 16 tokens: with more the one-wave-per-row-tile split this needs is slower than the quantizer launch it saves
// (Qwen3-4B W1|W3 at 64 tokens 31 -> 42 us; at 16 tokens the launch pair is 1.5 us per layer cheaper)
bool gemm_q80_g5_can_quantize_outputs(const GemvArgs &a) {
    return gemm_q80_g5_supports(a) && a.epi == GEMV_EPI_SWIGLU && a.seg[0].rows % 64u == 0 && a.nb <= 16;
}
hipError_t launch_gemm_q80_g5(const GemvArgs &a, int8_t *xf2, float *xsf2, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g5_supports(a)) return hipErrorInvalidValue;
    if (xf2 && (!xsf2 || !gemm_q80_g5_can_quantize_outputs(a))) return hipErrorInvalidValue;
    G5Dev d{};
    for (int i = 0; i < 3; i++) {
        const bool live = i < (int)a.nseg;
        d.w[i] = live ? reinterpret_cast<const int8_t *>(a.seg[i].w) : nullptr;
        d.ws[i] = live ? a.seg[i].ws : nullptr;
        d.out[i] = live ? a.seg[i].out : nullptr;
        d.rows[i] = live ? a.seg[i].rows : 0;
        d.out_bstride[i] = live ? a.seg[i].out_bstride : 0;
        d.out_pstride[i] = live ? a.seg[i].out_pstride : 0;
    }
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    d.n = a.n; d.ng = a.n / a.gs; d.epi = a.epi; d.nb = a.nb; d.nhc = (d.ng + 7) / 8;
    d.tt = (a.nb + 15) / 16;
    d.nmat = sw ? 2u : 1u;
    d.xf = a.xq_in; d.xsf = a.xs_in; d.pos = a.pos;
    d.stamps = a.stamps;
    d.canon = q80_canonical(a) ? 1u : 0u;
    { static const bool hoist = !(getenv("NANO_G5_HOIST") && *getenv("NANO_G5_HOIST") == '0'); d.hoist_ws = hoist ? 1u : 0u; }   // A/B knob
    // The split.  A workgroup = one row tile (SwiGLU: the W1/W3 pair) x nkw waves; a CU holds 12 waves (3 per SIMD at
    // <= 168 VGPRs).  Take the deepest split whose workgroups are ALL resident at once (no second round of workgroups, whose
    // tail would run on a mostly idle chip); matrices too tall for that (the classifier) get one wave per tile, 4 per group.
    const uint32_t TTc = d.tt <= 1 ? 1u : d.tt == 2 ? 2u : 4u;
    const int cus = a.cus ? (int)a.cus : 256;                          // compute units of the model's device (backend.hip fills it in)
    // BALANCED tiles (round 3).  A CU streams ~25 GB/s whatever it runs, so a launch lasts as long as the CU with the most
    // rows: a tile is trw <= 16 rows (even), fitted to minimise (tiles per CU, rounded up) x trw -- 2560 rows: 160 tiles of 16
    // keep 160 of 256 CUs busy, 256 tiles of 10 all of them.  Ties go to the taller tile (fewer waves re-reading the
    // activation fragments).  The output quantizer of SwiGLU launches (xf2) works on 64-row groups of four full tiles.
    // MEASURED (round 3, Qwen3-4B, whole steps on one box): 8 sequences 2.220 ms balanced vs 2.181 ms with 16-row tiles, 64
    // sequences 4.51 vs 4.06 ms -- the shorter tiles multiply the waves that each re-read the activation fragments and the
    // kernels are not CU-bandwidth bound at these sizes.  Default therefore OFF (NANO_G5_BALANCED=1 switches it on for A/B runs);
    // the balanced slabs of the batch-1 GEMV (gemv_q80_impl.h plan_slab) did pay: 1.588 -> 1.526 ms.
    static const bool balanced = [] { const char *e = getenv("NANO_G5_BALANCED"); return e && *e && *e != '0'; }();
    const uint32_t nseg5 = sw ? 1u : a.nseg;
    auto tiles_for = [&](uint32_t trw, uint32_t *tc) {
        uint32_t t = 0;
        for (uint32_t s2 = 0; s2 < nseg5; s2++) { t += (a.seg[s2].rows + trw - 1) / trw; if (tc && s2 < 2) tc[s2] = t; }
        return t;
    };
    uint32_t trw = 16;
    if (balanced && !xf2) {
        uint32_t best_cost = ~0u;
        for (uint32_t c = 4; c <= 16; c += 2) {
            const uint32_t t = tiles_for(c, nullptr), cost = ((t + (uint32_t)cus - 1) / (uint32_t)cus) * c;
            if (cost <= best_cost) { best_cost = cost; trw = c; }
        }
    }
    {
        uint32_t tc[2] = {0xffffffffu, 0xffffffffu};
        d.ntiles = tiles_for(trw, tc);
        d.trw = trw;
        d.tc0 = nseg5 > 1 ? tc[0] : 0xffffffffu;
        d.tc1 = nseg5 > 2 ? tc[1] : 0xffffffffu;
    }
    const uint32_t maxkw = G5_MAX_WAVES / d.nmat;
    uint32_t nkw = 1, groups = 1;                                      // groups: row tiles (pairs) per workgroup
    bool fits = false;
    // Measured with the phase stamps (round 3, `profiles/r03_g5_stamps.txt`): two workgroups of 5 or 8 waves do NOT share a CU
    // (Qwen3-4B's QKV, 384 tiles as 384 five-wave workgroups: the last workgroup entered 3.9 us after the first, the launch was
    // two rounds of ~5 us).  So when the tiles outnumber the CUs, a workgroup takes `groups` tiles -- one team of waves each --
    // and the whole launch is one round again.  NANO_G5_GROUPS=0 restores the one-tile workgroups.
    static const bool group_tiles = !(getenv("NANO_G5_GROUPS") && *getenv("NANO_G5_GROUPS") == '0');
    if (group_tiles && !xf2 && d.ntiles > (uint32_t)cus) {
        for (uint32_t gsz = 2; gsz <= 4 && !fits; gsz++) {
            if ((d.ntiles + gsz - 1) / gsz > (uint32_t)cus) continue;
            for (uint32_t k = G5_MAX_WAVES / (gsz * d.nmat); k >= 1; k--) {
                if (k > d.nhc) continue;
                const uint32_t cpw = (d.nhc + k - 1) / k, kk = (d.nhc + cpw - 1) / cpw;
                const uint32_t wg_waves = gsz * d.nmat * kk;
                const size_t wg_lds = (size_t)wg_waves * G5_LDS_WAVE + (size_t)gsz * d.nmat * TTc * 1040u + 1024u;
                if (wg_waves <= G5_MAX_WAVES && wg_lds <= 160u * 1024u) { nkw = kk; groups = gsz; fits = true; break; }
            }
        }
    }
    for (uint32_t k = maxkw < d.nhc ? maxkw : d.nhc; k >= 1 && !fits; k--) {
        const uint32_t cpw = (d.nhc + k - 1) / k, kk = (d.nhc + cpw - 1) / cpw;            // balanced: no wave owns more than cpw half chunks
        const uint32_t wg_waves = kk * d.nmat;
        const size_t wg_lds = (size_t)wg_waves * G5_LDS_WAVE + (size_t)d.nmat * TTc * 1040u;
        uint32_t slots = G5_MAX_WAVES / wg_waves;
        if ((size_t)slots * wg_lds > 160u * 1024u) slots = (uint32_t)(160u * 1024u / wg_lds);
        if (slots && d.ntiles <= (uint32_t)cus * slots) { nkw = kk; fits = true; break; }
    }
    if (!fits) { nkw = 1; groups = 4u / d.nmat; }
    if (xf2) { nkw = 1; groups = 4; d.xf2 = xf2; d.xsf2 = xsf2; d.ng2 = a.seg[0].rows / 64u; }     // one 64-row group of outputs per workgroup
    d.nkw = nkw; d.cpw = (d.nhc + nkw - 1) / nkw;
    d.teams = groups * d.nmat;
    const uint32_t waves = d.teams * d.nkw;
    const uint32_t nwg = (d.ntiles + groups - 1) / groups;
    size_t lds = (size_t)waves * G5_LDS_WAVE + (size_t)d.teams * TTc * 1024u + (((size_t)d.teams * TTc * 4u + 15u) & ~(size_t)15u) + (size_t)TTc * 256u + 16u;
    {   // fragments staged in LDS: one token tile, every wave walks the whole row, more than one wave to share them, and room
        static const bool stage = !(getenv("NANO_G5_STAGE") && *getenv("NANO_G5_STAGE") == '0');     // A/B knob
        const size_t need = (size_t)waves * G5_LDS_WAVE + (size_t)d.teams * 1024u + (((size_t)d.teams * 4u + 15u) & ~(size_t)15u) + 64u * 4u + 256u + (size_t)d.ng * 1024u + 64u;
        d.stage_x = (stage && TTc == 1 && nkw == 1 && waves >= 4 && need <= 160u * 1024u) ? 1u : 0u;
        if (d.stage_x) lds = need;
    }
    // Fast path (canonical fold), round 4: the unit sums of a row tile can go to a TABLE [team][token tile][half chunk][256] whose last
    // owner adds them in order -- one wait per (tile, token tile) instead of one chain link per half chunk.  MEASURED (one box, Qwen3-4B,
    // whole steps): 32 sequences 2.921 ms with the table, 2.910 with the chain; 64 sequences 4.260 vs 4.230; Qwen3-0.6B at 64: 1.975 vs
    // 1.992.  Neutral: at 2..4 token tiles the launch is bound by the serial MFMA -> scale work of a wave walking every token tile and by
    // the fragment traffic, not by the chain's links.  Off by default (NANO_G5_TABLE=1), kept because it is the same bits.
    d.t_off = 0;
    if (d.canon && d.nkw > 1u && getenv("NANO_G5_TABLE") && *getenv("NANO_G5_TABLE") == '1') {
        const size_t off = (lds + 15u) & ~(size_t)15u, tab = (size_t)d.teams * TTc * d.nhc * 1024u;
        if (off + tab <= 160u * 1024u) { d.t_off = (uint32_t)off; lds = off + tab; }
    }
    if (d.t_off) {
        if (TTc == 1) launch_tt<1, true>(d, nwg, waves, lds, st);
        else if (TTc == 2) launch_tt<2, true>(d, nwg, waves, lds, st);
        else launch_tt<4, true>(d, nwg, waves, lds, st);
    } else {
        if (TTc == 1) launch_tt<1, false>(d, nwg, waves, lds, st);
        else if (TTc == 2) launch_tt<2, false>(d, nwg, waves, lds, st);
        else launch_tt<4, false>(d, nwg, waves, lds, st);
    }
    return hipGetLastError();
}

}  // namespace nano
