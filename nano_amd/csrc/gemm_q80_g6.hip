// gemm_q80_g6.hip -- G6: the split-K Q80 (W8A8) projection kernel of the FAST path, group size 64, 1..16 tokens per weight read.
//
// What it replaces and why (round 4).  G5 (gemm_q80_g5.hip) kept the reference's ascending group order (infer/infer.c:668-674) for
// every output by handing a running value from wave to wave -- a serial chain through LDS (19 links for Qwen3-4B's W2) -- and its
// SwiGLU launch tied eight waves to one 64-row output group (152 of 256 CUs busy).  The fast path is now held to the float bar of
// SURVEY 7 tier ii (<= 1e-5 relative; the integer group sums and both quantizers stay bit-exact) with ONE fixed, split-independent
// reduction shape -- the CANONICAL fold, shared with the GEMV kernels (gemv_q80_impl.h) and G5's large-batch form:
//     products   p_g   = ((float)ival_g * ws_g) * xs_g                          (infer.c:672, unchanged)
//     unit sums  S_u   = ((p_8u + p_8u+1) + ... ) + p_8u+7                      (8 groups = 512 bytes of the row, ascending)
//     row value        = ((S_0 + S_1) + S_2) + ...                              (units ascending)
// so a batch is still bit for bit its sequences alone and batched prefill is bit for bit token-by-token ingestion.  Strict mode
// (nano_hip_set_strict) keeps the reference's order in the older kernels and stays the bit-exact certificate.
//
// Structure (MI355X: 256 CUs, 8 waves of one workgroup per CU, LDS 160 KB):
//   * a TILE is up to 16 matrix rows = two halves of `hh` <= 8 rows (SwiGLU: half 0 = rows of W1, half 1 = the same rows of W3, so
//     the pair meets in one matrix-core tile and no 64-row grouping is needed); hh is fitted on the host so that the tiles spread
//     evenly over the CUs (a launch lasts as long as the CU with the most rows).  Workgroup b owns tiles b, b + grid, ...
//   * an ITEM is (tile, unit): 16 rows x 512 B = 8 KB of weights.  The items of a workgroup are dealt round-robin to its waves;
//     a wave keeps D items in flight in registers (every load of a Qwen3-4B launch is issued at kernel entry) -- TRUE split-K:
//     no wave waits for another one's result before it multiplies.
//   * per item: registers -> wave-private LDS transposition buffer (row pitch 528 B) -> MFMA A fragments (ds_read_b128); one
//     v_mfma_i32_16x16x64_i8 per group gives the exact int32 group sums of (16 rows x 16 tokens); products; the unit sum S_u goes
//     to an LDS table; the wave that owns the row's LAST unit waits for the tile's counter, adds the units in order and runs the
//     epilogue (store | residual add | SwiGLU).  One counter wait per tile, no chain.
//   * the activation comes first in every wave's load queue (loads return in issue order; round 3 measured the activation of a
//     52.9 MB launch "arriving" after the whole weight burst when it was issued behind it):
//       MODE P  (1..8 sequences): rmsnorm | split-attention combine + Q80 quantization (tensor.c:21-46, bit-exact) run in the
//               kernel's prologue from registers while the weights are in flight, into a compact fragment layout in LDS --
//               no quantizer launch, 5 launches per layer at batch 1;
//       MODE F  (fragment-order activations from quant_rows_frag_kernel / the attention kernel): each item's 8 KB of B fragments
//               are requested right before its weights.
// MFMA operand layout as in gemm_q80.hip (verified on gfx950, tools/kbench/mfma_probe.hip).
#include <atomic>
#include <type_traits>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

enum : int { G6_F = 0, G6_P = 1 };
constexpr uint32_t G6_PITCH = 528, G6_WBUF = 16 * G6_PITCH;
constexpr uint32_t G6_LDS_WAVE = G6_WBUF + 512 + 512;          // + weight scales [8 groups][16 rows] + (F) activation scales [8][16 tokens]
constexpr uint32_t G6_NW = 8;                                   // waves of a workgroup (launches with fewer items use fewer)

struct G6Dev {
    GemvDev g;                          // segments, n, ng, epi, flags, nb, the fp32 activation / norm weight / attention partials (MODE P)
    const int8_t *xf; const float *xsf; // MODE F: activations in MFMA B-fragment order [group][lane][16 B], scales [group][16 tokens]
    uint32_t hh;                        // live rows per half tile (1..8)
    uint32_t nu, magic_nu;              // units per row; (it * magic_nu) >> 16 == it / nu for every item index of a workgroup
    uint32_t ntiles, tc0, tc1;          // tiles; tiles up to the end of segment 0 / 1
    uint32_t grid, tpw;                 // workgroups; tiles per workgroup (max)
    uint32_t nw, _pad;                  // waves per workgroup
};

__device__ __forceinline__ uint32_t g6_lds_load_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the fp32 activation items of a thread (MODE P): float4 item q = tid + j * 512 of every sequence
template <int NBC, int NV, bool COMB>
struct G6X {
    float4 x[NBC][NV];
    float4 nw[NV];
    float4 pv[COMB ? NV : 1][COMB ? 8 : 1];     // split-attention partials (one sequence): all splits of this thread's items
    float ml_m, ml_l;
};

template <int MODE, bool COMB, int NBC, int NV>
__global__ __launch_bounds__(512, 2) void gemm_q80_g6_kernel(const G6Dev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = MODE == G6_P ? 3 : 2;                             // items in flight per wave
    const GemvDev &a = d.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t m = lane & 15u, kq = lane >> 4;
    karg_touch(a.out[0]); karg_touch(a.out[1]); karg_touch(a.out[2]); karg_touch(a.pos); karg_touch(d.xf); karg_touch(d.xsf);
    karg_touch(a.out_bstride[0]); karg_touch(a.out_pstride[2]); karg_touch(d.tc0); karg_touch(d.magic_nu);
    const uint32_t n = a.n, ng = a.ng, nu = d.nu, hh = d.hh, NW = d.nw, nb = a.nb;
    const uint32_t epi = a.epi;
    const bool sw = epi == GEMV_EPI_SWIGLU;
    const uint32_t halfoff = sw ? 0u : hh;                              // rows between the two halves of a tile

    // ---- LDS ---------------------------------------------------------------------------------------------------------------
    int8_t *wbuf = reinterpret_cast<int8_t *>(smem) + (size_t)wid * G6_LDS_WAVE;
    float *wsl = reinterpret_cast<float *>(wbuf + G6_WBUF);            // [8 groups][16 rows]
    float *xslw = wsl + 128;                                           // MODE F: [8 groups][16 tokens]
    float *T = reinterpret_cast<float *>(smem + (size_t)NW * G6_LDS_WAVE);          // [tpw][nu - 1][256] unit sums
    uint32_t *cnt = reinterpret_cast<uint32_t *>(T + (size_t)d.tpw * (nu - 1u) * 256u);   // [tpw] units arrived
    unsigned char *pbase = reinterpret_cast<unsigned char *>(cnt + ((d.tpw + 3u) & ~3u));
    int8_t *xqc = reinterpret_cast<int8_t *>(pbase);                   // MODE P: [ng][4 k-quarters][NBC][16 B]
    float *xs_l = reinterpret_cast<float *>(pbase + (size_t)ng * 64u * NBC);        // MODE P: [ng][16] activation scales (slots >= NBC unused)
    float *red = xs_l + (size_t)ng * 16u;                              // [NBC][8] wave partials of the sums of squares
    float *wgt = red + NBC * 8;                                        // COMB: [n_head][8] combine weights

    // ---- the workgroup's items -----------------------------------------------------------------------------------------------
    const uint32_t bid = blockIdx.x;
    const uint32_t ntl = bid < d.ntiles ? (d.ntiles - bid + d.grid - 1u) / d.grid : 0u;   // tiles of this workgroup
    const uint32_t nitems = ntl * nu;

    struct TI { uint32_t tl, u, lrow0, rows0, obs, ops; const int8_t *wA, *wB; const float *sA, *sB; float *out; bool live; };
    auto decode = [&](uint32_t it) -> TI {
        TI t;
        t.live = it < nitems;
        t.tl = (it * d.magic_nu) >> 16; t.u = it - t.tl * nu;
        const uint32_t tile = bid + t.tl * d.grid;
        const int sel = sw ? 0 : (int)(tile >= d.tc0) + (int)(tile >= d.tc1);
        t.wA = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
        t.sA = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
        t.wB = sw ? a.w[1] : t.wA; t.sB = sw ? a.ws[1] : t.sA;
        t.out = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
        t.rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
        t.obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
        t.ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
        t.lrow0 = (tile - (sel == 0 ? 0u : sel == 1 ? d.tc0 : d.tc1)) * (sw ? hh : 2u * hh);
        return t;
    };

    // ---- the ring: D items of this wave in flight ---------------------------------------------------------------------------------
    struct Slot { int4 w[8]; float4 s0, s1; i32x4 b[MODE == G6_F ? 8 : 1]; float4 xs; };
    Slot ring[D];
    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(d.xf, MODE == G6_F ? ng * 1024u : 0u);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(d.xsf, MODE == G6_F ? ng * 64u : 0u);
    auto issue = [&](auto J, uint32_t it) {
        constexpr int sl = decltype(J)::value;
        const TI t = decode(it);
        const uint32_t g0 = t.u * 8u;
        if constexpr (MODE == G6_F) {               // the item's activation fragments FIRST (loads return in issue order)
#pragma unroll
            for (uint32_t j = 0; j < 8; j++)
                ring[sl].b[j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)((t.live && g0 + j < ng) ? lane * 16u : OOB), (int)((g0 + j) * 1024u), 0);
            const uint32_t xg = g0 + (lane >> 2);                       // lanes 0..31: group g0 + l/4, tokens 4 (l%4) .. +3
            ring[sl].xs = bload_f4(rxs, (t.live && lane < 32u && xg < ng) ? (xg * 16u + (lane & 3u) * 4u) * 4u : OOB);
        }
        const __amdgpu_buffer_rsrc_t rA = mkrsrc(t.wA, t.live ? t.rows0 * n : 0u), rB = mkrsrc(t.wB, t.live ? t.rows0 * n : 0u);
        const __amdgpu_buffer_rsrc_t qA = mkrsrc(t.sA, t.live ? t.rows0 * ng * 4u : 0u), qB = mkrsrc(t.sB, t.live ? t.rows0 * ng * 4u : 0u);
        const uint32_t col = t.u * 512u + (lane & 31u) * 16u;
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) {            // tile row 2 r8 + l/32: half r8 / 4 (compile time: one descriptor per instruction)
            const uint32_t r = 2u * ((uint32_t)r8 & 3u) + (lane >> 5);
            const uint32_t row = t.lrow0 + (r8 >= 4 ? halfoff : 0u) + r;
            const bool ok = r < hh && row < t.rows0 && col < n;
            ring[sl].w[r8] = bload_w(r8 >= 4 ? rB : rA, ok ? row * n + col : OOB);
        }
        {                                           // weight scales: lanes 0..15: row l/2 of the half, groups g0 + 4 (l%2) .. +3
            const uint32_t r = lane >> 1, g = g0 + (lane & 1u) * 4u;
            const uint32_t rowA = t.lrow0 + r, rowB = t.lrow0 + halfoff + r;
            const bool okl = lane < 16u && r < hh && g < ng;
            ring[sl].s0 = bload_f4(qA, (okl && rowA < t.rows0) ? (rowA * ng + g) * 4u : OOB);
            ring[sl].s1 = bload_f4(qB, (okl && rowB < t.rows0) ? (rowB * ng + g) * 4u : OOB);
        }
    };

    // ---- MODE P: the fp32 activation is asked for before any weight --------------------------------------------------------------
    G6X<NBC, NV, COMB> sx;
    const bool norm = (a.flags & F_NORM) != 0;
    if constexpr (MODE == G6_P) {
        const __amdgpu_buffer_rsrc_t rx = mkrsrc(a.xin, COMB ? 0u : ((nb - 1u) * a.xin_bstride + n) * 4u);
        const __amdgpu_buffer_rsrc_t rn = mkrsrc(a.norm_w, norm ? n * 4u : 0u);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (tid + (uint32_t)j * 512u) * 4u;
            const uint32_t off = (i < n) ? i * 4u : OOB;
            if constexpr (!COMB) {
#pragma unroll
                for (int b = 0; b < NBC; b++) sx.x[b][j] = bload_f4(rx, (b < (int)nb) ? off + (uint32_t)b * a.xin_bstride * 4u : OOB);
            }
            sx.nw[j] = bload_f4(rn, off);
        }
        if constexpr (COMB) {
            const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
            const __amdgpu_buffer_rsrc_t rp = mkrsrc(a.attn_part, ns * n * 4u);
            const __amdgpu_buffer_rsrc_t rm = mkrsrc(a.attn_ml, nh * ns * 8u);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const uint32_t i = (tid + (uint32_t)j * 512u) * 4u;
#pragma unroll
                for (int sp = 0; sp < 8; sp++) sx.pv[j][sp] = bload_f4(rp, (i < n && (uint32_t)sp < ns) ? ((uint32_t)sp * n + i) * 4u : OOB);
            }
            const uint32_t sp = tid & 7u, h = tid >> 3;
            const uint32_t mo = (h < nh && sp < ns) ? (h * ns + sp) * 8u : OOB;
            sx.ml_m = bload_f(rm, mo);
            sx.ml_l = bload_f(rm, mo == OOB ? OOB : mo + 4u);
        }
    }
    // ---- every wave's first D items ---------------------------------------------------------------------------------------------
    issue(std::integral_constant<int, 0>{}, wid);
    issue(std::integral_constant<int, 1>{}, wid + NW);
    if constexpr (D == 3) issue(std::integral_constant<int, 2>{}, wid + 2u * NW);
    if (tid < d.tpw) cnt[tid] = 0u;

    // ---- MODE P prologue: combine | rmsnorm, Q80 quantization (tensor.c:21-46) into the compact fragment layout ---------------------
    if constexpr (MODE == G6_P) {
        if constexpr (COMB) {
            const bool pre_ml = a.attn_n_head * 8u <= 512u;           // every (head, split) pair has its own thread
            if (pre_ml) combine_weights<1, true>(a, wgt, sx.ml_m, sx.ml_l); else combine_weights<1, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                const uint32_t i = (tid + (uint32_t)j * 512u) * 4u;
                const float *wg = wgt + (size_t)((i < n ? i : 0u) / a.attn_hd) * 8u;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sp = 0; sp < 8; sp++) {                      // splits >= nsplit: partial read as 0, weight 0 (gemv_q80_impl.h)
                    const float w = wg[sp];
                    acc.x += sx.pv[j][sp].x * w; acc.y += sx.pv[j][sp].y * w; acc.z += sx.pv[j][sp].z * w; acc.w += sx.pv[j][sp].w * w;
                }
                sx.x[0][j] = acc;
            }
        }
        float ss[NBC];
#pragma unroll
        for (int b = 0; b < NBC; b++) ss[b] = 1.0f;
        if (norm) {                     // rmsnorm scale (infer.c:603-609): the 512-thread tree (quant_rows_frag_kernel repeats it for batches > 8)
#pragma unroll
            for (int b = 0; b < NBC; b++) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc += sx.x[b][j].x * sx.x[b][j].x; acc += sx.x[b][j].y * sx.x[b][j].y;
                    acc += sx.x[b][j].z * sx.x[b][j].z; acc += sx.x[b][j].w * sx.x[b][j].w;
                }
                acc = dpp_wave_sum(acc);
                if (lane == 0) red[b * 8 + wid] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NBC; b++) {
                float t = 0.0f;
#pragma unroll
                for (int w = 0; w < 8; w++) t += red[b * 8 + w];
                t /= (float)n; t += 1e-5f;
                ss[b] = 1.0f / sqrtf(t);
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (tid + (uint32_t)j * 512u) * 4u;
            const uint32_t g = i >> 6, q4 = (i >> 4) & 3u, e = i & 15u;
#pragma unroll
            for (int b = 0; b < NBC; b++) {
                float4 v = sx.x[b][j];
                if (norm) {
                    v.x = sx.nw[j].x * (ss[b] * v.x); v.y = sx.nw[j].y * (ss[b] * v.y);
                    v.z = sx.nw[j].z * (ss[b] * v.z); v.w = sx.nw[j].w * (ss[b] * v.w);
                }
                float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                mx = dpp_group_max<16>(mx);                           // a group of 64 = 16 consecutive threads
                const float scale = div_const<127>(mx);
                if (i < n) {
                    const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
                    *reinterpret_cast<uint32_t *>(xqc + (size_t)g * 64u * NBC + (size_t)q4 * 16u * NBC + (size_t)b * 16u + e) =
                        (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
                    if ((tid & 15u) == 0u) xs_l[g * 16u + (uint32_t)b] = scale;
                }
            }
        }
    }
    __syncthreads();                                                   // counters armed, (P) the quantized activation is in LDS

    // ---- the items of this wave ---------------------------------------------------------------------------------------------------
    auto consume = [&](auto J, uint32_t it) {
        constexpr int sl = decltype(J)::value;
        const TI t = decode(it);
        const uint32_t g0 = t.u * 8u;
        // 1. weight pieces -> transposition buffer; weight scales (and, F, activation scales) -> LDS
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) *reinterpret_cast<int4 *>(wbuf + (size_t)(2 * r8 + (int)(lane >> 5)) * G6_PITCH + (lane & 31u) * 16u) = ring[sl].w[r8];
        if (lane < 16u) {
            const uint32_t r = lane >> 1, gq = (lane & 1u) * 4u;
            wsl[(gq + 0u) * 16u + r] = ring[sl].s0.x; wsl[(gq + 1u) * 16u + r] = ring[sl].s0.y; wsl[(gq + 2u) * 16u + r] = ring[sl].s0.z; wsl[(gq + 3u) * 16u + r] = ring[sl].s0.w;
            wsl[(gq + 0u) * 16u + 8u + r] = ring[sl].s1.x; wsl[(gq + 1u) * 16u + 8u + r] = ring[sl].s1.y; wsl[(gq + 2u) * 16u + 8u + r] = ring[sl].s1.z; wsl[(gq + 3u) * 16u + 8u + r] = ring[sl].s1.w;
        }
        if constexpr (MODE == G6_F) { if (lane < 32u) *reinterpret_cast<float4 *>(xslw + lane * 4u) = ring[sl].xs; }
        // 2. the finisher of the tile (owner of its last unit) asks for what its epilogue needs now
        const bool fin = t.u == nu - 1u;
        const uint32_t half = kq >> 1, rr0 = (kq & 1u) * 4u;
        const uint32_t orow0 = t.lrow0 + half * halfoff + rr0;          // output row of c[0] (SwiGLU: lanes kq < 2 write, half 0)
        float oldv[4] = {0.f, 0.f, 0.f, 0.f};
        uint32_t opos = 0;
        if (fin && m < nb) {
            if (t.ops) opos = a.pos[m];
            if (epi == GEMV_EPI_RESID) {
                const float *o = t.out + (size_t)m * t.obs + orow0;     // the residual stream is never position indexed
#pragma unroll
                for (int i = 0; i < 4; i++) if (rr0 + (uint32_t)i < hh && orow0 + (uint32_t)i < t.rows0) oldv[i] = o[i];
            }
        }
        // (P: the slot's registers are free once the pieces are in LDS -- the next item of this wave goes out now)
        if constexpr (MODE == G6_P) issue(J, it + (uint32_t)D * NW);
        // 3. eight groups: A fragment from LDS, one MFMA, products, the unit sum in ascending group order
        float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const i32x4 fa = *reinterpret_cast<const i32x4 *>(wbuf + (size_t)m * G6_PITCH + j * 64u + kq * 16u);
            i32x4 fb; float xsc;
            if constexpr (MODE == G6_F) { fb = ring[sl].b[j]; xsc = xslw[j * 16u + m]; }
            else {
                const bool okb = m < (uint32_t)NBC && g0 + j < ng;
                fb = okb ? *reinterpret_cast<const i32x4 *>(xqc + (size_t)(g0 + j) * 64u * NBC + (size_t)kq * 16u * NBC + (size_t)m * 16u) : i32x4{0, 0, 0, 0};
                xsc = okb ? xs_l[(g0 + j) * 16u + m] : 0.0f;
            }
            const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, v4i{0, 0, 0, 0}, 0, 0, 0);
            const float4 wv = *reinterpret_cast<const float4 *>(wsl + j * 16u + kq * 4u);
            const float p0 = ((float)cv[0] * wv.x) * xsc, p1 = ((float)cv[1] * wv.y) * xsc;      // infer.c:672
            const float p2 = ((float)cv[2] * wv.z) * xsc, p3 = ((float)cv[3] * wv.w) * xsc;
            if (j == 0) { S[0] = p0; S[1] = p1; S[2] = p2; S[3] = p3; }
            else if (g0 + j < ng) { S[0] += p0; S[1] += p1; S[2] += p2; S[3] += p3; }           // (wave-uniform: a row's last unit may hold 4 groups)
        }
        // 4. arrive, or fold the tile and finish it
        if (!fin) {
            *reinterpret_cast<float4 *>(T + ((size_t)t.tl * (nu - 1u) + t.u) * 256u + lane * 4u) = make_float4(S[0], S[1], S[2], S[3]);
            if (lane == 0u) __hip_atomic_fetch_add(cnt + t.tl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            float tot[4] = {S[0], S[1], S[2], S[3]};
            if (nu > 1u) {
                // (bounded: a miscounted tile must not hang the device -- 2^24 naps are ~0.5 s, the results are then wrong and the tests say so)
                for (uint32_t spin = 0; g6_lds_load_acq(cnt + t.tl) != nu - 1u && spin < (1u << 24); spin++) __builtin_amdgcn_s_sleep(1);
                const float *tp = T + (size_t)t.tl * (nu - 1u) * 256u + lane * 4u;
                float4 acc = *reinterpret_cast<const float4 *>(tp);
                for (uint32_t u0 = 1; u0 < nu - 1u; u0 += 4) {          // units ascending; the reads of four units go out together
                    float4 q[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) q[k] = (u0 + k < nu - 1u) ? *reinterpret_cast<const float4 *>(tp + (size_t)(u0 + k) * 256u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) if (u0 + k < nu - 1u) { acc.x += q[k].x; acc.y += q[k].y; acc.z += q[k].z; acc.w += q[k].w; }
                }
                tot[0] = acc.x + S[0]; tot[1] = acc.y + S[1]; tot[2] = acc.z + S[2]; tot[3] = acc.w + S[3];
            }
            float v3[4] = {0.f, 0.f, 0.f, 0.f};
            if (sw) {                                                  // W3's values live 32 lanes up (rows 8..15 of the tile)
#pragma unroll
                for (int i = 0; i < 4; i++) v3[i] = __shfl_xor(tot[i], 32, 64);
            }
            if (t.live && m < nb && (!sw || kq < 2u)) {
                float *o = t.out + (size_t)m * t.obs + (size_t)opos * t.ops + orow0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (rr0 + (uint32_t)i < hh && orow0 + (uint32_t)i < t.rows0) o[i] = finish_epi(epi, tot[i], v3[i], oldv[i]);
            }
        }
        if constexpr (MODE == G6_F) issue(J, it + (uint32_t)D * NW);  // (F: the fragments were read by the MFMAs above)
    };
    for (uint32_t s = 0;; s += (uint32_t)D) {
        const uint32_t it = wid + s * NW;
        if (it >= nitems) break;
        consume(std::integral_constant<int, 0>{}, it);
        if (it + NW >= nitems) break;
        consume(std::integral_constant<int, 1>{}, it + NW);
        if constexpr (D == 3) {
            if (it + 2u * NW >= nitems) break;
            consume(std::integral_constant<int, 2>{}, it + 2u * NW);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
static uint32_t g6_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}

struct G6Plan { uint32_t hh, ntiles, tc0, tc1, grid, tpw, nw, nu; size_t lds_common; };

// tile height fitted to the chip: minimise the rows the busiest workgroup streams (+ a per-tile overhead worth ~2 rows)
static bool g6_plan(const GemvArgs &a, G6Plan &p) {
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    const uint32_t cus = a.cus ? a.cus : 256u, nseg = sw ? 1u : a.nseg, ng = a.n / 64u;
    p.nu = (ng + 7u) / 8u;
    uint32_t best = 0, best_cost = ~0u;
    for (uint32_t hh = 1; hh <= 8; hh++) {
        const uint32_t trw = sw ? hh : 2u * hh;
        uint32_t tiles = 0;
        for (uint32_t s = 0; s < nseg; s++) tiles += (a.seg[s].rows + trw - 1) / trw;
        const uint32_t grid = tiles < cus ? tiles : cus, tpw = (tiles + grid - 1) / grid;
        const uint32_t cost = tpw * (trw * (sw ? 2u : 1u) + 2u);
        if (cost <= best_cost) { best_cost = cost; best = hh; }       // ties: the taller tile
    }
    if (const char *e = getenv("NANO_G6_HH")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v <= 8) best = v; }   // measurement knob
    p.hh = best;
    const uint32_t trw = sw ? best : 2u * best;
    uint32_t tiles = 0, tc[2] = {0xffffffffu, 0xffffffffu};
    for (uint32_t s = 0; s < nseg; s++) { tiles += (a.seg[s].rows + trw - 1) / trw; if (s < 2) tc[s] = tiles; }
    p.ntiles = tiles; p.tc0 = nseg > 1 ? tc[0] : 0xffffffffu; p.tc1 = nseg > 2 ? tc[1] : 0xffffffffu;
    p.grid = tiles < cus ? tiles : cus; p.tpw = (tiles + p.grid - 1) / p.grid;
    const uint32_t items = p.tpw * p.nu;
    p.nw = items < G6_NW ? items : G6_NW;
    p.lds_common = (size_t)p.nw * G6_LDS_WAVE + (size_t)p.tpw * (p.nu - 1u) * 1024u + (size_t)((p.tpw + 3u) & ~3u) * 4u;
    const uint32_t magic = (65536u + p.nu - 1u) / p.nu;
    for (uint32_t it = 0; it < items + 4u * G6_NW; it++) if (((it * magic) >> 16) != it / p.nu) return false;
    return true;
}

static bool g6_common_ok(const GemvArgs &a) {
    if (a.gs != 64 || a.nb == 0 || a.n % 256u || a.nseg == 0 || a.nseg > 3 || a.resid_add || a.tile_max) return false;
    if (a.epi == GEMV_EPI_SWIGLU && (a.nseg != 2 || a.seg[0].rows != a.seg[1].rows)) return false;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    for (uint32_t s = 0; s < nseg; s++) if ((uint64_t)a.seg[s].rows * a.n >= (1ull << 32) - (1u << 20)) return false;   // 32-bit buffer offsets per segment
    if (g6_rows(a) >= 65536u) return false;                            // (the classifier has kernels of its own: STREAM / GC)
    return true;
}

template <int MODE, bool COMB, int NBC, int NV>
static hipError_t g6_launch_t(const G6Dev &d, size_t lds, hipStream_t st) {
    auto kern = &gemm_q80_g6_kernel<MODE, COMB, NBC, NV>;
    static std::atomic<bool> armed[64];
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !armed[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) armed[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(d.grid), dim3(d.nw * 64u), lds, st, d);
    return hipGetLastError();
}

static G6Dev g6_dev(const GemvArgs &a, const G6Plan &p) {
    G6Dev d{};
    d.g = to_dev(a);
    d.g.nthr = p.nw * 64u;
    d.hh = p.hh; d.nu = p.nu; d.magic_nu = (65536u + p.nu - 1u) / p.nu;
    d.ntiles = p.ntiles; d.tc0 = p.tc0; d.tc1 = p.tc1; d.grid = p.grid; d.tpw = p.tpw; d.nw = p.nw;
    return d;
}

}  // namespace

// MODE F: activations already quantized, MFMA B-fragment order (a.xq_in / a.xs_in), up to 16 tokens
bool gemm_q80_g6_supports(const GemvArgs &a) {
    if (!g6_common_ok(a) || a.nb > 16 || a.attn_part) return false;
    G6Plan p;
    return g6_plan(a, p) && p.lds_common + 64 <= 160u * 1024u;
}
hipError_t launch_gemm_q80_g6(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g6_supports(a)) return hipErrorInvalidValue;
    G6Plan p;
    if (!g6_plan(a, p)) return hipErrorInvalidValue;
    G6Dev d = g6_dev(a, p);
    d.xf = a.xq_in; d.xsf = a.xs_in;
    return g6_launch_t<G6_F, false, 1, 1>(d, p.lds_common + 64, st);
}

// MODE P: fp32 activations (a.xin | the split-attention partials), 1..8 sequences; rmsnorm / combine + quantization in the prologue
static bool g6p_shape(const GemvArgs &a, uint32_t &nbc, uint32_t &nv) {
    nbc = a.nb <= 1 ? 1u : a.nb <= 2 ? 2u : a.nb <= 4 ? 4u : 8u;
    nv = a.n <= 4096u ? 2u : a.n <= 10240u ? 5u : 0u;
    if (!nv || nbc * nv > 16u) return false;
    if (a.attn_part && (a.nb != 1 || nv != 2u || a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4 || a.attn_n_head > 128)) return false;
    return true;
}
bool gemm_q80_g6p_supports(const GemvArgs &a) {
    if (!g6_common_ok(a) || a.nb > 8 || a.xq_in || (!a.xin && !a.attn_part)) return false;
    uint32_t nbc, nv;
    if (!g6p_shape(a, nbc, nv)) return false;
    G6Plan p;
    if (!g6_plan(a, p) || p.nw != G6_NW) return false;               // the prologue's tree is the 512-thread one
    const uint32_t ng = a.n / 64u;
    const size_t lds = p.lds_common + (size_t)ng * 64u * nbc + (size_t)ng * 64u + (size_t)nbc * 32u + (a.attn_part ? (size_t)a.attn_n_head * 32u : 0u) + 64u;
    return lds <= 160u * 1024u;
}
hipError_t launch_gemm_q80_g6p(const GemvArgs &a, hipStream_t st) {
    if (!gemm_q80_g6p_supports(a)) return hipErrorInvalidValue;
    uint32_t nbc, nv;
    G6Plan p;
    if (!g6p_shape(a, nbc, nv) || !g6_plan(a, p)) return hipErrorInvalidValue;
    G6Dev d = g6_dev(a, p);
    const uint32_t ng = a.n / 64u;
    const size_t lds = p.lds_common + (size_t)ng * 64u * nbc + (size_t)ng * 64u + (size_t)nbc * 32u + (a.attn_part ? (size_t)a.attn_n_head * 32u : 0u) + 64u;
    if (a.attn_part) return g6_launch_t<G6_P, true, 1, 2>(d, lds, st);
    if (nv == 2u) {
        if (nbc == 1) return g6_launch_t<G6_P, false, 1, 2>(d, lds, st);
        if (nbc == 2) return g6_launch_t<G6_P, false, 2, 2>(d, lds, st);
        if (nbc == 4) return g6_launch_t<G6_P, false, 4, 2>(d, lds, st);
        return g6_launch_t<G6_P, false, 8, 2>(d, lds, st);
    }
    if (nbc == 1) return g6_launch_t<G6_P, false, 1, 5>(d, lds, st);
    if (nbc == 2) return g6_launch_t<G6_P, false, 2, 5>(d, lds, st);
    return hipErrorInvalidValue;
}

}  // namespace nano
