// gemm_q80_g6_impl.h -- the G6 kernel (see gemm_q80_g6.hip for the design); included by the translation units that instantiate it
// (gemm_q80_g6.hip).
#pragma once
#include <atomic>
#include <type_traits>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

enum : int { G6_F = 0, G6_S = 2 };
constexpr uint32_t G6_PITCH = 528, G6_WBUF = 16 * G6_PITCH;
constexpr uint32_t G6_LDS_WAVE = G6_WBUF + 512 + 512;          // + weight scales [8 groups][16 rows] + (F) activation scales [8][16 tokens]
constexpr uint32_t G6_NW = 8;                                   // waves of a workgroup (launches with fewer items use fewer)

struct G6Dev {
    GemvDev g;                          // segments, n, ng, epi, flags, nb
    const int8_t *xf; const float *xsf; // MODE F: activations in MFMA B-fragment order [group][lane][16 B], scales [group][16 tokens]
    uint32_t hh;                        // live rows per half tile (1..8)
    uint32_t nu, magic_nu;              // units per row; (it * magic_nu) >> 16 == it / (nu * tts) for every item index of a workgroup
    uint32_t tts, tts_log2;             // MODE F, small launches: token tiles SPREAD over the waves (1 | 2 | 4): an item is (tile, unit, token tile)
    uint32_t ntiles, tc0, tc1;          // tiles; tiles up to the end of segment 0 / 1
    uint32_t grid, tpw;                 // workgroups; tiles per workgroup (max)
    uint32_t nw, full;                  // waves per workgroup (a power of two); workgroups that own tpw tiles (the others: tpw - 1)
};

__device__ __forceinline__ uint32_t g6_lds_load_acq(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int R0, int R1, class F> __device__ __forceinline__ void g6_static_for(F &&f) {
    if constexpr (R0 < R1) { f(std::integral_constant<int, R0>{}); g6_static_for<R0 + 1, R1>(f); }
}

// R  = rounds: the most items a wave of the launch owns (compile time: R issues and R consumes in straight-line code, nothing dead)
// MS = several weight segments share the launch (q | k | v): a tile looks its segment up; single-segment launches skip that
// TT = token tiles of 16 (MODE F: 1 | 2 | 4 -- up to 64 tokens; an item's weights are transposed once and multiplied with every token
//      tile's fragments, tile t + 1's being fetched from L2 while tile t is multiplied; the other modes: 1)
template <int MODE, int NV, int R, bool MS, int TT = 1>
__global__ __launch_bounds__(512, 2) void gemm_q80_g6_kernel(const G6Dev d) {
    static_assert(TT == 1 || MODE == G6_F, "token tiles beyond the first come from L2 (MODE F)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = MODE == G6_F ? 2 : 4;                             // register slots per wave (F: a slot also holds the item's 8 KB of B fragments)
    constexpr int LA = MODE == G6_F ? D : 2;                            // rounds requested ahead of the one being multiplied
    const GemvDev &a = d.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t m = lane & 15u, kq = lane >> 4;
    karg_touch(a.out[0]); karg_touch(a.out[1]); karg_touch(a.out[2]); karg_touch(a.pos); karg_touch(d.xf); karg_touch(d.xsf);
    karg_touch(a.out_bstride[0]); karg_touch(a.out_pstride[2]); karg_touch(d.tc0); karg_touch(d.magic_nu);
    NANO_STAMP(a.stamps, 0, tid);
    const uint32_t n = a.n, ng = a.ng, nu = d.nu, hh = d.hh, NW = d.nw, nb = a.nb;
    const uint32_t tts = MODE == G6_F ? d.tts : 1u, tsh = MODE == G6_F ? d.tts_log2 : 0u, ipt = nu << tsh;   // items per tile
    const uint32_t NT = (uint32_t)TT << tsh;                              // token tiles of the unit-sum table
    const uint32_t epi = a.epi;
    const bool sw = epi == GEMV_EPI_SWIGLU;
    const uint32_t halfoff = sw ? 0u : hh;                              // rows between the two halves of a tile
    const uint32_t ngp = nu * 8u;                                       // groups incl. the padding of a row's last unit (ng % 8 == 4)

    // ---- LDS ---------------------------------------------------------------------------------------------------------------
    int8_t *wbuf = reinterpret_cast<int8_t *>(smem) + (size_t)wid * G6_LDS_WAVE;
    float *wsl = reinterpret_cast<float *>(wbuf + G6_WBUF);            // [8 groups][16 rows]
    float *xslw = wsl + 128;                                           // MODE F: [8 groups][16 tokens]
    float *T = reinterpret_cast<float *>(smem + (size_t)NW * G6_LDS_WAVE);          // [tpw][nu][NT token tiles][256] unit sums
    uint32_t *cnt = reinterpret_cast<uint32_t *>(T + (size_t)d.tpw * nu * NT * 256u);   // [tpw] items arrived
    unsigned char *pbase = reinterpret_cast<unsigned char *>(cnt + ((d.tpw + 3u) & ~3u));
    // MODE S: the fragment-order activation itself [ngp][64 lanes][16 B] (16 token slots per group's 64-byte k-quarter), scales [ngp][16]
    constexpr uint32_t SLOTS = 16u;
    int8_t *xqc = reinterpret_cast<int8_t *>(pbase);
    unsigned char *zblk = pbase + (size_t)ngp * 64u * SLOTS;
    float *xs_l = reinterpret_cast<float *>(zblk + 64);

    // ---- lane parts of every address of the item loop (item-invariant) --------------------------------------------------------------
    // weight pieces of an item: 16 rows x 512 B, two rows per load instruction: load k (and k + 4, the other half) reads row
    // 2k + l/32 of its half, bytes 16 (l % 32) .. +15.  Rows >= hh: out of range -> 0.  Everything item dependent (tile row, unit,
    // the end of the segment) sits in the descriptor's base and size, which are scalars.
    const uint32_t r_lo = lane >> 5, c16 = (lane & 31u) * 16u;
    uint32_t voff[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { const uint32_t r = 2u * k + r_lo; voff[k] = r < hh ? r * n + c16 : OOB; }
    const uint32_t svoff = (lane < 16u && (lane >> 1) < hh) ? ((lane >> 1) * ng + (lane & 1u) * 4u) * 4u : OOB;   // weight scales: row l/2, 4 groups
    int8_t *wb_w = wbuf + (size_t)r_lo * G6_PITCH + c16;               // + 2 r8 rows
    const int8_t *wb_r = wbuf + (size_t)m * G6_PITCH + kq * 16u;       // + 64 j
    float *wsl_w = wsl + ((lane & 1u) * 4u) * 16u + (lane >> 1);       // lanes 0..15: [4 (l%2) + k][row l/2], half 1: + 8
    const float *wsl_r = wsl + kq * 4u;                                // + 16 j
    // MODE S: token slot m of a group's staged fragments
    const uint32_t pb_off = m < SLOTS ? kq * 16u * SLOTS + m * 16u : (uint32_t)(zblk - reinterpret_cast<unsigned char *>(xqc));
    const uint32_t pb_str = m < SLOTS ? 64u * SLOTS : 0u;
    const uint32_t px_off = m < SLOTS ? m : 15u, px_str = 16u;       // 

    // ---- the workgroup's items -----------------------------------------------------------------------------------------------
    const uint32_t bid = blockIdx.x;
    const uint32_t ntl = bid < d.full ? d.tpw : d.tpw - 1u;              // tiles of this workgroup
    const uint32_t nitems = ntl * ipt;

    struct TI { uint32_t lrow0, rows0, obs, ops; const int8_t *wA, *wB; const float *sA, *sB; float *out; };
    auto decode = [&](uint32_t tl) -> TI {
        TI t;
        const uint32_t tile = bid + tl * d.grid;
        const int sel = !MS ? 0 : (int)(tile >= d.tc0) + (int)(tile >= d.tc1);
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
    // MODE F with several token tiles per item (TT > 1; 17..64 tokens): the fragments of this wave's (item, token tile) BLOCKS run through a
    // register queue of their own -- the next block in flight behind the one being multiplied, across item boundaries (round 6, below);
    // the slots then hold weights only.  (A three-deep queue measured the same step time -- the blocks come from L2 in ~300 cycles,
    // TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ -- and its 32 registers make the 3-round variants spill.)
    constexpr bool STREAM_B = MODE == G6_F && TT > 1;
    constexpr int QD = 2;
    struct Slot { int4 w[8]; float4 s0, s1; i32x4 b[(MODE == G6_F && !STREAM_B) ? 8 : 1]; float4 xs; };
    Slot ring[D];
    auto issue = [&](auto J, uint32_t it) {
        constexpr int sl = decltype(J)::value;
        const bool live = it < nitems;
        const uint32_t tl = (it * d.magic_nu) >> 16, rem = it - tl * ipt, u = rem >> tsh, tk = rem & (tts - 1u);
        const TI t = decode(tl);
        if constexpr (MODE == G6_F && !STREAM_B) {  // the item's activation fragments FIRST (loads return in issue order)
            const uint32_t g0 = u * 8u;
            const bool lvb = live && tk * 16u < nb;  // (spread token tiles: tile tk of this item; serial ones: tile 0 here, the others in consume)
            const __amdgpu_buffer_rsrc_t rxf = mkrsrc(d.xf + ((size_t)tk * ng + g0) * 1024u, lvb ? (ng - g0) * 1024u : 0u);     // groups >= ng: out of range -> 0
            const __amdgpu_buffer_rsrc_t rxs = mkrsrc(d.xsf + ((size_t)tk * ng + g0) * 16u, lvb ? (ng - g0) * 64u : 0u);
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) ring[sl].b[j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(lane * 16u + j * 1024u), 0, 0);
            ring[sl].xs = bload_f4(rxs, lane < 32u ? lane * 16u : OOB);             // lanes 0..31: group g0 + l/4, tokens 4 (l%4) .. +3
        }
        // descriptors of the two halves: base = first row of the half, this unit's 512 bytes; size = what is left of the segment (a row
        // beyond it, i.e. a dead item or the ragged last tile, reads 0).  A row's last unit may be half a unit (ng % 8 == 4): its upper
        // 256 bytes then belong to the next row -- finite int8 values that meet activation bytes which are zero there.
        const uint32_t rowA = t.lrow0, rowB = t.lrow0 + halfoff;
        const uint32_t offA = rowA * n + u * 512u, offB = rowB * n + u * 512u, end = t.rows0 * n;
        const __amdgpu_buffer_rsrc_t rA = mkrsrc(t.wA + offA, (live && rowA < t.rows0) ? end - offA : 0u);
        const __amdgpu_buffer_rsrc_t rB = mkrsrc(t.wB + offB, (live && rowB < t.rows0) ? end - offB : 0u);
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) ring[sl].w[r8] = bload_w(r8 >= 4 ? rB : rA, voff[r8 & 3]);
        const uint32_t sofA = rowA * ng + u * 8u, sofB = rowB * ng + u * 8u, send = t.rows0 * ng;
        const __amdgpu_buffer_rsrc_t qA = mkrsrc(t.sA + sofA, (live && rowA < t.rows0) ? (send - sofA) * 4u : 0u);
        const __amdgpu_buffer_rsrc_t qB = mkrsrc(t.sB + sofB, (live && rowB < t.rows0) ? (send - sofB) * 4u : 0u);
        ring[sl].s0 = bload_f4(qA, svoff);
        ring[sl].s1 = bload_f4(qB, svoff);
    };

    // ---- MODE S: the workgroup's copy of the fragment-order activation (quant_rows_frag_kernel / the attention kernel wrote it): thread
    //      t fetches the 16-byte units t, t + 512, ... (NV of them; a group is 64 units) and one float4 of the scales -- asked for before
    //      any weight, parked in LDS behind the first barrier, read by every item of every tile of the workgroup -------------------------
    i32x4 sb[MODE == G6_S ? NV : 1]; float4 sxs = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == G6_S) {
        const __amdgpu_buffer_rsrc_t rxf = mkrsrc(d.xf, ng * 1024u);
        const __amdgpu_buffer_rsrc_t rxs = mkrsrc(d.xsf, ng * 64u);
#pragma unroll
        for (int j = 0; j < NV; j++) sb[j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)((tid + (uint32_t)j * 512u) * 16u), 0, 0);   // beyond ng groups: 0
        sxs = bload_f4(rxs, tid * 16u);
    }
    // ---- what the first tile this wave FINISHES needs for its epilogue (the old residual values, the position of a position-indexed
    //      output): asked for first, tiny, and only by launches that need them -- nothing at the end waits for a cold load of its own ---
    // FINISHING is dealt out too: pair q = (tile q / NT, token tile q % NT) is folded by wave NW - 1 - q % NW (the high waves own one item
    // less when the items do not divide evenly) once the tile's counter says every item has arrived -- round 4 first let the owner of a
    // tile's last unit fold all its token tiles: 2.8 ... 4.4 us of a 64-token launch with seven waves idle (profiles/r04_stamps_64.txt)
    constexpr uint32_t NTL2 = (TT == 4 ? 2u : TT == 2 ? 1u : 0u);
    const uint32_t ntsh = NTL2 + tsh, npairs = ntl << ntsh, q0 = NW - 1u - wid;
    const uint32_t ftl = q0 < npairs ? (q0 >> ntsh) : 0xffffffffu, ftt = q0 & (NT - 1u);
    const uint32_t half = kq >> 1, rr0 = (kq & 1u) * 4u;              // this lane's four output rows: rows rr0 .. rr0 + 3 of half `half`
    float oldv0[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t opos0 = 0;
    if (epi == GEMV_EPI_RESID || (a.out_pstride[0] | a.out_pstride[1] | a.out_pstride[2]) != 0u) {
        const TI t = decode(ftl == 0xffffffffu ? 0u : ftl);
        const uint32_t orow0 = t.lrow0 + half * halfoff + rr0;
        const uint32_t tok0 = ftt * 16u + m;
        const bool lv = ftl != 0xffffffffu && tok0 < nb;
        const __amdgpu_buffer_rsrc_t ro = mkrsrc(t.out, (ftl != 0xffffffffu && epi == GEMV_EPI_RESID) ? ((nb - 1u) * t.obs + t.rows0) * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rp = mkrsrc(a.pos, (ftl != 0xffffffffu && t.ops) ? nb * 4u : 0u);
        opos0 = __builtin_amdgcn_raw_buffer_load_b32(rp, (int)(lv ? tok0 * 4u : OOB), 0, 0);
#pragma unroll
        for (uint32_t i = 0; i < 4; i++)                               // (the residual stream is never position indexed)
            oldv0[i] = bload_f(ro, (lv && rr0 + i < hh && orow0 + i < t.rows0) ? (tok0 * t.obs + orow0 + i) * 4u : OOB);
    }
    // ---- the first items go out.  MODE F: every round that has a slot, right away (its one barrier -- arming the counters -- is behind
    //      it: nothing waits on a wave that the memory pipeline holds up while it issues).  MODE S: ROUND 0 ONLY -- the staging below has a
    //      barrier, and a wave stuck issuing 30 KB into a full memory pipeline keeps the whole workgroup from its first multiply ---------
    // STREAM_B: block s of this wave = (its round s / TT, token tile s % TT) -> queue slot s % QD.  Every load of the stream is unconditional
    // (a block of a dead item or of a token tile beyond the batch reads through a zero-sized descriptor: zeros, no memory access) and sits
    // OUTSIDE the branches around the arithmetic, so the compiler's vmcnt counts stay exact: multiplying block s waits for block s alone.
    constexpr int NBLK = STREAM_B ? R * TT : 0;
    i32x4 fq[STREAM_B ? QD : 1][8]; float4 fx[STREAM_B ? QD : 1];
    auto fissue = [&](auto S) {
        constexpr int sb_ = decltype(S)::value, qi = sb_ % QD, r = sb_ / TT, t = sb_ % TT;
        const uint32_t it = wid + (uint32_t)r * NW;
        const uint32_t tl = (it * d.magic_nu) >> 16, u = it - tl * ipt, g0 = u * 8u;          // (serial token tiles: ipt = nu)
        const bool lv = it < nitems && (uint32_t)t * 16u < nb;
        const __amdgpu_buffer_rsrc_t rxf = mkrsrc(d.xf + ((size_t)t * ng + g0) * 1024u, lv ? (ng - g0) * 1024u : 0u);      // groups >= ng: out of range -> 0
        const __amdgpu_buffer_rsrc_t rxs = mkrsrc(d.xsf + ((size_t)t * ng + g0) * 16u, lv ? (ng - g0) * 64u : 0u);
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) fq[qi][j] = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(lane * 16u + j * 1024u), 0, 0);
        fx[qi] = bload_f4(rxs, lane < 32u ? lane * 16u : OOB);                      // lanes 0..31: group g0 + l/4, tokens 4 (l%4) .. +3
    };
    if (tid < d.tpw) cnt[tid] = 0u;
    if constexpr (MODE == G6_F) __syncthreads();
    if constexpr (STREAM_B) {                                          // block 0, round 0's weights, round 1's weights: in the order they are needed
        fissue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 0>{}, wid);
        if constexpr (R > 1) issue(std::integral_constant<int, 1>{}, wid + NW);
    } else
    g6_static_for<0, (MODE == G6_F ? (R < D ? R : D) : 1)>([&](auto K) { issue(K, wid + (uint32_t)decltype(K)::value * NW); });
    NANO_STAMP(a.stamps, 1, opos0);                                 // the loads of the first round(s) issued

    if constexpr (MODE == G6_S) {                                      // (loads beyond the ng groups returned 0: the padding groups of a row's last unit)
#pragma unroll
        for (int j = 0; j < NV; j++) { const uint32_t un = tid + (uint32_t)j * 512u; if (un < ngp * 64u) *reinterpret_cast<i32x4 *>(xqc + (size_t)un * 16u) = sb[j]; }
        if (tid < ngp * 4u) *reinterpret_cast<float4 *>(xs_l + tid * 4u) = sxs;
    }
    if constexpr (MODE != G6_F) {
        __syncthreads();                                               // counters armed, the quantized activation is in LDS
        g6_static_for<1, (R < LA ? R : LA)>([&](auto K) { issue(K, wid + (uint32_t)decltype(K)::value * NW); });   // round 1 (see LA)
    }
    NANO_STAMP(a.stamps, 2, cnt[0]);                                // (S) the activation is staged

    // ---- the items of this wave ---------------------------------------------------------------------------------------------------
    // consume: slot registers -> LDS, eight MFMAs, the unit sum into the table, the tile's counter.  No load, no store, no wait for
    // another wave: straight-line code whose s_waitcnt counts the compiler gets exactly right (round 4's first build looped over the
    // slots -- at the loop header the compiler's counter merged to "wait for everything", and every wave multiplied only after ALL
    // its weights had landed: 21 us for a launch whose stream lasts 8).
    auto consume = [&](auto J, auto FIRST, uint32_t it) {
        constexpr int sl = decltype(J)::value;
        constexpr bool first = decltype(FIRST)::value;
        const uint32_t tl = (it * d.magic_nu) >> 16, rem = it - tl * ipt, u = rem >> tsh, tk = rem & (tts - 1u);
        const uint32_t g0 = u * 8u;
        // 1. weight pieces -> transposition buffer; weight scales (and, F, activation scales) -> LDS
#pragma unroll
        for (int r8 = 0; r8 < 8; r8++) *reinterpret_cast<int4 *>(wb_w + (size_t)(2 * r8) * G6_PITCH) = ring[sl].w[r8];
        if (lane < 16u) {
            wsl_w[0] = ring[sl].s0.x; wsl_w[16] = ring[sl].s0.y; wsl_w[32] = ring[sl].s0.z; wsl_w[48] = ring[sl].s0.w;
            wsl_w[8] = ring[sl].s1.x; wsl_w[24] = ring[sl].s1.y; wsl_w[40] = ring[sl].s1.z; wsl_w[56] = ring[sl].s1.w;
        }
        if constexpr (MODE == G6_F) { if (lane < 32u) *reinterpret_cast<float4 *>(xslw + lane * 4u) = ring[sl].xs; }
        if constexpr (first) NANO_STAMP(a.stamps, 3, (float)ring[sl].w[7].x + ring[sl].s1.x);     // this wave's first weights (and scales) arrived
        // 2. eight groups -- A fragment from LDS, one MFMA, products, the unit sum in ascending group order (groups >= ng of a row's last
        //    unit: zero activation bytes and scales -> products +0.0f); the sum goes to the tile's table
        float S[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            const i32x4 fa = *reinterpret_cast<const i32x4 *>(wb_r + j * 64u);
            i32x4 fb; float xsc;
            if constexpr (MODE == G6_F) { fb = ring[sl].b[j]; xsc = xslw[j * 16u + m]; }
            else {
                fb = *reinterpret_cast<const i32x4 *>(xqc + pb_off + (g0 + j) * pb_str);
                xsc = xs_l[px_off + (g0 + j) * px_str];
            }
            const v4i cv = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, v4i{0, 0, 0, 0}, 0, 0, 0);
            const float4 wv = *reinterpret_cast<const float4 *>(wsl_r + j * 16u);
            const float p0 = ((float)cv[0] * wv.x) * xsc, p1 = ((float)cv[1] * wv.y) * xsc;      // infer.c:672
            const float p2 = ((float)cv[2] * wv.z) * xsc, p3 = ((float)cv[3] * wv.w) * xsc;
            if (j == 0) { S[0] = p0; S[1] = p1; S[2] = p2; S[3] = p3; }
            else { S[0] += p0; S[1] += p1; S[2] += p2; S[3] += p3; }
        }
        if constexpr (first) NANO_STAMP(a.stamps, 4, S[3]);            // ... multiplied
        *reinterpret_cast<float4 *>(T + (((size_t)tl * nu + u) * NT + tk) * 256u + lane * 4u) = make_float4(S[0], S[1], S[2], S[3]);
        // 3. arrive
        if (lane == 0u) __hip_atomic_fetch_add(cnt + tl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // R rounds in straight-line code.  MODE F (a slot holds the item's fragments: two slots): the item of round r + 2 is requested when
    // round r's slot is free.  MODES P / S (four slots): LOOK-AHEAD of two rounds -- round r + 2 is requested right before round r is
    // multiplied.  (Round 4 first issued every round that had a slot at once: a wave then sits ~3 us in a full memory pipeline issuing
    // 30 loads while its first item has long landed -- W1|W3's first multiply 6.2 us after entry, profiles/r04_g6_stamps.txt.)
    if constexpr (STREAM_B) {
        // Round 6 -- 17..64 tokens.  Round 5 fetched token tile t + 1 of an item while it multiplied tile t: one 8-KB block in flight per wave,
        // asked for ~0.3 us before it was needed, and every tile waited out the rest of an L2 round trip (W2 of Qwen3-4B at 64 tokens: twelve
        // blocks per wave, 19.8 us where the weights stream in 4.4; SQ_WAIT_ANY + SQ_WAIT_INST_ANY 72 % of the wave cycles, profiles/
        // r06_4b_b64_pmc_before.txt).  Now the blocks form ONE stream per wave, the next one in flight behind the one being multiplied across
        // item boundaries too (64 registers where the slot copies + look-ahead pair took 128), and what the freed registers buy is below:
        // the item's A fragments and weight scales held in registers over its token tiles, eight matrix instructions back to back.
        g6_static_for<0, R>([&](auto K) {
            constexpr int r = decltype(K)::value;
            const uint32_t it = wid + (uint32_t)r * NW;
            const bool live = it < nitems;
            const uint32_t tl = (it * d.magic_nu) >> 16, u = it - tl * ipt;
            // the item's weight pieces -> transposition buffer, its weight scales -> LDS (a dead item's registers hold zeros)
#pragma unroll
            for (int r8 = 0; r8 < 8; r8++) *reinterpret_cast<int4 *>(wb_w + (size_t)(2 * r8) * G6_PITCH) = ring[r % D].w[r8];
            if (lane < 16u) {
                wsl_w[0] = ring[r % D].s0.x; wsl_w[16] = ring[r % D].s0.y; wsl_w[32] = ring[r % D].s0.z; wsl_w[48] = ring[r % D].s0.w;
                wsl_w[8] = ring[r % D].s1.x; wsl_w[24] = ring[r % D].s1.y; wsl_w[40] = ring[r % D].s1.z; wsl_w[56] = ring[r % D].s1.w;
            }
            if constexpr (r == 0) NANO_STAMP(a.stamps, 3, (float)ring[0].w[7].x + ring[0].s1.x);     // this wave's first weights (and scales) arrived
            // the item's A fragments and weight scales are the same for every token tile: read from LDS ONCE per item (round 5 re-read them per
            // tile: 16 KB of LDS reads and their latency in front of every tile's first matrix instruction; with two waves per SIMD nothing
            // covered it -- 3600 cycles per tile where the instructions need ~900, profiles/r06_4b_b64_pmc_before.txt)
            i32x4 fa[8]; float4 wv[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) fa[j] = *reinterpret_cast<const i32x4 *>(wb_r + j * 64u);
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) wv[j] = *reinterpret_cast<const float4 *>(wsl_r + j * 16u);
            g6_static_for<0, TT>([&](auto TK) {
                constexpr int t = decltype(TK)::value, sblk = r * TT + t, qi = sblk % QD;
                if (lane < 32u) *reinterpret_cast<float4 *>(xslw + lane * 4u) = fx[qi];          // this block's activation scales [8 groups][16 tokens]
                if constexpr (sblk + QD - 1 < NBLK) fissue(std::integral_constant<int, sblk + QD - 1>{});     // the next block: into the slot block sblk - 1 left
                if (live && (uint32_t)t * 16u < nb) {                                              // (wave-uniform; no load inside)
                    float xsc[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) xsc[j] = xslw[j * 16u + m];
                    v4i cv[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) cv[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[j], fq[qi][j], v4i{0, 0, 0, 0}, 0, 0, 0);     // eight matrix instructions back to back
                    float S[4];
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) {
                        const float p0 = ((float)cv[j][0] * wv[j].x) * xsc[j], p1 = ((float)cv[j][1] * wv[j].y) * xsc[j];      // infer.c:672
                        const float p2 = ((float)cv[j][2] * wv[j].z) * xsc[j], p3 = ((float)cv[j][3] * wv[j].w) * xsc[j];
                        if (j == 0) { S[0] = p0; S[1] = p1; S[2] = p2; S[3] = p3; }
                        else { S[0] += p0; S[1] += p1; S[2] += p2; S[3] += p3; }
                    }
                    if constexpr (sblk == 0) NANO_STAMP(a.stamps, 4, S[3]);                          // ... multiplied (token tile 0)
                    *reinterpret_cast<float4 *>(T + (((size_t)tl * nu + u) * NT + (uint32_t)t) * 256u + lane * 4u) = make_float4(S[0], S[1], S[2], S[3]);
                }
            });
            if (live && lane == 0u) __hip_atomic_fetch_add(cnt + tl, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (r + D < R) issue(std::integral_constant<int, r % D>{}, it + (uint32_t)D * NW);
        });
    } else
    g6_static_for<0, R>([&](auto K) {
        constexpr int r = decltype(K)::value;
        const uint32_t it = wid + (uint32_t)r * NW;
        if constexpr (MODE != G6_F && r + LA < R) issue(std::integral_constant<int, (r + LA) % D>{}, it + (uint32_t)LA * NW);
        if (it < nitems) consume(std::integral_constant<int, r % D>{}, std::integral_constant<bool, r == 0>{}, it);
        if constexpr (MODE == G6_F && r + D < R) issue(std::integral_constant<int, r % D>{}, it + (uint32_t)D * NW);
    });
    NANO_STAMP(a.stamps, 5, oldv0[0]);                                // this wave's items done
    // ---- the (tile, token tile) pairs this wave finishes: wait for the tile's items, add the units in ascending order, epilogue ---------
    for (uint32_t q = q0; q < npairs; q += NW) {
        const uint32_t tl = q >> ntsh, tt = q & (NT - 1u);
        if (tt * 16u >= nb) continue;                                  // (wave-uniform: a token tile beyond the batch)
        const TI t = decode(tl);
        const uint32_t orow0 = t.lrow0 + half * halfoff + rr0;        // output row of c[0] (SwiGLU: lanes kq < 2 write, half 0)
        const uint32_t tok = tt * 16u + m;
        float oldv[4] = {oldv0[0], oldv0[1], oldv0[2], oldv0[3]};
        uint32_t opos = opos0;
        if (q != q0 && tok < nb) {                                     // (only this wave's first pair was fetched up front)
            if (t.ops) opos = a.pos[tok];
            if (epi == GEMV_EPI_RESID) {
                const float *o = t.out + (size_t)tok * t.obs + orow0;
#pragma unroll
                for (int i = 0; i < 4; i++) if (rr0 + (uint32_t)i < hh && orow0 + (uint32_t)i < t.rows0) oldv[i] = o[i];
            }
        }
        // (bounded: a miscounted tile must not hang the device -- 2^24 naps are ~0.5 s; giving up is reported through the sticky error word)
        uint32_t spin = 0;
        for (; g6_lds_load_acq(cnt + tl) != ipt && spin < (1u << 24); spin++) __builtin_amdgcn_s_sleep(1);
        if (spin == (1u << 24) && d.g.err) __hip_atomic_fetch_or(d.g.err, NANO_DEVERR_G6_TILE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const float *tp = T + ((size_t)tl * nu * NT + tt) * 256u + lane * 4u;
        float4 acc = *reinterpret_cast<const float4 *>(tp);
        for (uint32_t u0 = 1; u0 < nu; u0 += 4) {                      // units ascending; the reads of four units go out together
            float4 qv[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) qv[k] = (u0 + k < nu) ? *reinterpret_cast<const float4 *>(tp + (size_t)(u0 + k) * NT * 256u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) if (u0 + k < nu) { acc.x += qv[k].x; acc.y += qv[k].y; acc.z += qv[k].z; acc.w += qv[k].w; }
        }
        const float tot[4] = {acc.x, acc.y, acc.z, acc.w};
        float v3[4] = {0.f, 0.f, 0.f, 0.f};
        if (sw) {                                                      // W3's values live 32 lanes up (rows 8..15 of the tile)
#pragma unroll
            for (int i = 0; i < 4; i++) v3[i] = __shfl_xor(tot[i], 32, 64);
        }
        if (tok < nb && (!sw || kq < 2u)) {
            float *o = t.out + (size_t)tok * t.obs + (size_t)opos * t.ops + orow0;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (rr0 + (uint32_t)i < hh && orow0 + (uint32_t)i < t.rows0) o[i] = finish_epi(epi, tot[i], v3[i], oldv[i]);
        }
    }
    NANO_STAMP_END(a.stamps, 6);                                    // the workgroup's last wave ends
}


// ---- launch plumbing shared by the translation units ---------------------------------------------------------------------------------
template <int MODE, int NV, int R, bool MS, int TT = 1>
static hipError_t g6_launch_t(const G6Dev &d, size_t lds, hipStream_t st) {
    auto kern = &gemm_q80_g6_kernel<MODE, NV, R, MS, TT>;
    static std::atomic<bool> armed[64];
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !armed[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) armed[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(d.grid), dim3(d.nw * 64u), lds, st, d);
    return hipGetLastError();
}

}  // namespace

}  // namespace nano
