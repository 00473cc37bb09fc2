// gemm_q80_g7.hip -- G7: the Q80 (W8A8) projection kernel of the FAST path for 17..64 tokens per weight read (decode steps of 17..64
// sequences, batched-prefill chunks) -- a loader / consumer engine: the WEIGHTS stream through a deep LDS ring by LDS-DMA, the
// activation fragments are staged in LDS once per workgroup by the consumer waves themselves, eight consumer waves multiply out of LDS
// and keep the canonical fold in registers.
//
// Why (round 5).  G6 MODE F / G5 at 2..4 token tiles spent 2.3 us per (8 KB of weights, token tile): a wave fetched the 8 KB of
// activation fragments of every (item, token tile) from L2 into registers with one tile of look-ahead (4 x 8 KB of fragments per 8 KB
// of weights, latency bound), and Qwen3-4B's W1|W3 did not fit G6 at all (100 KB of unit sums).  Here
//   * one workgroup per CU owns `tpw` row tiles (<= 16 rows each, fitted to the chip like G6's) and walks the row length in STEPS of
//     256 bytes (4 quantization groups);
//   * two loader waves DMA the tiles' weights of alternate steps (global_load_lds_dwordx4, non-temporal; 16-byte chunks XOR-swizzled
//     at the SOURCE so that the A-fragment ds_read_b128 is conflict free on row-major rows) and their scales into a ring of `nsa`
//     stages -- no VGPR holds a weight byte in flight, the ring IS the prefetch queue (tens of KB per CU, what the HBM latency asks
//     for), vmcnt is counted by hand (the loaders issue nothing else);
//   * the step's activation fragments (already in MFMA B order: lane l reads slot l) come from L2: every consumer wave fetches a
//     share of step k + 4's fragments into registers (vector loads: 64 B / clk / CU, where LDS-DMA measured ~30 GB/s per CU --
//     round 5's first build, profiles/r05_g7_first_build.txt) and parks step k + 1's in one of two LDS stages -- ONCE per
//     workgroup, whatever the number of row tiles that meet them;
//   * a consumer wave owns (row tile, token tile) PAIRS for the whole row length: one v_mfma_i32_16x16x64_i8 per group gives the exact
//     int32 group sums of (16 rows x 16 tokens), products ((float)ival * ws) * xs (infer.c:672, two roundings), the unit sum S_u of 8
//     groups in ascending order, the running row value += S_u in registers: the CANONICAL fold (kernels.h q80_canonical(),
//     tests/canon.py) without a table, a counter or a finishing pass -- a batch stays bit for bit its sequences alone;
//   * one s_barrier per step hands a landed weight stage and a parked fragment stage to the consumers and the stages consumed before
//     them back to their writers.
// Reference: matmul_quant infer/infer.c:654-679 (the arithmetic), the prompt loop :1258-1260 (what batched prefill replaces).
// MFMA operand layout as in gemm_q80.hip (verified on gfx950): lane l holds A[m = l%16][k = 16 (l/16) .. +15], B[k][n = l%16],
// c[i] = C[m = 4 (l/16) + i][n = l%16].
#include <atomic>
#include <type_traits>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr uint32_t G7_NCW = 14;                     // consumer waves
constexpr uint32_t G7_NLA = 2;                      // weight loader waves (steps k % 2)
constexpr uint32_t G7_NW = G7_NCW + G7_NLA;         // 16 waves: four per SIMD (<= 128 registers each)
constexpr uint32_t G7_MAXNSA = 32;
constexpr uint32_t G7_LDS = 160u * 1024u;
constexpr int G7_BD = 2;                            // steps of activation fragments a consumer wave keeps in flight (registers): asked for one whole step (>= 0.6 us: an
                                                    // L2 round trip is ~0.15 us) before they are parked; 2 = the steps of a unit, so slot and unit phase unroll together

struct G7Dev {
    GemvDev g;                          // segments, n, ng, epi, nb, pos
    const int8_t *xf; const float *xsf; // activations in MFMA B-fragment order [token tile][group][lane][16 B], scales [token tile][group][16 tokens]
    uint32_t hh;                        // live rows per half tile (1..8)
    uint32_t ntiles, tc0, tc1;          // tiles; tiles up to the end of segment 0 / 1
    uint32_t grid, tpw, full;           // workgroups; tiles per workgroup (max); workgroups that own tpw tiles (the others: tpw - 1)
    uint32_t nk, ttl, nsa, pre;         // steps (256 B of a row each); live token tiles; weight-ring stages; steps asked for before the first barrier
    uint32_t a_stage, a_ws;             // bytes of a weight stage (tiles' weights at 0, their scales at a_ws)
    uint32_t b_base, b_stage, b_xs;     // the two fragment stages: LDS offset of the first, bytes of one, offset of the activation scales inside
};

typedef float g7f2 __attribute__((ext_vector_type(2)));
template <int AUX> __device__ __forceinline__ void g7_dma16(const void *gsrc, unsigned char *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, AUX);
}
// the loaders' only wait: at most `n` of THIS wave's DMA instructions still in flight (n wave-uniform; loads land in issue order)
#define G7_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void g7_wait_vm(uint32_t n) {
    switch (n < 63u ? n : 63u) {
        G7_W(0) G7_W(1) G7_W(2) G7_W(3) G7_W(4) G7_W(5) G7_W(6) G7_W(7) G7_W(8) G7_W(9) G7_W(10) G7_W(11) G7_W(12) G7_W(13) G7_W(14) G7_W(15)
        G7_W(16) G7_W(17) G7_W(18) G7_W(19) G7_W(20) G7_W(21) G7_W(22) G7_W(23) G7_W(24) G7_W(25) G7_W(26) G7_W(27) G7_W(28) G7_W(29) G7_W(30) G7_W(31)
        G7_W(32) G7_W(33) G7_W(34) G7_W(35) G7_W(36) G7_W(37) G7_W(38) G7_W(39) G7_W(40) G7_W(41) G7_W(42) G7_W(43) G7_W(44) G7_W(45) G7_W(46) G7_W(47)
        G7_W(48) G7_W(49) G7_W(50) G7_W(51) G7_W(52) G7_W(53) G7_W(54) G7_W(55) G7_W(56) G7_W(57) G7_W(58) G7_W(59) G7_W(60) G7_W(61) G7_W(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}
#undef G7_W
// (a loader must not drain its DMA queue at the barrier: no __syncthreads(), whose fence is a vmcnt(0))
__device__ __forceinline__ void g7_loader_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <int R0, int R1, class F> __device__ __forceinline__ void g7_static_for(F &&f) {
    if constexpr (R0 < R1) { f(std::integral_constant<int, R0>{}); g7_static_for<R0 + 1, R1>(f); }
}

// TP = row tiles per workgroup (capacity of the loaders' address registers): 1 | 2 | 3 | 5 | 8
// PP = token tiles per consumer wave (1 | 2): the smallest with tpw x ceil(token tiles / PP) <= 14 waves
// MS = several weight segments share the launch (q | k | v)
template <int TP, int PP, bool MS>
__global__ __launch_bounds__(G7_NW * 64) void gemm_q80_g7_kernel(const G7Dev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const GemvDev &a = d.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    NANO_STAMP(a.stamps, 0, tid);
    const uint32_t n = a.n, ng = a.ng, hh = d.hh, nb = a.nb, nk = d.nk, nsa = d.nsa, ttl = d.ttl;
    const uint32_t epi = a.epi;
    const bool sw = epi == GEMV_EPI_SWIGLU;
    const uint32_t halfoff = sw ? 0u : hh;                             // rows between the two halves of a tile
    const uint32_t bid = blockIdx.x;
    const uint32_t ntl = bid < d.full ? d.tpw : d.tpw - 1u;             // tiles of this workgroup

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

    if (wid >= G7_NCW) {
        // ================================================ weight loaders ===============================================================
        // loader a = wid - 8 brings the steps k with k % 2 == a
        // A stage holds, per tile, 16 rows x 256 B row-major; the 16-byte chunk c of tile row r sits at position c ^ r (an A-fragment
        // ds_read_b128 -- lane (m, kq), group j: chunk 4 j + kq of row m -- then hits 16 distinct 16-byte bank slots per 16-lane group).
        // DMA instruction i of a tile covers tile rows 4 i .. 4 i + 3: lane p writes LDS slot p = row 4 i + p / 16, position p % 16, so it
        // FETCHES chunk (p % 16) ^ row of that row: the 16 lanes of a row still cover its 256 contiguous bytes (two whole 128-B lines).
        // Tile rows 0..7 are half 0 (live: the first hh), 8..15 half 1 (SwiGLU: the same rows of W3); dead rows are not fetched (their
        // LDS rows keep whatever they held: their results are never stored).
        const uint32_t lr = lane >> 4, cp = lane & 15u;
        const int8_t *src[TP][4];                                         // per lane: the address of step 0's chunk
        uint32_t lvm = 0, ilm = 0, ips = 0;                                // bit 4 t + i: this lane fetches / the instruction is issued at all
#pragma unroll
        for (int t = 0; t < TP; t++) {
            const bool tlive = (uint32_t)t < ntl;
            const TI ti = decode(tlive ? (uint32_t)t : 0u);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t r = 4u * (uint32_t)i + lr, half = r >> 3, rr = r & 7u;
                const uint32_t grow = ti.lrow0 + half * halfoff + rr;
                const bool live = tlive && rr < hh && grow < ti.rows0;
                src[t][i] = (half ? ti.wB : ti.wA) + ((size_t)grow * n + ((cp ^ (r & 15u)) << 4));
                const uint32_t rr_first = (4u * (uint32_t)i) & 7u, grow_first = ti.lrow0 + ((uint32_t)i >> 1) * halfoff + rr_first;
                const bool any = tlive && rr_first < hh && grow_first < ti.rows0;      // (wave-uniform: the instruction has a live lane)
                lvm |= live ? 1u << (4 * t + i) : 0u;
                ilm |= any ? 1u << (4 * t + i) : 0u;
                ips += any ? 1u : 0u;
            }
        }
        // weight scales of the step (16 rows x 4 groups per tile): one instruction per four tiles; lane p serves tile 4 s + p / 16, row p % 16
        // -> that row's 16 bytes, LDS [tile][row][4 groups]
        constexpr int NSI = (TP + 3) / 4;
        const float *ssrc[NSI]; bool sl[NSI], sil[NSI];
#pragma unroll
        for (int s = 0; s < NSI; s++) {
            const uint32_t t = 4u * (uint32_t)s + lr, r = cp;
            const bool tlive = t < ntl;
            const TI ti = decode(tlive ? t : 0u);
            const uint32_t half = r >> 3, rr = r & 7u, grow = ti.lrow0 + half * halfoff + rr;
            sl[s] = tlive && rr < hh && grow < ti.rows0;
            ssrc[s] = (half ? ti.sB : ti.sA) + (size_t)grow * ng;
            sil[s] = 4u * (uint32_t)s < ntl;
            ips += sil[s] ? 1u : 0u;
        }
        auto issue = [&](uint32_t k, unsigned char *st) {
            const uint32_t kb = k * 256u;
#pragma unroll
            for (int t = 0; t < TP; t++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (ilm & (1u << (4 * t + i))) { if (lvm & (1u << (4 * t + i))) g7_dma16<2>(src[t][i] + kb, st + (uint32_t)t * 4096u + (uint32_t)i * 1024u); }
#pragma unroll
            for (int s = 0; s < NSI; s++)
                if (sil[s]) { if (sl[s]) g7_dma16<0>(ssrc[s] + k * 4u, st + d.a_ws + (uint32_t)s * 1024u); }
        };
        // Step j lives in stage j % nsa.  Before barrier k the steps 0 .. k + nsa - 2 have been asked for (stage (k - 1) % nsa is still being
        // read); after it step k + nsa - 1 may go out.  `mine` = steps THIS loader has issued; its step k is its (k / 2)-th.
        const uint32_t la = wid - G7_NCW;
        uint32_t mine = 0;
        // Round 6: only d.pre steps (one per loader) go out before the first barrier -- a DMA instruction costs ~150 cycles of issue, and
        // the consumers' first multiply waited for the loaders to finish ISSUING nsa - 1 steps (Qwen3-4B's W1|W3 at 64 tokens: "first weights
        // land" 4.8 us after entry, profiles/r06_stamps_wide_b64_before.txt); the ring then fills two steps per barrier until it is nsa - 1 deep.
        const uint32_t pre = d.pre;
        for (uint32_t j = 0; j < pre; j++) if ((j & 1u) == la) { issue(j, smem + (j % nsa) * d.a_stage); mine++; }
        uint32_t jn = pre, stn = pre % nsa;                                // the next step to go out and its stage
        for (uint32_t k = 0; k < nk; k++) {
            if ((k & 1u) == la) g7_wait_vm((mine - 1u - (k >> 1)) * ips);      // step k has landed (this loader's later steps may still fly)
            g7_loader_barrier();
            // stage (k - 1) % nsa is free again: steps up to k + nsa - 1 may be in the ring
            for (uint32_t c = 0; c < 2u && jn < nk && jn < k + nsa; c++) {
                if ((jn & 1u) == la) { issue(jn, smem + stn * d.a_stage); mine++; }
                jn++; stn = stn + 1u == nsa ? 0u : stn + 1u;
            }
        }
        return;
    }
    // ==================================================== consumers ======================================================================
    // Wave w < nact multiplies row tile w / gpt with the PP consecutive token tiles (w % gpt) PP .. + PP - 1: the tile's A fragments and
    // weight scales are read from LDS once per step for all of them.  Straight-line code per step: the loads of all operands, the 4 PP
    // matrix instructions, then the products and sums (no branch between them: the scheduler overlaps the pairs' chains).
    const uint32_t m = lane & 15u, kq = lane >> 4;
    uint32_t a_off[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) a_off[j] = m * 256u + (((4u * j + kq) ^ m) << 4);
    const uint32_t half = kq >> 1, rr0 = (kq & 1u) * 4u;              // this lane's four output rows: rows rr0 .. rr0 + 3 of half `half`
    const uint32_t gpt = (ttl + (uint32_t)PP - 1u) / (uint32_t)PP;    // waves per row tile
    const bool active = wid < ntl * gpt;
    const uint32_t wtile = active ? wid / gpt : 0u, tt0 = active ? (wid - wtile * gpt) * (uint32_t)PP : 0u;
    float acc[PP][4], oldv[PP][4];
    g7f2 S01[PP], S23[PP];                                            // the running unit sums of rows 0 | 1 and 2 | 3 of each pair
    uint32_t opos[PP];
    const TI wt = decode(wtile);
    const uint32_t orow0 = wt.lrow0 + half * halfoff + rr0;           // output row of c[0] (SwiGLU: lanes kq < 2 write, half 0)
#pragma unroll
    for (int i = 0; i < PP; i++) {
        opos[i] = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) { acc[i][r] = 0.0f; oldv[i][r] = 0.0f; }
        S01[i] = g7f2{0.0f, 0.0f}; S23[i] = g7f2{0.0f, 0.0f};
        // what the epilogue needs from memory (the old residual values, the position of a position-indexed output): asked for now
        const uint32_t tok = (tt0 + (uint32_t)i) * 16u + m;
        if (active && tok < nb && (epi == GEMV_EPI_RESID || wt.ops != 0u)) {
            if (wt.ops) opos[i] = a.pos[tok];
            if (epi == GEMV_EPI_RESID) {
                const float *o = wt.out + (size_t)tok * wt.obs + orow0;
#pragma unroll
                for (int r = 0; r < 4; r++) if (rr0 + (uint32_t)r < hh && orow0 + (uint32_t)r < wt.rows0) oldv[i][r] = o[r];
            }
        }
    }
    // ---- the activation fragments: the step's 1-KB chunks -- c = 4 tt + j: token tile tt, group j of the step; c = 4 ttl: the step's
    //      activation scales (256 B per token tile) -- are fetched and parked by wave c % 14 (<= two chunks per wave).  A chunk sits in
    //      registers for G7_BD steps: asked for at step k - 1, parked in LDS stage (k + 1) % 2 at step k, multiplied at step k + 1.
    // The step loop below has NO branch around a load or an LDS store (a chunk that does not exist is read through an out-of-range offset
    // -- zeros, no memory access -- and parked in a dummy kilobyte): every s_waitcnt the compiler places is then the exact count (the
    // younger steps stay in flight), where a merge point made it wait for everything (the first build: vmcnt(0)).
    const uint32_t dummy = d.b_base + 2u * d.b_stage + lane * 16u;     // 1 KB behind the fragment stages
    uint32_t c_src[2], c_dst[2], c_step[2];
    __amdgpu_buffer_rsrc_t c_rs[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t c = wid + (uint32_t)i * G7_NCW, tt = c >> 2, j = c & 3u;
        const bool frag = c < 4u * ttl, scal = c == 4u * ttl;
        c_rs[i] = scal ? mkrsrc(d.xsf, ttl * ng * 64u) : mkrsrc(d.xf, ttl * ng * 1024u);
        // fragments: + 4 groups = 4096 bytes per step; scales: lane l -> token tile l / 16, the 16 bytes l % 16 of its 4 groups x 16 tokens: + 256 bytes per step
        c_src[i] = frag ? (tt * ng + j) * 1024u + lane * 16u : (scal && (lane >> 4) < ttl) ? (((lane >> 4) * ng) * 16u + (lane & 15u) * 4u) * 4u : OOB;
        c_step[i] = scal ? 256u : 4096u;
        c_dst[i] = frag ? tt * 4096u + j * 1024u + lane * 16u : scal ? d.b_xs + lane * 16u : 0xffffffffu;     // (inside a fragment stage; no chunk: the dummy)
    }
    i32x4 breg[G7_BD][2];
    auto b_issue = [&](auto SL, uint32_t k) {                          // the loads of step k (beyond the last step: nothing) into register slot SL
        constexpr int sl = decltype(SL)::value;
        const bool in = k < nk;
        breg[sl][0] = __builtin_amdgcn_raw_buffer_load_b128(c_rs[0], (int)((in && c_src[0] != OOB) ? c_src[0] + k * c_step[0] : OOB), 0, 0);
        breg[sl][1] = __builtin_amdgcn_raw_buffer_load_b128(c_rs[1], (int)((in && c_src[1] != OOB) ? c_src[1] + k * c_step[1] : OOB), 0, 0);
    };
    auto b_park = [&](auto SL, uint32_t k) {                           // register slot SL (step k) -> fragment stage k % 2 (a step beyond the last: harmless,
        constexpr int sl = decltype(SL)::value;                        //  that stage is not read again)
        const uint32_t bs = d.b_base + (k & 1u) * d.b_stage;
        *reinterpret_cast<i32x4 *>(smem + (c_dst[0] != 0xffffffffu ? bs + c_dst[0] : dummy)) = breg[sl][0];
        *reinterpret_cast<i32x4 *>(smem + (c_dst[1] != 0xffffffffu ? bs + c_dst[1] : dummy)) = breg[sl][1];
    };
    // one step of this wave's pairs: FIRST = the step opens a unit (its group 0 starts the unit sum)
    // FIRST (compile time): the step opens a unit.  Round 6: two fragment slots instead of three, so that the step loop unrolls over one unit
    // and neither the slot nor "does this step open a unit" is a run-time select (round 5: 20 v_cndmask + 16 v_mov among a step's 142 VALU
    // instructions in a kernel whose consumers are VALU bound -- SQ_ACTIVE_INST_VALU 45 % of the SIMD cycles, profiles/
    // r06_4b_b64_pmc_before.txt), and the weight scales are read as the (row r, row r + 1) pairs the packed products take.
    auto step = [&](auto FIRST, const unsigned char *st, const unsigned char *bs) {
        constexpr bool first = decltype(FIRST)::value;
        const unsigned char *A = st + wtile * 4096u;
        const float *WS = reinterpret_cast<const float *>(st + d.a_ws + wtile * 256u) + kq * 16u;              // rows 4 kq .. + 3: [row][4 groups]
        i32x4 fa[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fa[j] = *reinterpret_cast<const i32x4 *>(A + a_off[j]);
        g7f2 w01[4], w23[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { w01[j] = g7f2{WS[j], WS[4 + j]}; w23[j] = g7f2{WS[8 + j], WS[12 + j]}; }      // (ds_read2_b32: the pair lands in adjacent registers)
#pragma unroll
        for (int i = 0; i < PP; i++) {                                 // per token tile: 4 fragment reads, 4 matrix instructions back to back, then the VALU work
            const unsigned char *B = bs + (tt0 + (uint32_t)i) * 4096u + lane * 16u;
            const float *XS = reinterpret_cast<const float *>(bs + d.b_xs + (tt0 + (uint32_t)i) * 256u) + m;    // [group][16 tokens]
            v4i cv[4];
            float xsc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const i32x4 fb = *reinterpret_cast<const i32x4 *>(B + j * 1024);
                xsc[j] = XS[j * 16];
                cv[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[j], fb, v4i{0, 0, 0, 0}, 0, 0, 0);
            }
            // infer.c:672 per output: ((float)ival * ws) * xs, then the running unit sum -- as PACKED fp32 (two rows per instruction:
            // v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar forms): 4 conversions + 6 packed operations per group where
            // the scalar form took 16; this VALU work, not the matrix cores, is what bounds the kernel at 3-4 token tiles
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const g7f2 c01 = {(float)cv[j][0], (float)cv[j][1]}, c23 = {(float)cv[j][2], (float)cv[j][3]};
                const g7f2 x2 = {xsc[j], xsc[j]};
                const g7f2 p01 = (c01 * w01[j]) * x2, p23 = (c23 * w23[j]) * x2;
                if (first && j == 0) { S01[i] = p01; S23[i] = p23; }
                else { S01[i] += p01; S23[i] += p23; }
            }
        }
    };
    auto fold = [&](auto FIRSTU) {                                     // units ascending; the first unit is the row's starting value (not 0 + S_0: -0.0)
        constexpr bool firstu = decltype(FIRSTU)::value;
#pragma unroll
        for (int i = 0; i < PP; i++) {
            const float sv_[4] = {S01[i].x, S01[i].y, S23[i].x, S23[i].y};
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][r] = firstu ? sv_[r] : acc[i][r] + sv_[r];
        }
    };
    // prologue: the first G7_BD steps' fragments are asked for, step 0's parked (behind the first barrier everyone may read them)
    g7_static_for<0, G7_BD>([&](auto SL) { b_issue(SL, (uint32_t)decltype(SL)::value); });
    b_park(std::integral_constant<int, 0>{}, 0u);
    NANO_STAMP(a.stamps, 1, breg[0][0].x);                          // prologue done: step 0's fragments arrived and are parked
    uint32_t sta = 0;                                                  // weight stage of the current step
    auto one_step = [&](auto SI, auto FIRST, uint32_t k) {             // step k: fragment slot k % G7_BD == SI, FIRST = k is even
        constexpr int si = decltype(SI)::value;
        __syncthreads();                                               // weights of step k landed (the loaders), fragments of step k parked; everyone is done with step k - 1
        if (k == 0u) NANO_STAMP(a.stamps, 2, sta);                     // the first weights have landed
        b_park(std::integral_constant<int, (si + 1) % G7_BD>{}, k + 1u);       // stage (k + 1) % 2 was last read at step k - 1
        b_issue(SI, k + (uint32_t)G7_BD);                              // slot si was parked at step k - 1
        if (active) step(FIRST, smem + sta * d.a_stage, smem + d.b_base + (k & 1u) * d.b_stage);
        sta = sta + 1u == nsa ? 0u : sta + 1u;
        if (k == 0u) NANO_STAMP(a.stamps, 3, S01[0].x);                 // step 0 multiplied
    };
    static_assert(G7_BD == 2, "the step loop is unrolled over one unit = two steps = the two fragment slots");
    using T_ = std::true_type; using F_ = std::false_type;
    using S0_ = std::integral_constant<int, 0>; using S1_ = std::integral_constant<int, 1>;
    // unit 0 (steps 0 and 1) gives the rows their starting values; then unit by unit, straight-line code per unit
    one_step(S0_{}, T_{}, 0u);
    if (nk > 1u) one_step(S1_{}, F_{}, 1u);
    fold(T_{});
    uint32_t k2 = 2;
    for (; k2 + 2u <= nk; k2 += 2u) { one_step(S0_{}, T_{}, k2); one_step(S1_{}, F_{}, k2 + 1u); fold(F_{}); }
    if (k2 < nk) { one_step(S0_{}, T_{}, k2); fold(F_{}); }             // a row of an odd number of steps: its last unit is one step
    NANO_STAMP(a.stamps, 4, acc[0][0]);                             // every step done
    // ---- epilogue: store | residual add | SwiGLU -----------------------------------------------------------------------------------------
    if (!active) return;
#pragma unroll
    for (int i = 0; i < PP; i++) {
        const uint32_t tok = (tt0 + (uint32_t)i) * 16u + m;
        float v3[4] = {0.f, 0.f, 0.f, 0.f};
        if (sw) {                                                      // W3's values live 32 lanes up (rows 8..15 of the tile)
#pragma unroll
            for (int r = 0; r < 4; r++) v3[r] = __shfl_xor(acc[i][r], 32, 64);
        }
        if (tok < nb && (!sw || kq < 2u)) {
            float *o = wt.out + (size_t)tok * wt.obs + (size_t)opos[i] * wt.ops + orow0;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (rr0 + (uint32_t)r < hh && orow0 + (uint32_t)r < wt.rows0) o[r] = finish_epi(epi, acc[i][r], v3[r], oldv[i][r]);
        }
    }
    NANO_STAMP(a.stamps, 5, acc[0][0]);                             // stores issued
    NANO_STAMP_END(a.stamps, 6);
}

// =========================================================================================================================================
// G7K (round 6): ONE row tile per workgroup and a LONG row (Wo, W2 of Qwen3-4B at 17..64 tokens) -- the K-phase form.
// G7 above keeps a (row tile, token tile) pair in ONE wave for the whole row length: with one tile per CU that is `ttl` <= 4 of fourteen
// consumer waves at work, 16 / 38 serial steps of ~0.6 us each (round 5: 13.9 / 27.9 us, so these launches stayed with G6: 10.6 / 18.7 us,
// three 8-KB items per wave in sequence).  Here the row length is split among the waves BY UNITS (8 groups = 2 steps = the canonical fold's
// unit sum, so no sum crosses a wave): consumer wave (token tile tt, phase j) multiplies the units u = j (mod KS); KS x ttl waves work at
// once.  What changes with it:
//   * the activation fragments do NOT go through LDS: with one row tile per workgroup a fragment meets exactly one wave, so every wave
//     fetches its own units' fragments from L2 straight into registers (one unit ahead) -- no fragment stages, no parking;
//   * ONE barrier per SUPER-STEP (KS units = 2 KS steps of weights) instead of one per step: the loaders fill a ring of `ring`
//     super-steps by LDS-DMA exactly as above;
//   * the unit sums S_u go to an LDS table [unit][token tile][lane]; after the last super-step the phase-0 wave of a token tile adds them
//     in ascending order (the row's starting value is S_0) and runs the epilogue: the CANONICAL fold, bit for bit G7's / G6's.
// Reference: matmul_quant infer/infer.c:654-679.
constexpr uint32_t G7K_STAGE = 4096u + 256u;        // a stage: the tile's 16 rows x 256 B, then its 16 x 4 weight scales
struct G7KDev {
    GemvDev g;
    const int8_t *xf; const float *xsf;
    uint32_t hh, ntiles, nk, nu, ttl, ks, ncw, nss, nl; // live rows per half tile; tiles = workgroups; steps; units; token tiles; phases; consumer waves; super-steps; loader waves
    uint32_t tab, ring;                                 // LDS offset of the unit-sum table; super-steps of weights in the ring
};

__global__ __launch_bounds__(G7_NW * 64) void gemm_q80_g7k_kernel(const G7KDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const GemvDev &a = d.g;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n = a.n, ng = a.ng, hh = d.hh, nb = a.nb, nk = d.nk, nu = d.nu, ttl = d.ttl, ks = d.ks;
    const uint32_t epi = a.epi;
    const uint32_t ssl = 2u * ks, nsg = d.ring * ssl;                 // steps per super-step; stages of the ring
    const uint32_t lrow0 = blockIdx.x * 2u * hh, rows0 = a.rows[0];
    NANO_STAMP(a.stamps, 0, tid);
    if (wid >= d.ncw) {
        // ---- the weight loaders (nl = the waves the consumers leave, 2..6): loader la brings the steps k = la (mod nl); stage of step k =
        //      k % nsg (layout: G7's, one tile).  A DMA instruction costs ~150 cycles of issue: two loaders move a step per ~0.18 us ----------
        const uint32_t la = wid - d.ncw, nl = d.nl;
        const uint32_t lr = lane >> 4, cp = lane & 15u;
        const int8_t *src[4];
        uint32_t lvm = 0, ilm = 0, ips = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = 4u * (uint32_t)i + lr, half = r >> 3, rr = r & 7u;
            const uint32_t grow = lrow0 + half * hh + rr;
            const bool live = rr < hh && grow < rows0;
            src[i] = a.w[0] + ((size_t)grow * n + ((cp ^ (r & 15u)) << 4));
            const uint32_t rr_first = (4u * (uint32_t)i) & 7u, grow_first = lrow0 + ((uint32_t)i >> 1) * hh + rr_first;
            const bool any = rr_first < hh && grow_first < rows0;
            lvm |= live ? 1u << i : 0u; ilm |= any ? 1u << i : 0u; ips += any ? 1u : 0u;
        }
        // the step's weight scales: lane p < 16 serves tile row p -> that row's 16 bytes (4 groups)
        const uint32_t srow = lrow0 + (cp >> 3) * hh + (cp & 7u);
        const bool sl = lr == 0u && (cp & 7u) < hh && srow < rows0;
        const float *ssrc = a.ws[0] + (size_t)srow * ng;
        ips += 1u;
        auto issue = [&](uint32_t k) {
            unsigned char *st = smem + (k % nsg) * G7K_STAGE;
            const uint32_t kb = k * 256u;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (ilm & (1u << i)) { if (lvm & (1u << i)) g7_dma16<2>(src[i] + kb, st + (uint32_t)i * 1024u); }
            if (sl) g7_dma16<0>(ssrc + k * 4u, st + 4096u);
        };
        uint32_t mine = 0;
        const uint32_t pre = nsg < nk ? nsg : nk;
        for (uint32_t k = la; k < pre; k += nl) { issue(k); mine++; }
        for (uint32_t s = 0; s < d.nss; s++) {
            const uint32_t bound = (s + 1u) * ssl < nk ? (s + 1u) * ssl : nk;      // steps below `bound` must have landed
            const uint32_t need = bound > la ? (bound - la + nl - 1u) / nl : 0u;   // ... this loader's share of them
            g7_wait_vm((mine - need) * ips);
            g7_loader_barrier();
            if (s >= 1u) {                                                         // super-step s - 1 has been read: its stages take super-step s - 1 + RING
                const uint32_t k0 = (s - 1u + d.ring) * ssl, k1 = k0 + ssl < nk ? k0 + ssl : nk;
                for (uint32_t k = k0 + (la + nl - k0 % nl) % nl; k < k1; k += nl) { issue(k); mine++; }
            }
        }
        NANO_STAMP_END(a.stamps, 6);
        return;
    }
    // ---- consumers: wave (phase j, token tile tt) -----------------------------------------------------------------------------------------
    const uint32_t m = lane & 15u, kq = lane >> 4;
    uint32_t a_off[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) a_off[j] = m * 256u + (((4u * j + kq) ^ m) << 4);
    const uint32_t half = kq >> 1, rr0 = (kq & 1u) * 4u;
    const uint32_t ph = wid / ttl, tt = wid - ph * ttl;
    const uint32_t orow0 = lrow0 + half * hh + rr0;
    const uint32_t tok = tt * 16u + m;
    float oldv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t opos = 0u;
    const uint32_t obs = a.out_bstride[0], ops = a.out_pstride[0];
    if (ph == 0u && tok < nb && (epi == GEMV_EPI_RESID || ops != 0u)) {
        if (ops) opos = a.pos[tok];
        if (epi == GEMV_EPI_RESID) {
            const float *o = a.out[0] + (size_t)tok * obs + orow0;
#pragma unroll
            for (int r = 0; r < 4; r++) if (rr0 + (uint32_t)r < hh && orow0 + (uint32_t)r < rows0) oldv[r] = o[r];
        }
    }
    // this wave's fragments: token tile tt, groups 8 u .. 8 u + 7 of its units, lane l reads slot l; scales [group][16 tokens]
    const __amdgpu_buffer_rsrc_t rsf = mkrsrc(d.xf, ttl * ng * 1024u);
    const __amdgpu_buffer_rsrc_t rss = mkrsrc(d.xsf, ttl * ng * 64u);
    const uint32_t fbase = tt * ng * 1024u + lane * 16u, sbase = (tt * ng * 16u + m) * 4u;
    i32x4 fb[2][4];
    float xsc[2][4];
    auto b_issue = [&](auto H, uint32_t u) {                            // half H (step 2 u + H) of unit u; beyond the row: zeros, no access
        constexpr int h = decltype(H)::value;
        const uint32_t g0 = 8u * u + 4u * (uint32_t)h;
        const bool in = 2u * u + (uint32_t)h < nk && u < nu;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            fb[h][j] = __builtin_amdgcn_raw_buffer_load_b128(rsf, (int)(in ? fbase + (g0 + (uint32_t)j) * 1024u : OOB), 0, 0);
            xsc[h][j] = bload_f(rss, in ? sbase + (g0 + (uint32_t)j) * 64u : OOB);
        }
    };
    g7f2 S01, S23;
    auto step = [&](auto H, const unsigned char *st) {
        constexpr int h = decltype(H)::value;
        const float *WS = reinterpret_cast<const float *>(st + 4096u) + kq * 16u;
        i32x4 fa[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fa[j] = *reinterpret_cast<const i32x4 *>(st + a_off[j]);
        g7f2 w01[4], w23[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { w01[j] = g7f2{WS[j], WS[4 + j]}; w23[j] = g7f2{WS[8 + j], WS[12 + j]}; }
        v4i cv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) cv[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[j], fb[h][j], v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {                                   // infer.c:672 per output, then the running unit sum (packed fp32: G7's step)
            const g7f2 c01 = {(float)cv[j][0], (float)cv[j][1]}, c23 = {(float)cv[j][2], (float)cv[j][3]};
            const g7f2 x2 = {xsc[h][j], xsc[h][j]};
            const g7f2 p01 = (c01 * w01[j]) * x2, p23 = (c23 * w23[j]) * x2;
            if (h == 0 && j == 0) { S01 = p01; S23 = p23; }
            else { S01 += p01; S23 += p23; }
        }
    };
    using H0_ = std::integral_constant<int, 0>; using H1_ = std::integral_constant<int, 1>;
    float4 *tab = reinterpret_cast<float4 *>(smem + d.tab);
    b_issue(H0_{}, ph); b_issue(H1_{}, ph);
    NANO_STAMP(a.stamps, 1, fb[0][0].x);                            // (measurement builds: G7's phases) the first unit's fragments arrived
    for (uint32_t s = 0; s < d.nss; s++) {
        __syncthreads();                                                 // the weights of super-step s have landed; everyone is done with s - 1
        if (s == 0u) NANO_STAMP(a.stamps, 2, s);                         // the first weights have landed
        const uint32_t u = s * ks + ph;
        if (u < nu) {
            const uint32_t k0 = 2u * u;
            step(H0_{}, smem + (k0 % nsg) * G7K_STAGE);
            b_issue(H0_{}, u + ks);                                      // the next unit's first half into the registers just read
            if (k0 + 1u < nk) step(H1_{}, smem + ((k0 + 1u) % nsg) * G7K_STAGE);
            b_issue(H1_{}, u + ks);
            tab[(u * ttl + tt) * 64u + lane] = make_float4(S01.x, S01.y, S23.x, S23.y);
        }
        if (s == 0u) NANO_STAMP(a.stamps, 3, S01.x);                     // the first super-step multiplied
    }
    __syncthreads();                                                     // every unit sum is in the table
    NANO_STAMP(a.stamps, 4, S01.x);                                  // every super-step done
    if (ph != 0u) { NANO_STAMP_END(a.stamps, 6); return; }
    float acc[4];
    {
        const float4 t0 = tab[tt * 64u + lane];                          // units ascending; the first unit is the row's starting value (not 0 + S_0: -0.0)
        acc[0] = t0.x; acc[1] = t0.y; acc[2] = t0.z; acc[3] = t0.w;
        for (uint32_t u = 1; u < nu; u++) {
            const float4 t = tab[(u * ttl + tt) * 64u + lane];
            acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
        }
    }
    if (tok < nb) {
        float *o = a.out[0] + (size_t)tok * obs + (size_t)opos * ops + orow0;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (rr0 + (uint32_t)r < hh && orow0 + (uint32_t)r < rows0) o[r] = finish_epi(epi, acc[r], 0.0f, oldv[r]);
    }
    NANO_STAMP(a.stamps, 5, acc[0]);                                 // stores issued
    NANO_STAMP_END(a.stamps, 6);
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
struct G7Plan { uint32_t hh, ntiles, tc0, tc1, grid, tpw, nk, ttl, nsa, a_stage, a_ws, b_base, b_stage, b_xs, tp, pp; bool ms; size_t lds; };

static uint32_t g7_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}

// Tile height.  At 17..64 tokens a launch is bound by the consumers' VALU work as much as by its bytes (measured, round 5: ~120 SIMD-cycles
// per matrix-core result of 16 rows x 16 tokens x one group -- cvt, two multiplies and an add per output -- i.e. ~0.23 us per row tile and
// step at four token tiles, where the tile's 16 x 256 B stream in ~0.18 us), and a tile costs that whatever its live rows: minimise
// (tiles per CU) x max(VALU, bytes), ties to the taller tile (Qwen3-4B's q|k|v: 384 tiles of 16 rows, 2 on the busiest CU, instead of G6's
// 768 tiles of 8 rows, 3 per CU).
static bool g7_plan(const GemvArgs &a, G7Plan &p) {
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    const uint32_t cus = a.cus ? a.cus : 256u, nseg = sw ? 1u : a.nseg;
    const uint32_t ttl0 = (a.nb + 15u) / 16u;
    uint32_t best = 0, best_cost = ~0u;
    for (uint32_t hh = 1; hh <= 8; hh++) {
        const uint32_t trw = sw ? hh : 2u * hh;
        uint32_t tiles = 0;
        for (uint32_t s = 0; s < nseg; s++) tiles += (a.seg[s].rows + trw - 1) / trw;
        const uint32_t grid = tiles < cus ? tiles : cus, tpw = (tiles + grid - 1) / grid;
        const uint32_t valu = 6u * ttl0, bytes = trw * (sw ? 2u : 1u) + 2u;
        const uint32_t cost = tpw * (valu > bytes ? valu : bytes);
        if (cost <= best_cost) { best_cost = cost; best = hh; }       // ties: the taller tile
    }
    p.hh = best;
    const uint32_t trw = sw ? best : 2u * best;
    uint32_t tiles = 0, tc[2] = {0xffffffffu, 0xffffffffu};
    for (uint32_t s = 0; s < nseg; s++) { tiles += (a.seg[s].rows + trw - 1) / trw; if (s < 2) tc[s] = tiles; }
    p.ntiles = tiles; p.tc0 = nseg > 1 ? tc[0] : 0xffffffffu; p.tc1 = nseg > 2 ? tc[1] : 0xffffffffu;
    p.grid = tiles < cus ? tiles : cus; p.tpw = (tiles + p.grid - 1) / p.grid;
    p.ms = !sw && a.nseg > 1;
    p.nk = a.n / 256u; p.ttl = (a.nb + 15u) / 16u;
    p.tp = p.tpw <= 1 ? 1u : p.tpw == 2 ? 2u : p.tpw == 3 ? 3u : p.tpw <= 5 ? 5u : p.tpw <= 8 ? 8u : 0u;
    if (!p.tp) return false;
    p.pp = 0;
    for (uint32_t pp = 1; pp <= 2u && !p.pp; pp *= 2u) if (p.tpw * ((p.ttl + pp - 1u) / pp) <= G7_NCW) p.pp = pp;
    if (!p.pp) return false;
    // LDS: the weight ring (a stage = the tiles' 16 rows x 256 B + their scales, one 1-KB DMA instruction per four tiles), then the two
    // fragment stages (token tiles x 4 KB + 1 KB of activation scales)
    p.a_ws = p.tpw * 4096u; p.a_stage = p.a_ws + ((p.tpw + 3u) / 4u) * 1024u;
    p.b_xs = ((p.ttl + p.pp - 1u) / p.pp) * p.pp * 4096u; p.b_stage = p.b_xs + 1024u;    // (token tiles rounded up to the waves' PP: a wave reads all of its PP)
    if (2u * p.b_stage + 1024u + 2u * p.a_stage > G7_LDS) return false;
    uint32_t nsa = (G7_LDS - 2u * p.b_stage - 1024u) / p.a_stage;
    // vmcnt is a 6-bit counter per wave: a loader's steps in flight behind the one it waits for (every second step is its own)
    const uint32_t ips = 4u * p.tpw + (p.tpw + 3u) / 4u;
    if (nsa > 1u + 2u * (63u / ips)) nsa = 1u + 2u * (63u / ips);
    if (nsa > G7_MAXNSA) nsa = G7_MAXNSA;
    if (nsa > p.nk + 1u) nsa = p.nk + 1u;
    if (nsa < 2u) return false;
    p.nsa = nsa;
    p.b_base = nsa * p.a_stage;
    p.lds = (size_t)p.b_base + 2u * p.b_stage + 1024u;                  // + the dummy kilobyte dead chunks are parked in
    return true;
}

template <int TP, int PP, bool MS>
static hipError_t g7_launch_t(const G7Dev &d, size_t lds, hipStream_t st) {
    auto kern = &gemm_q80_g7_kernel<TP, PP, MS>;
    static std::atomic<bool> armed[64];
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !armed[dev].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G7_LDS);
        if (dev >= 0 && dev < 64) armed[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(d.grid), dim3(G7_NW * 64u), lds, st, d);
    return hipGetLastError();
}



// ---- G7K: plan ----------------------------------------------------------------------------------------------------------------------------
struct G7KPlan { uint32_t hh, ntiles, nk, nu, ttl, ks, ncw, nss, tab, ring, nl; size_t lds; };
static bool g7k_plan(const GemvArgs &a, G7KPlan &p) {
    if (a.gs != 64 || a.nb < 3u || a.nb > 64u || a.n % 256u || a.nseg != 1 || a.epi == GEMV_EPI_SWIGLU) return false;
    if (a.ordered || a.resid_add || a.tile_max || a.attn_part) return false;
    if ((uint64_t)a.seg[0].rows * a.n >= (1ull << 32) - (1u << 20) || a.seg[0].rows >= 65536u) return false;
    const uint32_t cus = a.cus ? a.cus : 256u, rows = a.seg[0].rows;
    p.hh = 0;
    // tile height: a workgroup's time does not depend on its live rows (a matrix-core tile and its VALU work cost the same): the lowest tile
    // that still gives every workgroup a CU of its own = the most CUs at work (Qwen3-0.6B's Wo / W2: 256 workgroups of 4 rows instead of 64 of 16:
    // 1.437 -> 1.413 ms per 64-sequence step; Qwen3-4B's: 256 of 10 rows instead of 160 of 16: 3.215 -> 3.19)
    for (uint32_t hh = 1; hh <= 8 && !p.hh; hh++) if ((rows + 2u * hh - 1u) / (2u * hh) <= cus) p.hh = hh;
    if (!p.hh) return false;
    p.ntiles = (rows + 2u * p.hh - 1u) / (2u * p.hh);
    p.nk = a.n / 256u; p.nu = (p.nk + 1u) / 2u; p.ttl = (a.nb + 15u) / 16u;
    if (p.nk < 8u) return false;                                       // short rows: G7 / G6
    // Where it pays (same-box A/Bs against G6 MODE F, profiles/r06_g7k.txt): up to three token tiles.  Qwen3-0.6B at 32 sequences 1.253 ->
    // 1.18-1.23 ms per step, Qwen3-4B at 48: 3.012 -> 2.981, at 32: even; with FOUR token tiles (49..64 tokens) it LOSES: Qwen3-4B at 64
    // sequences 3.127 -> 3.19 ms, Qwen3-0.6B even -- both kernels then sit at the same ~0.1 us per (16 rows x 16 tokens x 256 B) of a CU.
    if (p.ttl > 3u) return false;
    // (3..16 tokens, round 6: Qwen3-0.6B at 16 sequences 1.068 -> 1.016 ms per step, Qwen3-4B at 4 / 8 / 16: -0.9 % each; the loaders are the
    //  waves the consumers leave, up to six -- at 32..48 tokens four loaders instead of two changed nothing)
    // phases: as many as the fourteen consumer waves and the ring + table allow
    // the ring comes first: THREE super-steps (the weights of super-step s + 2 go out at barrier s; with two, every super-step pays a DMA issue
    // + an HBM round trip -- Qwen3-4B at 64 sequences 3.24-3.27 ms against 3.195-3.215), then as many phases as still fit; two only when no
    // phase count leaves room for three (the most phases first, the ring second, measured 0.6-0.8 % slower on Qwen3-4B at 8 / 16 / 32 sequences)
    const uint32_t ks_max = G7_NCW / p.ttl < 6u ? G7_NCW / p.ttl : 6u;
    auto fits = [&](uint32_t ks, uint32_t rg) {
        const uint32_t nl = G7_NW - ks * p.ttl < 6u ? G7_NW - ks * p.ttl : 6u;     // the waves the consumers leave load
        const size_t ring = (size_t)rg * 2u * ks * G7K_STAGE, tab = (size_t)p.nu * p.ttl * 1024u;
        if (ks > p.nu || ring + tab > G7_LDS) return false;
        if (rg * ((2u * ks + nl - 1u) / nl) * 5u > 60u) return false;  // a loader's instructions in flight (ring super-steps x its steps x <= 5) fit vmcnt's six bits
        p.ks = ks; p.ncw = ks * p.ttl; p.nss = (p.nu + ks - 1u) / ks; p.ring = rg; p.nl = nl;
        p.tab = (uint32_t)ring; p.lds = ring + tab;
        return true;
    };
    for (uint32_t rg = 3u; rg >= 2u; rg--) for (uint32_t ks = ks_max; ks >= 2u; ks--) if (fits(ks, rg)) return true;
    return false;
}

}  // namespace

bool gemm_q80_g7_supports(const GemvArgs &a) {
    { G7KPlan kp; if (g7k_plan(a, kp)) return true; }                   // one row tile per CU and a long row: the K-phase form
    if (a.gs != 64 || a.nb < 17u || a.nb > 64u || a.n % 256u || a.nseg == 0 || a.nseg > 3) return false;
    if (a.ordered || a.resid_add || a.tile_max || a.attn_part) return false;
    if (a.epi == GEMV_EPI_SWIGLU && (a.nseg != 2 || a.seg[0].rows != a.seg[1].rows)) return false;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    for (uint32_t s = 0; s < nseg; s++) if ((uint64_t)a.seg[s].rows * a.n >= (1ull << 32) - (1u << 20)) return false;   // 32-bit row offsets per segment
    if (g7_rows(a) >= 65536u) return false;                            // (the classifier has kernels of its own: STREAM / GC)
    G7Plan p;
    if (!g7_plan(a, p)) return false;
    // Where it pays (round 5, same-box A/B against G6 MODE F / G5, profiles/r05_g7_stamps.txt): launches with several row tiles per CU
    // (q|k|v, W1|W3: one weight stage feeds 8..20 matrix-core pairs) and very short rows.  A launch of ONE row tile per CU and a long
    // row (Wo, W2 of Qwen3-4B: 16 / 38 steps of ~0.6 us with four of the fourteen consumer waves at work) stays with G6, whose eight
    // waves split the row length: 8.3 / 15.8 us there against 13.9 / 27.9 here.
    return p.tpw * p.ttl >= 8u || p.nk <= 4u;
}

hipError_t launch_gemm_q80_g7(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g7_supports(a)) return hipErrorInvalidValue;
    G7KPlan kp;
    if (g7k_plan(a, kp)) {
        G7KDev d{};
        d.g = to_dev(a);
        d.g.nthr = (kp.ncw + kp.nl) * 64u;
        d.xf = a.xq_in; d.xsf = a.xs_in;
        d.hh = kp.hh; d.ntiles = kp.ntiles; d.nk = kp.nk; d.nu = kp.nu; d.ttl = kp.ttl; d.ks = kp.ks; d.ncw = kp.ncw; d.nss = kp.nss; d.tab = kp.tab; d.ring = kp.ring; d.nl = kp.nl;
        static std::atomic<bool> armed_k[64];
        int dev = 0; (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !armed_k[dev].load(std::memory_order_acquire)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_q80_g7k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G7_LDS);
            if (dev >= 0 && dev < 64) armed_k[dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(gemm_q80_g7k_kernel, dim3(kp.ntiles), dim3((kp.ncw + kp.nl) * 64u), kp.lds, st, d);
        return hipGetLastError();
    }
    G7Plan p;
    if (!g7_plan(a, p)) return hipErrorInvalidValue;
    G7Dev d{};
    d.g = to_dev(a);
    d.g.nthr = G7_NW * 64u;
    d.xf = a.xq_in; d.xsf = a.xs_in;
    d.hh = p.hh; d.ntiles = p.ntiles; d.tc0 = p.tc0; d.tc1 = p.tc1; d.grid = p.grid; d.tpw = p.tpw;
    d.full = p.ntiles - (p.tpw - 1u) * p.grid;
    d.nk = p.nk; d.ttl = p.ttl; d.nsa = p.nsa;
    d.pre = p.nsa - 1u < p.nk ? p.nsa - 1u : p.nk;
    if (d.pre > 3u) d.pre = 3u;         // (2 / 3 / all nsa - 1 before the first barrier: 3.249 / 3.248 / 3.279 ms per 64-sequence Qwen3-4B step, one box, two runs each)
    d.a_stage = p.a_stage; d.a_ws = p.a_ws; d.b_base = p.b_base; d.b_stage = p.b_stage; d.b_xs = p.b_xs;
#define G7_GO(TP_, PP_) do { return p.ms ? g7_launch_t<TP_, PP_, true>(d, p.lds, st) : g7_launch_t<TP_, PP_, false>(d, p.lds, st); } while (0)
#define G7_TP(PP_) do { if (p.tp == 1u) G7_GO(1, PP_); if (p.tp == 2u) G7_GO(2, PP_); if (p.tp == 3u) G7_GO(3, PP_); if (p.tp == 5u) G7_GO(5, PP_); G7_GO(8, PP_); } while (0)
    if (p.pp == 1u) G7_TP(1);
    G7_TP(2);
#undef G7_TP
#undef G7_GO
}

}  // namespace nano
