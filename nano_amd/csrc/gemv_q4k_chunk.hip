// gemv_q4k_chunk.hip -- Q4K (W4A4) fused GEMV for ONE sequence, weights read as 16-byte chunks (round 4).
//
// Same arithmetic as gemv_q4k.hip (reference infer/tensor.c:359-434 dot_two_blocks_q4k, 438-471 matmul_q4k: three integer sums
// per 32-weight group, the four-term float combine in the reference's operation order, the 8 groups of a block added in
// order, the blocks of a row added in order -> bit-identical fp32 for equal quantized inputs), another division of labour:
//
//   * gemv_q4k.hip deals (row, group) items to threads; each item costs three loads (16 nibble bytes + the block's 32 header
//     bytes, which the 8 groups of a block each fetch again) and twelve registers, a workgroup holds <= 4096 of them, and the
//     grid follows from that: 608 workgroups of 640 threads for Qwen3-4B's W1|W3, one resident per CU -> 2.4 rounds of a 7.9 us
//     workgroup (20 us per launch, `profiles/r04_stamps_q4k_before.txt`); 320 workgroups of 1024 for W2 -> two rounds, the second
//     one a quarter full, each workgroup quantizing all 9728 activations again (3.6 us).
//   * here a workgroup owns `rw` consecutive rows (any count: the grid is fitted to the chip, one round), which are ONE
//     contiguous run of 160-byte blocks.  Lane l < 60 of a wave reads chunk l of a 960-byte "wave-load" = six whole blocks:
//     one 16-byte load per lane, every byte of the matrix requested exactly once, four registers per load in flight.  Lane
//     l's role is fixed for the whole launch: c = l % 10 is the chunk inside the block (0, 1: the header; 2..9: the nibbles
//     of group c - 2), l / 10 the block of the wave-load.  The group lanes fetch the five header words they need from their
//     block's two header lanes (ds_bpermute), multiply against the staged activation group, and park the group value in a
//     wave-private LDS line; the block's first lane adds the eight in order and writes the block sum to the workgroup's
//     table [row][block]; one thread per row adds the blocks in order.  A ring of D loads per wave is in flight; the
//     classifier-like launches loop over it (persistent workgroups, the activation quantized once per workgroup).
//
// Restrictions (the router keeps gemv_q4k.hip for the rest): whole blocks (n % 256 == 0), n <= 16384.
//
// Round 5: 2 .. 8 sequences (template NB = 2 | 4 | 8).  Rounds 3-4 ran them through gemv_q4k.hip, whose workgroups each stage AND quantize
// every sequence's activation: Qwen3-4B's rows left room for one sequence per launch, i.e. no weight sharing at all (13.5 ms per 8-sequence
// step where one sequence takes 1.46).  Here the activations are normalised / combined and block-quantized ONCE, by q4k_quant_rows_kernel
// (one workgroup per sequence running exactly the one-sequence kernel's prologue: same thread count, same trees, same bits), which leaves
// the staged groups (XGroup: packed nibbles, sq, bq, nibble sum) in global memory; the projection workgroups copy them to LDS, read every
// weight byte once and multiply it against all NB sequences -- per wave-load the weight-only half once, the activation half per sequence.
#include "gemv_q4k_impl.h"
#include <hip/hip_ext.h>

namespace nano {

extern hipEvent_t g_q80_probe_start, g_q80_probe_stop;     // gemv_q80.hip: exact start / stop of the next classifier launch

namespace {

template <int ROLE, int NV, int D, bool LOOP, int NB = 1>
__global__ __launch_bounds__(1024) void gemv_q4k_chunk_kernel(const GemvDev a) {
    static_assert(NB == 1 || (ROLE == R_GENERIC && NV == 1), "several sequences: activations arrive quantized (q4k_quant_rows_kernel), generic role");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define CHUNK_A a
#define CHUNK_BID blockIdx.x
#define CHUNK_HAND 0
#define CHUNK_HANDV (SlabHand{})
#define CHUNK_PTAG 0u
#define CHUNK_XHAND 0
#define CHUNK_XHANDV (SlabHand{})
#define CHUNK_CTAG 0u
#define CHUNK_XWAIT 0u
#define CHUNK_PART 0
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_A
#undef CHUNK_BID
#undef CHUNK_HAND
#undef CHUNK_HANDV
#undef CHUNK_PTAG
#undef CHUNK_XHAND
#undef CHUNK_XHANDV
#undef CHUNK_CTAG
#undef CHUNK_XWAIT
#undef CHUNK_PART
}

}  // namespace
}  // namespace nano
#include "attn_impl.h"
namespace nano {
namespace {
// ---- q | k | v projection + attention in ONE launch (Qwen3 decode, one sequence, head_dim 128; round 6: what gemv_q80_impl.h's
//      qkv_attn_fused_kernel is for Q80) ---------------------------------------------------------------------------------------------------
// The first `ngemv` workgroups are the projection's chunk GEMV (role: rmsnorm + block quantizer + store), whose fold threads also store every
// result as an 8-byte {tag, value} granule; the LAST n_attn workgroups are the attention's (head x split): they ask for their K / V rows at
// entry, nap, then poll for q, the raw k row and the fresh v row of their KV group.  256 threads for both kinds (q4k_fused_shape).  Epoch tags, give-up and re-issue: device_common.h, backend.hip.  Reference: infer/infer.c:758-879.
struct Q4FusedArgs { GemvDev g; AttnArgs a; SlabHand hand; uint32_t n_attn, head_wgs, wait16, ngemv; };
template <int NV, int D>
__global__ __launch_bounds__(256) void q4k_qkv_attn_fused_kernel(const Q4FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);                       // the step's epoch: the first load of every workgroup
    if (blockIdx.x >= fa.ngemv) {
        const uint32_t ab = blockIdx.x - fa.ngemv;
        const uint32_t split = ab / fa.head_wgs, grp = ab - split * fa.head_wgs;
        attention_body<8, 4, 1, 1, false, false, 2, false, true>(fa.a, smem, grp, 0u, split, fa.hand, hand_ctag(tk_, fa.hand), fa.wait16);
        return;
    }
    constexpr int ROLE = R_NORM_STORE, NB = 1;
    constexpr bool LOOP = false;
#define CHUNK_A fa.g
#define CHUNK_BID blockIdx.x
#define CHUNK_HAND 1
#define CHUNK_HANDV fa.hand
#define CHUNK_PTAG hand_ptag(tk_, fa.hand)
#define CHUNK_XHAND 0
#define CHUNK_XHANDV (SlabHand{})
#define CHUNK_CTAG 0u
#define CHUNK_XWAIT 0u
#define CHUNK_PART 0
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_A
#undef CHUNK_BID
#undef CHUNK_HAND
#undef CHUNK_HANDV
#undef CHUNK_PTAG
#undef CHUNK_XHAND
#undef CHUNK_XHANDV
#undef CHUNK_CTAG
#undef CHUNK_XWAIT
#undef CHUNK_PART
}

// ---- Wo + W1|W3 in ONE launch (round 6: what gemv_q80_impl.h's wo_w13_fused_kernel is for Q80) ---------------------------------------------
// The launch has W1|W3's grid and thread count; its first `wo_wgs` workgroups run Wo's body first (results stored as usual AND as granules),
// then EVERY workgroup runs W1|W3's body with the activation polled from the granules.  Issue order: Wo's loads, W1|W3's weight and
// norm-weight loads (CHUNK_PART 1 of both bodies), Wo's arithmetic, W1|W3's.  Producers first, a grid of at most one workgroup per CU is
// resident as a whole.  Same bodies, same bits.  Reference: infer/infer.c:885-944.
struct Q4Wo13Args { GemvDev wo; GemvDev w13; SlabHand hand; uint32_t wo_wgs, wait16; };
template <int ROLE_A, int NV_A, int D_A, int NV_B, int D_B>
__global__ __launch_bounds__(1024) void q4k_wo_w13_fused_kernel(const Q4Wo13Args fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);
    constexpr int NB = 1;
    constexpr bool LOOP = false;
    {
        constexpr int ROLE = ROLE_A, NV = NV_A, D = D_A;
#define CHUNK_BID blockIdx.x
#define CHUNK_A fa.wo
#define CHUNK_HAND 1
#define CHUNK_HANDV fa.hand
#define CHUNK_PTAG hand_ptag(tk_, fa.hand)
#define CHUNK_XHAND 0
#define CHUNK_XHANDV (SlabHand{})
#define CHUNK_CTAG 0u
#define CHUNK_XWAIT 0u
#define CHUNK_PART 1
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_PART
        auto wo_rest = [&]() __attribute__((always_inline)) {
#define CHUNK_PART 2
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_PART
        };
#undef CHUNK_A
#undef CHUNK_HAND
#undef CHUNK_HANDV
#undef CHUNK_PTAG
#undef CHUNK_XHAND
#undef CHUNK_XHANDV
#undef CHUNK_CTAG
#undef CHUNK_XWAIT
        {
            constexpr int ROLE = R_NORM_SWIGLU, NV = NV_B, D = D_B;
#define CHUNK_A fa.w13
#define CHUNK_HAND 0
#define CHUNK_HANDV (SlabHand{})
#define CHUNK_PTAG 0u
#define CHUNK_XHAND 1
#define CHUNK_XHANDV fa.hand
#define CHUNK_CTAG hand_ctag(tk_, fa.hand)
#define CHUNK_XWAIT (blockIdx.x >= fa.wo_wgs ? fa.wait16 : 0u)
#define CHUNK_PART 1
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_PART
            if (blockIdx.x < fa.wo_wgs) wo_rest();
            __syncthreads();                // (LDS is W1|W3's from here)
#define CHUNK_PART 2
#include "gemv_q4k_chunk_body.inc"
#undef CHUNK_PART
#undef CHUNK_A
#undef CHUNK_HAND
#undef CHUNK_HANDV
#undef CHUNK_PTAG
#undef CHUNK_XHAND
#undef CHUNK_XHANDV
#undef CHUNK_CTAG
#undef CHUNK_XWAIT
        }
#undef CHUNK_BID
    }
}

// ---- the activations of a 2 .. 8-sequence launch, normalised / combined and block-quantized once -----------------------------------------
// One workgroup per sequence; its code is the one-sequence kernel's prologue (stage_issue -> stage_xn -> quantize_q4k_regs) run with the
// thread count the one-sequence launch of the same matrix would use, so every tree (rmsnorm sum of squares, split-attention combine) and
// therefore every quantized nibble is what that sequence gets when it is decoded alone.  Output: the staged groups, [sequence][GT] XGroups.
template <int ROLE, int NV>
__global__ __launch_bounds__(1024) void q4k_quant_rows_kernel(const GemvDev a, XGroup *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n, GT = (n >> 8) * 8u, b = blockIdx.x;
    XGroup *xg = reinterpret_cast<XGroup *>(smem);
    float *red = reinterpret_cast<float *>(smem + (size_t)GT * sizeof(XGroup));
    GemvDev s = a;                                                  // this workgroup's sequence as a launch of its own (route.hip gemv_slice)
    s.nb = 1;
    if (a.xin) s.xin = a.xin + (size_t)b * a.xin_bstride;
    if (a.attn_part) { s.attn_part = a.attn_part + (size_t)b * a.attn_nsplit * n; s.attn_ml = a.attn_ml + (size_t)b * a.attn_n_head * a.attn_nsplit * 2u; }
    Staged<1, NV> sx;
    stage_issue<ROLE, 1, NV>(s, sx);
    stage_xn<ROLE, 1, NV>(s, sx, nullptr, red, (n + 3u) & ~3u, true);
    quantize_q4k_regs<1, NV, false>(s, sx, xg);
    __syncthreads();
    const uint4 *src = reinterpret_cast<const uint4 *>(xg);
    uint4 *dst = reinterpret_cast<uint4 *>(out + (size_t)b * GT);
    for (uint32_t i = tid; i < GT * 2u; i += nthr) dst[i] = src[i];
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct ChunkPlan { uint32_t rw, nthr, d, loop, rounds, nv, wg[3], grid; size_t lds; };

size_t chunk_lds_bytes(uint32_t n, bool combine, uint32_t attn_n_head, uint32_t nmat, uint32_t rw, uint32_t nw, uint32_t sl, uint32_t nbq = 1) {
    const size_t bpl = n >> 8, GT = bpl * 8;
    return nbq * GT * sizeof(XGroup) + (16 + (combine ? (size_t)attn_n_head * 8 : 0) + (size_t)nw * sl * 64 + 2 * (size_t)nw + (size_t)nmat * nbq * rw * (bpl | 1)) * 4 + 16;
}

bool plan_chunk(const GemvArgs &a, ChunkPlan &p, uint32_t force_nw = 0) {      // force_nw: the fused launch's 256 threads (same bits: see q4k_fused_shape)
    if (a.nb == 0 || a.nb > 8 || a.n == 0 || (a.n & 255u) || a.n > 16384u || a.nseg == 0 || a.nseg > 3) return false;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return false;
    const uint32_t nbq = a.nb <= 1 ? 1u : a.nb <= 2 ? 2u : a.nb <= 4 ? 4u : 8u;     // the kernel's NB (sequences beyond a.nb are skipped)
    if (nbq > 1 && (a.x4_in || a.xq_in)) return false;                              // (caller-quantized activations: one sequence)
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
    const uint32_t nmat = sw ? 2u : 1u, nseg = sw ? 1u : a.nseg, bpl = a.n >> 8;
    uint32_t rows = 0;
    for (uint32_t s = 0; s < nseg; s++) rows += a.seg[s].rows;
    if (rows == 0) return false;
    constexpr uint32_t want_div = 4u;
    uint32_t want = ((a.n / want_div + 63) / 64) * 64;                 // the block quantizer: four elements per thread and pass
    if (want < 256) want = 256;
    if (want > 1024) want = 1024;
    const bool cls = nseg == 1 && a.epi == GEMV_EPI_STORE && rows >= 65536u;
    const uint32_t cus = a.cus ? a.cus : 256u;
    // workgroups per CU: the per-layer matrices get one (every workgroup quantizes the activation: once per CU); the classifier's
    // activation is short next to its rows, its workgroups are persistent and small (4 x 256 threads at n = 1024)
    if (cls && want > 512) want = 1024;
    // (Measured and not taken, Qwen3-0.6B on one box: half the workgroups with twice the rows for Wo / W2, whose workgroups each bring
    // n / 4 quantizer threads: 1686 vs 1720 tok/s; the quantizer on n / 8 threads, two blocks per wave: 1661; three wave-loads per wave: 1640.)
    const uint32_t k = cls ? 1024u / want : 1u, target = cus * k;
    uint32_t best = 0, best_cost = ~0u;
    for (uint32_t c = 1; c <= 1024; c++) {
        if ((uint64_t)c * bpl >= 65536u) break;                          // the kernel's ceil(nblk / 6) and b / bpl by multiplication
        if (chunk_lds_bytes(a.n, nbq == 1 && a.attn_part != nullptr, a.attn_n_head, nmat, c, 16, 8, nbq) * k > 150u * 1024u) break;
        uint32_t wgs = 0;
        for (uint32_t s = 0; s < nseg; s++) wgs += (a.seg[s].rows + c - 1) / c;
        // rows of the busiest CU slot; several rounds of workgroups pay the prologue (activation, norm, block quantizer) once per round
        uint32_t cost = ((wgs + target - 1) / target) * c * 100u;
        if (wgs > target) cost += cost * 15u / 100u;
        if (cost <= best_cost) { best_cost = cost; best = c; }          // ties: the larger slab (fewer workgroups staging the activation)
    }
    if (!best) return false;
    p.rw = best;
    const uint32_t TT = nmat * ((best * bpl + 5) / 6);
    uint32_t nw = want / 64;
    // one wave-load per wave where the workgroup's waves allow it (<= 16): every load at kernel entry, no serial second item
    constexpr uint32_t per_want = 2u;
    if (!cls) { uint32_t m = (TT + per_want - 1) / per_want; if (m > 16) m = 16; if (nw < m) nw = m; }
    if (force_nw) nw = force_nw;
    if (nw * 64 < best) nw = (best + 63) / 64;                           // one fold thread per row
    if (nw > 16) return false;
    uint32_t per = (TT + nw - 1) / nw;
    p.loop = per > 8;
    p.d = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
    p.rounds = p.loop ? (per + 7) / 8 : 1;
    p.nthr = nw * 64;
    p.nv = (a.n / 4 + p.nthr - 1) / p.nthr;
    if (nbq == 1 && p.nv > 4) return false;
    p.grid = 0;
    for (uint32_t s = 0; s < 3; s++) { p.wg[s] = s < nseg ? (a.seg[s].rows + best - 1) / best : 0; p.grid += p.wg[s]; }
    p.lds = chunk_lds_bytes(a.n, nbq == 1 && a.attn_part != nullptr, a.attn_n_head, nmat, best, nw, nbq > 1 ? nbq : p.loop ? 1u : p.d, nbq);
    if (nbq > 1) p.nv = 1;                                              // (nothing staged in registers)
    return p.lds <= 160u * 1024u;
}

template <int ROLE, int NV, int D, bool LOOP>
hipError_t launch_chunk_t(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    auto kern = &gemv_q4k_chunk_kernel<ROLE, NV, D, LOOP>;
    if (p.lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    // measurement (bench.py's `best_kernel`): the classifier launch between the kernel's own start / stop timestamps, as the Q80 STREAM
    // launch is timed (gemv_q80_impl.h); the backend arms the pair for the looping launch of one decode step only
    hipEvent_t e0 = LOOP ? g_q80_probe_start : nullptr, e1 = LOOP ? g_q80_probe_stop : nullptr;
    if (e0 && e1) {
        g_q80_probe_start = g_q80_probe_stop = nullptr;
        hipExtLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), (uint32_t)p.lds, st, e0, e1, 0, d);
    } else hipLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), p.lds, st, d);
    return hipGetLastError();
}
template <int ROLE, int NV>
hipError_t launch_chunk_d(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.loop) return launch_chunk_t<ROLE, NV, 8, true>(d, p, st);
    if (p.d == 1) return launch_chunk_t<ROLE, NV, 1, false>(d, p, st);
    if (p.d == 2) return launch_chunk_t<ROLE, NV, 2, false>(d, p, st);
    if (p.d == 4) return launch_chunk_t<ROLE, NV, 4, false>(d, p, st);
    return launch_chunk_t<ROLE, NV, 8, false>(d, p, st);
}
template <int ROLE>
hipError_t launch_chunk_r(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.nv <= 1) return launch_chunk_d<ROLE, 1>(d, p, st);
    if (p.nv <= 2) return launch_chunk_d<ROLE, 2>(d, p, st);
    return launch_chunk_d<ROLE, 4>(d, p, st);
}
// 2 .. 8 sequences: generic role, activations from q4k_quant_rows_kernel
template <int D, bool LOOP, int NB>
hipError_t launch_chunk_nb_t(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    auto kern = &gemv_q4k_chunk_kernel<R_GENERIC, 1, D, LOOP, NB>;
    if (p.lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(p.nthr), p.lds, st, d);
    return hipGetLastError();
}
template <int NB>
hipError_t launch_chunk_nb(const GemvDev &d, const ChunkPlan &p, hipStream_t st) {
    if (p.loop) return launch_chunk_nb_t<8, true, NB>(d, p, st);
    if (p.d == 1) return launch_chunk_nb_t<1, false, NB>(d, p, st);
    if (p.d == 2) return launch_chunk_nb_t<2, false, NB>(d, p, st);
    if (p.d == 4) return launch_chunk_nb_t<4, false, NB>(d, p, st);
    return launch_chunk_nb_t<8, false, NB>(d, p, st);
}
template <int ROLE, int NV>
hipError_t launch_quant_rows_t(const GemvDev &d, XGroup *out, uint32_t nb, size_t lds, hipStream_t st) {
    auto kern = &q4k_quant_rows_kernel<ROLE, NV>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(d.nthr), lds, st, d, out);
    return hipGetLastError();
}
// (only the prologue's flags matter to the quantizer: the role kernels' registers do not spill at four float4 per thread, the generic one's do)
template <int NV>
hipError_t launch_quant_rows_r(const GemvDev &d, XGroup *out, uint32_t nb, size_t lds, hipStream_t st) {
    if (d.flags == F_NORM) return launch_quant_rows_t<R_NORM_STORE, NV>(d, out, nb, lds, st);
    if (d.flags == 0) return launch_quant_rows_t<R_RESID, NV>(d, out, nb, lds, st);
    if (d.flags == F_COMBINE) return launch_quant_rows_t<R_RESID_COMBINE, NV>(d, out, nb, lds, st);
    return launch_quant_rows_t<R_GENERIC, NV>(d, out, nb, lds, st);
}

}  // namespace

// one sequence; 2 .. 8 sequences when the caller brings scratch for the staged groups (GemvArgs::q4_scratch) and the one-sequence launch of
// the same matrix is a chunk launch too (its plan gives the quantizer its thread count)
bool gemv_q4k_chunk_supports(const GemvArgs &a) {
    ChunkPlan p;
    if (a.nb <= 1) return plan_chunk(a, p);
    GemvArgs one = a; one.nb = 1;
    return a.q4_scratch && (size_t)a.nb * (a.n >> 5) * sizeof(XGroup) <= a.q4_scratch_bytes && plan_chunk(one, p) && plan_chunk(a, p);
}
bool gemv_q4k_chunk_loops(const GemvArgs &a) { ChunkPlan p; return a.nb == 1 && plan_chunk(a, p) && p.loop; }     // the persistent (classifier) variant
// (max, row) arg-max partials a classifier launch writes: one per workgroup (0: none, the arg-max kernel scans the logits)
uint32_t gemv_q4k_chunk_partials(const GemvArgs &a) {
    ChunkPlan p;
    if (a.nb != 1 || !a.tile_max || a.epi != GEMV_EPI_STORE || a.nseg != 1 || a.seg[0].out_pstride || !plan_chunk(a, p)) return 0;
    return p.grid;
}

// 2 .. 8 sequences: the quantizer launch (one workgroup per sequence, the one-sequence launch's prologue), then the projection
static hipError_t launch_gemv_q4k_chunk_batched(GemvArgs &a, hipStream_t st) {
    if (!gemv_q4k_chunk_supports(a)) return hipErrorInvalidValue;
    ChunkPlan p1, p;
    GemvArgs one = a; one.nb = 1;
    if (!plan_chunk(one, p1) || !plan_chunk(a, p)) return hipErrorInvalidValue;
    const uint32_t bpl = a.n >> 8, GT = bpl * 8u;
    XGroup *xg = reinterpret_cast<XGroup *>(a.q4_scratch);
    {
        GemvDev q = to_dev(a);                                       // flags: norm / combine as the launch asks
        q.nthr = p1.nthr; q.tile_max = nullptr;
        const size_t lds = (size_t)GT * sizeof(XGroup) + (16 + (a.attn_part ? (size_t)a.attn_n_head * 8 : 0)) * 4 + 16;
        const hipError_t e = p1.nv <= 1 ? launch_quant_rows_r<1>(q, xg, a.nb, lds, st) : p1.nv <= 2 ? launch_quant_rows_r<2>(q, xg, a.nb, lds, st) : launch_quant_rows_r<4>(q, xg, a.nb, lds, st);
        if (e != hipSuccess) return e;
    }
    GemvArgs g = a;
    g.norm_w = nullptr; g.attn_part = nullptr; g.attn_ml = nullptr; g.tile_max = nullptr;
    GemvDev d = to_dev(g);
    d.flags = F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(xg);
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    d.wg_c0 = nseg > 1 ? p.wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? p.wg[0] + p.wg[1] : 0xffffffffu;
    d.ntiles = 0;
    if (a.nb <= 2) return launch_chunk_nb<2>(d, p, st);
    if (a.nb <= 4) return launch_chunk_nb<4>(d, p, st);
    return launch_chunk_nb<8>(d, p, st);
}

// ---- the fused q | k | v + attention launch: host side -----------------------------------------------------------------------------------
static bool q4k_fused_shape(const GemvArgs &ga, const AttnArgs &aa, ChunkPlan &p) {
    if (ga.nb != 1 || ga.nseg != 3 || ga.epi != GEMV_EPI_STORE || !ga.norm_w || ga.xq_in || ga.x4_in || ga.attn_part || ga.tile_max || ga.resid_add) return false;
    if (ga.seg[0].out_pstride || ga.seg[1].out_pstride) return false;            // (only v is position indexed: its cache row)
    // 256 threads, like the attention workgroups: the kernel needs ~200 registers (the attention body), i.e. two waves per SIMD, and a CU must
    // seat a projection workgroup AND an attention workgroup (the first build ran the projection's own 384 threads: one workgroup per CU,
    // the attention started when the projection had left -- 0.575 -> 0.626 ms per step).  Same bits as the 384-thread launch: one float4 item
    // per thread on threads 0 .. n / 4 - 1 either way, a Q4K block is one wave's, the waves beyond add +0.0 to the norm's sum.
    if (ga.n > 1024u || !plan_chunk(ga, p, 4u) || p.loop || p.nthr != 256u || p.nv != 1u) return false;
    if (p.lds > 64u * 1024u) return false;
    return fused_attn_side_ok(aa, ga.seg[0].rows, ga.seg[1].rows, ga.seg[2].rows);
}
bool qkv_attn_fused_q4k_supports(const GemvArgs &ga, const AttnArgs &aa) { ChunkPlan p; return q4k_fused_shape(ga, aa, p); }

hipError_t launch_qkv_attn_fused_q4k(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    ChunkPlan p;
    if (!hand || !tick || !layer1 || layer1 > 127u || !q4k_fused_shape(ga, aa, p)) return hipErrorInvalidValue;
    GemvDev d = to_dev(ga);
    d.tile_max = nullptr; d.ntiles = 0;
    const uint32_t bpl = ga.n >> 8;
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    d.wg_c0 = p.wg[0]; d.wg_c1 = p.wg[0] + p.wg[1];
    AttnArgs a = aa;
    { uint32_t l2 = 0; while ((1u << l2) < a.n_kv_head) l2++; a.kv_log2 = l2; }
    { const uint32_t kv_mul = a.n_head / a.n_kv_head; uint32_t l2 = 0; while ((1u << l2) < kv_mul) l2++; a.kvmul_log2 = l2; }
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    h.base[0] = 0; h.base[1] = a.q_dim; h.base[2] = a.q_dim + a.kv_dim;
    const size_t hd4 = a.hd, lds_a = (hd4 + hd4 + 4 + 4 + 4 * hd4 + hd4) * sizeof(float);            // as gemv_q80_impl.h launch_qkv_attn_fused
    const size_t lds = p.lds > lds_a ? p.lds : lds_a;
    Q4FusedArgs fa{};
    fa.g = d; fa.a = a; fa.hand = h; fa.n_attn = a.n_head * a.nsplit; fa.head_wgs = a.n_head; fa.ngemv = p.grid;
    // naps of 16 x 64 cycles between the K / V requests and the first poll.  Same box, driver's flags (profiles/r06_q4k_fused.txt): the five
    // launches per layer 1733 / 1755 tok/s; fused with 2 naps 1773 / 1818, 4: 1782 / 1779, 6: 1742 / 1739, 8: 1684 / 1688
    // (re-swept at the end of round 6, producers first in the grid: 0 / 1 / 2 / 3 naps 1789 / 1761 / 1782 / 1780 and 1784 / 1778 / 1779 / 1775 tok/s: flat; none)
    fa.wait16 = 0u;
#define Q4F_GO(NV_, D_) do { hipLaunchKernelGGL((q4k_qkv_attn_fused_kernel<NV_, D_>), dim3(fa.n_attn + fa.ngemv), dim3(p.nthr), lds, st, fa); return hipGetLastError(); } while (0)
#define Q4F_NV(NV_) do { if (p.d == 1) Q4F_GO(NV_, 1); if (p.d == 2) Q4F_GO(NV_, 2); if (p.d == 4) Q4F_GO(NV_, 4); Q4F_GO(NV_, 8); } while (0)
    if (p.nv <= 1) Q4F_NV(1);
    if (p.nv <= 2) Q4F_NV(2);
    Q4F_NV(4);
#undef Q4F_NV
#undef Q4F_GO
}

// ---- the fused Wo + W1|W3 launch: host side ----------------------------------------------------------------------------------------------
struct Q4Wo13Plan { ChunkPlan a, b; };
static bool q4k_wo13_shape(const GemvArgs &wo, const GemvArgs &w13, Q4Wo13Plan &q) {
    if (wo.nb != 1 || w13.nb != 1) return false;
    if (wo.nseg != 1 || wo.epi != GEMV_EPI_RESID || wo.norm_w || wo.xq_in || wo.x4_in || wo.tile_max || wo.resid_add || wo.seg[0].out_pstride) return false;
    if (w13.nseg != 2 || w13.epi != GEMV_EPI_SWIGLU || !w13.norm_w || w13.xq_in || w13.x4_in || w13.attn_part || w13.tile_max || w13.resid_add) return false;
    if (w13.n != wo.seg[0].rows || w13.xin != wo.seg[0].out) return false;             // W1|W3's input is what Wo writes
    if (!plan_chunk(wo, q.a) || !plan_chunk(w13, q.b) || q.a.loop || q.b.loop) return false;
    // both bodies on the launch's threads: the two plans must agree (Wo has no tree, but its plan's registers per thread follow the count)
    if (q.a.nthr != q.b.nthr || q.a.grid > q.b.grid) return false;
    const uint32_t cus = w13.cus ? w13.cus : 256u;
    if (q.b.grid > cus) return false;                                                   // one workgroup per CU: the whole grid is resident
    if (q.a.lds > 64u * 1024u || q.b.lds > 64u * 1024u) return false;
    // instantiated: Qwen3-0.6B's shapes (Wo: one float4 item and one wave-load per thread / wave; W1|W3: one item, two wave-loads)
    return q.a.nv == 1u && q.a.d == 1u && q.b.nv == 1u && q.b.d == 2u;
}
bool wo_w13_fused_q4k_supports(const GemvArgs &wo, const GemvArgs &w13) { Q4Wo13Plan q; return q4k_wo13_shape(wo, w13, q); }

static void chunk_dev_fill(GemvDev &d, const GemvArgs &a, const ChunkPlan &p) {
    const uint32_t bpl = a.n >> 8;
    d.tile_max = nullptr; d.ntiles = 0;
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    d.wg_c0 = nseg > 1 ? p.wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? p.wg[0] + p.wg[1] : 0xffffffffu;
}

hipError_t launch_wo_w13_fused_q4k(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    Q4Wo13Plan q;
    if (!hand || !tick || !layer1 || layer1 > 127u || !q4k_wo13_shape(wo, w13, q)) return hipErrorInvalidValue;
    Q4Wo13Args fa{};
    fa.wo = to_dev(wo); fa.w13 = to_dev(w13);
    chunk_dev_fill(fa.wo, wo, q.a); chunk_dev_fill(fa.w13, w13, q.b);
    fa.wo_wgs = q.a.grid;
    fa.wait16 = 4u;                      // (workgroups that produce nothing: none on Qwen3-0.6B's shapes, where both grids are 256)
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    fa.hand = h;
    const size_t lds = q.a.lds > q.b.lds ? q.a.lds : q.b.lds;
    const bool comb = (fa.wo.flags & F_COMBINE) != 0;
    if (comb) hipLaunchKernelGGL((q4k_wo_w13_fused_kernel<R_RESID_COMBINE, 1, 1, 1, 2>), dim3(q.b.grid), dim3(q.b.nthr), lds, st, fa);
    else hipLaunchKernelGGL((q4k_wo_w13_fused_kernel<R_RESID, 1, 1, 1, 2>), dim3(q.b.grid), dim3(q.b.nthr), lds, st, fa);
    return hipGetLastError();
}

hipError_t launch_gemv_q4k_chunk(GemvArgs &a, hipStream_t st) {
    if (a.nb > 1) return launch_gemv_q4k_chunk_batched(a, st);
    ChunkPlan p;
    if (!plan_chunk(a, p)) return hipErrorInvalidValue;
    GemvDev d = to_dev(a);
    if (a.x4_in) { d.flags |= F_PRE; d.xq_in = reinterpret_cast<const int8_t *>(a.x4_in); }
    const uint32_t bpl = a.n >> 8;
    d.rw = p.rw; d.nthr = p.nthr; d.units = p.rounds;
    d.magic_nchunk = (uint32_t)(((1ull << 32) + bpl - 1) / bpl);
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    d.wg_c0 = nseg > 1 ? p.wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? p.wg[0] + p.wg[1] : 0xffffffffu;
    d.ntiles = gemv_q4k_chunk_partials(a);
    if (!d.ntiles) d.tile_max = nullptr;
    const uint32_t f = d.flags;
    if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_chunk_r<R_NORM_STORE>(d, p, st);
    if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_chunk_r<R_RESID>(d, p, st);
    if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_chunk_r<R_RESID_COMBINE>(d, p, st);
    if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU) return launch_chunk_r<R_NORM_SWIGLU>(d, p, st);
    return launch_chunk_r<R_GENERIC>(d, p, st);
}

}  // namespace nano
