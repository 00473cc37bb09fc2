// route.hip -- which kernel a projection launch (out = W . act, one weight tensor or a run of them sharing the activation) goes to.
// ONE router for the decode step (backend.hip), batched prefill and the operator entry points (ops.hip), so that the operator tests
// exercise exactly the launches a step issues.
//
//   FP32 / Q4K              the GEMV kernels (gemv_f32.hip, gemv_q4k.hip), more than 8 sequences in groups
//   Q80, fast path          SLAB GEMV (1..8 sequences on the small per-layer matrices; 1..2 on those of >= 8 M weights)   gemv_q80_impl.h
//                           G6 (fragment-order activations: MODE S staged in LDS, MODE F per item; 2..64 tokens) gemm_q80_g6.hip
//                           G7 (17..64 tokens: loader / consumer engine, both operands through LDS) gemm_q80_g7.hip
//                           G2 (group sizes other than 64, rows no multiple of 256: the reference's order) gemm_q80.hip
//                           STREAM GEMV / GC for the classifier                                   gemv_q80_impl.h, gemm_q80_cls.hip
//   Q80, strict mode        the kernels that keep the reference's ascending group order: SLAB, GC, G2 (a.ordered = 1)
#include <stdlib.h>
#include "kernels.h"

namespace nano {

static uint32_t route_rows(const GemvArgs &a) {
    if (a.epi == GEMV_EPI_SWIGLU) return a.seg[0].rows;
    uint32_t r = 0;
    for (uint32_t s = 0; s < a.nseg; s++) r += a.seg[s].rows;
    return r;
}
// every workgroup of a multi-sequence GEMV launch re-quantizes the nb x n activations: ~ workgroups x elements of redundant work
static bool gemv_is_heavy(const GemvArgs &a) { return (uint64_t)(route_rows(a) / 16) * a.nb * a.n > (4u << 20); }
// per-layer matrices of >= 8 M weights (Qwen3-4B's): bandwidth rather than latency bound
bool route_is_wide(const GemvArgs &a) { const uint32_t rows = route_rows(a); return rows < 65536u && (uint64_t)rows * a.n >= (8u << 20); }

// the rmsnorm sum-of-squares tree the activation quantizer launch must repeat for this matrix (launch_quant_rows_frag order)
// (512 threads on the wide matrices -- the order the >= 3-sequence launches of Qwen3-4B have had since round 4; the one- and
// two-sequence SLAB launches of those matrices run the tree of their own thread count, see kernels.h "what a batch shares")
uint32_t route_norm_order(const Q80Route &r, const GemvArgs &a) {
    (void)r;
    return (q80_canonical(a) && route_is_wide(a) && a.n <= 10240u) ? 512u : 256u;
}

RouteKind route_kind(const Q80Route &r, const GemvArgs &a) {
    if (r.quant == NANO_QUANT_Q4K) return ROUTE_Q4K;
    if (r.quant != NANO_QUANT_Q80) return a.nb > 8 ? ROUTE_GEMV_SLICED : ROUTE_GEMV;
    const bool scratch = r.gq && r.gxs;
    const bool canon = q80_canonical(a);
    const bool wide = route_is_wide(a);
    // two sequences on wide matrices: the balanced SLAB GEMV (capacity 2) -- measured against G6 MODE P on one box, round 5: Qwen3-4B 1.833 vs
    // 1.923 ms per step (profiles/r05_wide_two_sequences.txt); from three sequences on the batched route is the faster one (four: 1.99 vs 2.80)
    if (canon && wide && a.nb == 2 && !a.xq_in) return ROUTE_GEMV;
    if (canon) {
        const bool batched = a.nb >= r.mfma_min_nb || (r.mfma_min_nb == 9 && ((a.nb == 8 && gemv_is_heavy(a)) || (wide && a.nb >= 2)));
        if (batched && scratch && gemm_q80_g7_supports(a)) return ROUTE_FRAG_G7;      // 17..64 tokens, where it pays
        if (batched && scratch && !a.attn_part && !a.resid_add && gemm_q80_g6_supports(a)) return ROUTE_FRAG_G6;
    }
    // launches that are not canonical (strict mode, other group sizes, the classifier): 9..64 sequences always; 8 sequences when the matrix is
    // large; per-layer matrices of >= 8 M weights from 2 sequences on; the classifier keeps its STREAM GEMV up to 7 sequences
    bool mfma = false;
    if (scratch) {
        if (a.nb >= r.mfma_min_nb) mfma = true;
        else if (r.mfma_min_nb == 9) mfma = (a.nb == 8 && gemv_is_heavy(a)) || (a.nb >= 2 && wide);
    }
    // (a CANONICAL launch neither G6 nor G7 takes does not go to G2, whose fold is the reference's: it would no longer be bit for bit its
    //  sequences alone -- it runs through the GEMV kernels in groups of 8 below)
    if (mfma && !canon && !a.attn_part && !a.resid_add && gemm_q80_g2_supports(a)) return ROUTE_FRAG_OLD;
    if (a.nb > 8) return ROUTE_GEMV_SLICED;
    if (a.nb > 1 && !a.attn_part && !a.xq_in && scratch && gemv_is_heavy(a)) return ROUTE_GEMV_PREQ;
    return ROUTE_GEMV;
}

// the sequences [b0, b0 + cnt) of a launch, as a launch of their own (every per-sequence pointer advanced)
static GemvArgs gemv_slice(const GemvArgs &a, uint32_t b0, uint32_t cnt) {
    GemvArgs s = a;
    s.nb = cnt;
    for (uint32_t i = 0; i < a.nseg; i++) if (s.seg[i].out) s.seg[i].out += (size_t)b0 * a.seg[i].out_bstride;
    if (a.xin) s.xin += (size_t)b0 * a.xin_bstride;
    if (a.pos) s.pos += b0;
    if (a.xq_in) s.xq_in += (size_t)b0 * ((a.n + 15) & ~15u);
    if (a.xs_in) s.xs_in += (size_t)b0 * (a.n / a.gs);
    if (a.attn_part) { s.attn_part += (size_t)b0 * a.attn_nsplit * a.n; s.attn_ml += (size_t)b0 * a.attn_n_head * a.attn_nsplit * 2; }
    if (a.resid_add) s.resid_add += (size_t)b0 * a.resid_add_bstride;
    s.tile_max = nullptr;
    return s;
}

hipError_t route_projection(const Q80Route &r, GemvArgs &a, hipStream_t st) {
    const uint32_t max_wg = (r.cus ? (uint32_t)r.cus : 256u) * 8u;
    a.cus = (uint32_t)r.cus;
    const RouteKind k = route_kind(r, a);
    switch (k) {
    case ROUTE_Q4K: {
        // every workgroup stages the whole quantized activation of each sequence in LDS: long rows (Qwen3-4B's hidden size)
        // take fewer sequences per launch
        a.q4_scratch = r.q4x; a.q4_scratch_bytes = r.q4x_bytes;
        uint32_t fit = a.nb > 1 ? gemv_q4k_fit_batch(a) : 1u;
        if (a.nb > 1) {                                                  // round 5: whole-block launches share every weight byte among up to 8 sequences
            GemvArgs probe = a; probe.nb = a.nb < 8u ? a.nb : 8u;
            if (gemv_q4k_chunk_takes(probe)) fit = 8u;
        }
        if (a.nb <= fit) return launch_gemv_q4k(a, max_wg, st);
        for (uint32_t b0 = 0; b0 < a.nb; b0 += fit) {
            GemvArgs s = gemv_slice(a, b0, a.nb - b0 < fit ? a.nb - b0 : fit);
            const hipError_t e = launch_gemv_q4k(s, max_wg, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    case ROUTE_FRAG_G6:
    case ROUTE_FRAG_G7:
    case ROUTE_FRAG_OLD: {
        // quantize every sequence's activation once, straight into MFMA fragment order (unless the producing kernel already did), then
        // the GEMM
        if (!a.frag_ready) {
            const hipError_t e = launch_quant_rows_frag(a.xin, a.xin_bstride, a.norm_w, a.n, a.gs, a.nb, r.gq, r.gxs, st, route_norm_order(r, a));
            if (e != hipSuccess) return e;
        }
        a.xq_in = r.gq; a.xs_in = r.gxs;
        if (k == ROUTE_FRAG_G6) return launch_gemm_q80_g6(a, st);
        if (k == ROUTE_FRAG_G7) return launch_gemm_q80_g7(a, st);
        // the classifier of a batched step: GC (persistent waves, the activation fragments staged in LDS once per workgroup)
        if (gemm_q80_cls_supports(a)) return launch_gemm_q80_cls(a, st);
        return launch_gemm_q80_g2(a, st);
    }
    case ROUTE_GEMV_SLICED:
        // More sequences than a GEMV launch takes and a launch the GEMM does not take (row length / group size not a multiple of 4
        // groups, segment rows not multiples of 16, the LoRA o-branch addend): groups of 8 through the GEMV kernels.  Same arithmetic
        // per sequence, the weights are read once per group.
        for (uint32_t b0 = 0; b0 < a.nb; b0 += 8) {
            GemvArgs s = gemv_slice(a, b0, a.nb - b0 < 8 ? a.nb - b0 : 8u);
            const hipError_t e = launch_gemv(r.quant, s, max_wg, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    case ROUTE_GEMV_PREQ: {
        // when the redundant quantization outweighs a launch (~3 us) the activations are quantized once (quant_rows_kernel) and the GEMV
        // reads them back
        const hipError_t e = launch_quant_rows(a.xin, a.xin_bstride, a.norm_w, a.n, a.gs, a.nb, r.gq, r.gxs, st);
        if (e != hipSuccess) return e;
        a.xq_in = r.gq; a.xs_in = r.gxs; a.norm_w = nullptr;
        return launch_gemv(r.quant, a, max_wg, st);
    }
    case ROUTE_GEMV:
    default:
        return launch_gemv(r.quant, a, max_wg, st);
    }
}

}  // namespace nano
