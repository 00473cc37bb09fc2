// kernels.h -- host-visible argument blocks and launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/nano_mi355x.h"
#include "device_common.h"

namespace nano {

enum : uint32_t { GEMV_EPI_STORE = 0, GEMV_EPI_RESID = 1, GEMV_EPI_SWIGLU = 2 };

// One weight tensor (or a run of them sharing the input vector) of a fused GEMV launch.
struct GemvSeg {
    const void *w;          // FP32: float[rows][n]; Q80: int8[rows][n]; Q4K: 160-byte blocks, 16-B aligned
    const float *ws;        // Q80: float[rows][n/gs]
    float *out;             // output base
    uint32_t rows;
    uint32_t out_bstride;   // floats between sequence slots
    uint32_t out_pstride;   // floats per position (KV-cache rows); 0 = not position indexed
    uint32_t _pad;
};

struct GemvArgs {
    GemvSeg seg[3];
    uint32_t nseg;
    uint32_t n;             // row length (input vector length)
    uint32_t gs;            // Q80 group size
    uint32_t nb;            // live sequences (<= template capacity)
    const float *xin;       // fp32 input vectors
    uint32_t xin_bstride;
    uint32_t epi;
    const float *norm_w;    // rmsnorm weight or nullptr
    const uint32_t *pos;    // device positions [nb] (for out_pstride)
    uint32_t tiles;         // filled by the launcher
    uint32_t frag_ready;    // batched GEMM path: 1 = the fragment-order activations already exist in the step's scratch (the attention kernel wrote them)
    // operator-test inputs: an already quantized activation (skips the quantizing prologue)
    const int8_t *xq_in;    // Q80 int8[n] (batched GEMM path with frag_ready: all tokens, MFMA B-fragment order)
    const float *xs_in;     // Q80 float[n/gs]
    const uint8_t *x4_in;   // Q4K blocks[ceil(n/256)*160]
    uint8_t *q4_scratch; size_t q4_scratch_bytes;   // Q4K, 2 .. 8 sequences: room for the staged groups of every sequence (nb * n bytes), or nullptr
    // input = combination of split attention partials (attn.hip) instead of xin:
    //   x[b][i] = sum_s part[b][s][i] * w[b][head(i)][s],  w from the (max, sum) pairs in attn_ml
    const float *attn_part; // [nb][nsplit][n] unnormalised partial outputs, or nullptr
    const float *attn_ml;   // [nb][n_head][nsplit][2] (running max, exp-sum) per split
    uint32_t attn_nsplit, attn_n_head, attn_hd, _pad2;
    // residual epilogue only: an extra vector added to the GEMV result BEFORE the residual add, x += (W.act + resid_add)
    // (the LoRA o-branch, reference infer.c:898-908); [nb][resid_add_bstride] or nullptr
    const float *resid_add; uint32_t resid_add_bstride, _pad3;
    // optional per-tile arg-max partials of a STORE launch: tile_max[b][tile] = (max value, row index bits)
    float *tile_max;
    uint32_t cus;           // compute units of the device the launch goes to (0: assume 256); sizes the work split
    uint32_t ordered;       // 1: strict mode -- every fp32 group fold in the reference's ascending order (infer.c:668-674); 0: the fast
                            // path's CANONICAL fold where it applies (q80_canonical(): unit sums of 8 groups, units ascending)
    uint32_t *err;          // sticky error word of the model (device pointer to host-mapped memory), or nullptr: see NANO_DEVERR_*
    unsigned long long *stamps;   // measurement builds only (NANO_STAMPS): per-workgroup phase stamps, or nullptr
};

// Bounded waits inside kernels (G6's finisher on its tile counter, the fused launches' consumers on their granules) must not hang the device; a wait that gives up ORs its code (device_common.h NANO_DEVERR_*) into the model's
// sticky error word: the next synchronising C-ABI call re-issues the work through the plain launches (hand-offs) or returns NANO_HIP_ERUNTIME
// instead of results computed from whatever was there (round-4 / round-5 advice).

// The fast path's reduction shape of a Q80 projection (group size 64, row length a multiple of 256; not the classifier-like tall
// STORE launches, whose kernels hold whole rows per wave and keep the reference's order): row = ((S_0 + S_1) + ...), S_u = the 8
// group products of unit u added in ascending order.  Every kernel a launch can be routed to (SLAB GEMV, G6, G7) implements this
// one shape: given the same quantized activations, a batch's projections are bit for bit its sequences' alone whatever route each
// size takes.  What a batch shares beyond that: the rmsnorm sum-of-squares tree in front of the quantizer is 256 threads wide on
// every route of the small matrices (Qwen3-0.6B: batches ARE their sequences alone end to end, tests/test_gpu_fullsize.py::test_batch_equals_sequences_alone_and_runs_repeat asserts it);
// on the wide matrices (route_is_wide(), Qwen3-4B) the one- and two-sequence SLAB launches run the tree of their own thread count
// and the >= 3-sequence launches the 512-thread one (route_norm_order()), so a scale may differ in its last ulp between batch
// sizes there -- inside the fast path's stated tolerance, not a bit-for-bit promise.  Strict mode: never canonical.
inline bool q80_canonical(const GemvArgs &a) {
    if (a.ordered || a.gs != 64 || a.n % 256u) return false;
    if (a.nseg == 1 && a.epi == GEMV_EPI_STORE && a.seg[0].rows >= 16384u && a.seg[0].out_pstride == 0) return false;
    return true;
}
uint32_t gemv_tiles(uint32_t quant, const GemvArgs &a);   // tiles launch_gemv() will use (sizes tile_max)
hipError_t launch_gemv(uint32_t quant, GemvArgs &a, uint32_t max_wg, hipStream_t st);
hipError_t launch_gemv_q4k(GemvArgs &a, uint32_t max_wg, hipStream_t st);
bool gemv_q4k_chunk_supports(const GemvArgs &a);            // gemv_q4k_chunk.hip: one sequence, whole 256-value blocks
bool gemv_q4k_chunk_takes(const GemvArgs &a);               // ... or 2 .. 8 sequences where the chunk form is the faster one (gemv_q4k.hip)
bool gemv_q4k_chunk_loops(const GemvArgs &a);               // ... and the launch is the looping (classifier) variant
uint32_t gemv_q4k_chunk_partials(const GemvArgs &a);
hipError_t launch_gemv_q4k_chunk(GemvArgs &a, hipStream_t st);
uint32_t gemv_q4k_fit_batch(const GemvArgs &a);            // sequences per Q4K launch that fit in LDS (8 | 4 | 2 | 1)
hipError_t launch_gemv_q80(const GemvArgs &a, hipStream_t st);      // gemv_q80.hip
hipError_t launch_gemv_f32(const GemvArgs &a, hipStream_t st);      // gemv_f32.hip
// 9..64 tokens per weight read on the int8 matrix cores.  G2 (gemm_q80.hip): the general kernel in the reference's ascending group order --
// strict mode's batched route, and the launches the canonical-fold kernels do not take (group sizes other than 64, rows that are no multiple
// of 256); activations in MFMA B-fragment order (a.xq_in / a.xs_in = launch_quant_rows_frag's output)
bool gemm_q80_g2_supports(const GemvArgs &a);                       // host predicate: shapes / features the GEMM takes
hipError_t launch_gemm_q80_g2(const GemvArgs &a, hipStream_t st);
// G6 (gemm_q80_g6.hip): the fast path's split-K kernel, canonical fold, group size 64, fragment-order activations (MODE S: staged in LDS
// once per workgroup, <= 16 tokens; MODE F: fetched per item, 1 / 2 / 4 token tiles)
bool gemm_q80_g6_supports(const GemvArgs &a);
hipError_t launch_gemm_q80_g6(const GemvArgs &a, hipStream_t st);
// G7 (gemm_q80_g7.hip): the fast path's kernel for 17..64 tokens -- LDS-DMA loader waves stream weights AND activation fragments through an
// LDS ring, consumer waves own (row tile, token tile) pairs for the whole row length (canonical fold in registers); same inputs as MODE F
bool gemm_q80_g7_supports(const GemvArgs &a);
hipError_t launch_gemm_q80_g7(const GemvArgs &a, hipStream_t st);
// GC, tall matrices with short rows (the classifier): persistent waves, activation fragments staged in LDS (gemm_q80_cls.hip)
bool gemm_q80_cls_supports(const GemvArgs &a);
hipError_t launch_gemm_q80_cls(const GemvArgs &a, hipStream_t st);
// order: threads of the rmsnorm sum-of-squares tree -- 256 (the SLAB GEMV prologue's of the small matrices) or 512 (the wide matrices' batched launches, route_norm_order())
hipError_t launch_quant_rows_frag(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n, uint32_t gs, uint32_t nb,
                                  int8_t *xf, float *xsf, hipStream_t st, uint32_t order = 256);
hipError_t launch_quant_rows(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n, uint32_t gs, uint32_t nb,
                             int8_t *xq, float *xs, hipStream_t st);
uint32_t gemv_q80_partials(const GemvArgs &a);
uint32_t gemv_q4k_partials(const GemvArgs &a);   // Q4K: one (max, row) partial per workgroup of a one-segment STORE launch with tile_max

// ---- routing (route.hip): which kernel a projection launch goes to ------------------------------------------------------------
enum RouteKind : uint32_t {
    ROUTE_GEMV = 0,        // one GEMV launch (FP32 / Q80 SLAB or STREAM), activation quantized in its prologue
    ROUTE_GEMV_PREQ,       // Q80: row-major quantizer launch + GEMV reading the quantized rows (2..8 sequences on large inputs)
    ROUTE_GEMV_SLICED,     // more than 8 sequences through the GEMV kernels in groups of 8
    ROUTE_Q4K,
    ROUTE_RESERVED,        // (round 4's G6 MODE P; the value stays so that the route numbers the tests read do not move)
    ROUTE_FRAG_G6,         // fragment-order activations (quantizer launch unless frag_ready) + G6 MODE F
    ROUTE_FRAG_OLD,        // fragment-order activations + GC (the classifier) | G2 (the reference's group order)
    ROUTE_FRAG_G7,         // fragment-order activations + G7 (17..64 tokens, the fast path)
};
inline bool route_takes_fragments(RouteKind k) { return k == ROUTE_FRAG_G6 || k == ROUTE_FRAG_OLD || k == ROUTE_FRAG_G7; }
inline bool route_takes_attn_parts(RouteKind k) { return k == ROUTE_GEMV || k == ROUTE_Q4K; }
struct Q80Route {
    uint32_t quant; int cus;
    uint32_t mfma_min_nb;  // sequences from which the small Q80 matrices take the batched route (9; NANO_MFMA_MIN_NB)
    int8_t *gq; float *gxs;  // fragment-order activation scratch (nullptr: no batched route)
    uint8_t *q4x; size_t q4x_bytes;   // Q4K: scratch for the staged groups of 2 .. 8 sequences (gemv_q4k_chunk.hip), or nullptr
};
RouteKind route_kind(const Q80Route &r, const GemvArgs &a);
hipError_t route_projection(const Q80Route &r, GemvArgs &a, hipStream_t st);
uint32_t route_norm_order(const Q80Route &r, const GemvArgs &a);
bool route_is_wide(const GemvArgs &a);

// ---- attention ------------------------------------------------------------------------------------
// up to attention_split_cap() (default 32, capacity 64) splits of a range beyond attention_wide_from() positions (default 2048; <= 8
// below, which the Wo GEMV's prologue combines: attention_nsplit())
constexpr uint32_t ATTN_MAX_NSPLIT = 64;
uint32_t attention_wide_from();
uint32_t attention_split_cap();
struct AttnArgs {
    const float *q;         // [nb][q_dim] raw q from the QKV GEMV (normed + roped in LDS, head-local)
    float *q_out;           // optional [nb][q_dim]: finished q written back (debug / traces), or nullptr
    const float *kraw;      // [nb][kv_dim] raw k of the current position (nullptr: k row already final in cache)
    float *kcache;          // [nb][L][S][kv_dim]
    float *vcache;
    const uint32_t *pos;    // [nb]
    const float *q_norm;    // [hd] for this layer or nullptr
    const float *k_norm;
    const float *rope_cos;  // [rows][hd/2] or nullptr (no rope)
    const float *rope_sin;
    const float *rope_cur;  // optional [nb][2][hd/2]: the rows of pos[b], staged by the embed kernel (else read from the tables)
    float *out;             // nsplit > 1: [nb][nsplit][q_dim] UNNORMALISED partial outputs  sum_t exp(s_t - m) v_t
    float *ml;              // nsplit > 1: [nb][n_head][nsplit][2]  (m = max score of the split, l = sum exp(s_t - m))
    float *xba_out;         // nsplit == 1: [nb][q_dim] final (normalised) head outputs
    uint32_t nsplit;        // timestep blocks are dealt round-robin to `nsplit` workgroups per (KV group, sequence)
    uint32_t range_hint;    // host's upper bound of the attended range (>= pos+1 of every sequence; S when not causal)
    uint32_t layer, n_layer, S, hd, n_head, n_kv_head, q_dim, kv_dim;
    uint32_t rope_qwen3;    // 1: (i, i+hd/2) pairs, 0: adjacent pairs
    uint32_t is_causal;
    uint32_t cache_bstride_rows;   // = L*S rows between slots (in units of kv_dim floats)
    uint32_t fixed_range;          // op-test mode: attend over rows [0, fixed_range) of an externally filled cache
    uint32_t prep_only;            // 1: finish and store the k row of pos[b] (norm + RoPE), then return (batched prefill, pass 1)
    uint32_t kv_half;              // 1: kcache / vcache hold FP16 elements (opt-in); the fresh v row comes from vraw
    const float *vraw;             // FP16 cache: [nb][kv_dim] v of the current position from the QKV GEMV (FP32 scratch), else nullptr
    // single-split launches only, optional: the finished output also leaves as Q80 groups of 64 in MFMA B-fragment order (what
    // quant_rows_frag_kernel would make of xba_out), so the batched Wo GEMM needs no quantizer launch.  head_dim % 64 == 0.
    int8_t *xf_out; float *xsf_out;
    // filled by launch_attention(): workgroup x -> q heads so that the workgroups sharing a KV head run on ONE XCD (workgroup
    // index mod 8 = XCD, each XCD has its own L2): x = sub * n_kv_head + kv head, when n_kv_head is a power of two >= 8;
    // kv_log2 = log2(n_kv_head), else 0xffffffff = plain order (x = first head / heads per workgroup)
    uint32_t kv_log2, kvmul_log2;   // kvmul_log2 = log2(n_head / n_kv_head) (decode modes: both head counts are powers of two)
    // PAGED KV cache (opt-in): kcache / vcache are pools [layer][page][64 positions][kv_dim]; a sequence's 64-position
    // block j lives in the page whose first row (page * 64, rows counted inside a layer plane) is pt_rows[slot][j]
    const uint32_t *pt_rows;        // [slots][pt_stride], 0xffffffff = no page; nullptr = the contiguous cache
    const uint32_t *kvrow;          // [nb] pool row of position pos[b] (staged by the embed kernel next to the RoPE row)
    uint32_t pt_stride, pt_bstride; // entries per slot; entries between the sequences of this launch (0: batched prefill, one slot)
    uint32_t pool_rows, _pad6;      // rows of one layer plane = pages * 64
    uint32_t *err;          // sticky error word (as GemvArgs::err)
    unsigned long long *stamps;   // measurement builds only (NANO_STAMPS): per-workgroup phase stamps, or nullptr
};
hipError_t launch_attention(const AttnArgs &a, uint32_t nb, hipStream_t st);
// q | k | v projection (Q80 group size 64, one sequence) + Qwen3 decode attention as ONE launch (gemv_q80_impl.h): the attention workgroups wait
// for q / k / v as 8-byte {tag, value} granules in `hand` (q_dim + 2 kv_dim entries); tag = the step's tick * 128 + layer1 (device_common.h)
bool qkv_attn_fused_supports(const GemvArgs &ga, const AttnArgs &aa);
hipError_t launch_qkv_attn_fused(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
// Wo + W1|W3 of one sequence in ONE launch (gemv_q80_impl.h wo_w13_fused_kernel): x reaches W1|W3 as granules of the same launch
bool wo_w13_fused_supports(const GemvArgs &wo, const GemvArgs &w13);
hipError_t launch_wo_w13_fused(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
// W2 of a layer + q | k | v + attention of the next one in ONE launch (w2_qkv_attn_fused_kernel): x as granules (xhand, tag of layer1), then
// q / k / v as granules (hand, tag of layer1 + 1)
bool w2_qkv_attn_fused_supports(const GemvArgs &w2, const GemvArgs &ga, const AttnArgs &aa);
hipError_t launch_w2_qkv_attn_fused(const GemvArgs &w2, const GemvArgs &ga, const AttnArgs &aa, unsigned long long *xhand, unsigned long long *hand,
                                    uint32_t *tick, uint32_t layer1, hipStream_t st);
// the attention side of the fused one-sequence launches (Q80: gemv_q80_impl.h, Q4K: gemv_q4k_chunk.hip): Qwen3 decode attention on an FP32
// contiguous cache, head_dim 128, one head per workgroup, two timestep blocks in flight; qr / kr / vr = rows of the q | k | v segments
inline bool fused_attn_side_ok(const AttnArgs &aa, uint32_t qr, uint32_t kr, uint32_t vr) {
    if (aa.hd != 128u || !aa.q_norm || !aa.k_norm || !aa.rope_qwen3 || !aa.rope_cos || !aa.rope_cur || !aa.kraw || aa.fixed_range || !aa.is_causal || aa.q_out) return false;
    if (aa.kv_half || aa.pt_rows || aa.prep_only || aa.xf_out || aa.nsplit == 0 || aa.nsplit > 8u) return false;
    const uint32_t kv_mul = aa.n_kv_head ? aa.n_head / aa.n_kv_head : 0u;
    if (!aa.n_kv_head || (aa.n_kv_head & (aa.n_kv_head - 1u)) || !kv_mul || (kv_mul & (kv_mul - 1u))) return false;
    if ((uint64_t)aa.n_head * aa.nsplit > 256u) return false;                   // (beyond: the attention launcher puts several heads in a workgroup)
    if (aa.range_hint > aa.nsplit * 2u * 32u) return false;                     // (more than one round: the launcher may pick four blocks in flight)
    return aa.q_dim == qr && aa.kv_dim == kr && aa.kv_dim == vr && aa.q_dim == aa.n_head * aa.hd;
}
// ... and of the fused launch of FP32 models (gemv_f32.hip): the plain decode mode (no q / k norm, adjacent-pair RoPE: Nano), head_dim <= 64
inline bool fused_attn_side_ok_plain(const AttnArgs &aa, uint32_t qr, uint32_t kr, uint32_t vr) {
    if (aa.hd < 4u || aa.hd > 64u || aa.hd % 4u || aa.q_norm || aa.k_norm || aa.rope_qwen3 || !aa.rope_cos || !aa.rope_cur || !aa.kraw || aa.fixed_range || !aa.is_causal || aa.q_out) return false;
    if (aa.kv_half || aa.pt_rows || aa.prep_only || aa.xf_out || aa.nsplit == 0 || aa.nsplit > 8u) return false;
    const uint32_t kv_mul = aa.n_kv_head ? aa.n_head / aa.n_kv_head : 0u;
    if (!aa.n_kv_head || (aa.n_kv_head & (aa.n_kv_head - 1u)) || !kv_mul || (kv_mul & (kv_mul - 1u))) return false;
    if ((uint64_t)aa.n_head * aa.nsplit > 256u) return false;
    if (aa.range_hint > aa.nsplit * 2u * 32u) return false;
    return aa.q_dim == qr && aa.kv_dim == kr && aa.kv_dim == vr && aa.q_dim == aa.n_head * aa.hd;
}
bool qkv_attn_fused_f32_supports(const GemvArgs &ga, const AttnArgs &aa);
hipError_t launch_qkv_attn_fused_f32(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
bool wo_w13_fused_f32_supports(const GemvArgs &wo, const GemvArgs &w13);
hipError_t launch_wo_w13_fused_f32(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
// the same launch for Q4K (gemv_q4k_chunk.hip q4k_qkv_attn_fused_kernel, round 6)
bool qkv_attn_fused_q4k_supports(const GemvArgs &ga, const AttnArgs &aa);
hipError_t launch_qkv_attn_fused_q4k(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
bool wo_w13_fused_q4k_supports(const GemvArgs &wo, const GemvArgs &w13);
hipError_t launch_wo_w13_fused_q4k(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st);
uint32_t attention_nsplit(uint32_t range_hint, uint32_t hd);
hipError_t launch_attn_combine(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit, hipStream_t st);
hipError_t launch_attn_combine_tokens(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit, uint32_t nb,
                                      int8_t *xf_out, float *xsf_out, hipStream_t st);   // xf_out: also Q80 fragments (AttnArgs::xf_out), or nullptr

// ---- strict-parity kernels (strict.hip): every float reduction in the reference's own order --------------
struct StrictAttnArgs {
    float *q;               // [nb][q_dim] raw q, finished (norm + RoPE) in place
    const float *kraw;      // [nb][kv_dim] raw k of the current position
    float *kcache, *vcache; // [slots][L][S][kv_dim]
    const uint32_t *pos;    // [nb]
    const float *q_norm, *k_norm;          // [hd] of this layer or nullptr
    const float *rope_cos, *rope_sin;      // [rows][hd/2]
    float *att;             // [nb][n_head][S] scores -> probabilities
    float *xba;             // [nb][q_dim] head outputs
    uint32_t n_head, n_kv_head, hd, q_dim, kv_dim, layer, n_layer, S;
    uint32_t slot0;         // KV slot of sequence 0 of this step (sequence b lives in slot0 + b)
    uint32_t rope_qwen3, is_causal, _pad;
};
hipError_t launch_strict_rmsnorm(float *o, const float *x, const float *w, uint32_t n, uint32_t nvec, uint32_t x_stride, uint32_t o_stride, hipStream_t st);
hipError_t launch_strict_qk(const StrictAttnArgs &a, uint32_t nb, hipStream_t st);
hipError_t launch_strict_attention(const StrictAttnArgs &a, uint32_t nb, hipStream_t st);
hipError_t launch_strict_swiglu(float *hb, const float *hb2, uint32_t n, uint32_t nb, uint32_t bstride, hipStream_t st);
hipError_t launch_strict_matmul_f32(float *out, const float *x, const float *w, uint32_t n, uint32_t d, uint32_t nb, uint32_t x_bstride,
                                    uint32_t out_bstride, uint32_t out_pstride, const uint32_t *pos, int resid, hipStream_t st);

// ---- LoRA side branches (lora.hip) -------------------------------------------------------------------
struct LoraArgs {
    const float *x;         // qkv: residual stream x [nb][E]; o: attention output xba [nb][E]
    const float *norm_w;    // qkv: rms_attn weight of the layer
    const float *qa, *qb, *ka, *kb, *va, *vb;      // this layer's pairs; o: qa / qb = the o pair
    float *q, *kraw, *v;    // qkv: in-place targets (v = cache base of slot 0 / of the prefill slot); o: q = o1 out
    const uint32_t *pos;
    uint32_t E, KD, rank, alpha, v_bstride, _pad;
};
hipError_t launch_lora_qkv(const LoraArgs &a, uint32_t nb, hipStream_t st);
hipError_t launch_lora_o(const LoraArgs &a, uint32_t nb, hipStream_t st);

// ---- small kernels ----------------------------------------------------------------------------------
struct EmbedArgs {
    const void *tok;        // FP32 float[V][E] | Q80 int8[V][E] | Q4K blocks
    const float *tok_s;     // Q80 scales
    const uint32_t *tokens; // [nb]
    float *x;               // [nb][E]
    uint32_t E, gs, quant, x_bstride;
    // the step's first kernel also stages the RoPE row of each sequence's position at a FIXED address, so that
    // the attention kernels need no pos-dependent load: rope_cur[b] = { cos[pos[b]][0..half), sin[pos[b]][0..half) }
    const float *rope_cos; const float *rope_sin; const uint32_t *pos; float *rope_cur; uint32_t half, _pad;
    // paged KV cache: kvrow[b] = pt_rows[b * pt_bstride + pos[b] / 64] + pos[b] % 64 (the pool row the step writes), or nullptr
    const uint32_t *pt_rows; uint32_t *kvrow; uint32_t pt_bstride, pt_entries;   // pt_entries: table entries per slot (a position beyond them stages nothing)
    // the step's first kernel also advances the model's hand-off epoch (device_common.h: tick[0] += 1, workgroup 0), or nullptr
    uint32_t *tick;
};
hipError_t launch_embed(const EmbedArgs &a, uint32_t nb, hipStream_t st);

// argmax over logits[b][V] -> out[b]; optionally advances the decode loop state:
// tokens[b] = argmax, pos[b] += 1, trace[step*nb + b] = argmax
struct ArgmaxArgs {
    const float *logits; uint32_t V, bstride;
    uint32_t *out;
    uint32_t *tokens; uint32_t *pos; uint32_t *trace; const uint32_t *pos0; uint32_t nb;   // trace[(pos-pos0)*nb + b]
    const float *tile_max; uint32_t ntiles;     // optional (max, row) partials from the classifier GEMV
    // greedy loop only (tokens != nullptr): the SAME kernel then embeds the token it picked at its next position -- the next
    // step's first kernel (embed) and its launch boundary are gone.  emb.x == nullptr: not fused.  rope_rows: rows of the RoPE
    // tables (the position after the last one has no row: nothing is staged for it)
    EmbedArgs emb; uint32_t rope_rows, _pade;
};
hipError_t launch_argmax(const ArgmaxArgs &a, uint32_t nb, hipStream_t st);

hipError_t launch_rmsnorm(float *out, const float *x, const float *w, uint32_t n, hipStream_t st);
hipError_t launch_quantize_q80(const float *x, uint32_t n, uint32_t gs, int8_t *q, float *s, hipStream_t st);
hipError_t launch_quantize_q4k(const float *x, uint32_t n, uint8_t *blocks, hipStream_t st);
hipError_t launch_swiglu(float *hb, const float *hb2, uint32_t n, hipStream_t st);
hipError_t launch_rope(float *head, uint32_t hd, const float *fcr, const float *fci, int qwen3, hipStream_t st);
hipError_t launch_stream_read(const void *buf, size_t bytes, float *sink, hipStream_t st);
hipError_t launch_stream_read_masked(const void *buf, size_t bytes, float *sink, uint32_t xcd_mask, uint32_t wgs, hipStream_t st);

// ---- device-side sampler (sampler.hip) ----
constexpr uint32_t SAMPLE_CHUNK = 256;            // softmax numerators per chunk function (one wave x float4)
constexpr uint32_t SAMPLE_MAX_CHUNKS = 1024;      // vocabularies up to 262144
constexpr uint32_t SAMPLE_MAX_CANDIDATES = NANO_SAMPLE_MAX_CANDIDATES;
constexpr uint32_t SAMPLE_BINS = 256;             // histogram of the softmax numerators: 8 bins per binade, 2^0 .. 2^-31
struct SampleArgs {
    const float *logits; uint32_t V;
    uint32_t nch;                                 // chunks, rounded up to a multiple of 4; y/e hold nch*256 floats
    float *y, *e;                                 // penalised+tempered logits, softmax numerators
    const uint8_t *seen;                          // null when the penalty is 1
    float penalty, temperature, top_p, cutoff, coin;
    float *pmax;                                  // [nch/4] workgroup maxima of the prep kernel
    uint32_t *ncand, *ndrop, *dropmax, *bstar;    // accumulators, left at 0 by the last kernel (bstar: set by propagate)
    uint32_t *bin_cnt; unsigned long long *bin_mass;      // [SAMPLE_BINS], likewise
    float *approx; uint32_t *spec; uint2 *fn;     // per chunk: approximate sum, guessed exponent field, chunk function
    float *sum;
    unsigned long long *cand; uint32_t cap;
    NanoHipSample *res;
    // second phase, wide nuclei (sampler_wide.hip): every candidate as a key, the sorted keys, the sorted probabilities
    unsigned long long *wide_in, *wide_out; float *wide_p; uint32_t wide_cap;
};
hipError_t launch_seen_set(const uint32_t *ids, uint32_t n, uint8_t *seen, hipStream_t st);
hipError_t launch_sample_prep(const SampleArgs &a, hipStream_t st);   // penalty (and temperature) only
hipError_t launch_sample(const SampleArgs &a, hipStream_t st);
size_t sample_wide_temp_bytes(uint32_t n);                            // scratch of the device sort of n keys
hipError_t launch_sample_wide(const SampleArgs &a, void *temp, size_t temp_bytes, hipStream_t st);
hipError_t launch_sample_wide_cut(const SampleArgs &a, hipStream_t st);   // (its last three kernels: sampler.hip)   // after launch_sample reported NANO_SAMPLE_FALLBACK

}  // namespace nano
