// attn.hip -- single-token GQA attention over the FP32 KV cache (reference infer/infer.c:810-879).
//
// One workgroup per (KV head group, sequence, split).  Like the GEMVs of a batch-1 decode step this kernel is
// LATENCY bound (a KV head's rows are at most a few hundred KB), so it is built as ONE memory round trip:
// at entry every thread issues the loads of q, the raw k row, the norm weights and ALL the K and V rows
// its workgroup will ever need (buffer descriptors: rows beyond the cache are out of range and read as 0
// without touching memory; rows beyond `range_hint`, the host's upper bound of pos+1 for this launch, are
// skipped the same way); position-dependent work (RoPE row, causal mask, the fresh k row) is applied
// afterwards, when pos[b] has arrived.
//   * the q heads of a KV group share every K/V row they read (GQA: kv_mul heads per workgroup);
//   * Qwen3 q/k rmsnorm (infer.c:824-835) and RoPE (rope_qwen3 infer.c:692-706 / rope infer.c:681-690) are
//     head-local and done here by one wave per vector (DPP reductions, no barrier);
//   * the finished k row is written to cache row `pos` by split 0; every workgroup uses its own LDS copy
//     for timestep pos, so nobody reads the row while it is being written; v went to the cache directly
//     from the QKV GEMV;
//   * timestep blocks are dealt round-robin to `nsplit` workgroups ("flash-decoding"); with nsplit == 1 the
//     kernel normalises and writes the final head outputs, otherwise UNNORMALISED partials
//     o_s = sum_t exp(s_t - m_s) v_t with (m_s, l_s) which the Wo GEMV's prologue combines (gemv_q80.hip).
// Softmax is algebraically the reference's (max-subtracted, infer.c:616-634); summation order differs and the
// q.k / weighted-V accumulations use fused multiply-adds (tolerance 1e-5).
#include "attn_impl.h"

namespace nano {

namespace {

template <int LPR, int QV, int MODE, bool KVH>
static hipError_t launch_mode_kv(const AttnArgs &a_in, uint32_t nb, hipStream_t st) {
    const uint32_t kv_mul = a_in.n_head / a_in.n_kv_head;
    const uint32_t hd4 = (a_in.hd + 3) & ~3u;
    constexpr uint32_t R = 256 / LPR;                          // timesteps per block
    // (+ several heads per workgroup: the waves' transposition blocks of the sub-group combine, 4 waves x 64 / LPR sub-groups x LPR * QV * 4 floats)
    auto lds_for = [&](uint32_t kvm) { return (size_t)(kvm * hd4 + hd4 + 4 * kvm + 4 * kvm + (size_t)4 * kvm * hd4 + (kvm > 1 ? (size_t)4 * (64 / LPR) * (LPR * QV * 4) : 0)) * sizeof(float); };
    // q heads per workgroup (KVM; h0 = KVM grp, KV head h0 / kv_mul): fewer heads = more workgroups with less dependent work each,
    // the K/V rows' repeated reads come from L2 (and the KV head's fresh k row is written by each of its workgroups: same
    // bits).  Same per-head arithmetic whatever the choice.  One head per workgroup while that leaves at most one workgroup
    // per CU, two while at most two, else the whole KV group (measured: Qwen3-0.6B batch 1 6.15 -> 5.35 us per launch = +3.7 %
    // tokens/s; Qwen3-4B's kv_mul 4: 8.65 -> 6.03 us at batch 1, 10.05 -> 7.22 with two heads at 16 sequences, where one head
    // per workgroup costs 10.3; at 64 sequences the KV rows' bandwidth rules and four heads share them: 13.7 vs 16.4).
    constexpr bool xcd_order = true;
    AttnArgs a = a_in;
    a.kv_log2 = 0xffffffffu; a.kvmul_log2 = 0;
    const bool kv_pow2 = (a.n_kv_head & (a.n_kv_head - 1)) == 0, mul_pow2 = (kv_mul & (kv_mul - 1)) == 0;
    if (kv_pow2 && (MODE != 0 || (xcd_order && a.n_kv_head >= 8))) { uint32_t l2 = 0; while ((1u << l2) < a.n_kv_head) l2++; a.kv_log2 = l2; }
    if (mul_pow2) { uint32_t l2 = 0; while ((1u << l2) < kv_mul) l2++; a.kvmul_log2 = l2; }
    if (MODE != 0 && !(kv_pow2 && mul_pow2)) return hipErrorInvalidValue;       // launch_lpr() sends such shapes to the generic mode
    const uint64_t head_wgs = (uint64_t)a.n_head * nb * a.nsplit;
    uint32_t kvm = (head_wgs <= 256u || kv_mul % 2 != 0) ? 1u : (head_wgs <= 1024u || kv_mul % 4 != 0) ? 2u : 4u;
    // A workgroup that would walk exactly two rounds of NP blocks (513 .. 1024 positions at 8 splits) keeps four blocks in flight
    // instead: every row of the launch requested at kernel entry (Qwen3-0.6B at position 1023: 639 -> 632 us per step).  Measured
    // and NOT taken beyond: four rounds -> two at 2047 positions 754 -> 759 us, two -> one at 4095 with 32 splits 819 -> 836 (the
    // issue phase grows from 3.1 to 5.9 us: `profiles/r04_long_ctx_np4.txt`).  Decided by the range and the split count alone --
    // never by the batch: a token's attention is the same expression in a decode step and in a prefill chunk (the four-head
    // workgroups of large batches give way to two heads).
    const bool np4 = !a.prep_only && a.nsplit <= 8u && a.range_hint > a.nsplit * (uint32_t)NP * R && a.range_hint <= 2u * a.nsplit * (uint32_t)NP * R;
    if (np4 && kvm == 4) kvm = 2;
    constexpr bool CAN16 = KVH && QV % 4 == 0;
    const bool w16 = CAN16 && a.hd % 8u == 0u;
#define ATTN_GO3(KVM_, NP_, PG_) do { if constexpr (CAN16) { if (w16) { hipLaunchKernelGGL((attention_kernel<LPR, QV, KVM_, MODE, KVH, PG_, NP_, CAN16>), dim3(a.n_head / KVM_, nb, a.nsplit), dim3(256), lds_for(KVM_), st, a); break; } } \
                         hipLaunchKernelGGL((attention_kernel<LPR, QV, KVM_, MODE, KVH, PG_, NP_, false>), dim3(a.n_head / KVM_, nb, a.nsplit), dim3(256), lds_for(KVM_), st, a); } while (0)
#define ATTN_GO2(KVM_, NP_) do { if (a.pt_rows) ATTN_GO3(KVM_, NP_, true); else ATTN_GO3(KVM_, NP_, false); } while (0)
#define ATTN_GO(KVM_) do { if (np4) ATTN_GO2(KVM_, 4); else ATTN_GO2(KVM_, NP); } while (0)
    if (kvm == 4) ATTN_GO2(4, NP); else if (kvm == 2) ATTN_GO(2); else ATTN_GO(1);
#undef ATTN_GO
#undef ATTN_GO2
#undef ATTN_GO3
    return hipGetLastError();
}
template <int LPR, int QV, int MODE>
static hipError_t launch_mode(const AttnArgs &a, uint32_t nb, hipStream_t st) {
    return a.kv_half ? launch_mode_kv<LPR, QV, MODE, true>(a, nb, st) : launch_mode_kv<LPR, QV, MODE, false>(a, nb, st);
}
template <int LPR, int QV>
static hipError_t launch_lpr(const AttnArgs &a, uint32_t nb, hipStream_t st) {
    // the decode modes index with shifts and masks: power-of-two head counts (every BASELINE shape: 16 / 8 and 32 / 8 heads)
    const uint32_t kv_mul = a.n_kv_head ? a.n_head / a.n_kv_head : 0;
    const bool pow2 = a.n_kv_head && (a.n_kv_head & (a.n_kv_head - 1)) == 0 && kv_mul && (kv_mul & (kv_mul - 1)) == 0;
    const bool decode = pow2 && a.kraw && a.rope_cos && a.rope_cur && !a.fixed_range && a.is_causal && !a.q_out;
    if (decode && a.q_norm && a.rope_qwen3 && a.hd % 64 == 0 && a.hd == (uint32_t)(QV * LPR * 4)) return launch_mode<LPR, QV, 1>(a, nb, st);
    if (decode && !a.q_norm && !a.rope_qwen3) return launch_mode<LPR, QV, 2>(a, nb, st);
    return launch_mode<LPR, QV, 0>(a, nb, st);
}

}  // namespace

// timesteps one workgroup covers per round for this head size
static uint32_t steps_per_wg(uint32_t hd) { return NP * (256 / (hd > 128 ? 16 : 8)); }

// number of splits for an upper bound `range_hint` of the attended range: <= 8 up to attention_wide_from() positions (what the Wo
// GEMV's prologue combines); beyond, up to attention_split_cap() so that a long range still spreads over the chip (8 KV groups
// x 32 splits), combined by attn_combine_tokens_kernel -- a launch of its own, which pays from about four rounds per
// workgroup on (Qwen3-0.6B, tools/long_ctx_probe.py).
uint32_t attention_wide_from() {
    return 2048u;
}
uint32_t attention_split_cap() {
    return 32u;
}
uint32_t attention_nsplit(uint32_t range_hint, uint32_t hd) {
    const uint32_t per = steps_per_wg(hd), cap = range_hint > attention_wide_from() ? attention_split_cap() : 8u;
    uint32_t n = (range_hint + per - 1) / per;
    if (n < 1) n = 1;
    if (n > cap) n = cap;
    return n;
}

hipError_t launch_attention(const AttnArgs &a, uint32_t nb, hipStream_t st) {
    if (a.hd % 4 || a.hd > 256 || a.hd < 4 || a.nsplit == 0 || a.nsplit > ATTN_MAX_NSPLIT) return hipErrorInvalidValue;
    if (a.nsplit == 1 && !a.xba_out) return hipErrorInvalidValue;
    if (a.hd <= 32) return launch_lpr<8, 1>(a, nb, st);
    if (a.hd <= 64) return launch_lpr<8, 2>(a, nb, st);
    if (a.hd <= 128) return launch_lpr<8, 4>(a, nb, st);
    return launch_lpr<16, 4>(a, nb, st);
}

// stand-alone combine (operator tests, state read-back; the forward folds this into the Wo GEMV's prologue)
__global__ void attn_combine_kernel(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit) {
    const uint32_t h = blockIdx.x, q_dim = n_head * hd;
    const float *mlh = ml + (size_t)h * nsplit * 2;
    float M = -INFINITY;
    for (uint32_t s = 0; s < nsplit; s++) if (mlh[2 * s + 1] > 0.0f) M = fmaxf(M, mlh[2 * s]);
    float L = 0.0f;
    for (uint32_t s = 0; s < nsplit; s++) L += mlh[2 * s + 1] * ((mlh[2 * s + 1] > 0.0f) ? expf(mlh[2 * s] - M) : 0.0f);
    for (uint32_t i = threadIdx.x; i < hd; i += blockDim.x) {
        float acc = 0.0f;
        for (uint32_t s = 0; s < nsplit; s++) {
            const float e = (mlh[2 * s + 1] > 0.0f) ? expf(mlh[2 * s] - M) : 0.0f;
            acc += part[(size_t)s * q_dim + (size_t)h * hd + i] * (e / L);
        }
        out[(size_t)h * hd + i] = acc;
    }
}
hipError_t launch_attn_combine(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit, hipStream_t st) {
    hipLaunchKernelGGL(attn_combine_kernel, dim3(n_head), dim3(128), 0, st, part, ml, out, n_head, hd, nsplit);
    return hipGetLastError();
}

// The combine of the Wo GEMV's prologue (gemv_common.h combine_weights / combine4) as a kernel of its own, for steps
// whose Wo cannot fold it in (batched prefill through the MFMA GEMM, ranges split more than 8 ways): same weights (8 slots,
// pairwise-tree sum of l_s * exp(m_s - M), e / L), same ascending accumulation -- the bits of x are those of the decode path; more
// than 8 splits: the 8-slot tree per block of 8, blocks added in order (combine_weights_lds above).  A thread owns ONE output
// element and asks for all its partials before the weights exist: one memory round trip.
template <int NS>
__global__ __launch_bounds__(128) void attn_combine_tokens_kernel(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit,
                                                                  int8_t *xf_out, float *xsf_out) {
    __shared__ float wsh[64];
    const uint32_t b = blockIdx.y, q_dim = n_head * hd;
    for (uint32_t i0 = 0; i0 < hd; i0 += 128u) {                 // (head_dim 256: two passes)
        const uint32_t hh = blockIdx.x, i = i0 + threadIdx.x;
        const bool live = i < hd;
        const float *pb = part + (size_t)b * nsplit * q_dim + (size_t)hh * hd + i;
        float o[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) o[s] = (live && (uint32_t)s < nsplit) ? pb[(size_t)s * q_dim] : 0.0f;
        if (i0 == 0) combine_weights_lds(ml + (((size_t)b * n_head + hh) * nsplit) * 2, nsplit, wsh);
        float acc = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; s++) if ((uint32_t)s < nsplit) acc += o[s] * wsh[s];
        if (nsplit < (uint32_t)NS) acc += 0.0f;                  // (the empty slots' +0 of the register version: -0 becomes +0 as it always did)
        if (live) {
            out[(size_t)b * q_dim + (size_t)hh * hd + i] = acc;
            if (xf_out) {       // also as a Q80 group of 64 (= this wave's 64 lanes; head_dim % 64 == 0) in fragment order, see attention_kernel
                float mx = fabsf(acc);
#pragma unroll
                for (int o2 = 1; o2 < 64; o2 <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
                const float scale = div_const<127>(mx);
                const uint32_t el = hh * hd + i, g = el >> 6, jj = el & 63u, ng = q_dim >> 6;
                const size_t gb = (size_t)(b >> 4) * ng + g;
                xf_out[gb * 1024u + (size_t)((jj >> 4) * 16u + (b & 15u)) * 16u + (jj & 15u)] = (int8_t)q80_quant1(acc, scale);
                if (jj == 0) xsf_out[gb * 16u + (b & 15u)] = scale;
            }
        }
    }
}
hipError_t launch_attn_combine_tokens(const float *part, const float *ml, float *out, uint32_t n_head, uint32_t hd, uint32_t nsplit, uint32_t nb,
                                      int8_t *xf_out, float *xsf_out, hipStream_t st) {
    if (xf_out && (hd % 64u || !xsf_out)) return hipErrorInvalidValue;
    if (nsplit == 0 || nsplit > ATTN_MAX_NSPLIT) return hipErrorInvalidValue;
    if (nsplit <= 8) hipLaunchKernelGGL(attn_combine_tokens_kernel<8>, dim3(n_head, nb), dim3(128), 0, st, part, ml, out, n_head, hd, nsplit, xf_out, xsf_out);
    else if (nsplit <= 32) hipLaunchKernelGGL(attn_combine_tokens_kernel<32>, dim3(n_head, nb), dim3(128), 0, st, part, ml, out, n_head, hd, nsplit, xf_out, xsf_out);
    else hipLaunchKernelGGL(attn_combine_tokens_kernel<64>, dim3(n_head, nb), dim3(128), 0, st, part, ml, out, n_head, hd, nsplit, xf_out, xsf_out);
    return hipGetLastError();
}

}  // namespace nano
