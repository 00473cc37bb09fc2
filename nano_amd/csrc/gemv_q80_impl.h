// gemv_q80_impl.h -- body of the Q80 GEMV kernels; compiled once per group size (gemv_q80_gs*.hip define
// NANO_Q80_GS and NANO_Q80_ENTRY and include this file) so that the instantiations build in parallel.
// gemv_q80.hip -- Q80 (W8A8) fused decode GEMVs for gfx950 (MI355X), up to 8 sequences per launch.
//
// Restates for the device:
//   * quantize      (reference infer/tensor.c:21-46)   per group: s = max|x|/127, q = (int8)round(x/s)
//   * matmul_quant  (reference infer/infer.c:654-679)  per row, groups in ascending order:
//                                                      val += ((float)sum_i8xi8) * ws[g] * xs[g]
// with rmsnorm (infer.c:601-614), the split-attention combine (attn.hip), the residual adds
// (infer.c:906-908,963-965) and SwiGLU (infer.c:937-944) fused in as prologue / epilogue.
// Given identical int8 inputs the fp32 outputs are BIT-IDENTICAL to the reference: the integer group
// sums are exact, every group product is formed as ((float)ival * ws) * xs and one lane adds the
// groups of a row in ascending order (no tree, no FMA contraction).
//
// A batch-1 decode step is a chain of ~140 dependent kernels whose weights (2..6 MB each, except the
// classifier) stream in well under a microsecond: every per-layer kernel is LATENCY bound, not
// bandwidth bound.  Two kernels, built for the two regimes:
//
//   SLAB   (per-layer matrices).  A workgroup owns `rw` consecutive output rows x the whole row length.
//          Its work units (4 rows x one 1 KiB column chunk, x2 matrices for SwiGLU) are dealt to its
//          waves; every wave issues ALL of its loads right at kernel entry -- first the activation, then
//          weights and weight scales through buffer descriptors (32-bit offsets, hardware bounds check,
//          no exec-masked branches, no dependent scalar loads) -- so the whole kernel costs ONE memory
//          round trip.  rmsnorm + quantization run from registers while the weights are in flight; the
//          group products land in an LDS table and one thread per (row, sequence) folds them in order.
//   STREAM (classifier: vocab x n_embd).  1024 persistent workgroups; a wave owns 16-row tiles,
//          tile = wave + k * nwaves, so the chip sweeps memory linearly; four waves per SIMD keep 16 KiB each in
//          flight while one of them is consuming.  Lane l loads bytes [16l,16l+16) of each row chunk
//          (non-temporal, fully coalesced; the row-major int8 blocks stay exactly as in the model file);
//          DPP integer group sums; the leaders park them in a wave-private LDS table; then lane l owns
//          (row l/4, groups 4(l%4)..+3): the scales of a whole tile arrive as ONE coalesced load, all 64
//          lanes form products, and a 4-stage quad chain keeps the reference's group order.
//          Measured 5.6-5.7 TB/s (cold), 95 % of what a plain read of the same bytes achieves.
//
// MFMA is not used: 2 flop/byte at batch <= 8, no tile forms.
#include <string.h>
#include <stdio.h>
#include "gemv_common.h"
#include "gemv_q80_host.h"

namespace nano {

namespace {


__device__ __forceinline__ void quant_store4(float4 v, float scale, int8_t *dst) {
    const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
    *reinterpret_cast<uint32_t *>(dst) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
}

template <int ROLE, int GS, int B, int NV>
__device__ __forceinline__ void stage_finish(const GemvDev &a, Staged<B, NV> &r, int8_t *xq, float *xs, float *red, uint32_t n16, uint32_t ng4) {
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n, ng = a.ng;
    const uint32_t lane = tid & 63u, wid = tid >> 6, NW = nthr >> 6;
    if (has_flag<ROLE>(a, F_PRE)) { // the activations arrive quantized: xq_in[nb][n16], xs_in[nb][ng] (operator tests; large
                                    // batch x row-length products, where quant_rows_kernel quantizes ONCE instead of once per workgroup)
        for (uint32_t b = 0; b < a.nb; b++) {
            for (uint32_t i = tid * 16u; i < n; i += nthr * 16u) *reinterpret_cast<int4 *>(xq + b * n16 + i) = *reinterpret_cast<const int4 *>(a.xq_in + (size_t)b * n16 + i);
            for (uint32_t i = tid; i < ng; i += nthr) xs[b * ng4 + i] = a.xs_in[(size_t)b * ng + i];
        }
        __syncthreads();
        return;
    }
    const bool norm = has_flag<ROLE>(a, F_NORM), comb = has_flag<ROLE>(a, F_COMBINE);
    float *wgt = red + B * 16;
    if constexpr (NV == 0) {
        if (comb) combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
        // generic path (large n x B): two passes over memory per sequence
        for (uint32_t b = 0; b < a.nb; b++) {
            const float *x = a.xin + (size_t)b * a.xin_bstride;
            float ss = 1.0f;
            if (norm) {
                float acc = 0.0f;
                for (uint32_t i = tid * 4u; i < n; i += nthr * 4u) {
                    const float4 v = comb ? combine4(a, b, i, wgt) : *reinterpret_cast<const float4 *>(x + i);
                    acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
                }
                acc = dpp_wave_sum(acc);
                __syncthreads();
                if (lane == 0) red[wid] = acc;
                __syncthreads();
                float t = 0.0f;
                for (uint32_t w = 0; w < NW; w++) t += red[w];
                t /= (float)n; t += 1e-5f;
                ss = 1.0f / sqrtf(t);
            }
            for (uint32_t i = tid * 4u; i < n; i += nthr * 4u) {       // n % GS == 0, 4*nthr % GS == 0: groups are whole
                float4 v = comb ? combine4(a, b, i, wgt) : *reinterpret_cast<const float4 *>(x + i);
                if (norm) {
                    const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                    v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
                }
                float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                m = dpp_group_max<GS / 4>(m);
                const float scale = div_const<127>(m);
                quant_store4(v, scale, xq + b * n16 + i);
                if ((tid % (GS / 4)) == 0) xs[b * ng4 + i / GS] = scale;
            }
        }
        __syncthreads();
    } else {
        if (comb) {
            if constexpr (B == 1) {
                const bool pre_ml = a.attn_n_head * 8u <= nthr;       // every (head, split) pair has its own thread
                if (pre_ml) combine_weights<B, true>(a, wgt, r.ml_m, r.ml_l); else combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
                    const float *wg = wgt + (size_t)((i < n ? i : 0u) / a.attn_hd) * 8u;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int sp = 0; sp < 8; sp++) {          // splits >= nsplit: partial read as 0, weight 0
                        const float w = wg[sp];
                        acc.x += r.pv[j][sp].x * w; acc.y += r.pv[j][sp].y * w; acc.z += r.pv[j][sp].z * w; acc.w += r.pv[j][sp].w * w;
                    }
                    r.x[0][j] = acc;
                }
            } else {
                combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
                for (int b = 0; b < B; b++)
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
                        r.x[b][j] = (i < n && b < (int)a.nb) ? combine4(a, b, i, wgt) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
        }
        float ss[B];
#pragma unroll
        for (int b = 0; b < B; b++) ss[b] = 1.0f;
        if (norm) {                     // rmsnorm scale (infer.c:603-609); tree order, tolerance 1e-5
#pragma unroll
            for (int b = 0; b < B; b++) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    acc += r.x[b][j].x * r.x[b][j].x; acc += r.x[b][j].y * r.x[b][j].y;
                    acc += r.x[b][j].z * r.x[b][j].z; acc += r.x[b][j].w * r.x[b][j].w;
                }
                acc = dpp_wave_sum(acc);
                if (lane == 0) red[b * 16 + wid] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < B; b++) {
                float t = 0.0f;
                for (uint32_t w = 0; w < NW; w++) t += red[b * 16 + w];
                t /= (float)n; t += 1e-5f;
                ss[b] = 1.0f / sqrtf(t);
            }
        }
        NANO_STAMP(a.stamps, 2, ss[0] + r.x[0][0].x);                // the activation arrived (and its norm scale is known)
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
#pragma unroll
            for (int b = 0; b < B; b++) {
                float4 v = r.x[b][j];
                if (norm) {
                    v.x = r.nw[j].x * (ss[b] * v.x); v.y = r.nw[j].y * (ss[b] * v.y);
                    v.z = r.nw[j].z * (ss[b] * v.z); v.w = r.nw[j].w * (ss[b] * v.w);
                }
                float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
                m = dpp_group_max<GS / 4>(m);
                const float scale = div_const<127>(m);
                if (i < n) {
                    quant_store4(v, scale, xq + b * n16 + i);
                    if ((tid % (GS / 4)) == 0) xs[b * ng4 + i / GS] = scale;
                }
            }
        }
        __syncthreads();
    }
}


// ordered fold of one row's group products (infer.c:668-674): NQ float4 = 4 NQ groups, all LDS reads first, then the chain
template <int NQ> __device__ __forceinline__ float fold_row(const float *p) {
    float4 t[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) t[q] = *reinterpret_cast<const float4 *>(p + 4 * q);
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; q++) { v += t[q].x; v += t[q].y; v += t[q].z; v += t[q].w; }
    return v;
}
template <int NQ> __device__ __forceinline__ void fold_row2(const float *p0, const float *p1, float &v0, float &v1) {
    float4 t[NQ], u[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { t[q] = *reinterpret_cast<const float4 *>(p0 + 4 * q); u[q] = *reinterpret_cast<const float4 *>(p1 + 4 * q); }
    v0 = 0.0f; v1 = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; q++) { v0 += t[q].x; v1 += u[q].x; v0 += t[q].y; v1 += u[q].y; v0 += t[q].z; v1 += u[q].z; v0 += t[q].w; v1 += u[q].w; }
}

// CANONICAL fold of the fast path (kernels.h q80_canonical(); group size 64): unit sums S_u = the 8 group products of unit u added in
// ascending order, row = ((S_0 + S_1) + S_2) + ... -- the one reduction shape the split-K kernels (gemm_q80_g6.hip, G5) share, so
// that a batch stays bit for bit its sequences alone whichever kernel a batch size is routed to.  NQ float4 = 4 NQ groups.
__device__ __forceinline__ float unit_sum(const float4 &t0, const float4 &t1) {
    float s = t0.x; s += t0.y; s += t0.z; s += t0.w; s += t1.x; s += t1.y; s += t1.z; s += t1.w;
    return s;
}
// One fp32 add the optimizer cannot pair: the SLP vectorizer turned two neighbouring unit sums into v_pk_add_f32 behind a shuffle of
// v_mov and let the scheduler sink the LDS reads next to them (three LDS round trips per row).  Same instruction the compiler emits for
// a + b, same rounding; not volatile, so it schedules freely.
__device__ __forceinline__ float fadd(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// the NU = NQ / 2 unit sums of a row in lock step: NU independent add chains of depth 7 (a wave issues them back to back), then the
// NU - 1 adds of the units -- depth 7 + NU - 1 where the ordered chain has 8 NU - 1
template <int NU> __device__ __forceinline__ void unit_sums(const float4 *t, float *s) {
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(t[2 * u].x, t[2 * u].y);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u].z);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u].w);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u + 1].x);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u + 1].y);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u + 1].z);
#pragma unroll
    for (int u = 0; u < NU; u++) s[u] = fadd(s[u], t[2 * u + 1].w);
}
template <int NQ> __device__ __forceinline__ float fold_row_canon(const float *p) {
    static_assert(NQ % 2 == 0, "whole units");
    float4 t[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) t[q] = *reinterpret_cast<const float4 *>(p + 4 * q);
    __builtin_amdgcn_sched_barrier(0);          // every LDS read of the row goes out before the first add (one round trip, not one per unit pair)
    float s[NQ / 2];
    unit_sums<NQ / 2>(t, s);
    float v = s[0];
#pragma unroll
    for (int u = 1; u < NQ / 2; u++) v = fadd(v, s[u]);
    return v;
}
// any group count that is a multiple of 4 (the last unit may hold 4 groups); two units per trip, their reads first
__device__ __forceinline__ float fold_row_canon_any(const float *p, uint32_t ng) {
    float v = 0.0f;
    for (uint32_t g0 = 0; g0 < ng; g0 += 16) {
        float4 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) t[q] = (g0 + 4u * (uint32_t)q < ng) ? *reinterpret_cast<const float4 *>(p + g0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = t[0].x; s += t[0].y; s += t[0].z; s += t[0].w;
        if (g0 + 4 < ng) { s += t[1].x; s += t[1].y; s += t[1].z; s += t[1].w; }
        v = g0 == 0 ? s : v + s;
        if (g0 + 8 < ng) {
            s = t[2].x; s += t[2].y; s += t[2].z; s += t[2].w;
            if (g0 + 12 < ng) { s += t[3].x; s += t[3].y; s += t[3].z; s += t[3].w; }
            v += s;
        }
    }
    return v;
}
// two rows (W1's and W3's of a SwiGLU launch): every LDS read of both goes out before the first add
template <int NQ> __device__ __forceinline__ void fold_row2_canon(const float *p0, const float *p1, float &v0, float &v1) {
    static_assert(NQ % 2 == 0, "whole units");
    float4 t[NQ], u[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) { t[q] = *reinterpret_cast<const float4 *>(p0 + 4 * q); u[q] = *reinterpret_cast<const float4 *>(p1 + 4 * q); }
    __builtin_amdgcn_sched_barrier(0);
    float s0[NQ / 2], s1[NQ / 2];
    unit_sums<NQ / 2>(t, s0); unit_sums<NQ / 2>(u, s1);
    v0 = s0[0]; v1 = s1[0];
#pragma unroll
    for (int k = 1; k < NQ / 2; k++) { v0 = fadd(v0, s0[k]); v1 = fadd(v1, s1[k]); }
}
__device__ __forceinline__ void fold_row2_canon_ng(const float *p0, const float *p1, uint32_t ng, float &v0, float &v1) {
    if (ng == 16u) { fold_row2_canon<4>(p0, p1, v0, v1); return; }
    v0 = fold_row_canon_any(p0, ng); v1 = fold_row_canon_any(p1, ng);
}
__device__ __forceinline__ float fold_row_canon_ng(const float *p, uint32_t ng) {
    if (ng == 16u) return fold_row_canon<4>(p);
    if (ng == 32u) return fold_row_canon<8>(p);
    if (ng == 48u) return fold_row_canon<12>(p);
    return fold_row_canon_any(p, ng);
}

// ------------------------------------------------------------------------------------------------------------
// SLAB kernel
// ------------------------------------------------------------------------------------------------------------
// Hand-off of a launch's results to consumers INSIDE the same launch (the fused kernels below): every result is also stored as an 8-byte
// {tag, value} granule (ONE write-through store: the data is the flag); SlabHand and the epoch tags: device_common.h.
template <int ROLE, int GS, int B, int NV, int UPW, int EARLY = 0>
__global__ __launch_bounds__(1024) void gemv_q80_slab_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define SLAB_EARLY EARLY
#define SLAB_A a
#define SLAB_BID blockIdx.x
#define SLAB_HAND 0
#define SLAB_HANDV (SlabHand{})
#define SLAB_PTAG 0u
#define SLAB_XHAND 0
#define SLAB_XHANDV (SlabHand{})
#define SLAB_CTAG 0u
#define SLAB_PART 0
#include "gemv_q80_slab_body.inc"
#undef SLAB_A
#undef SLAB_BID
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
#undef SLAB_PART
#undef SLAB_EARLY
}

#if NANO_Q80_GS == 64
}  // namespace
}  // namespace nano
#include "attn_impl.h"
namespace nano {
namespace {
// ---- q | k | v projection + attention in ONE launch (Qwen3 decode, one sequence, head_dim 128; round 5) --------------------------------------
// The LAST `n_attn` workgroups are the attention's (head x split): they ask for their K / V rows at entry, exactly as the attention kernel
// does, and then wait for q, the raw k row and the fresh v row of their KV group as granules; the first `ngemv` workgroups are the projection's
// SLAB GEMV (role: rmsnorm + quantize + store), whose fold threads also store every result as a granule.  What the boundary between the two
// kernels cost -- the gap, the attention's entry ramp and its K / V round trip -- now overlaps the projection.  Bits: the same two bodies.
// Round 6: the PRODUCERS come first in the grid (round-5 advice: with the consumers in front, a chip that other work keeps busy could seat
// the pollers and leave the projection waiting for their slots); granules carry epoch tags (device_common.h), a consumer that gives up
// skips its stores.  Reference: infer/infer.c:758-879.
struct FusedArgs { GemvDev g; AttnArgs a; SlabHand hand; uint32_t n_attn, head_wgs, wait16, ngemv; };
template <int NV, int UPW>
__global__ __launch_bounds__(256) void qkv_attn_fused_kernel(const FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);                       // the step's epoch: the first load of every workgroup
    if (blockIdx.x >= fa.ngemv) {
        const uint32_t ab = blockIdx.x - fa.ngemv;
        const uint32_t split = ab / fa.head_wgs, grp = ab - split * fa.head_wgs;
        attention_body<8, 4, 1, 1, false, false, 2, false, true>(fa.a, smem, grp, 0u, split, fa.hand, hand_ctag(tk_, fa.hand), fa.wait16);
        return;
    }
    constexpr int ROLE = R_NORM_STORE, GS = 64, B = 1;
#define SLAB_A fa.g
#define SLAB_BID blockIdx.x
#define SLAB_HAND 1
#define SLAB_HANDV fa.hand
#define SLAB_PTAG hand_ptag(tk_, fa.hand)
#define SLAB_XHAND 0
#define SLAB_XHANDV (SlabHand{})
#define SLAB_CTAG 0u
#define SLAB_PART 0
#include "gemv_q80_slab_body.inc"
#undef SLAB_A
#undef SLAB_BID
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
#undef SLAB_PART
}

// ---- Wo + W1|W3 in ONE launch (round 5): the first all-to-all edge of the block as an in-launch all-gather -------------------------------
// Wo's epilogue produces the residual stream x; W1|W3 needs ALL of x (rmsnorm, then every row times the whole vector): an all-to-all edge,
// a kernel boundary in every earlier build.  Here the launch has W1|W3's grid; its first `wo_wgs` workgroups run Wo's body first (results
// stored as usual AND as 8-byte {tag, value} granules), then EVERY workgroup runs W1|W3's body with the activation polled from the granules
// (gemv_q80_slab_body.inc SLAB_XHAND).  W1|W3's weight and norm-weight loads go out at kernel entry (SLAB_PART 1), BEFORE Wo's body: they are
// in LDS-distance when the granules arrive, and -- younger than Wo's own loads -- do not hold up Wo's wait counts.  What the boundary cost
// (the gap, the entry ramp, the activation's round trip behind a cold launch) is traded for the polls.  Deadlock: producers are the first
// workgroups of the grid and never wait for consumers; a grid of <= one workgroup per CU is resident as a whole.  Same bodies, same bits
// (test_fused_wo_w13_launch_equals_the_two_launches).  Reference: infer/infer.c:885-944.
struct Wo13Args { GemvDev wo; GemvDev w13; SlabHand hand; uint32_t wo_wgs, wait16; };   // wait16: naps of 16 x 64 cycles before a non-producer workgroup starts polling
template <int ROLE_A, int NV_A, int UPW_A, int NV_B, int UPW_B, int NT>
__global__ __launch_bounds__(NT) void wo_w13_fused_kernel(const Wo13Args fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int GS = 64, B = 1;
    const uint2 tk_ = hand_tick(fa.hand);
    // Order of issue: Wo's loads, W1|W3's loads, Wo's arithmetic, W1|W3's.  (vmcnt counts in order: loads issued BEFORE Wo's would have to
    // land before Wo may use its own -- the first build had W1|W3's in front and Wo waited for the whole 64 MB of Qwen3-4B's two matrices.)
    // Workgroups beyond Wo's grid run Wo's part 1 on rows past the matrix (out-of-range buffer loads: no traffic) and skip its part 2.
    {
        constexpr int ROLE = ROLE_A, NV = NV_A, UPW = UPW_A;
#define SLAB_BID blockIdx.x
#define SLAB_A fa.wo
#define SLAB_HAND 1
#define SLAB_HANDV fa.hand
#define SLAB_PTAG hand_ptag(tk_, fa.hand)
#define SLAB_XHAND 0
#define SLAB_XHANDV (SlabHand{})
#define SLAB_CTAG 0u
#define SLAB_PART 1
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
        auto wo_rest = [&]() __attribute__((always_inline)) {
#define SLAB_PART 2
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
        };
#undef SLAB_A
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
        {
            constexpr int ROLE = R_NORM_SWIGLU, NV = NV_B, UPW = UPW_B;
#define SLAB_EARLY (NT == 1024 ? 1 : 0)     /* the wide matrices (round 6): W1|W3's first unit before Wo's body, the other three after x has arrived */
#define SLAB_A fa.w13
#define SLAB_HAND 0
#define SLAB_HANDV (SlabHand{})
#define SLAB_PTAG 0u
#define SLAB_XHAND 1
#define SLAB_XHANDV fa.hand
#define SLAB_CTAG hand_ctag(tk_, fa.hand)
#define SLAB_XHAND_WAIT (blockIdx.x >= fa.wo_wgs ? fa.wait16 : 0u)
#define SLAB_XHAND_NAP 2
#define SLAB_PART 1
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
            if (blockIdx.x < fa.wo_wgs) wo_rest();
            __syncthreads();                // (LDS is W1|W3's from here)
#define SLAB_PART 2
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
#undef SLAB_A
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
#undef SLAB_XHAND_WAIT
#undef SLAB_XHAND_NAP
#undef SLAB_EARLY
        }
#undef SLAB_BID
    }
}

// ---- W2 of layer l + q | k | v + attention of layer l + 1 in ONE launch (round 5) ---------------------------------------------------------
// The third hand-off: the residual stream x = x + W2 . hb reaches the NEXT layer's q | k | v projection as granules of the same launch (an
// all-gather, like Wo -> W1|W3 above), and q / k / v reach that layer's attention workgroups as in qkv_attn_fused_kernel.  With it a
// one-sequence step on the small matrices is TWO launches per layer.  The projection workgroups run W2's rows first (every one of them is a
// producer), the attention workgroups come LAST in the grid, ask for their K / V rows, nap (the q / k / v of the next layer are two bodies
// away) and poll.  Issue order: W2's loads, q|k|v's weight loads, W2's arithmetic, q|k|v's.  Reference: infer/infer.c:950-965, 758-879.
struct W2QkvArgs { GemvDev w2; GemvDev g; AttnArgs a; SlabHand xh; SlabHand hand; uint32_t n_attn, head_wgs, wait16, w2_wgs, xwait, ngemv; };
template <int NV_A, int UPW_A, int NV_B, int UPW_B>
__global__ __launch_bounds__(256) void w2_qkv_attn_fused_kernel(const W2QkvArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);
    if (blockIdx.x >= fa.ngemv) {
        const uint32_t ab = blockIdx.x - fa.ngemv;
        const uint32_t split = ab / fa.head_wgs, grp = ab - split * fa.head_wgs;
        attention_body<8, 4, 1, 1, false, false, 2, false, true>(fa.a, smem, grp, 0u, split, fa.hand, hand_ctag(tk_, fa.hand), fa.wait16);
        return;
    }
    constexpr int GS = 64, B = 1;
    {
        constexpr int ROLE = R_RESID, NV = NV_A, UPW = UPW_A;
#define SLAB_BID blockIdx.x
#define SLAB_A fa.w2
#define SLAB_HAND 1
#define SLAB_HANDV fa.xh
#define SLAB_PTAG hand_ptag(tk_, fa.xh)
#define SLAB_XHAND 0
#define SLAB_XHANDV (SlabHand{})
#define SLAB_CTAG 0u
#define SLAB_PART 1
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
        auto w2_rest = [&]() __attribute__((always_inline)) {
#define SLAB_PART 2
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
        };
#undef SLAB_A
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
        {
            constexpr int ROLE = R_NORM_STORE, NV = NV_B, UPW = UPW_B;
#define SLAB_A fa.g
#define SLAB_HAND 1
#define SLAB_HANDV fa.hand
#define SLAB_PTAG hand_ptag(tk_, fa.hand)
#define SLAB_XHAND 1
#define SLAB_XHANDV fa.xh
#define SLAB_CTAG hand_ctag(tk_, fa.xh)
#define SLAB_XHAND_WAIT fa.xwait
#define SLAB_XHAND_NAP 2
#define SLAB_PART 1
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
            if (blockIdx.x < fa.w2_wgs) w2_rest();
            __syncthreads();
#define SLAB_PART 2
#include "gemv_q80_slab_body.inc"
#undef SLAB_PART
#undef SLAB_A
#undef SLAB_HAND
#undef SLAB_HANDV
#undef SLAB_PTAG
#undef SLAB_XHAND
#undef SLAB_XHANDV
#undef SLAB_CTAG
#undef SLAB_XHAND_WAIT
#undef SLAB_XHAND_NAP
        }
#undef SLAB_BID
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------
// STREAM kernel (classifier)
// ------------------------------------------------------------------------------------------------------------
template <int ROLE, int GS, int B, int NV>
__global__ __launch_bounds__(256) void gemv_q80_stream_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LPG = GS / 16, GC = 1024 / GS, F = GC / 4;
    static_assert(F >= 1, "group size too large for the stream kernel");
    const uint32_t n = a.n, ng = a.ng;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n16 = (n + 15) & ~15u, ng4 = (ng + 3) & ~3u;
    int8_t *xq = reinterpret_cast<int8_t *>(smem);
    float *xs = reinterpret_cast<float *>(smem + B * n16);
    float *red = xs + B * ng4;
    int *tab = reinterpret_cast<int *>(red + B * 16) + wid * 16 * GC;

    Staged<B, NV> sx;
    stage_issue<ROLE, B, NV>(a, sx);

    const uint32_t rows = a.rows[0];
    const uint32_t nchunk = a.nchunk;
    const uint32_t ntiles = (rows + 15) >> 4;
    const uint32_t nwaves = gridDim.x * 4, wave_g = blockIdx.x * 4 + (uint32_t)wid;
    const uint32_t nunits = (wave_g < ntiles) ? ((ntiles - wave_g + nwaves - 1) / nwaves) * nchunk : 0;
    const __amdgpu_buffer_rsrc_t rw_ = mkrsrc(a.w[0], rows * n);
    const __amdgpu_buffer_rsrc_t rs_ = mkrsrc(a.ws[0], rows * ng * 4u);

    int4 wA[16];
    float sA[F];
    // unit u of this wave -> (tile, chunk); chunk fastest so a row's running value stays in the wave
    uint32_t tile_i = 0, chunk_i = 0;                      // cursor of the NEXT unit to issue
    auto issue = [&](int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + tile_i * nwaves;
        const uint32_t row0 = tile << 4;
        const uint32_t col = (chunk_i << 10) + (uint32_t)lane * 16u;
        const uint32_t base = (col < n) ? row0 * n + col : OOB;
#pragma unroll
        for (int r = 0; r < 16; r++) w[r] = bload_w(rw_, base + (uint32_t)r * n);       // rows beyond `rows` are out of range -> 0
        const uint32_t gb = chunk_i * GC + ((uint32_t)lane & 3u) * F;
        const uint32_t so = ((row0 + ((uint32_t)lane >> 2)) * ng + gb) * 4u;
#pragma unroll
        for (int f = 0; f < F; f++) s[f] = bload_f(rs_, (gb + f < ng) ? so + 4u * f : OOB);
        if (++chunk_i == nchunk) { chunk_i = 0; tile_i++; }
    };
    if (nunits) issue(wA, sA);

    stage_finish<ROLE, GS, B, NV>(a, sx, xq, xs, red, n16, ng4);

    float val[B], best[B];
    uint32_t besti[B];
#pragma unroll
    for (int b = 0; b < B; b++) { val[b] = 0.0f; best[b] = -INFINITY; besti[b] = 0xffffffffu; }
    uint32_t ctile = 0, cchunk = 0;                        // cursor of the unit being consumed
    auto consume = [&](int4 (&w)[16], float (&s)[F]) {
        const uint32_t tile = wave_g + ctile * nwaves;
        const uint32_t col = (cchunk << 10) + (uint32_t)lane * 16u;
        const uint32_t gb = cchunk * GC + ((uint32_t)lane & 3u) * F;
        const bool last = cchunk + 1 == nchunk;
        const uint32_t row = (tile << 4) + ((uint32_t)lane >> 2);
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < (int)a.nb) {
                const int4 xv = (col < n) ? *reinterpret_cast<const int4 *>(xq + b * n16 + col) : make_int4(0, 0, 0, 0);
                int iv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    int d = __builtin_amdgcn_sdot4(w[r].x, xv.x, 0, false);
                    d = __builtin_amdgcn_sdot4(w[r].y, xv.y, d, false);
                    d = __builtin_amdgcn_sdot4(w[r].z, xv.z, d, false);
                    d = __builtin_amdgcn_sdot4(w[r].w, xv.w, d, false);
                    iv[r] = dpp_group_sum<LPG>(d);
                }
                if ((lane % LPG) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) tab[r * GC + lane / LPG] = iv[r];
                }
                float p[F];
#pragma unroll
                for (int f = 0; f < F; f++) {
                    const int v = tab[(lane >> 2) * GC + (lane & 3) * F + f];
                    const float xsc = (gb + f < ng) ? xs[b * ng4 + gb + f] : 0.0f;
                    p[f] = ((float)v * s[f]) * xsc;                                          // infer.c:672
                }
                // ordered fold over the 4 lanes of a row: stage k adds lane k's products onto the running value
                float cur = val[b];
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    float t = cur;
#pragma unroll
                    for (int f = 0; f < F; f++) t += p[f];                                   // groups beyond ng add +0.0f (exact)
                    cur = (st == 0) ? DPP_F(t, 0x00) : (st == 1) ? DPP_F(t, 0x55) : (st == 2) ? DPP_F(t, 0xAA) : DPP_F(t, 0xFF);
                }
                val[b] = cur;
                if (last) {
                    if ((lane & 3) == 0 && row < rows) {
                        a.out[0][(size_t)b * a.out_bstride[0] + row] = cur;
                        if (cur > best[b] || besti[b] == 0xffffffffu) { best[b] = cur; besti[b] = row; }   // rows ascend per lane: first max kept
                    }
                    val[b] = 0.0f;
                }
            }
        }
        if (++cchunk == nchunk) { cchunk = 0; ctile++; }
    };
    // single register buffer: four waves per SIMD (128 VGPRs) keep the other tiles' loads in flight while one wave
    // consumes -- measured 4-5 % faster than a two-buffer software pipeline at two waves per SIMD (round-1 kernel laboratory)
    for (uint32_t u = 0; u < nunits; u++) {
        if (u) issue(wA, sA);
        consume(wA, sA);
    }
    // per-wave arg-max partial: larger value wins, equal values -> lower row (== the reference's first maximum)
    if (a.tile_max) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (b < (int)a.nb) {
                float bv = best[b]; uint32_t bi = besti[b];
#pragma unroll
                for (int o = 32; o >= 4; o >>= 1) {
                    const float ov = __shfl_xor(bv, o, 64);
                    const uint32_t oi = __shfl_xor(bi, o, 64);
                    if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
                if (lane == 0) {
                    float *tm = a.tile_max + ((size_t)b * a.ntiles + wave_g) * 2;
                    tm[0] = bv; tm[1] = __uint_as_float(bi);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------

// rows per workgroup / waves per workgroup of a slab launch (tuned on Qwen3-0.6B with the round-1 kernel laboratory: the chain
// time is flat within 3 % around these choices -- the kernels are latency bound)
struct SlabPlan { uint32_t rw, nw, upw, nv; };
static SlabPlan plan_slab(const GemvArgs &a, int B) {
    const uint32_t nchunk = (a.n + 1023) / 1024, nmat = a.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    uint32_t align = 0;                                   // a workgroup's rows must lie inside one segment
    if (nseg > 1) for (uint32_t s = 0; s < nseg; s++) align |= a.seg[s].rows;
    const uint32_t rows = total_rows(a);
    // Every workgroup re-stages the activations (B x n elements), so the row slab grows until one wave of workgroups
    // covers the chip: the largest power of two with >= 256 workgroups (small matrices: ~4 units = 16 KiB of weights per
    // matrix and >= 128 workgroups, the tuned batch-1 optimum), bounded by the LDS product table.
    uint32_t rw = 4;
    while (rw < 32 && (align % (rw * 2)) == 0 && (rw * 2 / 4) * nchunk <= 4 && rows / (rw * 2) >= 128) rw *= 2;
    while (rw < 64 && (align % (rw * 2)) == 0 && rows / (rw * 2) >= 256) rw *= 2;
    {
        const uint32_t ng = a.n / a.gs, pitch = ((ng + 47) / 64) * 64 + 16;
        while (rw > 4 && (size_t)B * nmat * rw * pitch * 4 > 64 * 1024) rw /= 2;
        while (rw > 4 && (rw / 4) * nchunk * nmat > 64) rw /= 2;          // <= 16 waves x 4 units
    }
    // One workgroup per CU when the power of two leaves CUs idle (round 3, Qwen3-0.6B's W1|W3: 3072 rows as 192 slabs of 16 ->
    // 256 slabs of 12: 1871-1879 -> 1896 tok/s; the slab kernel takes any row count).  One-segment launches only.
    if (B == 1 && nseg == 1) {
        const uint32_t cus = a.cus ? a.cus : 256u, c = (rows + cus - 1) / cus;
        if (c >= 5 && c < rw && rows / rw < cus && ((c + 3) / 4) * nchunk * nmat <= 64) rw = c;
    }
    // Large matrices (Qwen3-4B's layers: 10-50 MB each) are bandwidth rather than latency bound, and a CU pulls ~25 GB/s whatever
    // it runs: the launch ends when the CU with the most rows ends.  BALANCED slabs (round 3): rw = ANY row count, chosen to
    // minimise (rounds of `cus` workgroups) x rw = the rows the busiest CU streams; a power-of-two slab left 160 of 256 CUs
    // busy on a 2560-row matrix (rw 16) where rw = 10 gives every CU one workgroup.  On a tie the larger slab (fewer
    // workgroups re-staging the activation).
    uint32_t large_nw = 0;
    // (round 5: TWO sequences on these matrices take the same balanced slabs, the product table twice as large -- Qwen3-4B at 2 sequences
    //  1.833 ms per step against 1.923 through G6 MODE P, same box; four sequences: 2.80 against 1.99 through G6, so two is where it ends)
    if (B <= 2 && (uint64_t)rows * a.n * nmat >= (8u << 20)) {
        constexpr bool balanced = true;                            // (round 2's power-of-two rule below is kept for the record of what was measured)
        const uint32_t cus = a.cus ? a.cus : 256u;
        uint32_t best = 0, best_cost = ~0u;
        if (balanced) {
            const uint32_t ng = a.n / a.gs, pitch = (1024 / a.gs == 16) ? (((ng + 47) / 64) * 64 + 16) : (((ng + 3) & ~3u) + 4);
            for (uint32_t c = 4; c <= 64; c++) {
                const uint32_t tpw = (c + 3) / 4;
                if (tpw * nchunk * nmat > 64) break;                           // <= 16 waves x 4 units
                if ((size_t)B * nmat * tpw * 4 * pitch * 4 > 96 * 1024) break;     // product table
                uint32_t wgs = 0;
                if (nseg > 1) for (uint32_t s2 = 0; s2 < nseg; s2++) wgs += (a.seg[s2].rows + c - 1) / c; else wgs = (rows + c - 1) / c;
                // rows of the busiest CU; more than one workgroup per CU pays its prologue several times over on shared issue
                // slots (measured: QKV of Qwen3-4B, 768 workgroups of 8 rows 7.4 us vs 192 of 32 rows 6.9), so x 1.15 then
                uint32_t cost = ((wgs + cus - 1) / cus) * c * 100u;
                if (wgs > cus) cost += cost * 15u / 100u;
                if (cost <= best_cost) { best_cost = cost; best = c; }
            }
        } else {
            for (uint32_t c = 4; c <= 64; c *= 2) {
                const uint64_t bytes = (uint64_t)c * a.n * nmat;
                if (nseg > 1 && (align % c) != 0) continue;
                if (bytes < (64u << 10) || bytes > (160u << 10) || (c / 4) * nchunk * nmat > 64) continue;
                const uint32_t wgs = (rows + c - 1) / c, cost = ((wgs + 255) / 256) * c;
                if (cost <= best_cost) { best_cost = cost; best = c; }
            }
        }
        if (best) {
            rw = best;
            const uint32_t u = ((rw + 3) / 4) * nchunk * nmat;
            large_nw = u / 2 < 8 ? 8 : (u / 2 > 16 ? 16 : u / 2);
        }
    }
    const uint32_t force_nw = 0;
    const uint32_t units = ((rw + 3) / 4) * nchunk * nmat;
    uint32_t nw = units < 4 ? units : 4;
    // (one sequence, re-swept on round 6's last day with the three-launch layer: a wave per 384 activation values -- W2 of Qwen3-0.6B on 8 waves
    //  instead of 6 -- 1994 / 1978 tok/s against 1979 / 1966 with 512, 1984 / 1972 with 448, 1952 / 1959 with 320; the five-launch form and Qwen3-4B: even)
    const uint32_t want_div = B == 1 ? 384u : 512u;
    uint32_t want = (a.n * (uint32_t)(B > 2 ? B / 2 : 1) + want_div - 1) / want_div;     // idle waves still help the activation prologue
    if (want > 16) want = 16;
    if (nw < want) nw = want;
    if (nw * 64 < rw * (uint32_t)B) nw = (rw * (uint32_t)B + 63) / 64;      // one fold thread per (row, sequence)
    if (nw < 2) nw = 2;
    uint32_t upw = (units + nw - 1) / nw;
    while (upw > 4 && nw < 16) { nw++; upw = (units + nw - 1) / nw; }
    if (large_nw) { nw = large_nw; upw = (units + nw - 1) / nw; while (upw > 4 && nw < 16) { nw++; upw = (units + nw - 1) / nw; } }
    if (force_nw) { nw = force_nw; upw = (units + nw - 1) / nw; }
    // (Round 4 tried a raw barrier between the activation loads and the weight loads of the large slabs, so that every wave's activation
    // is asked for before any weight -- round 3 had measured the activation of Qwen3-4B's W1|W3 "arriving" with the end of the 52.9 MB
    // burst.  Measured on one box: 1.4707 ms per step with it, 1.4594 without.  The launch is bound by latency + stream + tail, not by
    // where the activation sits in the queue.  Removed.)
    SlabPlan p{rw, nw, upw, (a.n + 256 * nw - 1) / (256 * nw)};
    return p;
}

template <int ROLE, int GS, int B, int NV, int UPW>
static hipError_t launch_slab_t(const GemvDev &d, const SlabPlan &p, uint32_t nwg, hipStream_t st) {
    const uint32_t nmat = d.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n16 = (d.n + 15) & ~15u, ng4 = (d.ng + 3) & ~3u;
    const size_t pitch = (1024 / GS == 16) ? (((d.ng + 47) / 64) * 64 + 16) : (ng4 + 4);
    const size_t lds = B * n16 + B * ng4 * 4 + B * 64 + ((d.flags & F_COMBINE) ? (size_t)B * d.attn_n_head * 32 : 0) + (size_t)B * nmat * (d.tpw * 4) * pitch * 4;
    GemvDev dd = d; dd.nthr = 64 * p.nw;
    // the matrices of >= 8 M weights (d.early, one or two sequences, group size 64): the first unit of every wave before the activation is
    // quantized, the others after (gemv_q80_slab_body.inc SLAB_EARLY)
    if constexpr (GS == 64 && B <= 2 && UPW >= 2 && NV >= 1) if (d.early) {
        auto kern = &gemv_q80_slab_kernel<ROLE, GS, B, NV, UPW, 1>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * p.nw), lds, st, dd);
        return hipGetLastError();
    }
    auto kern = &gemv_q80_slab_kernel<ROLE, GS, B, NV, UPW>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * p.nw), lds, st, dd);
    return hipGetLastError();
}
template <int ROLE, int GS, int B>
static hipError_t launch_slab_r(const GemvDev &d, const SlabPlan &p, uint32_t rows, hipStream_t st) {
    if (p.upw > 4) return hipErrorInvalidValue;
#define SLAB_GO(NV_, UPW_) do { if constexpr (B * NV_ <= 8) return launch_slab_t<ROLE, GS, B, NV_, UPW_>(d, p, rows, st); } while (0)
    int nv = p.nv <= 1 ? 1 : p.nv <= 2 ? 2 : p.nv <= 4 ? 4 : 0;
    const int upw = p.upw <= 1 ? 1 : p.upw <= 2 ? 2 : 4;
    if (B * nv > 8) nv = 0;          // too many staged registers: loop path
    if (nv == 1) { if (upw == 1) SLAB_GO(1, 1); if (upw == 2) SLAB_GO(1, 2); SLAB_GO(1, 4); }
    if (nv == 2) { if (upw == 1) SLAB_GO(2, 1); if (upw == 2) SLAB_GO(2, 2); SLAB_GO(2, 4); }
    if (nv == 4) { if (upw == 1) SLAB_GO(4, 1); if (upw == 2) SLAB_GO(4, 2); SLAB_GO(4, 4); }
    if (upw == 1) SLAB_GO(0, 1);
    if (upw == 2) SLAB_GO(0, 2);
    SLAB_GO(0, 4);
    return hipErrorInvalidValue;
#undef SLAB_GO
}
template <int GS, int B>
static hipError_t launch_slab_b(GemvDev &d, const GemvArgs &a, hipStream_t st) {
    const SlabPlan p = plan_slab(a, B);
    d.rw = p.rw;
    // the matrices of >= 8 M weights, one or two sequences: the first unit of every wave's weights before the activation is quantized, the others
    // after (SLAB_EARLY).  Same box, interleaved (profiles/r06_slab_early_units.txt): Qwen3-4B one sequence 1.471 -> 1.427 ms per step, two
    // sequences 1.956 -> 1.825; the first TWO units early: no gain over none (one sequence), the same as one (two sequences).
    d.early = (B <= 2 && p.upw >= 2 && (uint64_t)total_rows(a) * a.n * (a.epi == GEMV_EPI_SWIGLU ? 2 : 1) >= (8u << 20)) ? 1u : 0u;
    d.tpw = (p.rw + 3) / 4;
    d.magic_rw = 65536u / p.rw + 1u;                                   // (tid * magic_rw) >> 16 == tid / rw for tid < 1024 <= 65536 / rw
    d.log2_tiles = 0;
    const bool sw = d.epi == GEMV_EPI_SWIGLU;
    d.units = d.tpw * d.nchunk * (sw ? 2 : 1);
    // workgroups per segment (a workgroup's rows lie inside one segment; the last one of a segment may be ragged)
    uint32_t wg[3] = {0, 0, 0};
    const uint32_t nseg = sw ? 1u : a.nseg;
    for (uint32_t s2 = 0; s2 < nseg; s2++) wg[s2] = (a.seg[s2].rows + p.rw - 1) / p.rw;
    d.wg_c0 = nseg > 1 ? wg[0] : 0xffffffffu;
    d.wg_c1 = nseg > 2 ? wg[0] + wg[1] : 0xffffffffu;
    const uint32_t rows = wg[0] + wg[1] + wg[2];                       // the grid
    // the per-layer launches of a batch-1 step: flags resolved at compile time.  Group size 64: the role kernels carry the canonical fold
    // only, so a launch that is not canonical (strict mode; a row length that is no multiple of 256) takes the generic kernel
    if constexpr (B == 1) if (GS != 64 || d.canon) {
        const uint32_t f = d.flags;
        if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_slab_r<R_NORM_STORE, GS, B>(d, p, rows, st);
        if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_slab_r<R_RESID, GS, B>(d, p, rows, st);
        if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_slab_r<R_RESID_COMBINE, GS, B>(d, p, rows, st);
        if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU) return launch_slab_r<R_NORM_SWIGLU, GS, B>(d, p, rows, st);
    }
    return launch_slab_r<R_GENERIC, GS, B>(d, p, rows, st);
}

template <int ROLE, int GS, int B>
static hipError_t launch_stream_r(GemvDev &d, hipStream_t st) {
    d.ntiles = STREAM_WGS * 4; d.nthr = 256;
    const size_t n16 = (d.n + 15) & ~15u, ng4 = (d.ng + 3) & ~3u;
    const size_t lds = B * n16 + B * ng4 * 4 + B * 64 + 4 * 16 * (1024 / GS) * 4;
    const uint32_t nv = (d.n + 1023) / 1024;
    hipEvent_t e0 = g_q80_probe_start, e1 = g_q80_probe_stop;
    g_q80_probe_start = g_q80_probe_stop = nullptr;
#define STREAM_GO(NV_) do { if (e0 && e1) hipExtLaunchKernelGGL((gemv_q80_stream_kernel<ROLE, GS, B, NV_>), dim3(STREAM_WGS), dim3(256), (uint32_t)lds, st, e0, e1, 0, d); \
                            else hipLaunchKernelGGL((gemv_q80_stream_kernel<ROLE, GS, B, NV_>), dim3(STREAM_WGS), dim3(256), lds, st, d); } while (0)
    bool done = false;
    if (nv <= 1) { STREAM_GO(1); done = true; }
    if constexpr (B <= 4) { if (!done && nv <= 2) { STREAM_GO(2); done = true; } }
    if constexpr (B <= 2) { if (!done && nv <= 4) { STREAM_GO(4); done = true; } }
    if (!done) { STREAM_GO(0); }
#undef STREAM_GO
    return hipGetLastError();
}
template <int GS, int B>
static hipError_t launch_stream_b(GemvDev &d, hipStream_t st) {
    if (d.flags == F_NORM) return launch_stream_r<R_NORM_STORE, GS, B>(d, st);      // the classifier
    return launch_stream_r<R_GENERIC, GS, B>(d, st);
}

#if NANO_Q80_GS == 64
// ---- the fused q | k | v + attention launch: host side -----------------------------------------------------------------------------------
static bool fused_shape(const GemvArgs &ga, const AttnArgs &aa, SlabPlan &p) {
    if (ga.gs != 64 || ga.nb != 1 || ga.nseg != 3 || ga.epi != GEMV_EPI_STORE || !ga.norm_w || ga.xq_in || ga.attn_part || ga.tile_max || ga.resid_add) return false;
    if (ga.n % 64u || ga.n > 4096u || use_stream(ga) || !q80_canonical(ga)) return false;        // (the role kernels of group size 64 carry the canonical fold only)
    if (ga.seg[0].out_pstride || ga.seg[1].out_pstride) return false;            // (only v is position indexed: its cache row)
    p = plan_slab(ga, 1);
    if (p.nw != 4u || p.upw > 4u || !(p.nv == 1u || p.nv == 2u || p.nv == 4u)) return false;        // 256 threads, like the attention workgroups
    if (!fused_attn_side_ok(aa, ga.seg[0].rows, ga.seg[1].rows, ga.seg[2].rows)) return false;          // (kernels.h)
    return true;
}

// ---- the fused Wo + W1|W3 launch: host side ----------------------------------------------------------------------------------------------
// Both bodies run on the launch's threads = W1|W3's own plan (its rmsnorm tree follows the thread count; Wo has no tree: any count gives its
// bits).  Instantiated: Qwen3-0.6B's shapes (256 threads) and Qwen3-4B's (1024 threads, every W1|W3 weight load of a workgroup in flight
// while Wo computes).
struct Wo13Plan { SlabPlan a, b; uint32_t nw, nv_a, upw_a, wa, wb; int sig; };
static bool wo13_shape(const GemvArgs &wo, const GemvArgs &w13, Wo13Plan &q) {
    if (wo.gs != 64 || w13.gs != 64 || wo.nb != 1 || w13.nb != 1 || !q80_canonical(wo) || !q80_canonical(w13)) return false;
    if (wo.nseg != 1 || wo.epi != GEMV_EPI_RESID || wo.norm_w || wo.xq_in || wo.tile_max || wo.resid_add || wo.seg[0].out_pstride) return false;
    if (wo.attn_part && (wo.attn_nsplit > 8u || wo.attn_hd % 4u)) return false;
    if (w13.nseg != 2 || w13.epi != GEMV_EPI_SWIGLU || !w13.norm_w || w13.xq_in || w13.attn_part || w13.tile_max || w13.resid_add) return false;
    if (use_stream(wo) || use_stream(w13) || w13.n != wo.seg[0].rows || w13.xin != wo.seg[0].out || w13.n % 4u) return false;
    q.a = plan_slab(wo, 1); q.b = plan_slab(w13, 1);
    q.nw = q.b.nw;
    const uint32_t units_a = ((q.a.rw + 3) / 4) * ((wo.n + 1023) / 1024);
    q.upw_a = (units_a + q.nw - 1) / q.nw;
    q.nv_a = (wo.n / 4 + 64 * q.nw - 1) / (64 * q.nw);
    q.wa = (wo.seg[0].rows + q.a.rw - 1) / q.a.rw; q.wb = (w13.seg[0].rows + q.b.rw - 1) / q.b.rw;
    const uint32_t cus = w13.cus ? w13.cus : 256u;
    if (q.wa > q.wb || q.wb > cus) return false;                                                // one workgroup per CU: the whole grid is resident
    if (q.a.rw > 64 * q.nw || q.b.rw > 64 * q.nw) return false;                                 // one fold thread per row
    q.sig = 0;
    if (q.nw == 4u && q.nv_a == 2u && q.upw_a == 1u && q.b.nv == 1u && q.b.upw == 2u) q.sig = 1;       // Qwen3-0.6B
    if (q.nw == 16u && q.nv_a == 1u && q.upw_a == 1u && q.b.nv == 1u && q.b.upw == 4u) q.sig = 2;      // Qwen3-4B
    if (q.sig == 1) {
        // Round 6, last day: the same two bodies on EIGHT waves where that leaves one float4 item per thread and one unit per wave in both (Qwen3-0.6B:
        // Wo's 512 items, 4 units; W1|W3's 256 items, 6 units) -- the four-wave form dates from the first fused build and had Wo's threads quantize
        // two items each.  Same bits (Wo has no tree; W1|W3's items sit on the same threads).  Same box, driver's flags | full window:
        // 1991.2 | 1894.2, 1992.5 | 1893.4, 1990.7 | 1895.2 with four waves, 2033.3 | 1919.6, 2039.2 | 1937.1, 2030.2 | 1919.2 with eight.
        const uint32_t units_b = ((q.b.rw + 3) / 4) * ((w13.n + 1023) / 1024) * 2u;
        if ((units_a + 7u) / 8u == 1u && (wo.n / 4 + 511u) / 512u == 1u && (units_b + 7u) / 8u == 1u && (w13.n / 4 + 511u) / 512u == 1u) { q.sig = 3; q.nw = 8u; }
    }
    return q.sig != 0;
}
static void slab_dev_fill(GemvDev &d, const GemvArgs &a, const SlabPlan &p, uint32_t nthr) {
    d.tile_max = nullptr;
    d.rw = p.rw; d.tpw = (p.rw + 3) / 4; d.magic_rw = 65536u / p.rw + 1u; d.log2_tiles = 0;
    d.units = d.tpw * d.nchunk * (a.epi == GEMV_EPI_SWIGLU ? 2u : 1u);
    d.wg_c0 = 0xffffffffu; d.wg_c1 = 0xffffffffu;
    d.nthr = nthr;
}
static size_t slab_lds(const GemvDev &d) {
    const uint32_t nmat = d.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n16 = (d.n + 15) & ~15u, ng4 = (d.ng + 3) & ~3u, pitch = ((d.ng + 47) / 64) * 64 + 16;
    return n16 + ng4 * 4 + 64 + ((d.flags & F_COMBINE) ? (size_t)d.attn_n_head * 32 : 0) + (size_t)nmat * (d.tpw * 4) * pitch * 4;
}

#endif

template <int GS, int B>
static hipError_t launch_b(const GemvArgs &a, hipStream_t st) {
    GemvDev d = to_dev(a);
    if (use_stream(a)) return launch_stream_b<GS, B>(d, st);
    d.tile_max = nullptr;
    return launch_slab_b<GS, B>(d, a, st);
}
template <int GS>
static hipError_t launch_gs(const GemvArgs &a, hipStream_t st) {
    if (a.nb <= 1) return launch_b<GS, 1>(a, st);
    if (a.nb <= 2) return launch_b<GS, 2>(a, st);
    if (a.nb <= 4) return launch_b<GS, 4>(a, st);
    return launch_b<GS, 8>(a, st);
}

}  // namespace

hipError_t NANO_Q80_ENTRY(const GemvArgs &a, hipStream_t st) { return launch_gs<NANO_Q80_GS>(a, st); }

#if NANO_Q80_GS == 64
bool qkv_attn_fused_supports(const GemvArgs &ga, const AttnArgs &aa) { SlabPlan p; return fused_shape(ga, aa, p); }

hipError_t launch_qkv_attn_fused(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    SlabPlan p;
    if (!hand || !tick || !layer1 || layer1 > 127u || !fused_shape(ga, aa, p)) return hipErrorInvalidValue;
    GemvDev d = to_dev(ga);
    d.tile_max = nullptr;
    d.rw = p.rw; d.tpw = (p.rw + 3) / 4; d.magic_rw = 65536u / p.rw + 1u; d.log2_tiles = 0;
    d.units = d.tpw * d.nchunk;
    uint32_t wg[3];
    for (uint32_t s2 = 0; s2 < 3; s2++) wg[s2] = (ga.seg[s2].rows + p.rw - 1) / p.rw;
    d.wg_c0 = wg[0]; d.wg_c1 = wg[0] + wg[1];
    const uint32_t ngemv = wg[0] + wg[1] + wg[2];
    d.nthr = 256;
    AttnArgs a = aa;
    { uint32_t l2 = 0; while ((1u << l2) < a.n_kv_head) l2++; a.kv_log2 = l2; }
    { const uint32_t kv_mul = a.n_head / a.n_kv_head; uint32_t l2 = 0; while ((1u << l2) < kv_mul) l2++; a.kvmul_log2 = l2; }
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    h.base[0] = 0; h.base[1] = a.q_dim; h.base[2] = a.q_dim + a.kv_dim;
    const uint32_t n_attn = a.n_head * a.nsplit;
    const size_t n16 = (d.n + 15) & ~15u, ng4 = (d.ng + 3) & ~3u, pitch = ((d.ng + 47) / 64) * 64 + 16;
    const size_t lds_g = n16 + ng4 * 4 + 64 + (size_t)(d.tpw * 4) * pitch * 4;
    const size_t hd4 = a.hd, lds_a = (hd4 + hd4 + 4 + 4 + 4 * hd4 + hd4) * sizeof(float);            // q | k | maxima | sums | 4 waves' partials | the fresh v row
    const size_t lds = lds_g > lds_a ? lds_g : lds_a;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const int upw = p.upw <= 1 ? 1 : p.upw <= 2 ? 2 : 4;
    FusedArgs fa{};
    fa.g = d; fa.a = a; fa.hand = h; fa.n_attn = n_attn; fa.head_wgs = a.n_head; fa.ngemv = ngemv;
    // The attention workgroups nap `wait16` x 16 x 64 cycles between asking for their K / V rows and the first poll.  Round 5 (consumers FIRST in
    // the grid: they started before the projection) tuned 3: 1881-1887 tok/s without, 1899-1901 with 3, 1851 with 5.  Round 6 put the PRODUCERS
    // first -- the attention workgroups start last and the nap is mostly dead time: same box, driver's flags, 0 / 1 / 2 / 3 naps: 1984 / 1983 /
    // 1961 / 1919, 1977 / 1986 / 1958 / 1920, 1982 / 1981 / 1960 / 1912 tok/s; positions 31..510: 1878 / 1880 / 1859 / 1821
    // (profiles/r06_handoff_naps.txt).
    fa.wait16 = 1u;
#define FUSED_GO(NV_, UPW_) do { hipLaunchKernelGGL((qkv_attn_fused_kernel<NV_, UPW_>), dim3(n_attn + ngemv), dim3(256), lds, st, fa); return hipGetLastError(); } while (0)
#define FUSED_NV(NV_) do { if (upw == 1) FUSED_GO(NV_, 1); if (upw == 2) FUSED_GO(NV_, 2); FUSED_GO(NV_, 4); } while (0)
    if (p.nv == 1u) FUSED_NV(1);
    if (p.nv == 2u) FUSED_NV(2);
    FUSED_NV(4);
#undef FUSED_NV
#undef FUSED_GO
}

// W2 (layer l) + q | k | v + attention (layer l + 1)
struct W2QkvPlan { SlabPlan a, b; uint32_t nv_a, upw_a, wa; };
static bool w2qkv_shape(const GemvArgs &w2, const GemvArgs &ga, const AttnArgs &aa, W2QkvPlan &q) {
    if (!fused_shape(ga, aa, q.b)) return false;
    if (w2.gs != 64 || w2.nb != 1 || !q80_canonical(w2) || w2.nseg != 1 || w2.epi != GEMV_EPI_RESID || w2.norm_w || w2.xq_in || w2.attn_part || w2.tile_max ||
        w2.resid_add || w2.seg[0].out_pstride || use_stream(w2)) return false;
    if (ga.n != w2.seg[0].rows || ga.xin != w2.seg[0].out) return false;              // the projection's input is what W2 writes
    q.a = plan_slab(w2, 1);
    const uint32_t units_a = ((q.a.rw + 3) / 4) * ((w2.n + 1023) / 1024);
    q.upw_a = (units_a + 3u) / 4u;                                                   // on the launch's 256 threads
    q.nv_a = (w2.n / 4 + 255u) / 256u;
    q.wa = (w2.seg[0].rows + q.a.rw - 1) / q.a.rw;
    uint32_t ngemv = 0;
    for (uint32_t s2 = 0; s2 < 3; s2++) ngemv += (ga.seg[s2].rows + q.b.rw - 1) / q.b.rw;
    if (q.wa > ngemv || q.a.rw > 256u) return false;
    // instantiated: Qwen3-0.6B's shapes (W2: three float4 items per thread -> NV 4, one unit per wave; q|k|v: NV 1, UPW 1)
    return q.upw_a == 1u && q.nv_a >= 3u && q.nv_a <= 4u && q.b.nv == 1u && q.b.upw == 1u;
}

bool w2_qkv_attn_fused_supports(const GemvArgs &w2, const GemvArgs &ga, const AttnArgs &aa) { W2QkvPlan q; return w2qkv_shape(w2, ga, aa, q); }

hipError_t launch_w2_qkv_attn_fused(const GemvArgs &w2, const GemvArgs &ga, const AttnArgs &aa, unsigned long long *xhand, unsigned long long *hand,
                                    uint32_t *tick, uint32_t layer1, hipStream_t st) {
    W2QkvPlan q;
    if (!xhand || !hand || !tick || !layer1 || layer1 > 126u || !w2qkv_shape(w2, ga, aa, q)) return hipErrorInvalidValue;
    W2QkvArgs fa{};
    fa.w2 = to_dev(w2); slab_dev_fill(fa.w2, w2, q.a, 256u);
    GemvDev d = to_dev(ga);
    d.tile_max = nullptr;
    d.rw = q.b.rw; d.tpw = (q.b.rw + 3) / 4; d.magic_rw = 65536u / q.b.rw + 1u; d.log2_tiles = 0;
    d.units = d.tpw * d.nchunk;
    uint32_t wg[3];
    for (uint32_t s2 = 0; s2 < 3; s2++) wg[s2] = (ga.seg[s2].rows + q.b.rw - 1) / q.b.rw;
    d.wg_c0 = wg[0]; d.wg_c1 = wg[0] + wg[1];
    const uint32_t ngemv = wg[0] + wg[1] + wg[2];
    d.nthr = 256;
    fa.g = d;
    AttnArgs a = aa;
    { uint32_t l2 = 0; while ((1u << l2) < a.n_kv_head) l2++; a.kv_log2 = l2; }
    { const uint32_t kv_mul = a.n_head / a.n_kv_head; uint32_t l2 = 0; while ((1u << l2) < kv_mul) l2++; a.kvmul_log2 = l2; }
    fa.a = a;
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1 + 1u;                // q / k / v of the NEXT layer
    h.base[0] = 0; h.base[1] = a.q_dim; h.base[2] = a.q_dim + a.kv_dim;
    SlabHand xh{};
    xh.buf = xhand; xh.tick = tick; xh.layer1 = layer1;
    xh.base[0] = 0; xh.base[1] = 0; xh.base[2] = 0;
    fa.hand = h; fa.xh = xh;
    fa.n_attn = a.n_head * a.nsplit; fa.head_wgs = a.n_head; fa.w2_wgs = q.wa; fa.ngemv = ngemv;
    // naps (x 16 x 64 cycles) before the first polls: the attention workgroups' q / k / v are two bodies away (swept 3 .. 11: 7-8 best),
    // the projection workgroups all finish W2 together and nap ~1 us before asking for the others' rows (swept 0 .. 3: 2 best).  Measured
    // against the default (two fused launches + W2) on one box: 1879-1898 vs 1877-1920 tok/s -- break-even, hence opt-in.
    fa.wait16 = 7u; fa.xwait = 2u;      // (round 6 re-sweep, attention naps 2 / 4 / 7 x activation naps 0 / 2: 1890-1933 tok/s where the three-launch form does 1968-1987: opt-in still)
    const size_t la = slab_lds(fa.w2), lb = slab_lds(fa.g);
    const size_t hd4 = a.hd, lds_a = (hd4 + hd4 + 4 + 4 + 4 * hd4 + hd4) * sizeof(float);
    size_t lds = la > lb ? la : lb; if (lds_a > lds) lds = lds_a;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((w2_qkv_attn_fused_kernel<4, 1, 1, 1>), dim3(fa.n_attn + ngemv), dim3(256), lds, st, fa);
    return hipGetLastError();
}

bool wo_w13_fused_supports(const GemvArgs &wo, const GemvArgs &w13) { Wo13Plan q; return wo13_shape(wo, w13, q); }

hipError_t launch_wo_w13_fused(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    Wo13Plan q;
    if (!hand || !tick || !layer1 || layer1 > 127u || !wo13_shape(wo, w13, q)) return hipErrorInvalidValue;
    Wo13Args fa{};
    fa.wo = to_dev(wo); fa.w13 = to_dev(w13);
    slab_dev_fill(fa.wo, wo, q.a, 64 * q.nw); slab_dev_fill(fa.w13, w13, q.b, 64 * q.nw);
    fa.wo_wgs = q.wa;
    // workgroups that produce nothing nap 4 x 16 x 64 cycles (~2 us) before their first poll, every workgroup 128 cycles between sweeps: same
    // box, driver's flags, Qwen3-0.6B: 1846-1852 tok/s without, 1855-1865 with 2, 1879-1889 with 3, 1882-1887 with 4, 1847-1852 with 5, 1806-1817 with 6
    fa.wait16 = 4u;             // (round 6, with one nap in the q|k|v + attention launch: 0 / 2 / 4 naps here 1959 / 1978 / 1982 and 1959 / 1975 / 1982 tok/s)
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    h.base[0] = 0; h.base[1] = 0; h.base[2] = 0;
    fa.hand = h;
    const size_t la = slab_lds(fa.wo), lb = slab_lds(fa.w13), lds = la > lb ? la : lb;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const bool comb = (fa.wo.flags & F_COMBINE) != 0;
#define WO13_GO(RA_, NVA_, UA_, NVB_, UB_, NT_) do { hipLaunchKernelGGL((wo_w13_fused_kernel<RA_, NVA_, UA_, NVB_, UB_, NT_>), dim3(q.wb), dim3(NT_), lds, st, fa); return hipGetLastError(); } while (0)
    if (q.sig == 1) { if (comb) WO13_GO(R_RESID_COMBINE, 2, 1, 1, 2, 256); WO13_GO(R_RESID, 2, 1, 1, 2, 256); }
    if (q.sig == 3) { if (comb) WO13_GO(R_RESID_COMBINE, 1, 1, 1, 1, 512); WO13_GO(R_RESID, 1, 1, 1, 1, 512); }
    if (comb) WO13_GO(R_RESID_COMBINE, 1, 1, 1, 4, 1024);
    WO13_GO(R_RESID, 1, 1, 1, 4, 1024);
#undef WO13_GO
}
#endif

}  // namespace nano
