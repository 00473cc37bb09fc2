// attn_impl.h -- the attention kernel's body (see attn.hip for the design); a header so that the fused q | k | v + attention launch
// (gemv_q80_impl.h, group size 64) can run it inside its own kernel.
#pragma once
#include <hip/hip_fp16.h>
#include "device_common.h"
#include "kernels.h"

namespace nano {

namespace {

#ifndef NANO_GEMV_COMMON_H          // (the same helper in gemv_common.h, for the translation unit that includes both)
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
#endif

template <int W> __device__ __forceinline__ float group_sum_t(float v) {      // aligned groups of W lanes
    if (W >= 2) v += DPP_F(v, 0xB1);
    if (W >= 4) v += DPP_F(v, 0x4E);
    if (W >= 8) v += DPP_F(v, 0x141);
    if (W >= 16) v += DPP_F(v, 0x140);
    if (W >= 32) v += __shfl_xor(v, 16, 64);
    if (W >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += DPP_F(v, 0xB1); v += DPP_F(v, 0x4E); v += DPP_F(v, 0x141); v += DPP_F(v, 0x140);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

#ifndef NANO_GEMV_COMMON_H
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB = 0x7ffffff0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float bload_f(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
#endif

// ---- KV element format: FP32 rows (the reference's, infer/infer.c:46-51) or, opt-in, FP16 rows ------------
// A lane's four consecutive elements of a row travel as one 16-byte (FP32) or 8-byte (FP16) load and stay in that raw
// form until they are used, so that the loads remain in flight.
template <bool KVH> struct KVRaw { float4 v; };
template <> struct KVRaw<true> { uint2 v; };
__device__ __forceinline__ float4 kv_cvt(const KVRaw<false> &r) { return r.v; }
__device__ __forceinline__ float4 kv_cvt(const KVRaw<true> &r) {
    const __half2 a = *reinterpret_cast<const __half2 *>(&r.v.x), b = *reinterpret_cast<const __half2 *>(&r.v.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
}
__device__ __forceinline__ uint2 kv_pack_half(const float4 &v) {       // round to nearest even, like every later reader will see the row
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 r; r.x = *reinterpret_cast<const uint32_t *>(&a); r.y = *reinterpret_cast<const uint32_t *>(&b);
    return r;
}
template <bool KVH> __device__ __forceinline__ float4 kv_round(const float4 &v) {      // what the cache will hold for v
    if constexpr (!KVH) return v;
    else { KVRaw<true> r; r.v = kv_pack_half(v); return kv_cvt(r); }
}
template <bool KVH> __device__ __forceinline__ float kv_round1(float v) {
    if constexpr (!KVH) return v; else return __half2float(__float2half_rn(v));
}
template <bool KVH> __device__ __forceinline__ void kv_store4(float *row_base, uint32_t elem, const float4 &v) {   // row_base: start of the row (in the cache's own format)
    if constexpr (!KVH) *reinterpret_cast<float4 *>(row_base + elem) = v;
    else *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(row_base) + elem) = kv_pack_half(v);
}
template <bool KVH> __device__ __forceinline__ void kv_store1(float *row_base, uint32_t elem, float v) {
    if constexpr (!KVH) row_base[elem] = v; else reinterpret_cast<__half *>(row_base)[elem] = __float2half_rn(v);
}
template <bool KVH> __device__ __forceinline__ KVRaw<KVH> kv_load4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    KVRaw<KVH> o;
    if constexpr (!KVH) o.v = bload_f4(r, off);
    else { typedef int i32x2 __attribute__((ext_vector_type(2))); const i32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0); o.v = make_uint2((uint32_t)t.x, (uint32_t)t.y); }
    return o;
}

// ---- reductions over the sub-groups of a wave (the lanes of equal l % LPR), result in every lane; VALU only ---------------
// lane ^ 8: DPP row rotate by 8 (inside a 16-lane row); lane ^ 16 / lane ^ 32: v_permlane16_swap / v_permlane32_swap (gfx950):
// with both operands = v the pair of results is { v of the even half, v of the odd half } in every lane of the pair of halves.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xlane8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)); }
template <int LPR> __device__ __forceinline__ float xsub_sum(float v) {
    if (LPR <= 8) v += xlane8(v);
    { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
    return v;
}
template <int LPR> __device__ __forceinline__ float xsub_max(float v) {
    if (LPR <= 8) v = fmaxf(v, xlane8(v));
    { const u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    { const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])); }
    return v;
}

// ---- combining the splits of a head (flash-decoding's second half) ---------------------------------------------------------
// The weights of the splits' partial outputs: w_s = e_s / L, e_s = exp(m_s - M) for the splits that saw a timestep, L = the 8-slot
// pairwise-tree sum of l_s e_s per block of 8 splits, the blocks added in order -- gemv_common.h combine_weights() (the Wo GEMV's
// prologue, <= 8 splits) has the same arithmetic, so whoever combines produces the same bits.  One wave does it (lane s <-> split
// s, <= 64 splits): the tree of a block is three DPP steps (quad_perm xor 1, xor 2, row_half_mirror: lane 8k ends with
// ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)), fp addition commutes), the blocks are read lane by lane.  (Round 3's version kept all
// slots in every thread's registers: 32 expf + 32 divisions per thread, 5 us per launch at 32 splits.)  Called by EVERY thread of
// the workgroup (>= 64 threads); ends with a barrier; wsh[0..63] then holds the weights (0 beyond nsplit).
__device__ __forceinline__ void combine_weights_lds(const float *mlh, uint32_t nsplit, float *wsh) {
    if (threadIdx.x < 64u) {
        const uint32_t s = threadIdx.x;
        const bool in = s < nsplit;
        const float mm = in ? mlh[2 * s] : -INFINITY;
        float v = in ? mlh[2 * s + 1] : 0.0f;
        float M = v > 0.0f ? mm : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
        const float e = v > 0.0f ? expf(mm - M) : 0.0f;
        v = v * e;
        v += DPP_F(v, 0xB1); v += DPP_F(v, 0x4E); v += DPP_F(v, 0x141);
        float L = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
        if (nsplit > 8u) {
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 8));
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 24));
        }
        if (nsplit > 32u) {
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 40));
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
            L += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 56));
        }
        wsh[s] = e / L;
    }
    __syncthreads();
}
// slots the register version of this arithmetic carried (8 / 32 / 64): a launch with fewer splits added `0 * w` for the rest
__device__ __forceinline__ uint32_t combine_slots(uint32_t nsplit) { return nsplit <= 8u ? 8u : nsplit <= 32u ? 32u : 64u; }

constexpr int NP = 2;            // timestep blocks a workgroup keeps in flight per round

// A KV row (head_dim floats) is shared by LPR lanes, QV float4 each (lane j owns float4 j, j+LPR, ...: every load
// instruction reads LPR*16 contiguous bytes per row); few lanes per row keep the per-row cross-lane reduction and the
// per-timestep exp() cheap.  KVM = q heads per workgroup.
// MODE resolves the feature flags at compile time for the decode launches (a taken branch costs ~40 cycles and every
// instruction of the one wave per SIMD is on the critical path): 0 generic (run-time flags), 1 Qwen3 decode (q/k-norm,
// half-split RoPE, staged RoPE row, fresh k, causal), 2 Nano/Qwen2 decode (adjacent-pair RoPE, staged row, fresh k, causal).
// PG: the paged KV cache (kernels.h AttnArgs::pt_rows) -- a template parameter so that the contiguous cache's code stays as it was
// NPT: timestep blocks a workgroup keeps in flight per round (NPT = 2; 4 for the launches that would otherwise walk several rounds:
// twice the rows requested at kernel entry, half the dependent round trips)
// W16 (FP16 rows, QV a multiple of 4, head_dim % 8 == 0): a lane's loads stay 16 bytes wide -- EIGHT halfs, i.e. the float4 slots
// 2c and 2c + 1 of chunk c = j + LPR * (q / 2) -- so a row costs half the load instructions and every instruction still asks for
// whole 128-byte lines (round 3's 8-byte loads: as many requests as FP32 rows for half the bytes, and not a microsecond saved).
// The RoPE partner of slot q stays slot q + QV / 2 of the same lane (f + 2 LPR = half a head further for QV = 4).
// FUSE (qkv_attn_fused_kernel, gemv_q80_impl.h): the workgroup runs INSIDE the launch that computes q | k | v -- its K / V rows are
// asked for at entry as always, q, the raw k row and the fresh v row then arrive as 8-byte {value, tag} granules that the projection's
// workgroups store write-through (hh.buf: q at 0, k at hh.base[1], v at hh.base[2]; tag = hand_tag, the epoch of this step and layer --
// device_common.h): one sweep per wave until every tag matches, the values go through LDS into the registers the loads used to fill.
// A workgroup that gives up (bounded wait) reports it and stores NOTHING: no k row, no output (round-5 advice).
template <int LPR, int QV, int KVM, int MODE, bool KVH, bool PG, int NPT, bool W16, bool FUSE>
__device__ __forceinline__ void attention_body(const AttnArgs &a, unsigned char *smem, const uint32_t grp, const uint32_t b, const uint32_t split,
                                               const SlabHand &hh, const uint32_t hand_tag = 0u, const uint32_t hand_wait16 = 0u) {
    static_assert(!W16 || (KVH && QV % 4 == 0), "16-byte FP16 loads: two float4 slots per load");
    static_assert(!FUSE || (((MODE == 1 && LPR * QV * 4 == 128) || MODE == 2) && KVM == 1 && !KVH && !PG && LPR * QV * 4 <= 128),
                  "the fused launch: Qwen3 decode at head_dim 128, or the plain decode mode (no q/k norm, adjacent-pair RoPE: Nano) at head_dim <= 128; FP32 contiguous cache");
    constexpr int R = 256 / LPR;                 // timesteps per block
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nsplit = a.nsplit;
    const uint32_t hd = (MODE == 1) ? (uint32_t)(QV * LPR * 4) : a.hd, half = hd >> 1, hd4 = (hd + 3) & ~3u;    // Qwen3 decode mode: the head fills the sub-group exactly (launch_lpr)
    // first q head of this workgroup and its KV head.  XCD-aware order (kv_log2 valid): x = sub * n_kv_head + KV head, so the
    // kv_mul / KVM workgroups that read the same K/V rows have equal x mod 8 = the same XCD = one L2 fetch of every row.
    // The decode modes (MODE 1 / 2) always run in that order with power-of-two head counts (launch_lpr() checks): shifts and
    // masks only -- the three integer divisions this used to take cost ~60 instructions before the first load was issued.
    uint32_t kv_mul, g, h0;
    bool first_of_group;                                       // this workgroup writes the KV head's fresh k (v) row
    if constexpr (MODE != 0) {
        kv_mul = 1u << a.kvmul_log2;
        g = grp & ((1u << a.kv_log2) - 1u);
        const uint32_t sub_wg = grp >> a.kv_log2;
        h0 = (g << a.kvmul_log2) + sub_wg * KVM;
        first_of_group = sub_wg == 0;
    } else {
        kv_mul = a.n_head / a.n_kv_head;
        const bool xcd = a.kv_log2 != 0xffffffffu;
        g = xcd ? (grp & ((1u << a.kv_log2) - 1u)) : (grp * KVM) / kv_mul;
        h0 = xcd ? g * kv_mul + (grp >> a.kv_log2) * KVM : grp * KVM;
        first_of_group = (h0 % kv_mul) == 0;
    }
    constexpr bool G = MODE == 0;
    constexpr int VR = (KVM + 1 + 3) / 4;                      // rounds of vectors per wave (q heads + the k row over 4 waves)
    constexpr int JJ = (LPR == 16) ? 2 : 1;                    // RoPE pairs per lane (head_dim > 128 needs two)
    const bool fresh_k = G ? a.kraw != nullptr : true;
    const bool has_norm = G ? a.q_norm != nullptr : MODE == 1;
    const bool rq3 = G ? a.rope_qwen3 != 0 : MODE == 1;
    const bool has_rope = G ? a.rope_cos != nullptr : true;
    const uint32_t fixed_range = G ? a.fixed_range : 0u;
    const bool causal = G ? a.is_causal != 0 : true;
    float *const q_out = G ? a.q_out : nullptr;

    // LDS: qh[KVM][hd4] kh[hd4] redm[KVM][4] redl[KVM][4] part[4 waves][KVM][hd4]
    float *qh = reinterpret_cast<float *>(smem);
    float *kh = qh + KVM * hd4;
    float *redm = kh + hd4;
    float *redl = redm + KVM * 4;
    float *part = redl + KVM * 4;

    NANO_STAMP(a.stamps, 0, tid);
    // Every kernel argument the prologue needs, fetched in ONE scalar round trip: left alone the compiler fetches the block in
    // two or three dependent batches (each a ~0.2 us scalar-cache miss on the fresh kernarg segment) between the first loads.
    if constexpr (MODE != 0)
        asm volatile("" :: "s"(a.q), "s"(a.kraw), "s"(a.kcache), "s"(a.vcache), "s"(a.pos), "s"(a.q_norm), "s"(a.k_norm), "s"(a.rope_cur),
                     "s"(a.out), "s"(a.ml), "s"(a.xba_out), "s"(a.nsplit), "s"(a.range_hint), "s"(a.layer), "s"(a.S), "s"(a.q_dim), "s"(a.kv_dim),
                     "s"(a.cache_bstride_rows), "s"(a.kv_log2), "s"(a.kvmul_log2), "s"(a.vraw), "s"(a.xf_out), "s"(a.pt_rows), "s"(a.kvrow));
    // the position first: it is waited for before everything else, and vector loads return in issue order
    const uint32_t pos_ld = (MODE == 0 && a.fixed_range) ? 0u : a.pos[b];
    constexpr bool paged = PG;                                 // paged KV cache: rows are reached through the sequence's page table
    uint32_t prow_ld = 0u;                                     // pool row of position pos
    if constexpr (PG) prow_ld = a.kvrow[b];
    // ---- 1. issue every load --------------------------------------------------------------------------------
    constexpr uint32_t ESZ = KVH ? 2u : 4u;                    // bytes per cache element
    const size_t slot_rows = paged ? (size_t)a.layer * a.pool_rows : (size_t)b * a.cache_bstride_rows + (size_t)a.layer * a.S;
    // start of this KV head's rows, in the cache's own element size (typed float* for the FP32 path's arithmetic)
    const float *kc = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.kcache) + (slot_rows * a.kv_dim + (size_t)g * hd) * ESZ);
    const float *vc = reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.vcache) + (slot_rows * a.kv_dim + (size_t)g * hd) * ESZ);
    const uint32_t cache_bytes = (paged ? a.pool_rows : fixed_range ? fixed_range : a.S) * a.kv_dim * ESZ;   // rows >= S (paged: beyond the layer plane): out of range
    const __amdgpu_buffer_rsrc_t rk = mkrsrc(kc, cache_bytes - g * hd * ESZ);
    const __amdgpu_buffer_rsrc_t rv = mkrsrc(vc, cache_bytes - g * hd * ESZ);
    const uint32_t sub = (uint32_t)tid / LPR, j = (uint32_t)tid % LPR;
    auto fidx = [&](int q) -> uint32_t { return W16 ? 2u * (j + (uint32_t)LPR * (uint32_t)(q >> 1)) + (uint32_t)(q & 1) : j + (uint32_t)LPR * (uint32_t)q; };   // float4 slot q of this lane
    const uint32_t range_hint = fixed_range ? fixed_range : a.range_hint;

    // ---- q heads and the raw k row ------------------------------------------------------------------------------------
    // Decode modes keep them in REGISTERS: the LPR lanes of a sub-group own the whole head (float4 j, j+LPR, ...), the
    // RoPE partner of float4 f (f + head_dim/8 for the half-split style, the same float4 for adjacent pairs) lives in
    // the same lane, so rmsnorm + RoPE need one DPP reduction and no LDS / barrier; every sub-group does it redundantly.
    constexpr bool REGQK = MODE != 0;
    // SHARE (round 6; decode modes with several q heads per workgroup = the batched steps): the KVM q heads and the k row are loaded,
    // normalised and rotated ONCE per workgroup -- sub-group v of wave 0 takes vector v (v < KVM: q head h0 + v; v == KVM: the k row) with
    // exactly the per-lane arithmetic below (same slots per lane, same DPP tree: same bits), parks it in LDS, and every lane reads the
    // finished vectors back.  Round 5 had all 32 sub-groups of a workgroup do all KVM + 1 vectors: at KVM = 4 that was 36 of a lane's 52
    // load instructions and ~500 of a wave's ~2300 VALU instructions in a launch that is VALU bound at 64 sequences (SQ_ACTIVE_INST_VALU
    // 64 % of the SIMD cycles, profiles/r06_4b_b64_pmc_before.txt).
    constexpr bool SHARE = REGQK && !FUSE && KVM > 1;
    float4 qv[KVM][QV], kfresh[QV], vfresh[QV];
    float4 sv[SHARE ? QV : 1], snw[SHARE ? QV : 1];            // SHARE: this sub-group's vector and its norm weight
    float4 qnw[QV], knw[QV], rcs[QV], rsn[QV];
    const __amdgpu_buffer_rsrc_t rq = mkrsrc(a.q + (size_t)b * a.q_dim, a.q_dim * 4u);
    const __amdgpu_buffer_rsrc_t rkr = mkrsrc(fresh_k ? a.kraw + (size_t)b * a.kv_dim : nullptr, fresh_k ? a.kv_dim * 4u : 0u);
    // FP16 cache: the QKV GEMV leaves the fresh v row in scratch (FP32) and this kernel rounds and stores it next to k
    const bool fresh_v = KVH && a.vraw != nullptr;
    const __amdgpu_buffer_rsrc_t rvr = mkrsrc(fresh_v ? a.vraw + (size_t)b * a.kv_dim : nullptr, fresh_v ? a.kv_dim * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rqn = mkrsrc(a.q_norm, has_norm ? hd * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rkn = mkrsrc(a.k_norm, has_norm ? hd * 4u : 0u);
    float e0[VR][JJ], e1[VR][JJ], nw0[VR][JJ], nw1[VR][JJ];     // generic path: [vector round][jj]
    float rc[JJ], rs[JJ];
    const bool rope_staged = G ? (a.rope_cur != nullptr && fresh_k) : true;
    if constexpr (SHARE) {
        if (wid == 0) {                                         // (wave-uniform: the other waves ask for none of this)
            const __amdgpu_buffer_rsrc_t rr = mkrsrc(a.rope_cur + (size_t)b * 2 * half, 2 * half * 4u);
            const bool isq = sub < (uint32_t)KVM, isk = sub == (uint32_t)KVM;
#pragma unroll
            for (int q = 0; q < QV; q++) {
                const uint32_t f = fidx(q);
                const uint32_t fo = (f * 4u < hd) ? f * 16u : OOB;
                const float4 xq = bload_f4(rq, (isq && fo != OOB) ? (h0 + sub) * hd * 4u + fo : OOB), xk = bload_f4(rkr, (isk && fo != OOB) ? g * hd * 4u + fo : OOB);
                sv[q] = isq ? xq : xk;
                if (MODE == 1) {
                    const float4 nq = bload_f4(rqn, isq ? fo : OOB), nk = bload_f4(rkn, isk ? fo : OOB);
                    snw[q] = isq ? nq : nk;
                    const uint32_t fr = f & (uint32_t)(QV * LPR / 2 - 1);
                    rcs[q] = bload_f4(rr, fo == OOB ? OOB : fr * 16u);
                    rsn[q] = bload_f4(rr, fo == OOB ? OOB : (half + fr * 4u) * 4u);
                } else {
                    const float c0 = bload_f(rr, fo == OOB ? OOB : (2u * f) * 4u), c1 = bload_f(rr, fo == OOB ? OOB : (2u * f + 1u) * 4u);
                    const float s0 = bload_f(rr, fo == OOB ? OOB : (half + 2u * f) * 4u), s1 = bload_f(rr, fo == OOB ? OOB : (half + 2u * f + 1u) * 4u);
                    rcs[q] = make_float4(c0, c1, 0.f, 0.f); rsn[q] = make_float4(s0, s1, 0.f, 0.f);
                }
            }
        }
        if constexpr (KVH) {                                    // FP16 cache: every lane keeps the fresh v row (it is not in the cache yet)
#pragma unroll
            for (int q = 0; q < QV; q++) { const uint32_t f = fidx(q); vfresh[q] = bload_f4(rvr, (f * 4u < hd) ? g * hd * 4u + f * 16u : OOB); }
        }
    } else if constexpr (REGQK) {
        const __amdgpu_buffer_rsrc_t rr = mkrsrc(a.rope_cur + (size_t)b * 2 * half, 2 * half * 4u);
#pragma unroll
        for (int q = 0; q < QV; q++) {
            const uint32_t f = fidx(q);
            const uint32_t fo = (f * 4u < hd) ? f * 16u : OOB;
            if constexpr (!FUSE) {
#pragma unroll
                for (int m = 0; m < KVM; m++) qv[m][q] = bload_f4(rq, fo == OOB ? OOB : (h0 + m) * hd * 4u + fo);
                kfresh[q] = bload_f4(rkr, fo == OOB ? OOB : g * hd * 4u + fo);
                if (KVH) vfresh[q] = bload_f4(rvr, fo == OOB ? OOB : g * hd * 4u + fo);
            }
            if (MODE == 1) {
                qnw[q] = bload_f4(rqn, fo); knw[q] = bload_f4(rkn, fo);
                const uint32_t fr = f & (uint32_t)(QV * LPR / 2 - 1);    // = f % (half / 4): cos/sin of element i and i+half are those of pair i
                rcs[q] = bload_f4(rr, fo == OOB ? OOB : fr * 16u);
                rsn[q] = bload_f4(rr, fo == OOB ? OOB : (half + fr * 4u) * 4u);
            } else {                                                  // adjacent pairs (2p, 2p+1), p = 2f, 2f+1: .xy = cos, .zw = sin
                const float c0 = bload_f(rr, fo == OOB ? OOB : (2u * f) * 4u), c1 = bload_f(rr, fo == OOB ? OOB : (2u * f + 1u) * 4u);
                const float s0 = bload_f(rr, fo == OOB ? OOB : (half + 2u * f) * 4u), s1 = bload_f(rr, fo == OOB ? OOB : (half + 2u * f + 1u) * 4u);
                rcs[q] = make_float4(c0, c1, 0.f, 0.f); rsn[q] = make_float4(s0, s1, 0.f, 0.f);
            }
        }
    } else {
    // the vectors this wave normalises / rotates: v = wid, wid+4, ... ; v < KVM: q head h0+v ; v == KVM: the k row
    // a lane holds the pair(s) RoPE rotates together: Qwen3 (i, i+half), Nano/Qwen2 (2i, 2i+1); pair index pi = lane + 64*jj
#pragma unroll
    for (int vr = 0; vr < VR; vr++) {
        const uint32_t v = (uint32_t)wid + 4u * vr;
        const bool isq = v < (uint32_t)KVM, isk = v == (uint32_t)KVM;
#pragma unroll
        for (int jj = 0; jj < JJ; jj++) {
            const uint32_t pi = (uint32_t)lane + 64u * jj;
            const uint32_t i0 = rq3 ? pi : 2 * pi, i1 = rq3 ? pi + half : 2 * pi + 1;
            const bool ok = pi < half;
            const uint32_t o0 = ok ? i0 * 4u : OOB, o1 = ok ? i1 * 4u : OOB;
            const uint32_t qb = (h0 + v) * hd * 4u, kb = g * hd * 4u;
            e0[vr][jj] = isq ? bload_f(rq, o0 + (ok ? qb : 0u)) : isk ? bload_f(rkr, o0 + (ok ? kb : 0u)) : 0.0f;
            e1[vr][jj] = isq ? bload_f(rq, o1 + (ok ? qb : 0u)) : isk ? bload_f(rkr, o1 + (ok ? kb : 0u)) : 0.0f;
            nw0[vr][jj] = !has_norm ? 1.0f : isq ? bload_f(rqn, o0) : isk ? bload_f(rkn, o0) : 0.0f;
            nw1[vr][jj] = !has_norm ? 1.0f : isq ? bload_f(rqn, o1) : isk ? bload_f(rkn, o1) : 0.0f;
        }
    }
    // RoPE row of pos[b]: staged at a fixed address by the step's first kernel (no pos-dependent load)
    {
        const __amdgpu_buffer_rsrc_t rr = mkrsrc(rope_staged ? a.rope_cur + (size_t)b * 2 * half : nullptr, rope_staged ? 2 * half * 4u : 0u);
#pragma unroll
        for (int jj = 0; jj < JJ; jj++) {
            const uint32_t pi = (uint32_t)lane + 64u * jj;
            rc[jj] = bload_f(rr, pi < half ? pi * 4u : OOB);
            rs[jj] = bload_f(rr, pi < half ? (half + pi) * 4u : OOB);
        }
    }
    }
    KVRaw<KVH> kreg[NPT][QV], vreg[NPT][QV];
    auto issue_kv = [&](uint32_t round) {
#pragma unroll
        for (int p = 0; p < NPT; p++) {
            const uint32_t tb = ((round * NPT + p) * nsplit + split) * R, t = tb + sub;
            uint32_t row = t;                                              // row of timestep t inside this sequence's / the pool's layer plane
            if constexpr (PG) {                                            // (a block of R <= 64 timesteps lies in one page)
                const uint32_t blk = tb >> 6;
                uint32_t rb = 0xffffffffu;
                if (blk < a.pt_stride && tb < range_hint) {                // wave-uniform: a SCALAR load (a vector load here would have to wait for
                    const uint64_t pa = reinterpret_cast<uint64_t>(a.pt_rows + (size_t)b * a.pt_bstride + blk);   // every K / V load issued before it -- vmcnt counts in order)
                    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa);   // (readfirstlane returns int: no sign extension into the high word)
                    asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rb) : "s"(pu) : "memory");
                }
                row = rb == 0xffffffffu ? 0x7fffffu : rb + (t & 63u);      // no page: beyond every plane -> out of range -> 0
            }
#pragma unroll
            for (int q = 0; q < QV; q++) {
                const uint32_t f = fidx(q);                                // float4 index inside the head
                const uint32_t off = (f * 4u < hd && t < range_hint && row < 0x7fffffu) ? (row * a.kv_dim + f * 4u) * ESZ : OOB;
                if constexpr (W16) {
                    if ((q & 1) == 0) {                                        // slots q, q + 1: one 16-byte load
                        const i32x4 kk = __builtin_amdgcn_raw_buffer_load_b128(rk, (int)off, 0, 0), vv = __builtin_amdgcn_raw_buffer_load_b128(rv, (int)off, 0, 0);
                        kreg[p][q].v = make_uint2((uint32_t)kk.x, (uint32_t)kk.y); kreg[p][q + 1].v = make_uint2((uint32_t)kk.z, (uint32_t)kk.w);
                        vreg[p][q].v = make_uint2((uint32_t)vv.x, (uint32_t)vv.y); vreg[p][q + 1].v = make_uint2((uint32_t)vv.z, (uint32_t)vv.w);
                    }
                } else {
                    kreg[p][q] = kv_load4<KVH>(rk, off);
                    vreg[p][q] = kv_load4<KVH>(rv, off);
                }
            }
        }
    };
    issue_kv(0);
    __builtin_amdgcn_sched_barrier(0);                         // the K / V loads go out BEFORE anything waits for q (the scheduler otherwise sinks them below the q wait)
    NANO_STAMP(a.stamps, 1, tid);                              // every load issued
    if constexpr (FUSE) {
        // q (this head: 128 granules, threads 0..127), the raw k row (128, threads 128..255) and the fresh v row (128, threads 0..127 again)
        // of this KV group.  Relaxed agent-scope 8-byte loads (sc1: past the L1) of granules stored the same way: the data is the flag.
        float *vh = part + 4 * KVM * hd4;                       // (an LDS row of its own behind the partials)
        const uint32_t t7 = (uint32_t)tid & 127u;
        const unsigned long long *hand = hh.buf;
        const unsigned long long *g0p = (uint32_t)tid < 128u ? hand + (size_t)h0 * hd + t7 : hand + hh.base[1] + (size_t)g * hd + t7;
        const unsigned long long *g1p = hand + hh.base[2] + (size_t)g * hd + t7;
        const unsigned long long tag_done = (unsigned long long)hand_tag << 32;
        // TWO sweeps in flight (A, B): a sweep is a memory round trip (~1 us), the next one is on its way while this one is looked at
        unsigned long long g0 = 0, g1 = 0;
        // (MODE 2, round 6: head_dim < 128 -- the threads beyond the head have nothing to wait for)
        const bool live7 = MODE == 1 || t7 < hd;
        const bool two = (uint32_t)tid < 128u && live7;
        auto sweep = [&](unsigned long long &x0, unsigned long long &x1) {
            x0 = live7 ? __hip_atomic_load(g0p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag_done;
            x1 = two ? __hip_atomic_load(g1p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag_done;
        };
        auto ready = [&](unsigned long long x0, unsigned long long x1) { return __all((uint32_t)(x0 >> 32) == hand_tag && (uint32_t)(x1 >> 32) == hand_tag) != 0; };
        unsigned long long a0, a1, b0, b1;
        for (uint32_t w_ = 0; w_ < hand_wait16; w_++) __builtin_amdgcn_s_sleep(16);      // (the projection needs ~3 us: polls before that only compete with it)
        sweep(a0, a1);
        bool got = false;
        for (uint32_t spin = 0; spin < (1u << 13); spin++) {   // (bounded, ~10 ms: a projection that never arrives must not hang the device)
            sweep(b0, b1);
            if (ready(a0, a1)) { g0 = a0; g1 = a1; got = true; break; }
            sweep(a0, a1);
            if (ready(b0, b1)) { g0 = b0; g1 = b1; got = true; break; }
            if ((spin & 63u) == 63u && hand_aborted(hh)) break;   // (somebody in this step already gave up: the call is lost)
        }
        if (live7) {
            if ((uint32_t)tid < 128u) { qh[t7] = __uint_as_float((uint32_t)g0); vh[t7] = __uint_as_float((uint32_t)g1); }
            else kh[t7] = __uint_as_float((uint32_t)g0);
        }
        if (__syncthreads_or(got ? 0 : 1)) {                   // (the staging barrier; a wave that gave up takes the whole workgroup out)
            if (!got && lane == 0) hand_give_up(hh, a.err);
            return;
        }
#pragma unroll
        for (int q = 0; q < QV; q++) {
            const uint32_t f = fidx(q);
            const bool in = MODE == 1 || f * 4u < hd;           // (slots beyond the head: zeros, as the loads of the unfused launch return)
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            qv[0][q] = in ? *reinterpret_cast<const float4 *>(qh + 4u * f) : z4;
            kfresh[q] = in ? *reinterpret_cast<const float4 *>(kh + 4u * f) : z4;
            vfresh[q] = in ? *reinterpret_cast<const float4 *>(vh + 4u * f) : z4;
        }
    }

    // ---- 2. position, RoPE row, norms ------------------------------------------------------------------------------
    const uint32_t pos = fixed_range ? (fixed_range - 1) : pos_ld;
    const uint32_t prow = paged ? prow_ld : pos;               // the cache row position pos is written to
    const uint32_t range = fixed_range ? fixed_range : (causal ? (pos + 1) : a.S);
    if constexpr (SHARE) {
        if (wid == 0) {
            const bool isq = sub < (uint32_t)KVM, isk = sub == (uint32_t)KVM;
            if (MODE == 1) {                     // rmsnorm over the head (infer.c:601-614, 824-835), tree order: the per-lane sums and the DPP tree of the code below
                float ssq = 0.0f;
#pragma unroll
                for (int q = 0; q < QV; q++) { ssq += sv[q].x * sv[q].x; ssq += sv[q].y * sv[q].y; ssq += sv[q].z * sv[q].z; ssq += sv[q].w * sv[q].w; }
                ssq = group_sum_t<LPR>(ssq); ssq /= (float)hd; ssq += 1e-5f; ssq = 1.0f / sqrtf(ssq);
#pragma unroll
                for (int q = 0; q < QV; q++) {
                    sv[q].x = snw[q].x * (ssq * sv[q].x); sv[q].y = snw[q].y * (ssq * sv[q].y); sv[q].z = snw[q].z * (ssq * sv[q].z); sv[q].w = snw[q].w * (ssq * sv[q].w);
                }
#pragma unroll
                for (int q = 0; q < QV / 2; q++) {   // half-split RoPE (rope_qwen3, infer.c:692-706)
                    const float4 l = sv[q], h = sv[q + QV / 2], c = rcs[q], sn = rsn[q];
                    sv[q].x = l.x * c.x - h.x * sn.x; sv[q + QV / 2].x = h.x * c.x + l.x * sn.x;
                    sv[q].y = l.y * c.y - h.y * sn.y; sv[q + QV / 2].y = h.y * c.y + l.y * sn.y;
                    sv[q].z = l.z * c.z - h.z * sn.z; sv[q + QV / 2].z = h.z * c.z + l.z * sn.z;
                    sv[q].w = l.w * c.w - h.w * sn.w; sv[q + QV / 2].w = h.w * c.w + l.w * sn.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < QV; q++) {       // adjacent-pair RoPE (rope, infer.c:681-690)
                    const float4 t = sv[q], c = rcs[q], sn = rsn[q];
                    sv[q].x = t.x * c.x - t.y * sn.x; sv[q].y = t.x * sn.x + t.y * c.x;
                    sv[q].z = t.z * c.y - t.w * sn.y; sv[q].w = t.z * sn.y + t.w * c.y;
                }
            }
            if (isk) {
#pragma unroll
                for (int q = 0; q < QV; q++) sv[q] = kv_round<KVH>(sv[q]);                          // what the cache holds
            }
            if (isq || isk) {
                float *dst = isq ? qh + sub * hd4 : kh;
#pragma unroll
                for (int q = 0; q < QV; q++) { const uint32_t f = fidx(q); if (f * 4u < hd) *reinterpret_cast<float4 *>(dst + 4u * f) = sv[q]; }
            }
            if (isk && split == 0 && first_of_group) {            // the finished k row (FP16 cache: and the v row) -> cache row pos
                float *krow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(kc)) + (size_t)prow * a.kv_dim * ESZ);
                float *vrow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(vc)) + (size_t)prow * a.kv_dim * ESZ);
#pragma unroll
                for (int q = 0; q < QV; q++) {
                    const uint32_t f = fidx(q);
                    if (f * 4u < hd) { kv_store4<KVH>(krow, 4u * f, sv[q]); if (KVH && fresh_v) kv_store4<KVH>(vrow, 4u * f, kv_round<KVH>(vfresh[q])); }
                }
            }
        }
        if (a.prep_only) return;                                  // batched prefill, pass 1: the k row is all that was wanted
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QV; q++) {
            const uint32_t f = fidx(q);
            const bool ok = f * 4u < hd;
#pragma unroll
            for (int m = 0; m < KVM; m++) qv[m][q] = ok ? *reinterpret_cast<const float4 *>(qh + m * hd4 + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
            kfresh[q] = ok ? *reinterpret_cast<const float4 *>(kh + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (KVH) vfresh[q] = kv_round<KVH>(vfresh[q]);
        }
    } else if constexpr (REGQK) {
        if (MODE == 1) {                         // rmsnorm over the head (infer.c:601-614, 824-835), tree order
            float sk = 0.0f, sq[KVM];
#pragma unroll
            for (int m = 0; m < KVM; m++) sq[m] = 0.0f;
#pragma unroll
            for (int q = 0; q < QV; q++) {
                sk += kfresh[q].x * kfresh[q].x; sk += kfresh[q].y * kfresh[q].y; sk += kfresh[q].z * kfresh[q].z; sk += kfresh[q].w * kfresh[q].w;
#pragma unroll
                for (int m = 0; m < KVM; m++) { sq[m] += qv[m][q].x * qv[m][q].x; sq[m] += qv[m][q].y * qv[m][q].y; sq[m] += qv[m][q].z * qv[m][q].z; sq[m] += qv[m][q].w * qv[m][q].w; }
            }
            sk = group_sum_t<LPR>(sk); sk /= (float)hd; sk += 1e-5f; sk = 1.0f / sqrtf(sk);
#pragma unroll
            for (int m = 0; m < KVM; m++) { sq[m] = group_sum_t<LPR>(sq[m]); sq[m] /= (float)hd; sq[m] += 1e-5f; sq[m] = 1.0f / sqrtf(sq[m]); }
#pragma unroll
            for (int q = 0; q < QV; q++) {
                kfresh[q].x = knw[q].x * (sk * kfresh[q].x); kfresh[q].y = knw[q].y * (sk * kfresh[q].y); kfresh[q].z = knw[q].z * (sk * kfresh[q].z); kfresh[q].w = knw[q].w * (sk * kfresh[q].w);
#pragma unroll
                for (int m = 0; m < KVM; m++) {
                    qv[m][q].x = qnw[q].x * (sq[m] * qv[m][q].x); qv[m][q].y = qnw[q].y * (sq[m] * qv[m][q].y);
                    qv[m][q].z = qnw[q].z * (sq[m] * qv[m][q].z); qv[m][q].w = qnw[q].w * (sq[m] * qv[m][q].w);
                }
            }
            // half-split RoPE (rope_qwen3, infer.c:692-706): float4 q pairs with float4 q + QV/2
            auto rot = [&](float4 &lo, float4 &hi, const float4 &c, const float4 &sn) {
                const float4 l = lo, h = hi;
                lo.x = l.x * c.x - h.x * sn.x; hi.x = h.x * c.x + l.x * sn.x;
                lo.y = l.y * c.y - h.y * sn.y; hi.y = h.y * c.y + l.y * sn.y;
                lo.z = l.z * c.z - h.z * sn.z; hi.z = h.z * c.z + l.z * sn.z;
                lo.w = l.w * c.w - h.w * sn.w; hi.w = h.w * c.w + l.w * sn.w;
            };
#pragma unroll
            for (int q = 0; q < QV / 2; q++) {
                rot(kfresh[q], kfresh[q + QV / 2], rcs[q], rsn[q]);
#pragma unroll
                for (int m = 0; m < KVM; m++) rot(qv[m][q], qv[m][q + QV / 2], rcs[q], rsn[q]);
            }
        } else {
            // adjacent-pair RoPE (rope, infer.c:681-690): (x,y) with (cos0,sin0), (z,w) with (cos1,sin1)
            auto rot = [&](float4 &v, const float4 &c, const float4 &sn) {
                const float4 t = v;
                v.x = t.x * c.x - t.y * sn.x; v.y = t.x * sn.x + t.y * c.x;
                v.z = t.z * c.y - t.w * sn.y; v.w = t.z * sn.y + t.w * c.y;
            };
#pragma unroll
            for (int q = 0; q < QV; q++) {
                rot(kfresh[q], rcs[q], rsn[q]);
#pragma unroll
                for (int m = 0; m < KVM; m++) rot(qv[m][q], rcs[q], rsn[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < QV; q++) { kfresh[q] = kv_round<KVH>(kfresh[q]); if (KVH) vfresh[q] = kv_round<KVH>(vfresh[q]); }   // what the cache holds
        if (split == 0 && sub == 0 && first_of_group) {          // the finished k row (FP16 cache: and the v row) -> cache row pos
            float *krow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(kc)) + (size_t)prow * a.kv_dim * ESZ);
            float *vrow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(vc)) + (size_t)prow * a.kv_dim * ESZ);
#pragma unroll
            for (int q = 0; q < QV; q++) {
                const uint32_t f = fidx(q);
                if (f * 4u < hd) { kv_store4<KVH>(krow, 4u * f, kfresh[q]); if (KVH && fresh_v) kv_store4<KVH>(vrow, 4u * f, vfresh[q]); }
            }
        }
        if (a.prep_only) return;                                  // batched prefill, pass 1: the k row is all that was wanted
    } else {
    if (!rope_staged) {
#pragma unroll
        for (int jj = 0; jj < JJ; jj++) {
            const uint32_t pi = (uint32_t)lane + 64u * jj;
            const bool ok = has_rope && pi < half && fresh_k;
            rc[jj] = ok ? a.rope_cos[(size_t)pos * half + pi] : 1.0f;
            rs[jj] = ok ? a.rope_sin[(size_t)pos * half + pi] : 0.0f;
        }
    }
#pragma unroll
    for (int vr = 0; vr < VR; vr++) {
        const uint32_t v = (uint32_t)wid + 4u * vr;
        if (v <= (uint32_t)KVM) {                                   // wave-uniform
            const bool isk = v == (uint32_t)KVM;
            float x0[JJ], x1[JJ];
#pragma unroll
            for (int jj = 0; jj < JJ; jj++) { x0[jj] = e0[vr][jj]; x1[jj] = e1[vr][jj]; }
            if (fresh_k && has_norm) {                               // rmsnorm over the head (infer.c:601-614), tree order
                float acc = 0.0f;
#pragma unroll
                for (int jj = 0; jj < JJ; jj++) { acc += x0[jj] * x0[jj]; acc += x1[jj] * x1[jj]; }
                float ss = wave_sum_dpp(acc);
                ss /= (float)hd; ss += 1e-5f; ss = 1.0f / sqrtf(ss);
#pragma unroll
                for (int jj = 0; jj < JJ; jj++) { x0[jj] = nw0[vr][jj] * (ss * x0[jj]); x1[jj] = nw1[vr][jj] * (ss * x1[jj]); }
            }
            float *dst = isk ? kh : qh + v * hd4;
#pragma unroll
            for (int jj = 0; jj < JJ; jj++) {
                const uint32_t pi = (uint32_t)lane + 64u * jj;
                if (pi < half && (!isk || fresh_k)) {
                    const uint32_t i0 = rq3 ? pi : 2 * pi, i1 = rq3 ? pi + half : 2 * pi + 1;
                    float y0 = x0[jj], y1 = x1[jj];
                    if (fresh_k && has_rope) {
                        const float c = rc[jj], s = rs[jj];
                        if (rq3) { y0 = x0[jj] * c - x1[jj] * s; y1 = x1[jj] * c + x0[jj] * s; }     // infer.c:700-703
                        else { y0 = x0[jj] * c - x1[jj] * s; y1 = x0[jj] * s + x1[jj] * c; }                 // infer.c:686-687
                    }
                    if (isk) { y0 = kv_round1<KVH>(y0); y1 = kv_round1<KVH>(y1); }
                    dst[i0] = y0; dst[i1] = y1;
                    if (isk && split == 0 && first_of_group) {
                        float *krow = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(kc)) + (size_t)prow * a.kv_dim * ESZ);
                        kv_store1<KVH>(krow, i0, y0); kv_store1<KVH>(krow, i1, y1);
                    }
                    if (!isk && split == 0 && q_out) { float *qo = q_out + (size_t)b * a.q_dim + (size_t)(h0 + v) * hd; qo[i0] = y0; qo[i1] = y1; }
                }
            }
        }
    }
    if constexpr (KVH) {                                          // generic path: the fresh v row straight from scratch, rounded, stored by split 0
#pragma unroll
        for (int q = 0; q < QV; q++) {
            const uint32_t f = fidx(q);
            const bool ok = f * 4u < hd;
            vfresh[q] = kv_round<KVH>(bload_f4(rvr, ok ? g * hd * 4u + f * 16u : OOB));
            if (fresh_v && ok && split == 0 && sub == 0 && first_of_group)
                kv_store4<KVH>(reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(const_cast<float *>(vc)) + (size_t)prow * a.kv_dim * ESZ), 4u * f, vfresh[q]);
        }
    }
    __syncthreads();
    if (a.prep_only) return;                                      // batched prefill, pass 1 (generic path)

    // ---- 3. scores, running softmax over rounds ------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < QV; q++) {
        const uint32_t f = fidx(q);
        const bool ok = f * 4u < hd;
#pragma unroll
        for (int m = 0; m < KVM; m++) qv[m][q] = ok ? *reinterpret_cast<const float4 *>(qh + m * hd4 + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
        kfresh[q] = (ok && fresh_k) ? *reinterpret_cast<const float4 *>(kh + 4 * f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    }
    NANO_STAMP(a.stamps, 2, qv[0][0].x);                       // q / k arrived, normalised and rotated
    const float sq_hd = sqrtf((float)hd);
    float mrun[KVM], lrun[KVM];
    float4 acc[KVM][QV];
#pragma unroll
    for (int m = 0; m < KVM; m++) {
        mrun[m] = -INFINITY; lrun[m] = 0.0f;
#pragma unroll
        for (int q = 0; q < QV; q++) acc[m][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint32_t per_round = NPT * nsplit * R;
    const uint32_t limit = range < range_hint ? range : range_hint;
    const uint32_t nround = (limit + per_round - 1) / per_round;
    for (uint32_t round = 0; round < nround; round++) {
        if (round) issue_kv(round);
        float sc[KVM][NPT];
#pragma unroll
        for (int p = 0; p < NPT; p++) {
            const uint32_t t = ((round * NPT + p) * nsplit + split) * R + sub;
            const bool fresh = fresh_k && t == pos;
#pragma unroll
            for (int m = 0; m < KVM; m++) {
                float d = 0.0f;
#pragma unroll
                for (int q = 0; q < QV; q++) {
                    const float4 kk = fresh ? kfresh[q] : kv_cvt(kreg[p][q]);
                    d = __builtin_fmaf(qv[m][q].x, kk.x, d); d = __builtin_fmaf(qv[m][q].y, kk.y, d); d = __builtin_fmaf(qv[m][q].z, kk.z, d); d = __builtin_fmaf(qv[m][q].w, kk.w, d);
                }
                d = group_sum_t<LPR>(d);
                sc[m][p] = (t < range) ? d / sq_hd : -INFINITY;                                   // infer.c:858
            }
        }
#pragma unroll
        for (int m = 0; m < KVM; m++) {
            float mx = mrun[m];
#pragma unroll
            for (int p = 0; p < NPT; p++) mx = fmaxf(mx, sc[m][p]);
            const float scale = (mrun[m] == -INFINITY) ? 0.0f : expf(mrun[m] - mx);
            float l = lrun[m] * scale;
#pragma unroll
            for (int q = 0; q < QV; q++) { acc[m][q].x *= scale; acc[m][q].y *= scale; acc[m][q].z *= scale; acc[m][q].w *= scale; }
#pragma unroll
            for (int p = 0; p < NPT; p++) {
                const float e = (sc[m][p] == -INFINITY) ? 0.0f : expf(sc[m][p] - mx);
                l += e;
                const uint32_t tv = ((round * NPT + p) * nsplit + split) * R + sub;
                const bool vf = (FUSE || (KVH && fresh_v)) && tv == pos;       // FP16 cache / fused launch: the fresh v row is not in the cache yet
#pragma unroll
                for (int q = 0; q < QV; q++) {
                    const float4 vv = vf ? vfresh[q] : kv_cvt(vreg[p][q]);
                    acc[m][q].x = __builtin_fmaf(e, vv.x, acc[m][q].x); acc[m][q].y = __builtin_fmaf(e, vv.y, acc[m][q].y);
                    acc[m][q].z = __builtin_fmaf(e, vv.z, acc[m][q].z); acc[m][q].w = __builtin_fmaf(e, vv.w, acc[m][q].w);
                }
            }
            mrun[m] = mx; lrun[m] = l;
        }
    }

    NANO_STAMP(a.stamps, 3, acc[0][0].x);                      // K / V rows arrived, scores + running softmax done
    // ---- 4. combine the R sub-groups of the workgroup ----------------------------------------------------------------
    // (a) inside each wave, in registers: the sub-groups of a wave hold the same head-dim slices in the lanes of equal
    //     l % LPR, so the wave's maximum, its exp-sum and its weighted-V slices are three cross-lane steps each (VALU only);
    // (b) across the four waves through LDS: 4 partial rows per head instead of one per sub-group.
    // (round 3: the former layout -- every sub-group's row through LDS, 32-term sums -- cost 1.7 of the kernel's 4.8 us)
    if constexpr (KVM > 1) {
        // Round 6, several heads per workgroup (the batched steps): the weighted accumulators of a wave's sub-groups meet through a
        // wave-private LDS block instead of three cross-lane steps per value (KVM x QV x 4 values x ~7 instructions = ~450 of a wave's VALU
        // instructions at KVM = 4, in a launch that is VALU bound): every lane parks its KVM x QV float4 products acc * w, then a lane
        // adds the NS sub-groups' values of two output elements in the SAME association the cross-lane tree has --
        // ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)), fp addition commutes -- so the bits are those of the KVM = 1 kernels.
        constexpr int NS = 64 / LPR, HP = LPR * QV * 4;          // sub-groups per wave; floats a sub-group holds per head (>= head_dim)
        float *tr = part + 4 * KVM * hd4 + (size_t)wid * NS * HP; // [sub-group][HP], wave-private: no barrier (a wave's LDS operations are in order)
        const uint32_t sw = (uint32_t)lane / LPR;                 // sub-group inside the wave
#pragma unroll
        for (int m = 0; m < KVM; m++) {
            const float Mw = xsub_max<LPR>(mrun[m]);
            const float w = (mrun[m] == -INFINITY) ? 0.0f : expf(mrun[m] - Mw);
            const float lw = xsub_sum<LPR>(lrun[m] * w);
#pragma unroll
            for (int q = 0; q < QV; q++)
                *reinterpret_cast<float4 *>(tr + sw * HP + 4u * fidx(q)) = make_float4(acc[m][q].x * w, acc[m][q].y * w, acc[m][q].z * w, acc[m][q].w * w);
            for (uint32_t o = (uint32_t)lane * 2u; o < (uint32_t)HP; o += 128u) {
                float2 v[NS];
#pragma unroll
                for (int k = 0; k < NS; k++) v[k] = *reinterpret_cast<const float2 *>(tr + k * HP + o);
                float2 t;
                if constexpr (NS == 8) {
                    t.x = ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
                    t.y = ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
                } else {
                    t.x = (v[0].x + v[1].x) + (v[2].x + v[3].x);
                    t.y = (v[0].y + v[1].y) + (v[2].y + v[3].y);
                }
                if (o < hd) *reinterpret_cast<float2 *>(part + ((size_t)wid * KVM + m) * hd4 + o) = t;
            }
            if (lane == 0) { redm[m * 4 + wid] = Mw; redl[m * 4 + wid] = lw; }
        }
    } else {
#pragma unroll
    for (int m = 0; m < KVM; m++) {
        const float Mw = xsub_max<LPR>(mrun[m]);
        const float w = (mrun[m] == -INFINITY) ? 0.0f : expf(mrun[m] - Mw);
        const float lw = xsub_sum<LPR>(lrun[m] * w);
#pragma unroll
        for (int q = 0; q < QV; q++) {
            const float4 t = make_float4(xsub_sum<LPR>(acc[m][q].x * w), xsub_sum<LPR>(acc[m][q].y * w), xsub_sum<LPR>(acc[m][q].z * w), xsub_sum<LPR>(acc[m][q].w * w));
            const uint32_t f = fidx(q);
            if (lane < LPR && f * 4u < hd) *reinterpret_cast<float4 *>(part + ((size_t)wid * KVM + m) * hd4 + 4 * f) = t;
        }
        if (lane == 0) { redm[m * 4 + wid] = Mw; redl[m * 4 + wid] = lw; }
    }
    }
    __syncthreads();
    NANO_STAMP(a.stamps, 4, redl[0]);                          // the four waves' partials are in LDS
    for (uint32_t idx = tid; idx < (uint32_t)KVM * hd; idx += 256) {
        uint32_t m = 0, i = idx;
        if (KVM > 1) { m = idx / hd; i = idx - m * hd; }
        const float4 mw = *reinterpret_cast<const float4 *>(redm + m * 4), lw4 = *reinterpret_cast<const float4 *>(redl + m * 4);
        const float M = fmaxf(fmaxf(mw.x, mw.y), fmaxf(mw.z, mw.w));
        const float e0 = (mw.x == -INFINITY) ? 0.0f : expf(mw.x - M), e1 = (mw.y == -INFINITY) ? 0.0f : expf(mw.y - M);
        const float e2 = (mw.z == -INFINITY) ? 0.0f : expf(mw.z - M), e3 = (mw.w == -INFINITY) ? 0.0f : expf(mw.w - M);
        const float *pp = part + (size_t)m * hd4 + i;
        const size_t ws_ = (size_t)KVM * hd4;
        float L = lw4.x * e0; L += lw4.y * e1; L += lw4.z * e2; L += lw4.w * e3;
        float o = pp[0] * e0; o += pp[ws_] * e1; o += pp[2 * ws_] * e2; o += pp[3 * ws_] * e3;
        const uint32_t h = h0 + m;
        if (nsplit == 1) {
            const float val = o / L;                                                                  // softmax normalisation (infer.c:631-633)
            // (plain store: write-through stores, which pay for the GEMVs' few results per workgroup, cost this kernel's 128 lanes
            // per head +0.3 us in round 3's A/B run)
            a.xba_out[(size_t)b * a.q_dim + (size_t)h * hd + i] = val;
            if (a.xf_out) {
                // Q80 group of 64 = the 64 lanes of this wave (head_dim % 64 == 0): quantize (infer/tensor.c:21-46) and store in
                // fragment order xf[token tile][group][kq * 16 + token % 16][16 bytes], byte j of a group at kq = j / 16
                float mx = fabsf(val);
#pragma unroll
                for (int o2 = 1; o2 < 64; o2 <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
                const float scale = div_const<127>(mx);
                const uint32_t e = h * hd + i, g = e >> 6, jj = e & 63u, ng = a.q_dim >> 6;
                const size_t gb = (size_t)(b >> 4) * ng + g;
                a.xf_out[gb * 1024u + (size_t)((jj >> 4) * 16u + (b & 15u)) * 16u + (jj & 15u)] = (int8_t)q80_quant1(val, scale);
                if (jj == 0) a.xsf_out[gb * 16u + (b & 15u)] = scale;
            }
        } else {
            a.out[((size_t)b * nsplit + split) * a.q_dim + (size_t)h * hd + i] = o;
            if (i == 0) { float *ml = a.ml + (((size_t)b * a.n_head + h) * nsplit + split) * 2; ml[0] = M; ml[1] = L; }
        }
    }
    NANO_STAMP_END(a.stamps, 5);                               // combined and stored: the workgroup's last wave ends
}

template <int LPR, int QV, int KVM, int MODE, bool KVH, bool PG, int NPT, bool W16>
__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    attention_body<LPR, QV, KVM, MODE, KVH, PG, NPT, W16, false>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z, SlabHand{});
}


}  // namespace

}  // namespace nano
