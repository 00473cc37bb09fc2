// gemv_common.h -- pieces shared by the decode GEMV kernels of every weight format (gemv_q80_impl.h, gemv_f32.hip,
// gemv_q4k.hip): DPP reductions, buffer-descriptor loads, the device-side argument block, kernel roles, and the
// workgroup-cooperative activation loads (plain vector or split-attention combine).
#pragma once
#define NANO_GEMV_COMMON_H 1
#include <stdlib.h>
#include "device_common.h"
#include "kernels.h"

namespace nano {

namespace {

#define DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))

template <int W> __device__ __forceinline__ int dpp_group_sum(int v) {       // aligned groups of W lanes
    if (W >= 2) v += DPP_I(v, 0xB1);      // quad_perm [1,0,3,2]
    if (W >= 4) v += DPP_I(v, 0x4E);      // quad_perm [2,3,0,1]
    if (W >= 8) v += DPP_I(v, 0x141);     // row_half_mirror
    if (W >= 16) v += DPP_I(v, 0x140);    // row_mirror
    return v;
}
template <int W> __device__ __forceinline__ float dpp_group_max(float v) {
    if (W >= 2) v = fmaxf(v, DPP_F(v, 0xB1));
    if (W >= 4) v = fmaxf(v, DPP_F(v, 0x4E));
    if (W >= 8) v = fmaxf(v, DPP_F(v, 0x141));
    if (W >= 16) v = fmaxf(v, DPP_F(v, 0x140));
    if (W >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (W >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float dpp_wave_sum(float v) {
    v += DPP_F(v, 0xB1); v += DPP_F(v, 0x4E); v += DPP_F(v, 0x141); v += DPP_F(v, 0x140);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- buffer-descriptor loads: wave-uniform base, 32-bit byte offset, out-of-range -> 0 ------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t OOB = 0x7ffffff0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mkrsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int4 bload_w(__amdgpu_buffer_rsrc_t r, uint32_t off) {          // streamed once: non-temporal
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 bload_f4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ float bload_f(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

enum : uint32_t { F_NORM = 1u, F_PRE = 2u, F_COMBINE = 4u };

// Kernel roles.  A taken branch costs ~40 cycles and every instruction of the single wave a SIMD runs is on the
// critical path of these latency-bound kernels, so the per-layer launches get kernels with their feature flags
// resolved at compile time; R_GENERIC keeps them as run-time (wave-uniform) flags for everything else.
enum : int { R_GENERIC = 0, R_NORM_STORE = 1, R_RESID = 2, R_RESID_COMBINE = 3, R_NORM_SWIGLU = 4 };
struct GemvDev;
template <int ROLE> __device__ __forceinline__ bool has_flag(const GemvDev &a, uint32_t f);
template <int ROLE> __device__ __forceinline__ uint32_t role_epi(const GemvDev &a);

// Device-side argument block (one kernarg fetch, everything scalar).
struct GemvDev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, rw, log2_tiles, nchunk, magic_nchunk, units, epi, flags, nb;
    // Q80 slab launches: rw is ANY row count (balanced slabs: workgroups x rows fitted to the chip, not a power of two);
    // tpw = ceil(rw / 4) four-row tiles per workgroup, magic_rw = 65536 / rw + 1 (thread -> (sequence, row) by multiply-shift),
    // wg_c0 / wg_c1 = workgroups up to the end of segment 0 / 1 (a workgroup's rows lie inside one segment)
    uint32_t tpw, magic_rw, wg_c0, wg_c1;
    const float *xin; const float *norm_w; const uint32_t *pos;
    uint32_t xin_bstride, nthr;     // nthr = threads per workgroup (blockDim.x lives in the dispatch packet: one more scalar-cache line, read late)
    const int8_t *xq_in; const float *xs_in;
    const float *attn_part; const float *attn_ml;
    uint32_t attn_nsplit, attn_n_head, attn_hd, ntiles;
    float *tile_max;
    const float *resid_add; uint32_t resid_add_bstride;
    uint32_t early;                 // SLAB launches of the wide matrices: units of a wave's weights asked for BEFORE the activation is normalised + quantized (0 = all)
    unsigned long long *stamps;     // measurement builds only (-DNANO_STAMPS=1, tools/stamp_probe.py): [workgroup][8] shader-clock stamps, or nullptr
    uint32_t dbg;                   // measurement builds only: experiment bits (NANO_DBG): 1 = no norm-weight load, 2 = weights issued before the activation
    uint32_t canon;                 // 1: the fast path's canonical fold (kernels.h q80_canonical()), 0: the reference's ascending group order
    uint32_t *err;                  // sticky error word (host-mapped; nullptr in operator tests): a kernel that gives up a bounded wait ORs its code in
};

template <int ROLE> __device__ __forceinline__ bool has_flag(const GemvDev &a, uint32_t f) {
    if (ROLE == R_GENERIC) return (a.flags & f) != 0;
    if (f == F_NORM) return ROLE == R_NORM_STORE || ROLE == R_NORM_SWIGLU;
    if (f == F_COMBINE) return ROLE == R_RESID_COMBINE;
    return false;       // F_PRE: generic only
}
template <int ROLE> __device__ __forceinline__ uint32_t role_epi(const GemvDev &a) {
    if (ROLE == R_GENERIC) return a.epi;
    if (ROLE == R_NORM_STORE) return GEMV_EPI_STORE;
    if (ROLE == R_NORM_SWIGLU) return GEMV_EPI_SWIGLU;
    return GEMV_EPI_RESID;
}

// ------------------------------------------------------------------------------------------------------------
// Activation staging, workgroup-cooperative.  Thread t owns the float4 items t, t+nthr, ... (NV of them) of
// every sequence; a quantization group is GS/4 consecutive threads.  stage_issue() only issues the loads
// (call it first thing in the kernel), stage_finish() does rmsnorm + quantization from the registers into
// LDS (xq[B][n16] int8, xs[B][ng4] float) and ends with a workgroup barrier.
// ------------------------------------------------------------------------------------------------------------
template <int B, int NV>
struct Staged {
    float4 x[B][NV];
    float4 nw[NV];
    // split-attention combine, single sequence: every partial of this thread's items and its (max, sum) pair
    // are fetched at kernel entry too (one round trip instead of nsplit dependent ones)
    float4 pv[B == 1 ? NV : 1][B == 1 ? 8 : 1];
    float ml_m, ml_l;
};
template <int B>
struct Staged<B, 0> {};          // NV == 0: nothing is kept in registers, stage_finish() re-reads memory in loops

template <int ROLE, int B, int NV>
__device__ __forceinline__ void stage_issue(const GemvDev &a, Staged<B, NV> &r) {
    if constexpr (NV == 0) { (void)a; (void)r; return; } else {
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n;
    const bool plain = !has_flag<ROLE>(a, F_PRE) && !has_flag<ROLE>(a, F_COMBINE);
    const __amdgpu_buffer_rsrc_t rx = mkrsrc(a.xin, plain ? ((a.nb - 1) * a.xin_bstride + n) * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rn = mkrsrc(a.norm_w, has_flag<ROLE>(a, F_NORM) ? n * 4u : 0u);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
        const uint32_t off = (i < n) ? i * 4u : OOB;
        if (plain) {
#pragma unroll
            for (int b = 0; b < B; b++) r.x[b][j] = bload_f4(rx, (b < (int)a.nb) ? off + (uint32_t)b * a.xin_bstride * 4u : OOB);
        }
        if (has_flag<ROLE>(a, F_NORM)) r.nw[j] = (NANO_STAMPS && (a.dbg & 1u)) ? make_float4(1.f, 1.f, 1.f, 1.f) : bload_f4(rn, off);
    }
    if constexpr (B == 1) {
      if (has_flag<ROLE>(a, F_COMBINE)) {     // uniform branch: an out-of-range load is not free, do not issue 8*NV of them
        const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
        const __amdgpu_buffer_rsrc_t rp = mkrsrc(a.attn_part, ns * n * 4u);
        const __amdgpu_buffer_rsrc_t rm = mkrsrc(a.attn_ml, nh * ns * 8u);
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
#pragma unroll
            for (int sp = 0; sp < 8; sp++) r.pv[j][sp] = bload_f4(rp, (i < n && (uint32_t)sp < ns) ? ((uint32_t)sp * n + i) * 4u : OOB);
        }
        const uint32_t sp = tid & 7u, h = tid >> 3;
        const uint32_t mo = (h < nh && sp < ns) ? (h * ns + sp) * 8u : OOB;
        r.ml_m = bload_f(rm, mo);
        r.ml_l = bload_f(rm, mo == OOB ? OOB : mo + 4u);
      }
    }
    }
}

// the norm weights of a launch whose activation values arrive another way (granules of the same launch: gemv_q80_slab_body.inc SLAB_XHAND)
template <int ROLE, int B, int NV>
__device__ __forceinline__ void stage_issue_nw(const GemvDev &a, Staged<B, NV> &r) {
    if constexpr (NV == 0) { (void)a; (void)r; return; } else {
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n;
    const __amdgpu_buffer_rsrc_t rn = mkrsrc(a.norm_w, has_flag<ROLE>(a, F_NORM) ? n * 4u : 0u);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
        if (has_flag<ROLE>(a, F_NORM)) r.nw[j] = bload_f4(rn, (i < n) ? i * 4u : OOB);
    }
    }
}

// x[b][i] = sum_s part[b][s][i] * wgt[b][head(i)][s]  (attn.hip split partials), wgt from (max, sum) pairs
// PRE: the (max, sum) pair of thread (head, split) came with the kernel's first loads (pm, pl) -- one pass, no load in the
// loop (a runtime flag left a load on the other path, and the wait the compiler put at the join was for EVERY load in flight,
// the weights included)
template <int B, bool PRE>
__device__ __forceinline__ void combine_weights(const GemvDev &a, float *wgt /* LDS [B][n_head][8] */, float pm, float pl) {
    const uint32_t tid = threadIdx.x, nthr = a.nthr;
    const uint32_t ns = a.attn_nsplit, nh = a.attn_n_head;
    // thread (b, h, s<8): e_s = exp(m_s - M) / sum_s l_s exp(m_s - M); the 8 lanes of a head are one DPP half-row
    for (uint32_t t = tid; t < (uint32_t)B * nh * 8u; t += nthr) {
        const uint32_t s = t & 7u, h = (t >> 3) % nh, b = (t >> 3) / nh;
        float m = -INFINITY, l = 0.0f;
        if constexpr (PRE) { m = pm; l = pl; (void)s; (void)h; (void)b; (void)ns; }
        else if (s < ns && b < a.nb) { const float *ml = a.attn_ml + (((size_t)b * nh + h) * ns + s) * 2; m = ml[0]; l = ml[1]; }
        const bool live = l > 0.0f;
        float M = live ? m : -INFINITY;
        M = fmaxf(M, DPP_F(M, 0xB1)); M = fmaxf(M, DPP_F(M, 0x4E)); M = fmaxf(M, DPP_F(M, 0x141));
        const float e = live ? expf(m - M) : 0.0f;
        float L = l * e;
        L += DPP_F(L, 0xB1); L += DPP_F(L, 0x4E); L += DPP_F(L, 0x141);
        wgt[t] = e / L;
        if constexpr (PRE) break;       // every (head, split) pair has its own thread
    }
    __syncthreads();
}
__device__ __forceinline__ float4 combine4(const GemvDev &a, uint32_t b, uint32_t i, const float *wgt) {
    const uint32_t ns = a.attn_nsplit, n = a.n;
    const float *part = a.attn_part + (size_t)b * ns * n + i;
    const float *wg = wgt + ((size_t)b * a.attn_n_head + i / a.attn_hd) * 8u;
    float4 o[8];                        // all splits in flight at once (one round trip), then the ordered accumulation
#pragma unroll
    for (int s = 0; s < 8; s++) o[s] = ((uint32_t)s < ns) ? *reinterpret_cast<const float4 *>(part + (size_t)s * n) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const float w = wg[s];          // splits >= nsplit carry weight 0
        acc.x += o[s].x * w; acc.y += o[s].y * w; acc.z += o[s].z * w; acc.w += o[s].w * w;
    }
    return acc;
}

__device__ __forceinline__ float finish_epi(uint32_t epi, float v0, float v1, float old) {
    if (epi == GEMV_EPI_STORE) return v0;
    if (epi == GEMV_EPI_RESID) return old + v0;                 // x[i] += xb2[i]
    float h = v0;                                               // SwiGLU: silu(w1 x) * (w3 x)
    h *= (1.0f / (1.0f + expf(-h)));
    h *= v1;
    return h;
}

// host: GemvArgs -> device argument block (plan-dependent fields are filled by the launchers)
static GemvDev to_dev(const GemvArgs &a) {
    GemvDev d{};
    for (int i = 0; i < 3; i++) {
        const bool live = i < (int)a.nseg;
        d.w[i] = live ? reinterpret_cast<const int8_t *>(a.seg[i].w) : nullptr;
        d.ws[i] = live ? a.seg[i].ws : nullptr;
        d.out[i] = live ? a.seg[i].out : nullptr;
        d.rows[i] = live ? a.seg[i].rows : 0;
        d.out_bstride[i] = live ? a.seg[i].out_bstride : 0;
        d.out_pstride[i] = live ? a.seg[i].out_pstride : 0;
    }
    if (a.epi == GEMV_EPI_SWIGLU) { d.rows[1] = 0; d.rows[2] = 0; }       // segment 1 is the second matrix, not more rows
    d.n = a.n; d.ng = a.gs ? a.n / a.gs : 0;
    d.nchunk = (a.n + 1023) / 1024;
    d.magic_nchunk = (65536 + d.nchunk - 1) / d.nchunk;
    d.epi = a.epi; d.nb = a.nb;
    d.flags = (a.norm_w ? F_NORM : 0) | (a.xq_in ? F_PRE : 0) | (a.attn_part ? F_COMBINE : 0);
    d.xin = a.xin; d.norm_w = a.norm_w; d.pos = a.pos; d.xin_bstride = a.xin_bstride;
    d.xq_in = a.xq_in; d.xs_in = a.xs_in;
    d.attn_part = a.attn_part; d.attn_ml = a.attn_ml; d.attn_nsplit = a.attn_nsplit; d.attn_n_head = a.attn_n_head; d.attn_hd = a.attn_hd;
    d.tile_max = a.tile_max;
    d.resid_add = a.resid_add; d.resid_add_bstride = a.resid_add_bstride;
    d.stamps = a.stamps;
    d.canon = q80_canonical(a) ? 1u : 0u;
    d.err = a.err;
#if NANO_STAMPS
    { static const uint32_t dbg = getenv("NANO_DBG") ? (uint32_t)strtoul(getenv("NANO_DBG"), nullptr, 0) : 0u; d.dbg = dbg; }      // measurement builds only
#endif
    return d;
}

}  // namespace

}  // namespace nano
