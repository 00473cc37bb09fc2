// gemv_f32.hip -- FP32 decode GEMV for gfx950 (reference matmul, infer/infer.c:637-651), same latency-oriented SLAB
// structure as the Q80 kernels (gemv_q80_impl.h): a workgroup owns `rw` consecutive rows, its work units (4 rows x one
// 256-float column chunk, x2 matrices for SwiGLU) are dealt to its waves, every wave issues the activation loads and
// then ALL its weight loads at kernel entry through buffer descriptors (one memory round trip), rmsnorm / the
// split-attention combine run from registers while the weights are in flight.
// The reference adds the n products of a row sequentially; here a lane accumulates its float4 slices with fused
// multiply-adds, a DPP tree sums the 64 lanes and one thread adds the chunk partials in order -- the summation ORDER
// differs, so results match the reference to rounding (stated tolerance 1e-5 relative, DESIGN.md "Parity").
// HBM-bound byte work (2 flop / 4 bytes): no MFMA.
#include "gemv_common.h"

namespace nano {

namespace {

template <int ROLE, int B, int NV>
__device__ __forceinline__ void stage_finish_f32(const GemvDev &a, Staged<B, NV> &r, float *xf, float *red, uint32_t n4) {
    const uint32_t tid = threadIdx.x, nthr = a.nthr, n = a.n;
    const uint32_t lane = tid & 63u, wid = tid >> 6, NW = nthr >> 6;
    const bool norm = has_flag<ROLE>(a, F_NORM), comb = has_flag<ROLE>(a, F_COMBINE);
    float *wgt = red + B * 16;
    if constexpr (NV == 0) {
        if (comb) combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
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
            for (uint32_t i = tid * 4u; i < n; i += nthr * 4u) {
                float4 v = comb ? combine4(a, b, i, wgt) : *reinterpret_cast<const float4 *>(x + i);
                if (norm) {
                    const float4 w = *reinterpret_cast<const float4 *>(a.norm_w + i);
                    v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
                }
                *reinterpret_cast<float4 *>(xf + b * n4 + i) = v;
            }
        }
        __syncthreads();
    } else {
        if (comb) {
            if constexpr (B == 1) {
                const bool pre_ml = a.attn_n_head * 8u <= nthr;
                if (pre_ml) combine_weights<B, true>(a, wgt, r.ml_m, r.ml_l); else combine_weights<B, false>(a, wgt, 0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    const uint32_t i = (tid + (uint32_t)j * nthr) * 4u;
                    const float *wg = wgt + (size_t)((i < n ? i : 0u) / a.attn_hd) * 8u;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int sp = 0; sp < 8; sp++) {
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
        if (norm) {                     // rmsnorm scale (infer.c:603-609); tree order
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
                if (i < n) *reinterpret_cast<float4 *>(xf + b * n4 + i) = v;
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float4 bload_wf(__amdgpu_buffer_rsrc_t r, uint32_t off) {          // streamed once: non-temporal
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 2);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

template <int ROLE, int B, int NV, int UPW>
__global__ __launch_bounds__(1024) void gemv_f32_slab_kernel(const GemvDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#define F32_A a
#define F32_BID blockIdx.x
#define F32_HAND 0
#define F32_HANDV (SlabHand{})
#define F32_PTAG 0u
#include "gemv_f32_slab_body.inc"
#undef F32_A
#undef F32_BID
#undef F32_HAND
#undef F32_HANDV
#undef F32_PTAG
}

}  // namespace
}  // namespace nano
#include "attn_impl.h"
namespace nano {
namespace {
// ---- q | k | v projection + attention in ONE launch for FP32 models (Nano: head_dim <= 64, no q / k norm, adjacent-pair RoPE; round 6: what
//      qkv_attn_fused_kernel is for Q80 and q4k_qkv_attn_fused_kernel for Q4K) ---------------------------------------------------------------
// The first `ngemv` workgroups run the projection's SLAB body (results stored as usual AND as granules), the last n_attn the attention's
// plain decode mode (attention_body MODE 2), 256 threads for both.  Epoch tags, give-up and re-issue: device_common.h, backend.hip.
// Reference: infer/infer.c:637-651, 758-879.
struct F32FusedArgs { GemvDev g; AttnArgs a; SlabHand hand; uint32_t n_attn, head_wgs, wait16, ngemv; };
template <int NV, int UPW, int QV>            // QV: float4 slots per lane of the attention's 8-lane sub-groups (1: head_dim <= 32, 2: <= 64 -- launch_attention's choice)
__global__ __launch_bounds__(256) void f32_qkv_attn_fused_kernel(const F32FusedArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);
    if (blockIdx.x >= fa.ngemv) {
        const uint32_t ab = blockIdx.x - fa.ngemv;
        const uint32_t split = ab / fa.head_wgs, grp = ab - split * fa.head_wgs;
        attention_body<8, QV, 1, 2, false, false, 2, false, true>(fa.a, smem, grp, 0u, split, fa.hand, hand_ctag(tk_, fa.hand), fa.wait16);
        return;
    }
    constexpr int ROLE = R_NORM_STORE, B = 1;
#define F32_A fa.g
#define F32_BID blockIdx.x
#define F32_HAND 1
#define F32_HANDV fa.hand
#define F32_PTAG hand_ptag(tk_, fa.hand)
#include "gemv_f32_slab_body.inc"
#undef F32_A
#undef F32_BID
#undef F32_HAND
#undef F32_HANDV
#undef F32_PTAG
}

// ---- Wo + W1|W3 in ONE launch for FP32 models (round 6: what wo_w13_fused_kernel is for Q80) -------------------------------------------------
// The launch has W1|W3's grid and thread count; its first `wo_wgs` workgroups run Wo's body first (results stored as usual AND as granules), then
// EVERY workgroup runs W1|W3's body with the activation polled from the granules.  Issue order: Wo's loads, W1|W3's weight and norm-weight loads,
// Wo's arithmetic, W1|W3's.  Producers first; a grid of at most one workgroup per CU is resident as a whole.  Same bodies, same bits.
struct F32Wo13Args { GemvDev wo; GemvDev w13; SlabHand hand; uint32_t wo_wgs, wait16; };
template <int ROLE_A, int NV_A, int UPW_A, int NV_B, int UPW_B>
__global__ __launch_bounds__(1024) void f32_wo_w13_fused_kernel(const F32Wo13Args fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint2 tk_ = hand_tick(fa.hand);
    constexpr int B = 1;
    {
        constexpr int ROLE = ROLE_A, NV = NV_A, UPW = UPW_A;
#define F32_BID blockIdx.x
#define F32_A fa.wo
#define F32_HAND 1
#define F32_HANDV fa.hand
#define F32_PTAG hand_ptag(tk_, fa.hand)
#define F32_XHAND 0
#define F32_XHANDV (SlabHand{})
#define F32_CTAG 0u
#define F32_XWAIT 0u
#define F32_PART 1
#include "gemv_f32_slab_body.inc"
#undef F32_PART
        auto wo_rest = [&]() __attribute__((always_inline)) {
#define F32_PART 2
#include "gemv_f32_slab_body.inc"
#undef F32_PART
        };
#undef F32_A
#undef F32_HAND
#undef F32_HANDV
#undef F32_PTAG
#undef F32_XHAND
#undef F32_XHANDV
#undef F32_CTAG
#undef F32_XWAIT
        {
            constexpr int ROLE = R_NORM_SWIGLU, NV = NV_B, UPW = UPW_B;
#define F32_A fa.w13
#define F32_HAND 0
#define F32_HANDV (SlabHand{})
#define F32_PTAG 0u
#define F32_XHAND 1
#define F32_XHANDV fa.hand
#define F32_CTAG hand_ctag(tk_, fa.hand)
#define F32_XWAIT (blockIdx.x >= fa.wo_wgs ? fa.wait16 : 0u)
#define F32_PART 1
#include "gemv_f32_slab_body.inc"
#undef F32_PART
            if (blockIdx.x < fa.wo_wgs) wo_rest();
            __syncthreads();                // (LDS is W1|W3's from here)
#define F32_PART 2
#include "gemv_f32_slab_body.inc"
#undef F32_PART
#undef F32_A
#undef F32_HAND
#undef F32_HANDV
#undef F32_PTAG
#undef F32_XHAND
#undef F32_XHANDV
#undef F32_CTAG
#undef F32_XWAIT
        }
#undef F32_BID
    }
}

struct F32Plan { uint32_t rw, nw, upw, nv; };
static F32Plan plan_f32(const GemvArgs &a, int B) {
    const uint32_t nchunk = (a.n + 255) / 256, nmat = a.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const uint32_t nseg = a.epi == GEMV_EPI_SWIGLU ? 1u : a.nseg;
    uint32_t align = 0;
    if (nseg > 1) for (uint32_t s = 0; s < nseg; s++) align |= a.seg[s].rows;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    uint32_t rw = 4;                 // 4 KiB of weights per unit: a few units per workgroup, >= 256 workgroups
    while (rw < 32 && (align % (rw * 2)) == 0 && (rw * 2 / 4) * nchunk <= 8 && rows / (rw * 2) >= 256) rw *= 2;
    const uint32_t units = (rw / 4) * nchunk * nmat;
    uint32_t nw = units < 8 ? units : 8;
    uint32_t want = (a.n * (uint32_t)(B > 2 ? B / 2 : 1) + 1023) / 1024;
    if (want > 16) want = 16;
    if (nw < want) nw = want;
    if (nw * 64 < rw * (uint32_t)B) nw = (rw * (uint32_t)B + 63) / 64;
    if (nw < 2) nw = 2;
    uint32_t upw = (units + nw - 1) / nw;
    while (upw > 4 && nw < 16) { nw++; upw = (units + nw - 1) / nw; }
    return F32Plan{rw, nw, upw, (a.n + 256 * nw - 1) / (256 * nw)};
}

template <int ROLE, int B, int NV, int UPW>
static hipError_t launch_f32_t(const GemvDev &d, const F32Plan &p, uint32_t rows, hipStream_t st) {
    const uint32_t nmat = d.epi == GEMV_EPI_SWIGLU ? 2 : 1;
    const size_t n4 = (d.n + 3) & ~3u, pc = (d.nchunk + 3) & ~3u;
    const size_t lds = (B * n4 + B * 16 + ((d.flags & F_COMBINE) ? (size_t)B * d.attn_n_head * 8 : 0) + (size_t)B * nmat * p.rw * pc) * 4;
    auto kern = &gemv_f32_slab_kernel<ROLE, B, NV, UPW>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemvDev dd = d; dd.nthr = 64 * p.nw;
    hipLaunchKernelGGL(kern, dim3((rows + p.rw - 1) / p.rw), dim3(64 * p.nw), lds, st, dd);
    return hipGetLastError();
}
template <int ROLE, int B>
static hipError_t launch_f32_r(const GemvDev &d, const F32Plan &p, uint32_t rows, hipStream_t st) {
    if (p.upw > 4) return hipErrorInvalidValue;
#define F32_GO(NV_, UPW_) do { if constexpr (B * NV_ <= 8) return launch_f32_t<ROLE, B, NV_, UPW_>(d, p, rows, st); } while (0)
    int nv = p.nv <= 1 ? 1 : p.nv <= 2 ? 2 : p.nv <= 4 ? 4 : 0;
    const int upw = p.upw <= 1 ? 1 : p.upw <= 2 ? 2 : 4;
    if (B * nv > 8) nv = 0;
    if (nv == 1) { if (upw == 1) F32_GO(1, 1); if (upw == 2) F32_GO(1, 2); F32_GO(1, 4); }
    if (nv == 2) { if (upw == 1) F32_GO(2, 1); if (upw == 2) F32_GO(2, 2); F32_GO(2, 4); }
    if (nv == 4) { if (upw == 1) F32_GO(4, 1); if (upw == 2) F32_GO(4, 2); F32_GO(4, 4); }
    if (upw == 1) F32_GO(0, 1);
    if (upw == 2) F32_GO(0, 2);
    F32_GO(0, 4);
    return hipErrorInvalidValue;
#undef F32_GO
}
template <int B>
static hipError_t launch_f32_b(const GemvArgs &a, hipStream_t st) {
    GemvDev d = to_dev(a);
    d.tile_max = nullptr;
    const F32Plan p = plan_f32(a, B);
    d.nchunk = (a.n + 255) / 256;
    d.magic_nchunk = (65536 + d.nchunk - 1) / d.nchunk;
    d.rw = p.rw;
    uint32_t l2 = 0; while ((1u << l2) < p.rw / 4) l2++;
    d.log2_tiles = l2;
    d.units = (p.rw / 4) * d.nchunk * (d.epi == GEMV_EPI_SWIGLU ? 2 : 1);
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    if constexpr (B == 1) {
        const uint32_t f = d.flags;
        if (f == F_NORM && d.epi == GEMV_EPI_STORE) return launch_f32_r<R_NORM_STORE, B>(d, p, rows, st);
        if (f == 0 && d.epi == GEMV_EPI_RESID) return launch_f32_r<R_RESID, B>(d, p, rows, st);
        if (f == F_COMBINE && d.epi == GEMV_EPI_RESID) return launch_f32_r<R_RESID_COMBINE, B>(d, p, rows, st);
        if (f == F_NORM && d.epi == GEMV_EPI_SWIGLU) return launch_f32_r<R_NORM_SWIGLU, B>(d, p, rows, st);
    }
    return launch_f32_r<R_GENERIC, B>(d, p, rows, st);
}

// the fused launch's plan: the projection's own, on 256 threads (four waves like the attention's workgroups; same bits -- the float4 items sit
// on the same threads, the fourth wave adds +0.0 to the norm's sum)
static bool f32_fused_shape(const GemvArgs &ga, const AttnArgs &aa, F32Plan &p) {
    if (ga.nb != 1 || ga.nseg != 3 || ga.epi != GEMV_EPI_STORE || !ga.norm_w || ga.xq_in || ga.attn_part || ga.tile_max || ga.resid_add || ga.n % 4u) return false;
    if (ga.seg[0].out_pstride || ga.seg[1].out_pstride) return false;            // (only v is position indexed: its cache row)
    for (uint32_t s2 = 0; s2 < 3; s2++) if (ga.seg[s2].rows % 4u) return false;
    p = plan_f32(ga, 1);
    if (p.nw > 4u || p.upw > 4u || ga.n > 1024u) return false;                   // one float4 item per thread on <= 256 threads
    p.nw = 4u; p.nv = 1u;
    const uint32_t units = (p.rw / 4) * ((ga.n + 255) / 256);
    p.upw = (units + 3u) / 4u;
    if (p.upw > 4u) return false;
    return fused_attn_side_ok_plain(aa, ga.seg[0].rows, ga.seg[1].rows, ga.seg[2].rows);
}

}  // namespace

bool qkv_attn_fused_f32_supports(const GemvArgs &ga, const AttnArgs &aa) { F32Plan p; return f32_fused_shape(ga, aa, p); }

// ---- the fused Wo + W1|W3 launch: host side (both bodies on W1|W3's threads; Wo has no tree: any thread count gives its bits) ------------------
namespace {
struct F32Wo13Plan { F32Plan a, b; uint32_t upw_a, nv_a, wa, wb; };
static void f32_dev_fill(GemvDev &d, const GemvArgs &a, const F32Plan &p, uint32_t nthr) {
    d.tile_max = nullptr;
    d.nchunk = (a.n + 255) / 256;
    d.magic_nchunk = (65536 + d.nchunk - 1) / d.nchunk;
    d.rw = p.rw;
    uint32_t l2 = 0; while ((1u << l2) < p.rw / 4) l2++;
    d.log2_tiles = l2;
    d.units = (p.rw / 4) * d.nchunk * (d.epi == GEMV_EPI_SWIGLU ? 2 : 1);
    d.nthr = nthr;
}
static bool f32_wo13_shape(const GemvArgs &wo, const GemvArgs &w13, F32Wo13Plan &q) {
    if (wo.nb != 1 || w13.nb != 1 || wo.n % 4u || w13.n % 4u) return false;
    if (wo.nseg != 1 || wo.epi != GEMV_EPI_RESID || wo.norm_w || wo.xq_in || wo.tile_max || wo.resid_add || wo.seg[0].out_pstride || wo.seg[0].rows % 4u) return false;
    if (wo.attn_part && (wo.attn_nsplit > 8u || wo.attn_hd % 4u)) return false;
    if (w13.nseg != 2 || w13.epi != GEMV_EPI_SWIGLU || !w13.norm_w || w13.xq_in || w13.attn_part || w13.tile_max || w13.resid_add || w13.seg[0].rows != w13.seg[1].rows) return false;
    if (w13.n != wo.seg[0].rows || w13.xin != wo.seg[0].out) return false;             // W1|W3's input is what Wo writes
    q.a = plan_f32(wo, 1); q.b = plan_f32(w13, 1);
    const uint32_t nw = q.b.nw;
    const uint32_t units_a = (q.a.rw / 4) * ((wo.n + 255) / 256);
    q.upw_a = (units_a + nw - 1) / nw;
    q.nv_a = (wo.n / 4 + 64 * nw - 1) / (64 * nw);
    q.wa = (wo.seg[0].rows + q.a.rw - 1) / q.a.rw; q.wb = (w13.seg[0].rows + q.b.rw - 1) / q.b.rw;
    const uint32_t cus = w13.cus ? w13.cus : 256u;
    if (q.wa > q.wb || q.wb > cus) return false;                                        // one workgroup per CU: the whole grid is resident
    if (q.a.rw > 64 * nw || q.b.rw > 64 * nw) return false;                             // one fold thread per row
    // instantiated: Nano-168M's shapes (Wo: one float4 item per thread, one unit per wave; W1|W3: one item, two units)
    return q.nv_a == 1u && q.upw_a == 1u && q.b.nv == 1u && q.b.upw == 2u;
}
}  // namespace
bool wo_w13_fused_f32_supports(const GemvArgs &wo, const GemvArgs &w13) { F32Wo13Plan q; return f32_wo13_shape(wo, w13, q); }

hipError_t launch_wo_w13_fused_f32(const GemvArgs &wo, const GemvArgs &w13, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    F32Wo13Plan q;
    if (!hand || !tick || !layer1 || layer1 > 127u || !f32_wo13_shape(wo, w13, q)) return hipErrorInvalidValue;
    F32Wo13Args fa{};
    fa.wo = to_dev(wo); fa.w13 = to_dev(w13);
    const uint32_t nthr = 64 * q.b.nw;
    f32_dev_fill(fa.wo, wo, q.a, nthr); f32_dev_fill(fa.w13, w13, q.b, nthr);
    fa.wo_wgs = q.wa;
    fa.wait16 = 4u;                      // workgroups that produce nothing nap ~2 us before their first poll (as the Q80 launch)
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    fa.hand = h;
    auto lds_of = [&](const GemvDev &d, const F32Plan &p) {
        const size_t n4 = (d.n + 3) & ~3u, pc = (d.nchunk + 3) & ~3u, nmat = d.epi == GEMV_EPI_SWIGLU ? 2 : 1;
        return (n4 + 16 + ((d.flags & F_COMBINE) ? (size_t)d.attn_n_head * 8 : 0) + nmat * p.rw * pc) * 4;
    };
    const size_t la = lds_of(fa.wo, q.a), lb = lds_of(fa.w13, q.b), lds = la > lb ? la : lb;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (fa.wo.flags & F_COMBINE) hipLaunchKernelGGL((f32_wo_w13_fused_kernel<R_RESID_COMBINE, 1, 1, 1, 2>), dim3(q.wb), dim3(nthr), lds, st, fa);
    else hipLaunchKernelGGL((f32_wo_w13_fused_kernel<R_RESID, 1, 1, 1, 2>), dim3(q.wb), dim3(nthr), lds, st, fa);
    return hipGetLastError();
}

hipError_t launch_qkv_attn_fused_f32(const GemvArgs &ga, const AttnArgs &aa, unsigned long long *hand, uint32_t *tick, uint32_t layer1, hipStream_t st) {
    F32Plan p;
    if (!hand || !tick || !layer1 || layer1 > 127u || !f32_fused_shape(ga, aa, p)) return hipErrorInvalidValue;
    GemvDev d = to_dev(ga);
    d.tile_max = nullptr;
    d.nchunk = (ga.n + 255) / 256;
    d.magic_nchunk = (65536 + d.nchunk - 1) / d.nchunk;
    d.rw = p.rw;
    uint32_t l2 = 0; while ((1u << l2) < p.rw / 4) l2++;
    d.log2_tiles = l2;
    d.units = (p.rw / 4) * d.nchunk;
    d.nthr = 256;
    uint32_t rows = 0;
    for (uint32_t s2 = 0; s2 < 3; s2++) rows += ga.seg[s2].rows;
    const uint32_t ngemv = (rows + p.rw - 1) / p.rw;
    AttnArgs a = aa;
    { uint32_t k2 = 0; while ((1u << k2) < a.n_kv_head) k2++; a.kv_log2 = k2; }
    { const uint32_t kv_mul = a.n_head / a.n_kv_head; uint32_t k2 = 0; while ((1u << k2) < kv_mul) k2++; a.kvmul_log2 = k2; }
    SlabHand h{};
    h.buf = hand; h.tick = tick; h.layer1 = layer1;
    h.base[0] = 0; h.base[1] = a.q_dim; h.base[2] = a.q_dim + a.kv_dim;
    const size_t n4 = (d.n + 3) & ~3u, pc = (d.nchunk + 3) & ~3u;
    const size_t lds_g = (n4 + 16 + (size_t)p.rw * pc) * 4;
    const size_t hd4 = (a.hd + 3) & ~3u, lds_a = (hd4 + hd4 + 4 + 4 + 4 * hd4 + hd4) * sizeof(float);    // q | k | maxima | sums | 4 waves' partials | the fresh v row
    const size_t lds = lds_g > lds_a ? lds_g : lds_a;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    F32FusedArgs fa{};
    fa.g = d; fa.a = a; fa.hand = h; fa.n_attn = a.n_head * a.nsplit; fa.head_wgs = a.n_head; fa.ngemv = ngemv;
    fa.wait16 = 1u;             // naps of 16 x 64 cycles before the attention's first poll: Nano-168M, one box, 0 / 1 / 2 / 3 naps: 2387 / 2384 / 2380 / 2364 and 2383 / 2381 / 2380 / 2366 tok/s
    const int upw = p.upw <= 1 ? 1 : p.upw <= 2 ? 2 : 4;
#define F32F_GO(UPW_) do { if (a.hd <= 32u) hipLaunchKernelGGL((f32_qkv_attn_fused_kernel<1, UPW_, 1>), dim3(fa.n_attn + ngemv), dim3(256), lds, st, fa); \
                           else hipLaunchKernelGGL((f32_qkv_attn_fused_kernel<1, UPW_, 2>), dim3(fa.n_attn + ngemv), dim3(256), lds, st, fa); return hipGetLastError(); } while (0)
    if (upw == 1) F32F_GO(1);
    if (upw == 2) F32F_GO(2);
    F32F_GO(4);
#undef F32F_GO
}

hipError_t launch_gemv_f32(const GemvArgs &a, hipStream_t st) {
    if (a.nb == 0 || a.nb > 8 || a.n % 4 || a.nseg == 0 || a.nseg > 3 || a.xq_in) return hipErrorInvalidValue;
    if (a.attn_part && (a.norm_w || a.attn_nsplit > 8 || a.attn_hd % 4)) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s < a.nseg; s++) if (a.seg[s].rows % 4) return hipErrorInvalidValue;
    if (a.nb <= 1) return launch_f32_b<1>(a, st);
    if (a.nb <= 2) return launch_f32_b<2>(a, st);
    if (a.nb <= 4) return launch_f32_b<4>(a, st);
    return launch_f32_b<8>(a, st);
}

}  // namespace nano
