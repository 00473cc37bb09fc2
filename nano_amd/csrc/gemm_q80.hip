// gemm_q80.hip -- Q80 (W8A8) skinny GEMM on the matrix cores for 9..64 tokens per weight read (large decode batches,
// batched prefill): the general kernel (G2: any group size 32..256, any group count) and the
// activation quantizers of the batched path.  gemm_q80_g6.hip / gemm_q80_g7.hip hold the faster kernels for group size 64.
// out[t][r] = matmul_quant(W[r,:], x_t) for every token t, BIT-IDENTICAL to the per-token reference
// (infer/infer.c:654-679): one v_mfma_i32_16x16x64_i8 forms the exact int32 group sums of a 16-row x 16-token tile for one
// 64-wide run of a quantization group, the group product ((float)ival * ws[r][g]) * xs[t][g] is applied on the VALU and
// accumulated per (row, token) in ascending group order, exactly like the GEMV kernels.
// MFMA operand layout (verified on gfx950, tools/kbench/mfma_probe.hip): lane l holds
// A[m = l%16][k = 16*(l/16) .. +15], B[k = same][n = l%16]; result c[i] = C[m = 4*(l/16) + i][n = l%16].
#include <type_traits>
#include "gemv_common.h"

namespace nano {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// ---- per-token activation quantization ------------------------------------------------------------------------------
template <int GS>
__global__ __launch_bounds__(256) void quant_rows_kernel(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n,
                                                         int8_t *xq, float *xs, uint32_t n16, uint32_t ng) {
    __shared__ float red[8];
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const float *xr = x + (size_t)t * x_bstride;
    float ss = 1.0f;
    if (norm_w) {                       // rmsnorm scale (infer.c:603-609), tree order
        float acc = 0.0f;
        for (uint32_t i = tid * 4u; i < n; i += 1024u) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            acc += v.x * v.x; acc += v.y * v.y; acc += v.z * v.z; acc += v.w * v.w;
        }
        acc = dpp_wave_sum(acc);
        if (lane == 0) red[wid] = acc;
        __syncthreads();
        float s = ((red[0] + red[1]) + red[2]) + red[3];          // the GEMV prologue's order for 256 threads (gemv_q80_impl.h)
        s /= (float)n; s += 1e-5f;
        ss = 1.0f / sqrtf(s);
    }
    for (uint32_t i = tid * 4u; i < n; i += 1024u) {      // n % GS == 0 and 1024 % GS == 0: groups are whole
        float4 v = *reinterpret_cast<const float4 *>(xr + i);
        if (norm_w) {
            const float4 w = *reinterpret_cast<const float4 *>(norm_w + i);
            v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
        }
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = dpp_group_max<GS / 4>(m);
        const float scale = div_const<127>(m);
        const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
        *reinterpret_cast<uint32_t *>(xq + (size_t)t * n16 + i) =
            (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
        if ((tid % (GS / 4)) == 0) xs[(size_t)t * ng + i / GS] = scale;
    }
}

// One quantization group of a 16x16 tile needs GS bytes of K per weight row and per token: FR = fragment registers
// (one per MFMA) per group.  GS >= 64: FR = GS/64 MFMAs of K = 64, a lane holds 16 bytes per fragment; GS = 32: one
// MFMA of K = 32, 8 bytes per lane.
template <int GS> struct Frag { i32x4 v; };
template <> struct Frag<32> { long v; };

template <int GS>
__device__ __forceinline__ v4i mma(const Frag<GS> &a, const Frag<GS> &b, v4i c) {
    if constexpr (GS == 32) return __builtin_amdgcn_mfma_i32_16x16x32_i8(a.v, b.v, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_i32_16x16x64_i8(a.v, b.v, c, 0, 0, 0);
}

// LDS fragment read of a staged weight row (row pitch KP bytes)
template <int GS>
__device__ __forceinline__ Frag<GS> lds_frag(const int8_t *row, uint32_t koff, uint32_t kq, int ks) {
    Frag<GS> f;
    if constexpr (GS == 32) f.v = *reinterpret_cast<const long *>(row + koff + kq * 8u);
    else f.v = *reinterpret_cast<const i32x4 *>(row + koff + (uint32_t)ks * 64u + kq * 16u);
    return f;
}

// =====================================================================================================================
// G2: the batched kernel of the BANDWIDTH-BOUND regime (large matrices, 8..64 tokens per weight read).
//
// (What bounded round 1's two kernels on Qwen3-4B-size matrices, 1.3-2.6 TB/s: one pass of weights in flight per
// workgroup with the load latency exposed at every pass, one workgroup per CU, a single wave doing all MFMA work of a
// token tile.)  G2 keeps the arithmetic (exact int32 group sums on the matrix cores, products
// ((float)ival * ws) * xs, ascending-group fold: bit-identical to the GEMV path and the reference) and changes the data
// flow:
//   * a workgroup = 8 waves = one 16-row tile (W1 and W3 tiles of the same rows for SwiGLU) x ALL tokens (TT tiles of 16);
//   * the row length is walked in passes of 512 bytes; every wave owns two rows of the tile and keeps its 1 KiB piece of
//     each of the next D = 8 passes IN FLIGHT in a register ring (64-128 KB per workgroup, re-issued as it is consumed):
//     no pass ever waits for a cold load except the first;
//   * a pass goes registers -> LDS stage (the coalesced row pieces land as rows of pitch 528: conflict-free ds_read_b128
//     of the MFMA A fragments) and is multiplied by all waves at once: work item (group of the pass, token tile) -> wave,
//     so with 16 tokens eight waves share the groups and with 64 tokens each wave reuses its A fragment on four token tiles;
//   * the activations arrive in MFMA B-fragment order (quant_rows_frag_kernel writes them that way): one coalesced 1 KiB
//     load per (group, token tile), prefetched a pass ahead from L2 -- no LDS staging, no transposition;
//   * products go to an LDS table; thread (row, token) adds them in ascending group order after the pass's barrier.
// Any group count per row (no multiple-of-4 rule), any number of segments whose interior sizes are multiples of 16.
// =====================================================================================================================
struct G2Dev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, epi, nb, npass, magic_ng;
    const int8_t *xf; const float *xsf; const uint32_t *pos;
};

constexpr uint32_t G2_PK = 512, G2_PITCH = 528;
// Ring depth: passes in flight per wave.  Loads complete in issue order (one vmcnt counter), so the activation
// fragments of a pass must be issued no later than the weight pieces that should still be in flight when the pass
// waits for them: BOTH rings run D passes ahead.  More token tiles = more fragment registers per pass = a shorter ring.
// Up to 32 tokens the kernel is held to 128 VGPRs so that TWO workgroups share a CU: one's start-up round trip (its ring
// filling) and epilogue overlap the other's passes.
constexpr uint32_t g2_depth(int tt, bool sw) { return tt == 1 ? (sw ? 6u : 8u) : 4u; }      // even: stage / table parity = slot parity

template <int GS, bool SW, int TT>
__global__ __launch_bounds__(512, ((TT <= 2 && GS == 64) ? 4 : 2)) void gemm_q80_g2_kernel(const G2Dev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int FR = GS >= 64 ? GS / 64 : 1;
    constexpr uint32_t GPP = G2_PK / (uint32_t)GS, NT = 16u * TT, NTP = NT + 1u, nmat = SW ? 2u : 1u;
    constexpr uint32_t NI = GPP * TT, IPW = (NI + 7u) / 8u;                 // MFMA work items per pass / per wave
    constexpr uint32_t PPT = (16u * NT + 511u) / 512u;                       // (row, token) accumulators per thread
    constexpr uint32_t FB = GS == 32 ? 512u : 1024u;                         // bytes of one B fragment block
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n = a.n, ng = a.ng, ngp = ng | 1u, npass = a.npass;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1];
    const uint32_t grow0 = blockIdx.x * 16u;
    const int sel = SW ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
    const float *ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);

    // LDS: stage[2][nmat][16][528] int8 | wsl[nmat][16][ngp] float | prod[nmat][GPP][16][NTP] float
    int8_t *stage = reinterpret_cast<int8_t *>(smem);
    float *wsl = reinterpret_cast<float *>(smem + 2u * nmat * 16u * G2_PITCH);
    float *prod = wsl + nmat * 16u * ngp;

    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0, rows0 * n), rw1 = mkrsrc(SW ? a.w[1] : nullptr, SW ? rows0 * n : 0u);
    const __amdgpu_buffer_rsrc_t rs0 = mkrsrc(ws0, rows0 * ng * 4u), rs1 = mkrsrc(SW ? a.ws[1] : nullptr, SW ? rows0 * ng * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rxf = mkrsrc(a.xf, (uint32_t)TT * ng * (uint32_t)FR * FB);
    const __amdgpu_buffer_rsrc_t rxs = mkrsrc(a.xsf, (uint32_t)TT * ng * 64u);

    // ---- first in the load queue (loads complete in issue order): the tile's weight scales (<= 8 elements per thread in
    // registers: ng <= 256; longer rows finish with a second round after the rings) and, for the residual epilogue, the
    // old output values -- so that nothing at the end of the kernel waits for a cold load
    constexpr uint32_t WSR = 8;
    float wv0[WSR], wv1[SW ? WSR : 1];
#pragma unroll
    for (uint32_t k = 0; k < WSR; k++) {
        const uint32_t e = k * 512u + tid;
        const uint32_t r = (e * a.magic_ng) >> 20, g = e - r * ng;               // e / ng (checked on the host)
        const uint32_t off = e < 16u * ng ? ((lrow0 + r) * ng + g) * 4u : OOB;
        wv0[k] = bload_f(rs0, off);
        if (SW) wv1[k] = bload_f(rs1, off);
    }
    float oldv[PPT];
#pragma unroll
    for (uint32_t q = 0; q < PPT; q++) {
        const uint32_t pr = tid + 512u * q, r = pr / NT, t = pr % NT;
        oldv[q] = 0.0f;
        if (a.epi == GEMV_EPI_RESID && pr < 16u * NT && t < a.nb && lrow0 + r < rows0)
            oldv[q] = out0[(size_t)t * obs + (ops ? (size_t)a.pos[t] * ops : 0) + lrow0 + r];
    }

    // ---- the weight ring: this wave's piece (rows 2w, 2w+1 x 512 bytes) of the next D passes ------------------------------
    const uint32_t lrow = lrow0 + wid * 2u + (lane >> 5), lcol = (lane & 31u) * 16u;
    constexpr uint32_t D = g2_depth(TT, SW);
    int4 ring[SW ? 2 : 1][D];
    auto issue_w = [&](uint32_t p, int slot) {
        const uint32_t col = p * G2_PK + lcol;
        const uint32_t off = (col < n) ? lrow * n + col : OOB;                 // passes beyond the row read as 0 (no branch: the compiler counts the loads)
#pragma unroll
        for (int mt = 0; mt < (int)nmat; mt++) {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(mt ? rw1 : rw0, (int)off, 0, 2);
            ring[mt][slot] = make_int4(v.x, v.y, v.z, v.w);
        }
    };

    // ---- this wave's MFMA work items of a pass: it = wid + 8 j -> (group of the pass, token tile) -----------------------
    const uint32_t m = lane & 15u, kq = lane >> 4;
    Frag<GS> fb[D][IPW][FR];
    float fxs[D][IPW];
    auto issue_b = [&](uint32_t p, int buf) {
#pragma unroll
        for (uint32_t j = 0; j < IPW; j++) {
            const uint32_t it = wid + 8u * j, gl = it % GPP, tt = it / GPP, g = p * GPP + gl;
            const bool live = it < NI && g < ng;
            const uint32_t blk = (tt * ng + g) * (uint32_t)FR;
#pragma unroll
            for (int ks = 0; ks < FR; ks++) {
                if constexpr (GS == 32) {
                    const uint32_t o = live ? blk * FB + lane * 8u : OOB;
                    const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(rxf, (int)o, 0, 0), hi = __builtin_amdgcn_raw_buffer_load_b32(rxf, (int)(o == OOB ? OOB : o + 4u), 0, 0);
                    fb[buf][j][ks].v = (long)(((unsigned long)hi << 32) | lo);
                } else {
                    fb[buf][j][ks].v = __builtin_amdgcn_raw_buffer_load_b128(rxf, (int)(live ? (blk + (uint32_t)ks) * FB + lane * 16u : OOB), 0, 0);
                }
            }
            fxs[buf][j] = bload_f(rxs, live ? ((tt * ng + g) * 16u + m) * 4u : OOB);
        }
    };
#pragma unroll
    for (int d = 0; d < (int)D; d++) { issue_w((uint32_t)d, d); issue_b((uint32_t)d, d); }

    // ---- the tile's weight scales -> LDS (any group count) ------------------------------------------------------------------
    // scales -> LDS (this waits for the scale loads only: the rings stay in flight)
#pragma unroll
    for (uint32_t k = 0; k < WSR; k++) {
        const uint32_t e = k * 512u + tid;
        const uint32_t r = (e * a.magic_ng) >> 20, g = e - r * ng;
        if (e < 16u * ng) { wsl[r * ngp + g] = wv0[k]; if (SW) wsl[16u * ngp + r * ngp + g] = wv1[k]; }
    }
    for (uint32_t e0 = WSR * 512u; e0 < 16u * ng; e0 += 512u) {               // rows of more than 256 groups (uniform trip count)
        const uint32_t e = e0 + tid;
        const uint32_t r = (e * a.magic_ng) >> 20, g = e - r * ng;
        const uint32_t off = e < 16u * ng ? ((lrow0 + r) * ng + g) * 4u : OOB;
        const float v0 = bload_f(rs0, off), v1 = SW ? bload_f(rs1, off) : 0.0f;
        if (e < 16u * ng) { wsl[r * ngp + g] = v0; if (SW) wsl[16u * ngp + r * ngp + g] = v1; }
    }

    float acc0[PPT], acc1[PPT];
#pragma unroll
    for (uint32_t q = 0; q < PPT; q++) { acc0[q] = 0.0f; acc1[q] = 0.0f; }

    // Passes are unrolled FLAT (ring slot and stage parity are compile-time) with one forward exit per pass, and every
    // load inside is unconditional: the compiler's s_waitcnt counting stays exact, so a pass waits only for its own
    // piece (issued D passes earlier) while the later pieces stay in flight.
    // A pass goes through three stages, software-pipelined so that ONE barrier per pass orders all of them (the three
    // stages between two barriers belong to three different passes and touch different LDS buffers):
    //   B(p-1)  MFMA work items of pass p-1 from stage[(p-1)&1] -> products prod[(p-1)&1]
    //   C(p-2)  ordered fold of pass p-2's products from prod[p&1]
    //   A(p)    this wave's piece of pass p: registers -> stage[p&1]; the ring slots are re-issued for pass p + D
    // The loop runs npass + 2 steps; stages of passes outside [0, npass) run on zeros (their loads are out of range) and
    // fold nothing -- no branch around a load or an MFMA, so the compiler's s_waitcnt counting stays exact: a step waits
    // only for its own pieces (issued D passes earlier) while the later ones stay in flight.  The body is unrolled by D
    // (ring slot, stage and table parity are compile-time) and small enough for the instruction cache.
    // PIPE = false (64 tokens with SwiGLU: two product tables do not fit in LDS): the stages of one pass run back to back
    // with two barriers.
    constexpr bool PIPE = !(TT == 4 && SW);
    constexpr uint32_t PRODSZ = nmat * GPP * 16u * NTP;                        // floats of one product table
    Frag<GS> fbc[IPW][FR]; float fxc[IPW];
#pragma unroll
    for (uint32_t j = 0; j < IPW; j++) { fxc[j] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < FR; ks++) fbc[j][ks] = Frag<GS>{}; }
    auto stA = [&](auto J, uint32_t p) {
        constexpr int d = decltype(J)::value, cur = d & 1;
        int8_t *st = stage + (size_t)cur * nmat * 16u * G2_PITCH;
#pragma unroll
        for (int mt = 0; mt < (int)nmat; mt++)
            *reinterpret_cast<int4 *>(st + ((size_t)mt * 16u + wid * 2u + (lane >> 5)) * G2_PITCH + lcol) = ring[mt][d];
        // the pass's activation fragments move to scratch registers: their slot is re-issued together with the weights
#pragma unroll
        for (uint32_t j = 0; j < IPW; j++) { fxc[j] = fxs[d][j];
#pragma unroll
            for (int ks = 0; ks < FR; ks++) fbc[j][ks] = fb[d][j][ks]; }
        issue_w(p + D, d);
        issue_b(p + D, d);
    };
    auto stB = [&](auto PAR, uint32_t p) {      // PAR = parity of pass p; groups beyond the row multiply zeros, the fold skips them
        constexpr int cur = decltype(PAR)::value;
        const int8_t *st = stage + (size_t)cur * nmat * 16u * G2_PITCH;
        float *pt = prod + (PIPE ? (size_t)cur * PRODSZ : 0);
#pragma unroll
        for (uint32_t j = 0; j < IPW; j++) {
            const uint32_t it = wid + 8u * j, gl = it % GPP, tt = it / GPP, g = p * GPP + gl;
            if (NI % 8u == 0 || it < NI) {
                const float xsc = fxc[j];
                v4i c0v = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < FR; ks++) c0v = mma<GS>(lds_frag<GS>(st + (size_t)m * G2_PITCH, gl * (uint32_t)GS, kq, ks), fbc[j][ks], c0v);
                const uint32_t gw = g < ng ? g : 0u;                           // scale index of a group beyond the row: any valid one (times zero)
#pragma unroll
                for (int i = 0; i < 4; i++)                                                          // infer.c:672
                    pt[((size_t)gl * 16u + kq * 4u + i) * NTP + tt * 16u + m] = ((float)c0v[i] * wsl[(kq * 4u + i) * ngp + gw]) * xsc;
                if (SW) {
                    v4i c1v = {0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < FR; ks++) c1v = mma<GS>(lds_frag<GS>(st + ((size_t)16u + m) * G2_PITCH, gl * (uint32_t)GS, kq, ks), fbc[j][ks], c1v);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        pt[((size_t)(GPP + gl) * 16u + kq * 4u + i) * NTP + tt * 16u + m] = ((float)c1v[i] * wsl[16u * ngp + (kq * 4u + i) * ngp + gw]) * xsc;
                }
            }
        }
    };
    auto stC = [&](auto PAR, uint32_t p) {      // ordered fold (infer.c:668-674): ascending groups; p outside [0, npass): nothing
        constexpr int cur = decltype(PAR)::value;
        const float *pt = prod + (PIPE ? (size_t)cur * PRODSZ : 0);
        const uint32_t gcnt = p >= npass ? 0u : (ng - p * GPP < GPP ? ng - p * GPP : GPP);      // (p = -1, -2 wrap to huge values)
#pragma unroll
        for (uint32_t q = 0; q < PPT; q++) {
            const uint32_t pr = tid + 512u * q, r = pr / NT, t = pr % NT;
            if (16u * NT % 512u == 0 || pr < 16u * NT) {
                float v0[GPP], v1[GPP];
#pragma unroll
                for (uint32_t gl = 0; gl < GPP; gl++) {
                    v0[gl] = pt[((size_t)gl * 16u + r) * NTP + t];
                    v1[gl] = SW ? pt[((size_t)(GPP + gl) * 16u + r) * NTP + t] : 0.0f;
                }
#pragma unroll
                for (uint32_t gl = 0; gl < GPP; gl++) {                         // a skipped group leaves the bits alone (x + 0.0f would turn -0.0f into +0.0f)
                    acc0[q] = gl < gcnt ? acc0[q] + v0[gl] : acc0[q];
                    if (SW) acc1[q] = gl < gcnt ? acc1[q] + v1[gl] : acc1[q];
                }
            }
        }
    };
    const uint32_t nstep = PIPE ? npass + 2u : npass;
    for (uint32_t p0 = 0;; p0 += D) {
#define G2_STEP(J) if constexpr ((J) < (int)D) { const uint32_t p = p0 + (uint32_t)(J); if (p >= nstep) goto g2_done; \
            using SJ = std::integral_constant<int, (J)>; using PC = std::integral_constant<int, ((J) & 1)>; using PO = std::integral_constant<int, (((J) & 1) ^ 1)>; \
            if constexpr (PIPE) { stB(PO{}, p - 1u); stC(PC{}, p - 2u); stA(SJ{}, p); __syncthreads(); } \
            else { stA(SJ{}, p); __syncthreads(); stB(PC{}, p); __syncthreads(); stC(PC{}, p); } }
        G2_STEP(0) G2_STEP(1) G2_STEP(2) G2_STEP(3) G2_STEP(4) G2_STEP(5) G2_STEP(6) G2_STEP(7)
#undef G2_STEP
        static_assert(D <= 8 && D % 2 == 0, "G2_STEP lines / parity");
    }
g2_done:
#pragma unroll
    for (uint32_t q = 0; q < PPT; q++) {
        const uint32_t pr = tid + 512u * q, r = pr / NT, t = pr % NT;
        if (pr < 16u * NT && t < a.nb && lrow0 + r < rows0) {
            float *o = out0 + (size_t)t * obs + (ops ? (size_t)a.pos[t] * ops : 0) + lrow0 + r;
            *o = finish_epi(a.epi, acc0[q], acc1[q], oldv[q]);
        }
    }
}

// rmsnorm (optional) + Q80 quantization of nb activation rows straight into MFMA B-fragment order:
//   xf  [tt][g][ks][lane = kq*16 + n][16 bytes]   (GS = 32: [tt][g][lane][8 bytes]),  token = tt*16 + n
//   xsf [tt][g][16 tokens]
template <int GS>
__global__ __launch_bounds__(256) void quant_rows_frag_kernel(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n,
                                                              int8_t *xf, float *xsf, uint32_t ng, uint32_t order512) {
    __shared__ float red[8];
    karg_touch(xf); karg_touch(xsf); karg_touch(ng);                  // the output pointers come with the first arguments, not after the quantizer
    constexpr uint32_t FR = GS >= 64 ? GS / 64 : 1, FB = GS == 32 ? 512u : 1024u, KB = GS == 32 ? 8u : 16u;
    const uint32_t t = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t tt = t >> 4, nn = t & 15u;
    const float *xr = x + (size_t)t * x_bstride;
    // workgroup (c, t) quantizes the 1024 elements [1024 c, 1024 c + 1024) of token t; with a norm every workgroup of the
    // token forms the whole row's sum of squares itself (same values in the same order: same bits in every workgroup)
    const uint32_t i = blockIdx.x * 1024u + tid * 4u;
    const bool live = i < n;
    float4 v = live ? *reinterpret_cast<const float4 *>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w = (live && norm_w) ? *reinterpret_cast<const float4 *>(norm_w + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 1.0f;
    if (norm_w && order512) {           // rmsnorm scale in the 512-thread tree order of route_norm_order() (512 threads: thread T adds the float4
        // items T, T + 512, ...; eight wave sums added in order): this thread plays T = tid and T = tid + 256 -- the same additions,
        // so 1..8 sequences (quantized in that prologue) and 9..64 (quantized here) see the same bits
        float acc0 = 0.0f, acc1 = 0.0f;
        for (uint32_t k0 = tid * 4u; k0 < n; k0 += 2048u) {
            const uint32_t k1 = k0 + 1024u;
            const float4 u0 = *reinterpret_cast<const float4 *>(xr + k0);
            const float4 u1 = k1 < n ? *reinterpret_cast<const float4 *>(xr + k1) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc0 += u0.x * u0.x; acc0 += u0.y * u0.y; acc0 += u0.z * u0.z; acc0 += u0.w * u0.w;
            if (k1 < n) { acc1 += u1.x * u1.x; acc1 += u1.y * u1.y; acc1 += u1.z * u1.z; acc1 += u1.w * u1.w; }
        }
        acc0 = dpp_wave_sum(acc0); acc1 = dpp_wave_sum(acc1);
        if (lane == 0) { red[wid] = acc0; red[4 + wid] = acc1; }
        __syncthreads();
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) s += red[k];
        s /= (float)n; s += 1e-5f;
        ss = 1.0f / sqrtf(s);
        v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
    } else
    if (norm_w) {                       // rmsnorm scale (infer.c:603-609), the GEMV prologue's tree order for 256 threads
        // four row pieces per trip, all four loads in flight before the first is used (a load per trip would cost a memory
        // round trip per 1024 elements); the additions keep their order: k ascending, x y z w
        float acc = 0.0f;
        for (uint32_t k0 = tid * 4u; k0 < n; k0 += 4096u) {
            float4 u[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t k = k0 + 1024u * (uint32_t)q;
                u[q] = k < n ? *reinterpret_cast<const float4 *>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (k0 + 1024u * (uint32_t)q < n) { acc += u[q].x * u[q].x; acc += u[q].y * u[q].y; acc += u[q].z * u[q].z; acc += u[q].w * u[q].w; }
        }
        acc = dpp_wave_sum(acc);
        if (lane == 0) red[wid] = acc;
        __syncthreads();
        float s = ((red[0] + red[1]) + red[2]) + red[3];
        s /= (float)n; s += 1e-5f;
        ss = 1.0f / sqrtf(s);
        v.x = w.x * (ss * v.x); v.y = w.y * (ss * v.y); v.z = w.z * (ss * v.z); v.w = w.w * (ss * v.w);
    }
    float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    mx = dpp_group_max<GS / 4>(mx);                       // n % GS == 0 and 1024 % GS == 0: groups are whole
    if (!live) return;
    const float scale = div_const<127>(mx);
    const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
    const uint32_t g = i / GS, j = i % GS;                                  // byte j of group g
    const uint32_t ks = GS >= 64 ? j / 64u : 0u, jj = GS >= 64 ? j % 64u : j, kq = jj / KB, b = jj % KB;
    const size_t dst = ((size_t)(tt * ng + g) * FR + ks) * FB + (size_t)(kq * 16u + nn) * KB + b;
    *reinterpret_cast<uint32_t *>(xf + dst) =
        (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    if ((tid % (GS / 4)) == 0) xsf[(size_t)(tt * ng + g) * 16u + nn] = scale;
}

template <int GS, bool SW, int TT>
static hipError_t launch_g2_t(const G2Dev &d, uint32_t rows, hipStream_t st) {
    constexpr uint32_t GPP = G2_PK / (uint32_t)GS, nmat = SW ? 2u : 1u, NTP = 16u * TT + 1u;
    constexpr bool PIPE = !(TT == 4 && SW);                                  // see the kernel: two product tables
    const size_t lds = (size_t)2 * nmat * 16 * G2_PITCH + (size_t)nmat * 16 * (d.ng | 1u) * 4 + (size_t)(PIPE ? 2 : 1) * nmat * GPP * 16 * NTP * 4;
    auto kern = &gemm_q80_g2_kernel<GS, SW, TT>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((rows + 15) / 16), dim3(512), lds, st, d);
    return hipGetLastError();
}
template <int GS>
static hipError_t launch_g2(const GemvArgs &a, hipStream_t st) {
    G2Dev d{};
    for (int i = 0; i < 3; i++) {
        const bool live = i < (int)a.nseg;
        d.w[i] = live ? reinterpret_cast<const int8_t *>(a.seg[i].w) : nullptr;
        d.ws[i] = live ? a.seg[i].ws : nullptr;
        d.out[i] = live ? a.seg[i].out : nullptr;
        d.rows[i] = live ? a.seg[i].rows : 0;
        d.out_bstride[i] = live ? a.seg[i].out_bstride : 0;
        d.out_pstride[i] = live ? a.seg[i].out_pstride : 0;
    }
    if (a.epi == GEMV_EPI_SWIGLU) { d.rows[1] = 0; d.rows[2] = 0; }
    d.n = a.n; d.ng = a.n / a.gs; d.epi = a.epi; d.nb = a.nb; d.npass = (a.n + G2_PK - 1) / G2_PK;
    d.magic_ng = ((1u << 20) + d.ng - 1) / d.ng;
    d.xf = a.xq_in; d.xsf = a.xs_in; d.pos = a.pos;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    const uint32_t tt = (a.nb + 15) / 16;
    const bool sw = a.epi == GEMV_EPI_SWIGLU;
#define G2_GO(TT_) do { return sw ? launch_g2_t<GS, true, TT_>(d, rows, st) : launch_g2_t<GS, false, TT_>(d, rows, st); } while (0)
    if (tt <= 1) G2_GO(1);
    if (tt <= 2) G2_GO(2);
    G2_GO(4);
#undef G2_GO
}

}  // namespace

// Host-side predicate: does the GEMM take this launch?  (What it does not take goes through the GEMV kernels in groups
// of 8 sequences, backend.hip gemv().)  Not taken: interior segments that are not multiples of the 16-row tile, a
// split-attention input, the LoRA o-branch addend.
bool gemm_q80_g2_supports(const GemvArgs &a) {
    if (a.nb == 0 || a.nb > 64 || a.gs == 0 || a.n % a.gs || a.n % 16 || a.nseg == 0 || a.nseg > 3 || a.attn_part || a.resid_add) return false;
    if (!(a.gs == 32 || a.gs == 64 || a.gs == 128 || a.gs == 256)) return false;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s + 1 < a.nseg; s++) if (a.seg[s].rows % 16) return false;     // a 16-row tile stays inside one segment
    const uint32_t ng = a.n / a.gs;
    const uint32_t magic = ((1u << 20) + ng - 1) / ng;
    for (uint32_t e = 0; e < 16 * ng + 4096; e++) if (((e * magic) >> 20) != e / ng) return false;
    return true;
}
// a.xq_in / a.xs_in: the activations in fragment order (launch_quant_rows_frag)
hipError_t launch_gemm_q80_g2(const GemvArgs &a, hipStream_t st) {
    if (!a.xq_in || !a.xs_in || !gemm_q80_g2_supports(a)) return hipErrorInvalidValue;
    switch (a.gs) {
    case 32: return launch_g2<32>(a, st);
    case 64: return launch_g2<64>(a, st);
    case 128: return launch_g2<128>(a, st);
    case 256: return launch_g2<256>(a, st);
    default: return hipErrorInvalidValue;
    }
}
// bytes of the fragment-order activation scratch for up to `tokens` tokens of row length n
size_t gemm_q80_frag_bytes(uint32_t tokens, uint32_t n) { return (size_t)((tokens + 15) / 16) * 16 * ((n + 15) & ~15u); }
hipError_t launch_quant_rows_frag(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n, uint32_t gs, uint32_t nb,
                                  int8_t *xf, float *xsf, hipStream_t st, uint32_t order) {
    if (!nb || gs == 0 || n % gs || n % 4 || (order != 256u && order != 512u)) return hipErrorInvalidValue;
    const uint32_t ng = n / gs, o5 = order == 512u ? 1u : 0u;
    switch (gs) {
    case 32: hipLaunchKernelGGL((quant_rows_frag_kernel<32>), dim3((n + 1023) / 1024, nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xf, xsf, ng, o5); break;
    case 64: hipLaunchKernelGGL((quant_rows_frag_kernel<64>), dim3((n + 1023) / 1024, nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xf, xsf, ng, o5); break;
    case 128: hipLaunchKernelGGL((quant_rows_frag_kernel<128>), dim3((n + 1023) / 1024, nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xf, xsf, ng, o5); break;
    case 256: hipLaunchKernelGGL((quant_rows_frag_kernel<256>), dim3((n + 1023) / 1024, nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xf, xsf, ng, o5); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// rmsnorm (optional) + Q80 quantization of nb activation rows, row-major (the quantize-once GEMV path of backend.hip)
hipError_t launch_quant_rows(const float *x, uint32_t x_bstride, const float *norm_w, uint32_t n, uint32_t gs, uint32_t nb,
                             int8_t *xq, float *xs, hipStream_t st) {
    if (!nb || gs == 0 || n % gs || n % 4) return hipErrorInvalidValue;
    const uint32_t n16 = (n + 15) & ~15u, ng = n / gs;
    switch (gs) {
    case 32: hipLaunchKernelGGL((quant_rows_kernel<32>), dim3(nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xq, xs, n16, ng); break;
    case 64: hipLaunchKernelGGL((quant_rows_kernel<64>), dim3(nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xq, xs, n16, ng); break;
    case 128: hipLaunchKernelGGL((quant_rows_kernel<128>), dim3(nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xq, xs, n16, ng); break;
    case 256: hipLaunchKernelGGL((quant_rows_kernel<256>), dim3(nb), dim3(256), 0, st, x, x_bstride, norm_w, n, xq, xs, n16, ng); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace nano
