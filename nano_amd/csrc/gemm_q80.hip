// gemm_q80.hip -- Q80 (W8A8) skinny GEMM on the matrix cores for 9..64 tokens per weight read (large decode batches;
// SURVEY 7 step 7 / 8f-1).  out[t][r] = matmul_quant(W[r,:], x_t) for every token t, BIT-IDENTICAL to the per-token
// reference (infer/infer.c:654-679): one v_mfma_i32_16x16x64_i8 forms the exact int32 group sums of a 16-row x 16-token
// tile for one 64-wide quantization group (K = group size -- this is where an int8 MFMA tile actually forms), the
// group product ((float)ival * ws[r][g]) * xs[t][g] is applied on the VALU and accumulated per (row, token) in
// ascending group order, exactly like the GEMV kernels.
//
// Mapping: a wave owns one (16-row tile, 16-token tile) pair and walks all groups of the row in order; the waves of a
// workgroup are the token tiles of the same rows, so the weight bytes come from HBM once and from the CU's L1 for
// the other token tiles.  MFMA operand layout (verified on gfx950, tools/kbench/mfma_probe.hip): lane l holds
// A[m = l%16][k = 16*(l/16) .. +15], B[k = same][n = l%16]; result c[i] = C[m = 4*(l/16) + i][n = l%16].
// The activations of all tokens are quantized once per GEMM by quant_rows_kernel (rmsnorm + quantize, reference
// infer/tensor.c:21-46, infer.c:601-614) into a global scratch the GEMM reads through L2.
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
        const float scale = m / 127.0f;
        const int q0 = q80_quant1(v.x, scale), q1 = q80_quant1(v.y, scale), q2 = q80_quant1(v.z, scale), q3 = q80_quant1(v.w, scale);
        *reinterpret_cast<uint32_t *>(xq + (size_t)t * n16 + i) =
            (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
        if ((tid % (GS / 4)) == 0) xs[(size_t)t * ng + i / GS] = scale;
    }
}

struct GemmDev {
    const int8_t *w[3]; const float *ws[3]; float *out[3];
    uint32_t rows[3], out_bstride[3], out_pstride[3];
    uint32_t n, ng, epi, nb, ttiles, n16;
    const int8_t *xq; const float *xs; const uint32_t *pos;
};

// integer group sums of a 16x16 tile for one quantization group: GS/64 MFMAs of K = 64 (GS = 32: one of K = 32)
template <int GS>
__device__ __forceinline__ v4i group_mma(__amdgpu_buffer_rsrc_t rw, uint32_t woff, __amdgpu_buffer_rsrc_t rx, uint32_t xoff, uint32_t kq) {
    v4i c = {0, 0, 0, 0};
    if constexpr (GS == 32) {
        // K = 32: lane holds 8 bytes, k = 8*kq .. +7
        const uint32_t wo = woff == OOB ? OOB : woff + kq * 8u, xo = xoff == OOB ? OOB : xoff + kq * 8u;
        const uint32_t a0 = __builtin_amdgcn_raw_buffer_load_b32(rw, (int)wo, 0, 0), a1 = __builtin_amdgcn_raw_buffer_load_b32(rw, (int)(wo == OOB ? OOB : wo + 4u), 0, 0);
        const uint32_t b0 = __builtin_amdgcn_raw_buffer_load_b32(rx, (int)xo, 0, 0), b1 = __builtin_amdgcn_raw_buffer_load_b32(rx, (int)(xo == OOB ? OOB : xo + 4u), 0, 0);
        const long a = (long)(((unsigned long)a1 << 32) | a0), b = (long)(((unsigned long)b1 << 32) | b0);
        c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
    } else {
#pragma unroll
        for (int ks = 0; ks < GS / 64; ks++) {
            const uint32_t wo = woff == OOB ? OOB : woff + ks * 64u + kq * 16u, xo = xoff == OOB ? OOB : xoff + ks * 64u + kq * 16u;
            const i32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)wo, 0, 0);
            const i32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)xo, 0, 0);
            c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
        }
    }
    return c;
}

template <int GS>
__global__ __launch_bounds__(256) void gemm_q80_mfma_kernel(const GemmDev a) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n = a.n, ng = a.ng;
    const bool swiglu = a.epi == GEMV_EPI_SWIGLU;
    // (row tile, token tile) pair of this wave; token tile fastest: the waves of a workgroup share their rows
    const uint32_t pair = blockIdx.x * 4u + (uint32_t)wid;
    const uint32_t tt = pair % a.ttiles, rt = pair / a.ttiles;
    const uint32_t b0 = a.rows[0], b1 = b0 + a.rows[1], total = swiglu ? a.rows[0] : b1 + a.rows[2];
    const uint32_t grow0 = rt * 16u;
    if (grow0 >= total) return;                                  // wave-uniform
    const int sel = swiglu ? 0 : (int)(grow0 >= b0) + (int)(grow0 >= b1);
    const int8_t *w0 = sel == 0 ? a.w[0] : sel == 1 ? a.w[1] : a.w[2];
    const float *ws0 = sel == 0 ? a.ws[0] : sel == 1 ? a.ws[1] : a.ws[2];
    float *out0 = sel == 0 ? a.out[0] : sel == 1 ? a.out[1] : a.out[2];
    const uint32_t rows0 = sel == 0 ? a.rows[0] : sel == 1 ? a.rows[1] : a.rows[2];
    const uint32_t obs = sel == 0 ? a.out_bstride[0] : sel == 1 ? a.out_bstride[1] : a.out_bstride[2];
    const uint32_t ops = sel == 0 ? a.out_pstride[0] : sel == 1 ? a.out_pstride[1] : a.out_pstride[2];
    const uint32_t lrow0 = grow0 - (sel == 0 ? 0u : sel == 1 ? b0 : b1);

    const uint32_t m = (uint32_t)lane & 15u, kq = (uint32_t)lane >> 4;
    const __amdgpu_buffer_rsrc_t rw0 = mkrsrc(w0, rows0 * n), rw1 = mkrsrc(swiglu ? a.w[1] : nullptr, swiglu ? rows0 * n : 0u);
    const __amdgpu_buffer_rsrc_t rs0 = mkrsrc(ws0, rows0 * ng * 4u), rs1 = mkrsrc(swiglu ? a.ws[1] : nullptr, swiglu ? rows0 * ng * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rx = mkrsrc(a.xq, a.nb * a.n16), rxs = mkrsrc(a.xs, a.nb * ng * 4u);
    const uint32_t wrow = (lrow0 + m) * n;                        // A operand: weight row lrow0 + m  (rows beyond the segment: out of range -> 0)
    const uint32_t tok = tt * 16u + m;                           // B operand / result column: token
    const uint32_t xrow = tok < a.nb ? tok * a.n16 : OOB;
    const uint32_t srow = (lrow0 + kq * 4u) * ng * 4u;            // result rows 4*kq + i: their weight scales
    const uint32_t xsrow = tok < a.nb ? tok * ng * 4u : OOB;

    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (uint32_t g = 0; g < ng; g++) {                           // ascending groups: the reference's order (infer.c:668-674)
        const uint32_t koff = g * GS;
        const v4i c0 = group_mma<GS>(rw0, wrow + koff, rx, xrow == OOB ? OOB : xrow + koff, kq);
        const float xsc = bload_f(rxs, xsrow == OOB ? OOB : xsrow + g * 4u);
        float wsc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) wsc[i] = bload_f(rs0, srow + ((uint32_t)i * ng + g) * 4u);
#pragma unroll
        for (int i = 0; i < 4; i++) acc0[i] += ((float)c0[i] * wsc[i]) * xsc;                   // infer.c:672
        if (swiglu) {
            const v4i c1 = group_mma<GS>(rw1, wrow + koff, rx, xrow == OOB ? OOB : xrow + koff, kq);
#pragma unroll
            for (int i = 0; i < 4; i++) acc1[i] += ((float)c1[i] * bload_f(rs1, srow + ((uint32_t)i * ng + g) * 4u)) * xsc;
        }
    }
    if (tok < a.nb) {
        float *o = out0 + (size_t)tok * obs + (ops ? (size_t)a.pos[tok] * ops : 0);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t r = lrow0 + kq * 4u + (uint32_t)i;
            if (r < rows0) o[r] = finish_epi(a.epi, acc0[i], acc1[i], a.epi == GEMV_EPI_RESID ? o[r] : 0.0f);
        }
    }
}

template <int GS>
static hipError_t launch_gs(const GemvArgs &a, hipStream_t st) {
    GemmDev d{};
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
    d.n = a.n; d.ng = a.n / a.gs; d.epi = a.epi; d.nb = a.nb; d.ttiles = (a.nb + 15) / 16; d.n16 = (a.n + 15) & ~15u;
    d.xq = a.xq_in; d.xs = a.xs_in; d.pos = a.pos;
    uint32_t rows = 0;
    if (a.epi == GEMV_EPI_SWIGLU) rows = a.seg[0].rows; else for (uint32_t s = 0; s < a.nseg; s++) rows += a.seg[s].rows;
    const uint32_t pairs = ((rows + 15) / 16) * d.ttiles;
    hipLaunchKernelGGL((gemm_q80_mfma_kernel<GS>), dim3((pairs + 3) / 4), dim3(256), 0, st, d);
    return hipGetLastError();
}

}  // namespace

// xq_in / xs_in of `a`: the quantized activations of all a.nb tokens, [nb][(n+15)&~15] int8 and [nb][n/gs] float
hipError_t launch_gemm_q80(const GemvArgs &a, hipStream_t st) {
    if (a.nb == 0 || a.nb > 64 || !a.xq_in || !a.xs_in || a.gs == 0 || a.n % a.gs || a.n % 16 || a.nseg == 0 || a.nseg > 3 || a.attn_part) return hipErrorInvalidValue;
    if (a.epi != GEMV_EPI_SWIGLU && a.nseg > 1)
        for (uint32_t s = 0; s + 1 < a.nseg; s++) if (a.seg[s].rows % 16) return hipErrorInvalidValue;     // a 16-row tile stays inside one segment
    switch (a.gs) {
    case 32: return launch_gs<32>(a, st);
    case 64: return launch_gs<64>(a, st);
    case 128: return launch_gs<128>(a, st);
    case 256: return launch_gs<256>(a, st);
    default: return hipErrorInvalidValue;
    }
}

// rmsnorm (optional) + Q80 quantization of nb activation rows into the GEMM's scratch
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
