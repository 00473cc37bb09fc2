// strict.hip -- the float reductions of the forward in the REFERENCE'S OWN ORDER ("strict parity" mode, NANO_STRICT=1 /
// nano_hip_set_strict).
//
// The fused decode kernels evaluate rmsnorm, q.k, the softmax denominator, the weighted V sum and the FP32 matmul as
// wave-parallel trees with device expf (tolerance 1e-5 per operator); a last-ulp difference there flips round(x/scale)
// decisions of the activation quantizers and the Q80 / Q4K logits drift to the reference's own inter-build noise floor
// (SURVEY F3).  The kernels below restate every one of those chains sequentially, exactly as the reference's C loops
// run them (no FMA contraction: the library is built with -ffp-contract=off; IEEE divide / sqrt; expf = exact_math.h's
// restatement of the pinned host libm).  Together with the bit-exact pieces the fast path already has (Q80 / Q4K
// quantizers and GEMVs, embedding dequantization, RoPE, residual adds, arg-max) a strict forward returns the
// reference's logits BIT FOR BIT -- the proof that nothing but summation order separates the fast path from it.
// Slow by construction (one thread walks each chain); never used on the timed path.
//
//   strict_rmsnorm_kernel     infer/infer.c:601-614     ss += x[j]*x[j] in index order; w[j] * (ss * x[j])
//   strict_qk_kernel          infer/infer.c:810-835     per head: [rmsnorm] + rope / rope_qwen3 (infer.c:681-706); k -> cache row pos
//   strict_scores_kernel      infer/infer.c:850-861     score += q[i]*k[i] in index order; score /= sqrtf(head_dim)
//   strict_softmax_kernel     infer/infer.c:616-634     max, expf(x - max), sum in index order, x /= sum
//   strict_av_kernel          infer/infer.c:866-877     xb[i] += a[t] * v[t][i], t ascending
//   strict_swiglu_kernel      infer/infer.c:937-944     val *= 1/(1+expf(-val)); val *= hb2
//   strict_matmul_f32_kernel  infer/infer.c:637-651     val += w[i*n+j] * x[j], j ascending
#include "device_common.h"
#include "exact_math.h"
#include "kernels.h"

namespace nano {

namespace {

__global__ __launch_bounds__(256) void strict_rmsnorm_kernel(float *o, const float *x, const float *w, uint32_t n, uint32_t x_stride, uint32_t o_stride) {
    extern __shared__ float sh[];                                  // n + 1
    const float *xv = x + (size_t)blockIdx.x * x_stride;
    float *ov = o + (size_t)blockIdx.x * o_stride;
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) sh[j] = xv[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        float ss = 0.0f;
        for (uint32_t j = 0; j < n; j++) ss += sh[j] * sh[j];
        ss /= (float)n;
        ss += 1e-5f;
        ss = 1.0f / sqrtf(ss);
        sh[n] = ss;
    }
    __syncthreads();
    const float ss = sh[n];
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) ov[j] = w[j] * (ss * sh[j]);
}

// one workgroup per head vector: blockIdx.x < n_head -> q head (in place), else the k head (raw k -> finished cache row)
__global__ __launch_bounds__(128) void strict_qk_kernel(const StrictAttnArgs a) {
    __shared__ float sh[257];
    const uint32_t v = blockIdx.x, b = blockIdx.y, hd = a.hd, half = hd >> 1, tid = threadIdx.x;
    const bool isq = v < a.n_head;
    const uint32_t pos = a.pos[b];
    const size_t krow = (((size_t)(a.slot0 + b) * a.n_layer + a.layer) * a.S + pos) * a.kv_dim;
    const float *src = isq ? a.q + (size_t)b * a.q_dim + (size_t)v * hd : a.kraw + (size_t)b * a.kv_dim + (size_t)(v - a.n_head) * hd;
    float *dst = isq ? a.q + (size_t)b * a.q_dim + (size_t)v * hd : a.kcache + krow + (size_t)(v - a.n_head) * hd;
    for (uint32_t j = tid; j < hd; j += blockDim.x) sh[j] = src[j];
    __syncthreads();
    const float *nw = isq ? a.q_norm : a.k_norm;
    if (nw) {                                                      // Qwen3 q/k-norm (infer.c:824-835)
        if (tid == 0) {
            float ss = 0.0f;
            for (uint32_t j = 0; j < hd; j++) ss += sh[j] * sh[j];
            ss /= (float)hd;
            ss += 1e-5f;
            ss = 1.0f / sqrtf(ss);
            sh[256] = ss;
        }
        __syncthreads();
        const float ss = sh[256];
        for (uint32_t j = tid; j < hd; j += blockDim.x) sh[j] = nw[j] * (ss * sh[j]);
        __syncthreads();
    }
    if (!a.rope_cos) { for (uint32_t j = tid; j < hd; j += blockDim.x) dst[j] = sh[j]; return; }
    const float *fcr = a.rope_cos + (size_t)pos * half, *fci = a.rope_sin + (size_t)pos * half;
    for (uint32_t p = tid; p < half; p += blockDim.x) {
        const float c = fcr[p], s = fci[p];
        if (a.rope_qwen3) {                                        // infer.c:692-706
            const float x0 = sh[p], x1 = sh[p + half];
            dst[p] = x0 * c - x1 * s;
            dst[p + half] = x1 * c + x0 * s;
        } else {                                                   // infer.c:681-690
            const float x0 = sh[2 * p], x1 = sh[2 * p + 1];
            dst[2 * p] = x0 * c - x1 * s;
            dst[2 * p + 1] = x0 * s + x1 * c;
        }
    }
}

__device__ __forceinline__ uint32_t strict_range(const StrictAttnArgs &a, uint32_t b) { return a.is_causal ? a.pos[b] + 1u : a.S; }

__global__ __launch_bounds__(256) void strict_scores_kernel(const StrictAttnArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, h = blockIdx.y, b = blockIdx.z, hd = a.hd;
    if (t >= strict_range(a, b)) return;
    const uint32_t kv_mul = a.n_head / a.n_kv_head;
    const float *qh = a.q + (size_t)b * a.q_dim + (size_t)h * hd;
    const float *kt = a.kcache + (((size_t)(a.slot0 + b) * a.n_layer + a.layer) * a.S + t) * a.kv_dim + (size_t)(h / kv_mul) * hd;
    float score = 0.0f;
    for (uint32_t i = 0; i < hd; i++) score += qh[i] * kt[i];
    score /= sqrtf((float)hd);
    a.att[((size_t)b * a.n_head + h) * a.S + t] = score;
}

__global__ __launch_bounds__(256) void strict_softmax_kernel(const StrictAttnArgs a) {
    __shared__ float red[256];
    const uint32_t h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const uint32_t range = strict_range(a, b);
    float *x = a.att + ((size_t)b * a.n_head + h) * a.S;
    float m = -INFINITY;                                           // the maximum does not depend on the scan order
    for (uint32_t t = tid; t < range; t += blockDim.x) m = fmaxf(m, x[t]);
    red[tid] = m;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    m = red[0];
    __syncthreads();
    for (uint32_t t = tid; t < range; t += blockDim.x) x[t] = nano_exact::exact_expf(x[t] - m, nano_exact::kExp2Tab);
    __syncthreads();
    if (tid == 0) {
        float sum = 0.0f;
        for (uint32_t t = 0; t < range; t++) sum += x[t];
        red[0] = sum;
    }
    __syncthreads();
    const float sum = red[0];
    for (uint32_t t = tid; t < range; t += blockDim.x) x[t] /= sum;
}

__global__ __launch_bounds__(256) void strict_av_kernel(const StrictAttnArgs a) {
    const uint32_t h = blockIdx.x, b = blockIdx.y, i = threadIdx.x, hd = a.hd;
    if (i >= hd) return;
    const uint32_t kv_mul = a.n_head / a.n_kv_head, range = strict_range(a, b);
    const float *att = a.att + ((size_t)b * a.n_head + h) * a.S;
    const float *vt = a.vcache + (((size_t)(a.slot0 + b) * a.n_layer + a.layer) * a.S) * a.kv_dim + (size_t)(h / kv_mul) * hd + i;
    float o = 0.0f;
    for (uint32_t t = 0; t < range; t++) o += att[t] * vt[(size_t)t * a.kv_dim];
    a.xba[(size_t)b * a.q_dim + (size_t)h * hd + i] = o;
}

__global__ __launch_bounds__(256) void strict_swiglu_kernel(float *hb, const float *hb2, uint32_t n, uint32_t bstride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= n) return;
    float val = hb[(size_t)b * bstride + i];
    val *= (1.0f / (1.0f + nano_exact::exact_expf(-val, nano_exact::kExp2Tab)));
    val *= hb2[(size_t)b * bstride + i];
    hb[(size_t)b * bstride + i] = val;
}

// 64 rows per workgroup, thread r owns row r: 64x64 tiles go through LDS (coalesced loads), the row's products are
// added in column order.  out[b][pos[b]*pstride + row] (= old + val with `resid`).
__global__ __launch_bounds__(64) void strict_matmul_f32_kernel(float *out, const float *x, const float *w, uint32_t n, uint32_t d,
                                                               uint32_t x_bstride, uint32_t out_bstride, uint32_t out_pstride, const uint32_t *pos, int resid) {
    __shared__ float tile[64][65];
    __shared__ float xs[64];
    const uint32_t r = threadIdx.x, row0 = blockIdx.x * 64u, b = blockIdx.y;
    const float *xb = x + (size_t)b * x_bstride;
    float val = 0.0f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        __syncthreads();
        for (uint32_t k = 0; k < 64; k++) tile[k][r] = (row0 + k < d && c0 + r < n) ? w[(size_t)(row0 + k) * n + c0 + r] : 0.0f;
        xs[r] = (c0 + r < n) ? xb[c0 + r] : 0.0f;
        __syncthreads();
        const uint32_t lim = (n - c0 < 64u) ? n - c0 : 64u;
        for (uint32_t j = 0; j < lim; j++) val += tile[r][j] * xs[j];
    }
    if (row0 + r < d) {
        float *o = out + (size_t)b * out_bstride + (out_pstride ? (size_t)pos[b] * out_pstride : 0) + row0 + r;
        *o = resid ? *o + val : val;
    }
}

}  // namespace

hipError_t launch_strict_rmsnorm(float *o, const float *x, const float *w, uint32_t n, uint32_t nvec, uint32_t x_stride, uint32_t o_stride, hipStream_t st) {
    hipLaunchKernelGGL(strict_rmsnorm_kernel, dim3(nvec), dim3(256), (n + 1) * sizeof(float), st, o, x, w, n, x_stride, o_stride);
    return hipGetLastError();
}
hipError_t launch_strict_qk(const StrictAttnArgs &a, uint32_t nb, hipStream_t st) {
    if (a.hd > 256 || (a.hd & 1u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(strict_qk_kernel, dim3(a.n_head + a.n_kv_head, nb), dim3(128), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_strict_attention(const StrictAttnArgs &a, uint32_t nb, hipStream_t st) {
    if (a.hd > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(strict_scores_kernel, dim3((a.S + 255) / 256, a.n_head, nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(strict_softmax_kernel, dim3(a.n_head, nb), dim3(256), 0, st, a);
    hipLaunchKernelGGL(strict_av_kernel, dim3(a.n_head, nb), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_strict_swiglu(float *hb, const float *hb2, uint32_t n, uint32_t nb, uint32_t bstride, hipStream_t st) {
    hipLaunchKernelGGL(strict_swiglu_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, st, hb, hb2, n, bstride);
    return hipGetLastError();
}
hipError_t launch_strict_matmul_f32(float *out, const float *x, const float *w, uint32_t n, uint32_t d, uint32_t nb, uint32_t x_bstride,
                                    uint32_t out_bstride, uint32_t out_pstride, const uint32_t *pos, int resid, hipStream_t st) {
    hipLaunchKernelGGL(strict_matmul_f32_kernel, dim3((d + 63) / 64, nb), dim3(64), 0, st, out, x, w, n, d, x_bstride, out_bstride, out_pstride, pos, resid);
    return hipGetLastError();
}

}  // namespace nano
