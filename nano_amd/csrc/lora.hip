// lora.hip -- LoRA side branches of the Nano architecture on the device (reference infer/infer.c:792-808, 898-903;
// file format / loader infer.c:434-498).  FP32 low-rank pairs (A [rank][E], B [out][rank]) on q, k, v and o:
//     q += (alpha/rank) * B_q (A_q xb),  same for k and v,  xb2 += (alpha/rank) * B_o (A_o xba)
// where xb = rmsnorm(x, rms_attn) and xba = the attention output.  The matrices are tiny (rank x E floats): one
// workgroup per sequence, everything through LDS; two extra launches per layer, only when a module is attached.
//   lora_qkv_kernel : after the QKV GEMV, before attention (q / raw k live in scratch, v in its cache row);
//   lora_o_kernel   : after attention, before the Wo GEMV; writes o1 = (alpha/rank) * B_o (A_o xba) which the Wo
//                     GEMV's residual epilogue adds in the reference's order  x += (Wo xba + o1).
// A-side dot products (length E) are wave tree sums (the reference's matmul adds sequentially: tolerance 1e-5);
// the B-side (length rank), the scaling and the accumulation follow the reference's operation order exactly.
#include "device_common.h"
#include "kernels.h"

namespace nano {

namespace {

// t[j] = A[j,:] . v  for j < nj, A row-major [nj][n]; v in LDS; result to LDS.  One wave per output, round-robin.
__device__ __forceinline__ void lowrank_down(const float *A, uint32_t nj, uint32_t n, const float *v, float *t) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (uint32_t j = wid; j < nj; j += nw) {
        const float *row = A + (size_t)j * n;
        float acc = 0.0f;
        for (uint32_t i = lane * 4u; i < n; i += 256u) {
            const float4 w = *reinterpret_cast<const float4 *>(row + i);
            const float4 x = *reinterpret_cast<const float4 *>(v + i);
            acc = __builtin_fmaf(w.x, x.x, acc); acc = __builtin_fmaf(w.y, x.y, acc); acc = __builtin_fmaf(w.z, x.z, acc); acc = __builtin_fmaf(w.w, x.w, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) t[j] = acc;
    }
}
// (alpha/rank) * B[i,:] . t  with the reference's sequential order (matmul then scale, infer.c:637-651, 589-593)
__device__ __forceinline__ float lowrank_up(const float *B, uint32_t i, uint32_t rank, const float *t, float s) {
    float val = 0.0f;
    const float *row = B + (size_t)i * rank;
    for (uint32_t j = 0; j < rank; j++) val += row[j] * t[j];
    return val * s;
}

__global__ __launch_bounds__(256) void lora_qkv_kernel(const LoraArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, E = a.E, KD = a.KD, rank = a.rank;
    float *xb = sm, *t = sm + ((E + 3) & ~3u), *red = t + 3 * rank;
    const float *x = a.x + (size_t)b * E;
    float acc = 0.0f;                                   // rmsnorm (infer.c:601-614), tree order
    for (uint32_t i = tid; i < E; i += 256u) acc += x[i] * x[i];
    float ss = block_sum(acc, red);
    ss /= (float)E; ss += 1e-5f; ss = 1.0f / sqrtf(ss);
    for (uint32_t i = tid; i < E; i += 256u) xb[i] = a.norm_w[i] * (ss * x[i]);
    __syncthreads();
    lowrank_down(a.qa, rank, E, xb, t);
    lowrank_down(a.ka, rank, E, xb, t + rank);
    lowrank_down(a.va, rank, E, xb, t + 2 * rank);
    __syncthreads();
    const float s = (float)a.alpha / (float)rank;
    float *q = a.q + (size_t)b * E, *k = a.kraw + (size_t)b * KD;
    float *v = a.v + (size_t)b * a.v_bstride + (size_t)a.pos[b] * KD;
    for (uint32_t i = tid; i < E; i += 256u) q[i] += lowrank_up(a.qb, i, rank, t, s);
    for (uint32_t i = tid; i < KD; i += 256u) { k[i] += lowrank_up(a.kb, i, rank, t + rank, s); v[i] += lowrank_up(a.vb, i, rank, t + 2 * rank, s); }
}

__global__ __launch_bounds__(256) void lora_o_kernel(const LoraArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, E = a.E, rank = a.rank;
    float *xba = sm, *t = sm + ((E + 3) & ~3u);
    for (uint32_t i = tid; i < E; i += 256u) xba[i] = a.x[(size_t)b * E + i];
    __syncthreads();
    lowrank_down(a.qa, rank, E, xba, t);
    __syncthreads();
    const float s = (float)a.alpha / (float)rank;
    for (uint32_t i = tid; i < E; i += 256u) a.q[(size_t)b * E + i] = lowrank_up(a.qb, i, rank, t, s);
}

}  // namespace

hipError_t launch_lora_qkv(const LoraArgs &a, uint32_t nb, hipStream_t st) {
    const size_t lds = (((size_t)a.E + 3) & ~(size_t)3) * 4 + 3 * (size_t)a.rank * 4 + 32 * 4;
    hipLaunchKernelGGL(lora_qkv_kernel, dim3(nb), dim3(256), lds, st, a);
    return hipGetLastError();
}
// x = xba [nb][E] in, q = o1 [nb][E] out, qa / qb = the o pair
hipError_t launch_lora_o(const LoraArgs &a, uint32_t nb, hipStream_t st) {
    const size_t lds = (((size_t)a.E + 3) & ~(size_t)3) * 4 + (size_t)a.rank * 4;
    hipLaunchKernelGGL(lora_o_kernel, dim3(nb), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace nano
