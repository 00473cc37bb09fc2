// attn.hip -- single-token GQA attention over the FP32 KV cache (reference infer/infer.c:810-879),
// split over the sequence ("flash-decoding"): nsplit workgroups per (head, sequence) each take every
// nsplit-th block of timesteps and emit an UNNORMALISED partial  o_s = sum_t exp(s_t - m_s) v_t  with
// its (m_s, l_s = sum_t exp(s_t - m_s)); the consumer (the Wo GEMV's prologue, gemv.hip, or
// attn_combine_kernel below) forms  sum_s o_s e^{m_s - M} / sum_s l_s e^{m_s - M}  -- algebraically the
// reference's softmax(q.k/sqrt(hd)) . V; rounding differs at the 1e-7 level (tolerance 1e-5).
//
// The kernel also owns the per-head work the reference does between the QKV GEMVs and the attention
// loop, because it is head-local:
//   * Qwen3: rmsnorm(q_head, q_norm), rmsnorm(k_head, k_norm) (infer.c:824-835) then half-split RoPE
//     (rope_qwen3, infer.c:692-706);  Nano/Qwen2: adjacent-pair RoPE (rope, infer.c:681-690);
//   * the finished k row is written to cache row `pos` by split 0 of the first head of each KV group
//     (raw k comes from the QKV GEMV through a scratch row; v goes to the cache directly); every
//     workgroup uses its own LDS copy of that row, so nobody reads the row while it is written.
// KV rows are read as float4 by sub-groups of hd/4 lanes (coalesced 4*hd-byte rows), 4 rows in flight
// per sub-group.
#include "device_common.h"
#include "kernels.h"

namespace nano {

__device__ __forceinline__ int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__global__ __launch_bounds__(256) void attention_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int h = blockIdx.x, b = blockIdx.y, split = blockIdx.z, tid = threadIdx.x;
    const int nsplit = (int)a.nsplit;
    const int hd = (int)a.hd, half = hd >> 1;
    const int kv_mul = (int)(a.n_head / a.n_kv_head);
    const int g = h / kv_mul;
    const uint32_t p = a.fixed_range ? (a.fixed_range - 1) : a.pos[b];
    const uint32_t range = a.fixed_range ? a.fixed_range : (a.is_causal ? (p + 1) : a.S);

    // LDS carve (all multiples of 16 bytes): qh[hd] kh[hd] red[32] part[256*4] att[local scores]
    const int hd4 = (hd + 3) & ~3;
    float *qh = reinterpret_cast<float *>(smem);
    float *kh = qh + hd4;
    float *red = kh + hd4;
    float *part = red + 32;
    float *att = part + 1024;

    const float *qg = a.q + (size_t)b * a.q_dim + (size_t)h * hd;
    const size_t slot_rows = (size_t)b * a.cache_bstride_rows + (size_t)a.layer * a.S;
    float *kc = a.kcache + slot_rows * a.kv_dim + (size_t)g * hd;      // row t at kc + t*kv_dim
    const float *vc = a.vcache + slot_rows * a.kv_dim + (size_t)g * hd;
    const bool fresh_k = a.kraw != nullptr;

    for (int i = tid; i < hd; i += 256) {
        qh[i] = qg[i];
        kh[i] = fresh_k ? a.kraw[(size_t)b * a.kv_dim + (size_t)g * hd + i] : 0.0f;
    }
    __syncthreads();

    if (fresh_k) {
        if (a.q_norm) {
            float sq = 0.0f, sk = 0.0f;
            for (int i = tid; i < hd; i += 256) { sq += qh[i] * qh[i]; sk += kh[i] * kh[i]; }
            sq = block_sum(sq, red);
            sk = block_sum(sk, red + 16);
            sq /= (float)hd; sq += 1e-5f; sq = 1.0f / sqrtf(sq);
            sk /= (float)hd; sk += 1e-5f; sk = 1.0f / sqrtf(sk);
            __syncthreads();
            for (int i = tid; i < hd; i += 256) {
                qh[i] = a.q_norm[i] * (sq * qh[i]);
                kh[i] = a.k_norm[i] * (sk * kh[i]);
            }
            __syncthreads();
        }
        if (a.rope_cos) {
            const float *fcr = a.rope_cos + (size_t)p * half;
            const float *fci = a.rope_sin + (size_t)p * half;
            for (int i = tid; i < half; i += 256) {
                const float c = fcr[i], s = fci[i];
                if (a.rope_qwen3) {
                    const float q0 = qh[i], q1 = qh[i + half];
                    qh[i] = q0 * c - q1 * s;  qh[i + half] = q1 * c + q0 * s;
                    const float k0 = kh[i], k1 = kh[i + half];
                    kh[i] = k0 * c - k1 * s;  kh[i + half] = k1 * c + k0 * s;
                } else {
                    const float q0 = qh[2 * i], q1 = qh[2 * i + 1];
                    qh[2 * i] = q0 * c - q1 * s;  qh[2 * i + 1] = q0 * s + q1 * c;
                    const float k0 = kh[2 * i], k1 = kh[2 * i + 1];
                    kh[2 * i] = k0 * c - k1 * s;  kh[2 * i + 1] = k0 * s + k1 * c;
                }
            }
            __syncthreads();
        }
        if (split == 0 && (h % kv_mul) == 0)
            for (int i = tid; i < hd; i += 256) kc[(size_t)p * a.kv_dim + i] = kh[i];
        if (split == 0 && a.q_out)
            for (int i = tid; i < hd; i += 256) a.q_out[(size_t)b * a.q_dim + (size_t)h * hd + i] = qh[i];
    }

    // ---- scores of this split's timestep blocks ---------------------------------------------------
    const int lanes = hd >> 2;                    // float4 lanes per KV row
    const int LPR = next_pow2(lanes);             // sub-group width
    const int nsub = 256 / LPR;                   // timesteps per block
    const int sub = tid / LPR, j = tid % LPR;
    const bool jact = j < lanes;
    const float4 qv = jact ? *reinterpret_cast<const float4 *>(qh + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float sq_hd = sqrtf((float)hd);
    const uint32_t nblk = (range + nsub - 1) / nsub;           // timestep blocks overall
    const uint32_t myblk = (nblk > (uint32_t)split) ? (nblk - split + nsplit - 1) / nsplit : 0;   // blocks of this split
    const uint32_t nloc = myblk * nsub;                         // local score slots

    for (uint32_t i0 = 0; i0 < myblk; i0 += 4) {
        float4 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = ((i0 + u) * nsplit + split) * nsub + sub;
            const bool ok = jact && (i0 + u) < myblk && t < range && !(fresh_k && t == p);
            kv[u] = ok ? *reinterpret_cast<const float4 *>(kc + (size_t)t * a.kv_dim + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = ((i0 + u) * nsplit + split) * nsub + sub;
            if (fresh_k && t == p && jact) kv[u] = *reinterpret_cast<const float4 *>(kh + 4 * j);
            float acc = qv.x * kv[u].x;
            acc += qv.y * kv[u].y; acc += qv.z * kv[u].z; acc += qv.w * kv[u].w;
            const float d = group_sum(acc, LPR);
            if (j == 0 && (i0 + u) < myblk) att[(i0 + u) * nsub + sub] = (t < range) ? d / sq_hd : -INFINITY;
        }
    }
    __syncthreads();

    // ---- local softmax numerators (reference infer.c:616-634, normalisation deferred to the combine) ----
    float m = -INFINITY;
    for (uint32_t i = tid; i < nloc; i += 256) m = fmaxf(m, att[i]);
    m = block_max(m, red);
    float sum = 0.0f;
    for (uint32_t i = tid; i < nloc; i += 256) {
        const float e = (att[i] == -INFINITY) ? 0.0f : expf(att[i] - m);
        att[i] = e; sum += e;
    }
    sum = block_sum(sum, red + 16);
    __syncthreads();

    // ---- weighted V sum ---------------------------------------------------------------------------------
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t i0 = 0; i0 < myblk; i0 += 4) {
        float4 vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t t = ((i0 + u) * nsplit + split) * nsub + sub;
            vv[u] = (jact && (i0 + u) < myblk && t < range) ? *reinterpret_cast<const float4 *>(vc + (size_t)t * a.kv_dim + 4 * j)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float w = ((i0 + u) < myblk) ? att[(i0 + u) * nsub + sub] : 0.0f;
            acc.x += w * vv[u].x; acc.y += w * vv[u].y; acc.z += w * vv[u].z; acc.w += w * vv[u].w;
        }
    }
    if (jact) *reinterpret_cast<float4 *>(part + (size_t)sub * hd4 + 4 * j) = acc;
    __syncthreads();
    float *po = a.out + ((size_t)b * nsplit + split) * a.q_dim + (size_t)h * hd;
    for (int i = tid; i < hd; i += 256) {
        float s = 0.0f;
        for (int sb = 0; sb < nsub; sb++) s += part[(size_t)sb * hd4 + i];
        po[i] = s;
    }
    if (tid == 0) {
        float *ml = a.ml + (((size_t)b * a.n_head + h) * nsplit + split) * 2;
        ml[0] = m; ml[1] = sum;
    }
}

uint32_t attention_nsplit(uint32_t S) {
    uint32_t n = S / 64;
    if (n < 1) n = 1;
    if (n > 8) n = 8;
    return n;
}

hipError_t launch_attention(const AttnArgs &a, uint32_t nb, hipStream_t st) {
    const uint32_t hd4 = (a.hd + 3) & ~3u;
    const uint32_t max_range = a.fixed_range ? a.fixed_range : a.S;
    const uint32_t lanes = a.hd >> 2;
    uint32_t LPR = 1; while (LPR < lanes) LPR <<= 1;
    const uint32_t nsub = 256 / LPR;
    const uint32_t nblk = (max_range + nsub - 1) / nsub;
    const uint32_t nloc = ((nblk + a.nsplit - 1) / a.nsplit + 4) * nsub;
    const size_t lds = ((size_t)2 * hd4 + 32 + 1024 + ((nloc + 3) & ~3u)) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attention_kernel, dim3(a.n_head, nb, a.nsplit), dim3(256), lds, st, a);
    return hipGetLastError();
}

// stand-alone combine (operator tests; the forward folds this into the Wo GEMV's prologue)
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

}  // namespace nano
