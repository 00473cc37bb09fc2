// misc.hip -- the small kernels around the GEMVs: embedding row fetch (with on-the-fly dequantization),
// arg-max sampling, and stand-alone versions of the fused prologue pieces used by the operator tests.
#include <float.h>

#include "device_common.h"
#include "kernels.h"

namespace nano {

// ---- embedding lookup ------------------------------------------------------------------------------
// The reference dequantizes the whole table to fp32 at load (infer/infer.c:126-127,147-149) and
// memcpy's one row per token (infer.c:987-988).  Here the row is dequantized when it is needed; the
// floats are the same: Q80 q[i]*s[i/gs] (tensor.c:15-19), Q4K (float)nibble*s - b (tensor.c:253-278).
// sequence b: token `tok` at position `p` -> x[b] (and the staged RoPE row / pool row of p); the calling workgroup's threads share the work
__device__ __forceinline__ void embed_row(const EmbedArgs &a, uint32_t b, uint32_t tok, uint32_t p, bool stage_pos) {
    float *x = a.x + (size_t)b * a.x_bstride;
    const uint32_t E = a.E;
    // paged KV cache: the pool row of this step's position.  (p >> 6) < pt_entries: the greedy loop's arg-max kernel embeds the token of
    // the position AFTER the last step too -- with max_seq_len a multiple of 64 that position's block lies one entry past the slot's row
    // of the table (round-3 advice: a device read out of bounds; the value was never used)
    if (stage_pos && a.kvrow && threadIdx.x == 0 && (p >> 6) < a.pt_entries)
        a.kvrow[b] = a.pt_rows[(size_t)b * a.pt_bstride + (p >> 6)] + (p & 63u);
    if (stage_pos && a.rope_cur) {
        for (uint32_t i = threadIdx.x; i < a.half; i += blockDim.x) {
            a.rope_cur[(size_t)b * 2 * a.half + i] = a.rope_cos[(size_t)p * a.half + i];
            a.rope_cur[(size_t)b * 2 * a.half + a.half + i] = a.rope_sin[(size_t)p * a.half + i];
        }
    }
    if (a.quant == 0x00u) {
        const float4 *row = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a.tok) + (size_t)tok * E);
        for (uint32_t i = threadIdx.x; i < E / 4; i += blockDim.x) reinterpret_cast<float4 *>(x)[i] = row[i];
    } else if (a.quant == 0x80u) {
        const int8_t *row = reinterpret_cast<const int8_t *>(a.tok) + (size_t)tok * E;      // E % 4 == 0, gs % 4 == 0
        const float *s = a.tok_s + ((size_t)tok * E) / a.gs;
        for (uint32_t i = threadIdx.x * 4; i < E; i += blockDim.x * 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(row + i);
            const float sc = s[i / a.gs];
            reinterpret_cast<float4 *>(x + i)[0] = make_float4((float)(int8_t)(w & 0xff) * sc, (float)(int8_t)((w >> 8) & 0xff) * sc,
                                                               (float)(int8_t)((w >> 16) & 0xff) * sc, (float)(int8_t)(w >> 24) * sc);
        }
    } else {
        const uint32_t bpl = (E + 255) / 256;
        const uint8_t *blocks = reinterpret_cast<const uint8_t *>(a.tok) + (size_t)tok * bpl * 160;
        for (uint32_t it = threadIdx.x; it < bpl * 256; it += blockDim.x) {       // all blocks in flight at once
            const uint32_t j = it >> 8, k = it & 255;
            const uint8_t *blk = blocks + (size_t)j * 160;
            const uint32_t d = (E >= (j + 1) * 256) ? 256 : (E - j * 256);
            const uint32_t len = *reinterpret_cast<const uint32_t *>(blk + 4);
            if (k < len) {
                const float s_scale = *reinterpret_cast<const float *>(blk + 12);
                const float s_bias = *reinterpret_cast<const float *>(blk + 16);
                uint32_t s6, b6;
                q4k_unpack6(*reinterpret_cast<const uint32_t *>(blk + 20), *reinterpret_cast<const uint32_t *>(blk + 24),
                            *reinterpret_cast<const uint32_t *>(blk + 28), (int)(k >> 5), s6, b6);
                const float sc = (float)s6 * s_scale, bb = (float)b6 * s_bias;
                const uint8_t byte = blk[32 + (k >> 1)];
                const uint32_t nib = (k & 1) ? (uint32_t)(byte >> 4) : (uint32_t)(byte & 0x0f);
                x[(size_t)j * d + k] = (float)nib * sc - bb;        // j*d destination offset as tensor.c:339
            }
        }
    }
}

__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs a) {
    const int b = blockIdx.x;
    if (a.tick && b == 0 && threadIdx.x == 0) a.tick[0] = a.tick[0] + 1u;       // a new step: a new epoch for its in-launch hand-offs
    embed_row(a, (uint32_t)b, a.tokens[b], a.pos ? a.pos[b] : 0u, true);
}

hipError_t launch_embed(const EmbedArgs &a, uint32_t nb, hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3(nb), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- arg-max (reference sample_argmax, infer.c:1026-1037: first maximum, strict '>') -----------------
__global__ __launch_bounds__(1024) void argmax_kernel(const ArgmaxArgs a) {
    __shared__ float sval[16];
    __shared__ uint32_t sidx[16];
    __shared__ uint32_t s_next[2];                            // greedy loop, fused embedding: the picked token and its position
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *x = a.logits + (size_t)b * a.bstride;
    float best = -INFINITY;
    uint32_t bi = 0xffffffffu;
    if (a.tile_max) {
        // the classifier GEMV already reduced every tile to (max, first row): scan ntiles pairs, not V logits
        const float *tm = a.tile_max + (size_t)b * a.ntiles * 2;
        for (uint32_t t = tid; t < a.ntiles; t += blockDim.x) {
            const float v = tm[2 * t];
            const uint32_t i = __float_as_uint(tm[2 * t + 1]);
            if (i != 0xffffffffu && (bi == 0xffffffffu || v > best || (v == best && i < bi))) { best = v; bi = i; }   // partials are NOT row-ordered
        }
    } else {
        // thread t scans float4 t, t+1024, ...: ascending indices per thread, 4 loads in flight
        const uint32_t V4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? a.V / 4 : 0;
        auto upd = [&](float v, uint32_t i) { if (bi == 0xffffffffu || v > best) { best = v; bi = i; } };
        uint32_t i = tid;
        for (; i + 3 * blockDim.x < V4; i += 4 * blockDim.x) {
            const float4 v0 = reinterpret_cast<const float4 *>(x)[i], v1 = reinterpret_cast<const float4 *>(x)[i + blockDim.x];
            const float4 v2 = reinterpret_cast<const float4 *>(x)[i + 2 * blockDim.x], v3 = reinterpret_cast<const float4 *>(x)[i + 3 * blockDim.x];
            upd(v0.x, 4 * i); upd(v0.y, 4 * i + 1); upd(v0.z, 4 * i + 2); upd(v0.w, 4 * i + 3);
            const uint32_t i1 = i + blockDim.x, i2 = i + 2 * blockDim.x, i3 = i + 3 * blockDim.x;
            upd(v1.x, 4 * i1); upd(v1.y, 4 * i1 + 1); upd(v1.z, 4 * i1 + 2); upd(v1.w, 4 * i1 + 3);
            upd(v2.x, 4 * i2); upd(v2.y, 4 * i2 + 1); upd(v2.z, 4 * i2 + 2); upd(v2.w, 4 * i2 + 3);
            upd(v3.x, 4 * i3); upd(v3.y, 4 * i3 + 1); upd(v3.z, 4 * i3 + 2); upd(v3.w, 4 * i3 + 3);
        }
        for (; i < V4; i += blockDim.x) {
            const float4 v0 = reinterpret_cast<const float4 *>(x)[i];
            upd(v0.x, 4 * i); upd(v0.y, 4 * i + 1); upd(v0.z, 4 * i + 2); upd(v0.w, 4 * i + 3);
        }
        for (uint32_t k = V4 * 4 + tid; k < a.V; k += blockDim.x) upd(x[k], k);
    }
    // combine: larger value wins; equal values -> smaller index (== first maximum in scan order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    const int lane = tid & 63, wid = tid >> 6;
    if (lane == 0) { sval[wid] = best; sidx[wid] = bi; }
    __syncthreads();
    if (tid == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; w++) {
            const float ov = sval[w]; const uint32_t oi = sidx[w];
            if (oi != 0xffffffffu && (bi == 0xffffffffu || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
        }
        if (bi == 0xffffffffu) bi = 0;
        a.out[b] = bi;
        if (a.tokens) a.tokens[b] = bi;
        if (a.trace) a.trace[(size_t)(a.pos[b] - a.pos0[b]) * a.nb + b] = bi;
        if (a.tokens) { const uint32_t np = a.pos[b] + 1; a.pos[b] = np; s_next[0] = bi; s_next[1] = np; }
    }
    if (a.tokens && a.emb.x) {                                // greedy loop: embed the picked token at its next position right here
        __syncthreads();
        const uint32_t np = s_next[1];
        if (a.emb.tick && b == 0 && tid == 0) a.emb.tick[0] = a.emb.tick[0] + 1u;      // (this kernel is the next step's first)
        embed_row(a.emb, (uint32_t)b, s_next[0], np, np < a.rope_rows);
    }
}

hipError_t launch_argmax(const ArgmaxArgs &a, uint32_t nb, hipStream_t st) {
    hipLaunchKernelGGL(argmax_kernel, dim3(nb), dim3(1024), 0, st, a);
    return hipGetLastError();
}


// ---- stand-alone operator kernels (operator parity tests) ----------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_kernel(float *out, const float *x, const float *w, uint32_t n) {
    __shared__ float red[32];
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += x[i] * x[i];
    float ss = block_sum(acc, red);
    ss /= (float)n; ss += 1e-5f; ss = 1.0f / sqrtf(ss);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) out[i] = w[i] * (ss * x[i]);
}
hipError_t launch_rmsnorm(float *out, const float *x, const float *w, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(1), dim3(256), 0, st, out, x, w, n);
    return hipGetLastError();
}

// same thread mapping as the fused GEMV prologue (4 elements per thread, gs/4 threads per group)
__global__ __launch_bounds__(256) void quantize_q80_kernel(const float *x, uint32_t n, uint32_t gs, int8_t *q, float *s) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int tpg = (int)gs / 4;
    const int iters = ((int)n + nthr * 4 - 1) / (nthr * 4);
    for (int it = 0; it < iters; it++) {
        const int i = (it * nthr + tid) * 4;
        const bool act = i < (int)n;
        float4 v = act ? *reinterpret_cast<const float4 *>(x + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = group_max(m, tpg);
        const float scale = div_const<127>(m);
        if (act) {
            q[i] = (int8_t)q80_quant1(v.x, scale); q[i + 1] = (int8_t)q80_quant1(v.y, scale);
            q[i + 2] = (int8_t)q80_quant1(v.z, scale); q[i + 3] = (int8_t)q80_quant1(v.w, scale);
            if ((tid % tpg) == 0) s[i / (int)gs] = scale;
        }
    }
}
hipError_t launch_quantize_q80(const float *x, uint32_t n, uint32_t gs, int8_t *q, float *s, hipStream_t st) {
    hipLaunchKernelGGL(quantize_q80_kernel, dim3(1), dim3(256), 0, st, x, n, gs, q, s);
    return hipGetLastError();
}

__global__ void swiglu_kernel(float *hb, const float *hb2, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = hb[i];
        v *= (1.0f / (1.0f + expf(-v)));
        v *= hb2[i];
        hb[i] = v;
    }
}
hipError_t launch_swiglu(float *hb, const float *hb2, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(swiglu_kernel, dim3((n + 255) / 256), dim3(256), 0, st, hb, hb2, n);
    return hipGetLastError();
}

__global__ void rope_kernel(float *h, uint32_t hd, const float *fcr, const float *fci, int qwen3) {
    const uint32_t i = threadIdx.x, half = hd / 2;
    if (i >= half) return;
    const float c = fcr[i], s = fci[i];
    if (qwen3) {
        const float a = h[i], b = h[i + half];
        h[i] = a * c - b * s; h[i + half] = b * c + a * s;
    } else {
        const float a = h[2 * i], b = h[2 * i + 1];
        h[2 * i] = a * c - b * s; h[2 * i + 1] = a * s + b * c;
    }
}
hipError_t launch_rope(float *head, uint32_t hd, const float *fcr, const float *fci, int qwen3, hipStream_t st) {
    hipLaunchKernelGGL(rope_kernel, dim3(1), dim3(256), 0, st, head, hd, fcr, fci, qwen3);
    return hipGetLastError();
}

// ---- read-bandwidth microbenchmark -------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *buf, size_t n16, float *sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(buf + i);
        const u32x4 b = __builtin_nontemporal_load(buf + i + stride);
        const u32x4 c = __builtin_nontemporal_load(buf + i + 2 * stride);
        const u32x4 d = __builtin_nontemporal_load(buf + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= buf[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = 1.0f;   // keep the loads alive
}
// the same reader on SOME of the chip only (tests of the in-launch hand-offs under uneven load): workgroup b runs on XCD b % 8 (observed
// placement); the workgroups of the XCDs outside `xcd_mask` leave at once
__global__ __launch_bounds__(256) void stream_read_masked_kernel(const u32x4 *buf, size_t n16, float *sink, uint32_t xcd_mask) {
    if (!((xcd_mask >> (blockIdx.x & 7u)) & 1u)) return;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) acc ^= __builtin_nontemporal_load(buf + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) *sink = 1.0f;
}
hipError_t launch_stream_read_masked(const void *buf, size_t bytes, float *sink, uint32_t xcd_mask, uint32_t wgs, hipStream_t st) {
    hipLaunchKernelGGL(stream_read_masked_kernel, dim3(wgs), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(buf), bytes / 16, sink, xcd_mask);
    return hipGetLastError();
}
hipError_t launch_stream_read(const void *buf, size_t bytes, float *sink, hipStream_t st) {
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(buf), bytes / 16, sink);
    return hipGetLastError();
}

}  // namespace nano
