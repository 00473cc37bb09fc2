// device_common.h -- shared device-side helpers for the gfx950 kernels (wave64, CDNA4).
//
// Numerics contract (DESIGN.md "Parity"): the build uses -ffp-contract=off and correctly rounded
// fp32 divide/sqrt, so every elementwise expression below evaluates exactly like the reference's
// C expression compiled without FMA contraction.  Reductions that the reference performs as one
// sequential float chain (rmsnorm sum of squares, attention dot / softmax sums) are done as
// wave-parallel trees here (stated tolerance 1e-5 relative); the quantized GEMVs keep the
// reference's group order (bit-exact float combine).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NANO_WAVE 64

namespace nano {

// ---- streaming (non-temporal) 16-byte loads: weights are read exactly once per step ---------------
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_stream_i4(const void *p) {
    const i32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 ld_stream_f4(const void *p) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---- cross-lane --------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over aligned sub-groups of `width` lanes (width power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int group_sum_i(int v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum / max for blocks of up to 1024 threads. `red` is >= 16 floats of LDS scratch.
// All threads get the result.  Two __syncthreads().
__device__ __forceinline__ float block_sum(float v, float *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float *red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; i++) t = fmaxf(t, red[i]);
    return t;
}

// ---- Q80 activation quantizer (reference infer/tensor.c:21-46) -----------------------------------
// q = (int8) round(x / scale), round = half away from zero evaluated in double by the reference;
// for a float argument roundf() returns the same integer.  0/0 (all-zero group) -> NaN -> the
// reference's cast is UB and x86-64 yields 0: we return 0.
__device__ __forceinline__ int q80_quant1(float x, float scale) {
    float qv = x / scale;
    float r = roundf(qv);
    return (r != r) ? 0 : (int)r;
}

// ---- Q4K helpers (reference infer/tensor.c:4-9,113-141) -------------------------------------------
__device__ __forceinline__ int nearest_int_magic(float v) {
    float t = v + 12582912.f;
    return (__float_as_int(t) & 0x007fffff) - 0x00400000;
}

// 6-bit scale / bias of group g (0..7) from the 12 packed bytes sb[0..11] given as three dwords
__device__ __forceinline__ void q4k_unpack6(uint32_t sb0, uint32_t sb1, uint32_t sb2, int g, uint32_t &s6, uint32_t &b6) {
    const int i = g & 3;
    const uint32_t s_lo = (sb0 >> (8 * i)) & 0xffu;     // sb[i]
    const uint32_t b_lo = (sb1 >> (8 * i)) & 0xffu;     // sb[4+i]
    const uint32_t mix  = (sb2 >> (8 * i)) & 0xffu;     // sb[8+i]
    if (g < 4) { s6 = s_lo & 0x3fu; b6 = b_lo & 0x3fu; }
    else {
        s6 = (((s_lo >> 6) << 4) | (mix & 0x0fu)) & 0x3fu;
        b6 = (((b_lo >> 6) << 4) | ((mix & 0xf0u) >> 4)) & 0x3fu;
    }
}

}  // namespace nano
