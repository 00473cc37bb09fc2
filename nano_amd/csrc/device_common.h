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

// Phase stamps (measurement builds only: make stamps -> libnano_mi355x_stamps.so, tools/stamp_probe.py).  In the product build
// NANO_STAMPS is 0 and every stamp site compiles to nothing.  A stamp is the shader clock (s_memtime) read by the first wave
// of a workgroup; `dep` ties the read behind the value it is meant to follow (the compiler cannot hoist it above the wait).
// NANO_STAMPS == 2 keeps only the entry stamp (slots 0 and 7) and the end stamp: the product's own schedule, timed.
#ifndef NANO_STAMPS
#define NANO_STAMPS 0
#endif
#if NANO_STAMPS
#define NANO_STAMP(buf, k, dep) do { if ((NANO_STAMPS == 1 || (k) == 0) && (buf) && threadIdx.x < 64) { unsigned long long t_; \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory"); \
        if (threadIdx.x == 0) (buf)[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (k)] = t_; \
        if ((k) == 0) { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); /* slot 7: the device-wide 100 MHz clock at entry */ \
            if (threadIdx.x == 0) (buf)[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + 7] = t_; } } } while (0)
// the last instruction of a kernel, every wave: slot k = the latest shader clock at which a wave of the workgroup ended
#define NANO_STAMP_END(buf, k) do { if (buf) { unsigned long long t_; \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0) atomicMax((buf) + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (k), t_); } } while (0)
#else
#define NANO_STAMP(buf, k, dep) do { } while (0)
#define NANO_STAMP_END(buf, k) do { } while (0)
#endif

namespace nano {

// A kernel argument the compiler would fetch (s_load) right before its first use -- in the middle of the dependent chain, one
// scalar-cache round trip each -- is asked for at the top of the kernel instead: every argument load is issued together and
// waited for once, while nothing else could run anyway.
template <class T> __device__ __forceinline__ void karg_touch(const T &v) { asm volatile("" :: "s"(v)); }

// ---- in-launch hand-offs (the fused decode launches, gemv_q80_impl.h / attn_impl.h) -----------------------------------------------
// Results travel between workgroups of ONE launch as 8-byte {tag, value} granules (one relaxed agent-scope store / load each: the
// data is the flag -- MI355X guide, Guideline 16 R2).  The tag is an EPOCH: tick * 128 + (layer + 1), where `tick` is a device
// word the step's first kernel (embed_kernel, or the arg-max kernel that embeds the next token) increments -- read from memory,
// never a launch argument (those are frozen under graph replay).  A granule left by any earlier launch carries another tag: a
// consumer that meets it WAITS (round 5 tagged with a constant 1 and relied on the other buffer of a pair being zeroed by the
// launch before: an aborted step could flip that parity and stale values were then taken for fresh ones -- round-5 advice).
// No buffer is ever zeroed, one buffer per edge, no restriction on the layer count's parity.
// tick[0] = the step counter, tick[1] = fault word XORed into the PRODUCERS' tag (0; nano_hip_debug_fault sets it so that every
// hand-off of a step fails: the test of the give-up path), tick[2] = abort flag: set by the first consumer that gives up, polled by
// the others so that a lost step does not sit out the bound once per launch; cleared by the host.
constexpr uint32_t NANO_DEVERR_G6_TILE = 1u, NANO_DEVERR_HANDOFF = 2u;
struct SlabHand { unsigned long long *buf; uint32_t *tick; uint32_t base[3], layer1; };
__device__ __forceinline__ uint2 hand_tick(const SlabHand &h) { return *reinterpret_cast<const uint2 *>(h.tick); }
__device__ __forceinline__ uint32_t hand_ctag(const uint2 t, const SlabHand &h) { return t.x * 128u + h.layer1; }              // what a consumer waits for
__device__ __forceinline__ uint32_t hand_ptag(const uint2 t, const SlabHand &h) { return (t.x * 128u + h.layer1) ^ t.y; }      // what a producer writes
__device__ __forceinline__ bool hand_aborted(const SlabHand &h) { return __hip_atomic_load(h.tick + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
__device__ __forceinline__ void hand_give_up(const SlabHand &h, uint32_t *err) {
    __hip_atomic_store(h.tick + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err) __hip_atomic_fetch_or(err, NANO_DEVERR_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- x / C for the quantizers' constant divisors (15, 63, 127) -----------------------------------------
// Three operations (multiply by RN(1/C), exact remainder by FMA, one correction) instead of the ~12 of the IEEE division
// expansion.  The result is the correctly rounded quotient -- bit-identical to x / C -- for EVERY finite float x (denormals
// included; kernels run with denormals on) except x = -0, which gives +0: checked exhaustively over all 2^32 bit patterns on
// the host for C = 15, 63, 127 (tools/div_const_check.c; the quantizers only divide maxima, which are never -0).
template <int C>
__device__ __forceinline__ float div_const(float x) {
    static_assert(C == 15 || C == 63 || C == 127, "checked constants only");
    constexpr float rc = 1.0f / (float)C;
    const float q0 = x * rc;
    const float r = __builtin_fmaf(-q0, (float)C, x);
    return __builtin_fmaf(r, rc, q0);
}

// ---- cross-lane --------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
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
