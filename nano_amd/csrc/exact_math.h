// exact_math.h — bit-exact restatements of the host arithmetic the reference's sampler relies on, usable from device
// code and (for the CPU checks in tests/) from plain C++.
//
// The reference's softmax (infer/infer.c:616-634, restated in oracle/) calls libm's expf() and sums the results in index
// order; its sampled token therefore depends on the host libm.  The build this repo is pinned against is glibc 2.35 on
// x86-64 with FMA (this image, and the GPU hosts): expf there is the table-driven double-precision algorithm published
// as ARM optimized-routines `expf` (N = 32 table, cubic polynomial), compiled with fused multiply-adds.  exact_expf()
// evaluates the same sequence of IEEE double operations, so it returns the same float for every input
// (tests/test_exact_math.py checks it against the running libm).  Table entry i is bits(2^(i/32)) - (i << 47).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NANO_HD __host__ __device__ __forceinline__
#else
#define NANO_HD static inline
#endif

namespace nano_exact {

#if defined(__HIPCC__)
__device__ __constant__
#endif
static const uint64_t kExp2Tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

NANO_HD double bits_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
NANO_HD uint64_t f64_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
NANO_HD uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
NANO_HD float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// expf for any float: the whole published routine (x <= 0 is the softmax argument x - max; SwiGLU's expf(-h) takes
// either sign and may overflow to +inf above log(2^128), reference infer/infer.c:941).
NANO_HD float exact_expf(float x, const uint64_t *tab) {
    const uint32_t ix = f32_bits(x);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop > 0x42a) {                                   // |x| >= 88: the slow cases of the published routine
        if (ix == 0xff800000u) return 0.0f;                 // -inf
        if (abstop > 0x7f7) return x + x;                   // nan, +inf
        if (x > 0x1.62e42ep6f) return bits_f32(0x7f800000u);// overflow: 2^97 * 2^97 rounds to +inf
        if (x < -0x1.9fe368p6f) return 0.0f;                // underflow: 2^-95 * 2^-95 rounds to +0
        if (x < -0x1.9d1d9ep6f) return bits_f32(1u);        // "may underflow": 0x1.4p-75^2 rounds to 2^-149
    }
    const double xd = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+5, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const double kd0 = __builtin_fma(InvLn2N, xd, Shift);
    const uint64_t ki = f64_bits(kd0);
    const double kd = kd0 - Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    const double s = bits_f64(tab[ki & 31] + (ki << 47));
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}
// the sampler's name for it (its arguments are never positive)
NANO_HD float exact_expf_nonpos(float x, const uint64_t *tab) { return exact_expf(x, tab); }


// ---- sequential float sums, evaluated in parallel ----------------------------------------------------------------
// The reference adds the V softmax numerators into one float in index order (infer/infer.c:625-629).  That sum is
// not associative, but while the running sum stays inside one binade it is integer arithmetic: with the sum written
// as M * 2^(E-150) (M < 2^24 the mantissa with its hidden bit, E the exponent field, denormals folded into E = 1),
// adding x >= 0 gives M + RN(x / 2^(E-150)), the rounding being to nearest with ties to the even *result*.  Only a tie
// depends on the sum so far, and only through the parity of M — so a run of additions is a function of that parity:
// a pair of mantissa increments (start even, start odd).  Such pairs compose associatively, which is what lets a wave
// and then a chunk be folded as a tree.  The pair is valid as long as the sum did not leave the binade it was
// computed for (M + d < 2^24); the caller checks that and otherwise adds the chunk's elements one by one.
struct ChunkFn { uint32_t dE, dO; };
constexpr uint32_t kSat = 1u << 25;                           // any increment >= 2^24 already means "left the binade"

NANO_HD uint32_t sum_exp(uint32_t b) { const uint32_t e = b >> 23; return e ? e : 1u; }
NANO_HD uint32_t sum_man(uint32_t b) { return (b & 0x7fffffu) | ((b >> 23) ? 0x800000u : 0u); }
NANO_HD uint32_t sat_add(uint32_t a, uint32_t b) { const uint32_t s = a + b; return s > kSat ? kSat : s; }

// fold one more addend (bits xb of a finite float >= 0) into f, for sums whose exponent field is Es
NANO_HD void chunk_push(ChunkFn &f, uint32_t xb, uint32_t Es) {
    const uint32_t Mx = sum_man(xb);
    const int sh = (int)Es - (int)sum_exp(xb);
    uint32_t q, up = 0, tie = 0;
    if (sh <= 0) {
        q = sh < -1 ? kSat : (Mx << (-sh));                  // exact, no rounding
    } else {
        const int s = sh > 25 ? 25 : sh;
        const uint32_t rem = Mx & ((1u << s) - 1u), half = 1u << (s - 1);
        q = Mx >> s; up = rem > half; tie = rem == half;
    }
    const uint32_t a = q + up;
    f.dE = sat_add(f.dE, a + (tie & (f.dE + q) & 1u));
    f.dO = sat_add(f.dO, a + (tie & (f.dO + 1u + q) & 1u));
}

// g applied after f
NANO_HD ChunkFn chunk_then(ChunkFn f, ChunkFn g) {
    ChunkFn h;
    h.dE = sat_add(f.dE, (f.dE & 1u) ? g.dO : g.dE);
    h.dO = sat_add(f.dO, ((f.dO + 1u) & 1u) ? g.dO : g.dE);
    return h;
}

// apply f (computed for exponent field Es) to the sum with bits sb; false when f does not apply to this sum
NANO_HD bool chunk_apply(uint32_t &sb, ChunkFn f, uint32_t Es) {
    if (sum_exp(sb) != Es) return false;
    const uint32_t M = sum_man(sb);
    const uint32_t M2 = M + ((M & 1u) ? f.dO : f.dE);
    if (M2 >= (1u << 24)) return false;
    sb = ((Es - 1u) << 23) + M2;
    return true;
}

}  // namespace nano_exact
