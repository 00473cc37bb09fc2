// Compares nano_exact::exact_expf with the running libm's expf.
// usage: expf_check [stride]   (stride 1 = every float of either sign; default 64 + all of a few binades)
#include "../../nano_amd/csrc/exact_math.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <omp.h>
#include <initializer_list>
int main(int argc, char **argv) {
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 64;
    uint64_t bad = 0, total = 0;
    uint32_t first_bad = 0;
    // every float from +0 up to +inf (0 .. 0x7f800000) and from -0 down to -inf (0x80000000 .. 0xff800000)
#pragma omp parallel for reduction(+ : bad, total) schedule(static)
    for (int64_t v = 0; v <= 2 * 0x7f800000ll + 1; v += stride) {
        const int64_t u = v <= 0x7f800000ll ? v : 0x80000000ll + (v - 0x7f800001ll);
        const float x = nano_exact::bits_f32((uint32_t)u);
        const float a = expf(x), b = nano_exact::exact_expf_nonpos(x, nano_exact::kExp2Tab);
        total++;
        if (nano_exact::f32_bits(a) != nano_exact::f32_bits(b)) {
            bad++;
#pragma omp critical
            if (!first_bad) { first_bad = (uint32_t)u; fprintf(stderr, "mismatch x=%a libm=%a mine=%a\n", x, a, b); }
        }
    }
    // dense windows: the last binades before the underflow thresholds and around -1
    for (uint32_t lo : {0xc2ce0000u, 0xc2b00000u, 0xbf800000u, 0xc1200000u, 0x42b00000u, 0x3f800000u, 0x41200000u}) {
#pragma omp parallel for reduction(+ : bad, total) schedule(static)
        for (int64_t u = lo; u < (int64_t)lo + (1 << 20); u++) {
            const float x = nano_exact::bits_f32((uint32_t)u);
            const float a = expf(x), b = nano_exact::exact_expf_nonpos(x, nano_exact::kExp2Tab);
            total++;
            if (nano_exact::f32_bits(a) != nano_exact::f32_bits(b)) bad++;
        }
    }
    printf("checked %llu values, %llu mismatches\n", (unsigned long long)total, (unsigned long long)bad);
    return bad ? 1 : 0;
}
