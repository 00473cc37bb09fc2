// Exhaustive host check behind div_const<C>() (nano_amd/csrc/device_common.h): for C in {15, 63, 127} and EVERY float bit pattern x,
//   q0 = x * RN(1/C);  r = fma(-q0, C, x);  q = fma(r, RN(1/C), q0)   equals   x / C   (IEEE, round to nearest)
// except x = -0 (q = +0).  gcc -O2 -fopenmp -mfma -ffp-contract=off -o chk tools/div_const_check.c -lm && ./chk   (about a minute;
// an argument N checks every N-th bit pattern instead: tests/test_div_const.py)
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
// exhaustive: for every float x, is fmaf(fmaf(-q0, c, x), rc, q0) with q0 = x * rc, rc = RN(1/c) equal to x / c ?
int main(int argc, char **argv) {
    const int64_t stride = argc > 1 ? atoll(argv[1]) : 1;
    float consts[3] = {15.0f, 63.0f, 127.0f};
    for (int ci = 0; ci < 3; ci++) {
        const float c = consts[ci]; const float rc = 1.0f / c;
        uint64_t bad = 0; uint32_t first_bad = 0, min_bad_exp = 255, max_bad_exp = 0;
        #pragma omp parallel for reduction(+:bad) schedule(static)
        for (int64_t i = 0; i < (1ll << 32); i += stride) {
            uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
            if (isnan(x) || isinf(x)) continue;
            volatile float q0 = x * rc;
            volatile float r = fmaf(-q0, c, x);
            volatile float q = fmaf(r, rc, q0);
            float ref = x / c;
            uint32_t a, b; float qq = q; memcpy(&a, &qq, 4); memcpy(&b, &ref, 4);
            if (a != b) {
                bad++;
                #pragma omp critical
                { uint32_t e = (u >> 23) & 255; if (e < min_bad_exp) min_bad_exp = e; if (e > max_bad_exp) max_bad_exp = e; first_bad = u; }
            }
        }
        printf("c=%g rc=%a mismatches=%llu exp range of mismatching x: [%u, %u] sample %08x\n", c, rc, (unsigned long long)bad, min_bad_exp, max_bad_exp, first_bad);
    }
    return 0;
}
