// mfma_probe.hip -- verifies the operand / result lane mapping of v_mfma_i32_16x16x64_i8 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int8_t *A /*[16][64] row-major M x K*/, const int8_t *Bt /*[16][64] N x K (token-major)*/, int *C /*[16][16] M x N*/) {
    const int l = threadIdx.x;
    // hypothesis: lane l holds A[m = l%16][k = 16*(l/16) .. +15] and B[k = 16*(l/16)..+15][n = l%16]
    const v4i a = *reinterpret_cast<const v4i *>(A + (l % 16) * 64 + 16 * (l / 16));
    const v4i b = *reinterpret_cast<const v4i *>(Bt + (l % 16) * 64 + 16 * (l / 16));
    v4i c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    // hypothesis: c[i] = C[m = (l/16)*4 + i][n = l%16]
    for (int i = 0; i < 4; i++) C[((l / 16) * 4 + i) * 16 + (l % 16)] = c[i];
}
int main() {
    std::vector<int8_t> A(16 * 64), B(16 * 64);
    for (int m = 0; m < 16; m++) for (int k = 0; k < 64; k++) A[m * 64 + k] = (int8_t)((m * 7 + k * 3) % 23 - 11);
    for (int n = 0; n < 16; n++) for (int k = 0; k < 64; k++) B[n * 64 + k] = (int8_t)((n * 5 + k * 11 + n * k) % 19 - 9);
    int8_t *dA, *dB; int *dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dC);
    std::vector<int> C(256); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0, badT = 0;
    for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
        int ref = 0; for (int k = 0; k < 64; k++) ref += (int)A[m * 64 + k] * (int)B[n * 64 + k];
        if (C[m * 16 + n] != ref) bad++;
        if (C[n * 16 + m] != ref) badT++;
    }
    printf("mfma_i32_16x16x64_i8: %d mismatches with C[m=(l/16)*4+i][n=l%%16]; %d with the transposed reading\n", bad, badT);
    return 0;
}
