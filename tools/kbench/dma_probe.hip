// dma_probe.hip -- what gemm_q80_g7.hip assumes about gfx950's LDS-DMA (global_load_lds_dwordx4), checked on the device:
//   1. lane p of a wave-instruction writes LDS bytes [base + 16 p, base + 16 p + 16) with the 16 bytes at ITS OWN global address
//      (any per-lane address: the kernel swizzles chunks at the source);
//   2. lanes masked off by EXEC leave their LDS slots untouched;
//   3. a destination above 64 KB (the ring uses up to 160 KB) works;
//   4. the cache-policy operand (nt) changes nothing about 1..3.
// Build: hipcc --offload-arch=gfx950 -O2 tools/kbench/dma_probe.hip -o tools/kbench/dma_probe ; prints one line per check.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int AUX> __device__ __forceinline__ void dma16(const void *g, unsigned char *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, AUX);
}

// out[0..1023]: LDS after check 1 (permuted sources); out[1024..2047]: after check 2 (even lanes only, sentinel 0xEE elsewhere);
// out[2048..3071]: the same permuted fetch landed at LDS offset `hi`; out[3072..4095]: nt policy
__global__ void probe(const unsigned char *src, unsigned char *out, uint32_t hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) { smem[i] = 0xEE; smem[hi + (i & 1023u)] = 0xEE; }
    __syncthreads();
    if (wid == 1) {                                 // a loader wave, like the kernel's: nothing but DMA + its own waits
        const uint32_t perm = (lane * 37u + 11u) & 63u;           // a permutation of 0..63
        dma16<0>(src + perm * 16u, smem);
        if ((lane & 1u) == 0u) dma16<0>(src + 1024u + lane * 16u, smem + 1024);
        dma16<0>(src + perm * 16u, smem + hi);
        dma16<2>(src + 2048u + perm * 16u, smem + 3072);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) {
        out[i] = smem[i]; out[1024 + i] = smem[1024 + i]; out[2048 + i] = smem[hi + i]; out[3072 + i] = smem[3072 + i];
    }
}

int main() {
    std::vector<unsigned char> h(4096);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)((i * 131u + (i >> 4) * 7u) & 0xff);
    unsigned char *ds = nullptr, *dout = nullptr;
    if (hipMalloc(&ds, 4096) != hipSuccess || hipMalloc(&dout, 4096) != hipSuccess) { printf("dma_probe: no device\n"); return 2; }
    hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const uint32_t hi = 150u * 1024u;
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 160 * 1024, 0, ds, dout, hi);
    if (hipDeviceSynchronize() != hipSuccess) { printf("dma_probe: launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<unsigned char> o(4096);
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, bad3 = 0, bad4 = 0;
    for (uint32_t p = 0; p < 64; p++) {
        const uint32_t perm = (p * 37u + 11u) & 63u;
        for (uint32_t b = 0; b < 16; b++) {
            bad1 += o[p * 16 + b] != h[perm * 16 + b];
            bad2 += o[1024 + p * 16 + b] != ((p & 1u) ? 0xEE : h[1024 + p * 16 + b]);
            bad3 += o[2048 + p * 16 + b] != h[perm * 16 + b];
            bad4 += o[3072 + p * 16 + b] != h[2048 + perm * 16 + b];
        }
    }
    printf("dma_probe 1 lane p -> LDS slot p, per-lane source address: %s (%d bad bytes)\n", bad1 ? "FAIL" : "ok", bad1);
    printf("dma_probe 2 EXEC-masked lanes leave their slots untouched: %s (%d bad bytes)\n", bad2 ? "FAIL" : "ok", bad2);
    printf("dma_probe 3 destination at LDS offset %u: %s (%d bad bytes)\n", hi, bad3 ? "FAIL" : "ok", bad3);
    printf("dma_probe 4 nt policy: %s (%d bad bytes)\n", bad4 ? "FAIL" : "ok", bad4);
    return (bad1 || bad2 || bad3 || bad4) ? 1 : 0;
}
