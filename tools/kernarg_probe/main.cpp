// Does kernarg preloading (the CP puts the first kernel arguments into SGPRs at wave launch; clang: -mllvm
// -amdgpu-kernarg-preload-count=N, plain -- not by-value struct -- arguments only) shorten a chain of short dependent kernels?
// A graph of 200 dependent tiny steps, three builds of the same step: by-value struct / plain arguments / plain + preload.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern "C" void launch_struct(const float *, float *, const float *, unsigned, float, hipStream_t);
extern "C" void launch_plain(const float *, float *, const float *, unsigned, float, hipStream_t);
extern "C" void launch_preload(const float *, float *, const float *, unsigned, float, hipStream_t);
typedef void (*launch_t)(const float *, float *, const float *, unsigned, float, hipStream_t);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const unsigned n = 256 * 256, steps = 200;
    float *a, *b, *w; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&w, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, n * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    const char *names[3] = {"by-value struct", "plain arguments", "plain + preload"};
    launch_t fns[3] = {launch_struct, launch_plain, launch_preload};
    hipGraphExec_t ge[3];
    for (int v = 0; v < 3; v++) {
        fns[v](a, b, w, n, 1.0f, st); CK(hipStreamSynchronize(st));
        hipGraph_t g; CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (unsigned s = 0; s < steps; s++) fns[v]((s & 1) ? b : a, (s & 1) ? a : b, w, n, 1.0f, st);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge[v], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; round++)
        for (int v = 0; v < 3; v++) {
            std::vector<float> ms;
            for (int r = 0; r < 12; r++) {
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge[v], st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("round %d  %-16s  %.3f us per step (median of 12 replays of %u dependent steps)\n", round, names[v], ms[6] * 1e3 / steps, steps);
        }
    return 0;
}
