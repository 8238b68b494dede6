// Per-kernel cost of a chain of dependent launches on one stream: plain launches vs the same chain replayed from a hipGraph.
// hipcc --offload-arch=gfx950 -O2 -o probe_launch probe_launch.hip && ./probe_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_touch(double *p, int n) {   // a little dependent work: every launch reads what the previous one wrote
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0000001 + 1.0;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 64 * 256, chain = 700, reps = 5;
    double *p; CK(hipMalloc(&p, n * sizeof(double))); CK(hipMemset(p, 0, n * sizeof(double)));
    for (int grid : {1, 64, 1024}) {
        // plain launches
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, p, n);
        CK(hipStreamSynchronize(s));
        double best = 1e30;
        for (int r = 0; r < reps; r++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, p, n);
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            if (us < best) best = us;
        }
        // the same chain from a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        double bestG = 1e30;
        for (int r = 0; r < reps; r++) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
            if (us < bestG) bestG = us;
        }
        printf("grid %5d x 256: plain %.2f us per dependent launch, graph %.2f us\n", grid, best, bestG);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
