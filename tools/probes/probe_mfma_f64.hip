// Issue rate of v_mfma_f64_16x16x4_f64 on one CU and on the whole chip: ns per MFMA and per SIMD with 1, 2, 4 waves per SIMD and
// 1, 2, 3 independent accumulators per wave (no memory traffic: operands are registers).
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_f64 probe_mfma_f64.hip && ./probe_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double acc_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double *out, int iters, double a0, double b0) {
    acc_t acc[NACC];
    for (int c = 0; c < NACC; c++) acc[c] = acc_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NACC; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < NACC; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 1.2345e-300) out[0] = s;
}
template <int NACC>
static void run(int blocks, int threads, const char *what) {
    double *d; hipMalloc(&d, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double perWave = (double)iters * 8 * NACC, wavesPerSimd = threads / 64 / 4.0;
    printf("%-28s acc %d: %.3f ms, %.1f ns per MFMA per wave, %.1f ns per MFMA per SIMD\n", what, NACC, ms, 1e6 * ms / perWave,
           1e6 * ms / (perWave * (wavesPerSimd < 1 ? 1 : wavesPerSimd)));
    hipFree(d);
}
int main() {
    run<1>(1, 256, "1 WG, 1 wave/SIMD");
    run<3>(1, 256, "1 WG, 1 wave/SIMD");
    run<3>(1, 512, "1 WG, 2 waves/SIMD");
    run<3>(1, 1024, "1 WG, 4 waves/SIMD");
    run<1>(256, 256, "256 WGs, 1 wave/SIMD");
    run<3>(256, 256, "256 WGs, 1 wave/SIMD");
    run<3>(256, 512, "256 WGs, 2 waves/SIMD");
    run<3>(256, 1024, "256 WGs, 4 waves/SIMD");
    run<3>(768, 384, "768 WGs x 6 waves (3 per CU)");
    return 0;
}
