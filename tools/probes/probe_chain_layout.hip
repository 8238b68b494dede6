// What would a chain-major layout of the helper vectors gain?  (round 5, review item 5)
// The two chain walks of the sweep (k_up_chain: running sums leaf -> chain top; k_down_chain: running sums chain top -> leaf, Hx written
// node-major) re-built here with their real shapes -- 493 chains x 22 stages, nv = 97, nx = 63, nu = 114, 256-thread workgroups, 12
// stages requested per round trip -- on arrays indexed either stage-major ([stage][chain][dim]: what the library does; a chain's nodes lie
// K nodes apart) or chain-major ([chain][stage][dim]: a chain's nodes are consecutive).  Between the timed kernels a streaming kernel
// reads 2 GiB non-temporally (the cache state the real walks start in: behind k_stream_gemv's 4 GB).  Hx is written node-major in both
// forms, as the dual update needs it.
// hipcc --offload-arch=gfx950 -O3 -o probe_chain_layout probe_chain_layout.hip && ./probe_chain_layout
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int PF = 12, THREADS = 256;
struct Dims { int K, L, nv, nx, nu, ny; };
template <bool CHAIN_MAJOR>
__device__ __forceinline__ size_t slot(const Dims &d, int s, int k) { return CHAIN_MAJOR ? (size_t)s * d.L + k : (size_t)k * d.K + s; }
template <bool CM>
__global__ void __launch_bounds__(THREADS) k_up(Dims d, const double *__restrict__ beta, const double *__restrict__ my, const double *__restrict__ qa, double *sk, double *top) {
    const int s = blockIdx.x, nv = d.nv, nx = d.nx;
    for (int t = threadIdx.x; t < nv + nx; t += THREADS) {
        if (t < nv) {
            double rho = 0;
            for (int k = d.L - 1; k >= 0; k -= PF) {
                double b[PF], m[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) { const size_t n = slot<CM>(d, s, k - j >= 0 ? k - j : 0); b[j] = beta[n * nv + t]; m[j] = my[n * 2 * nv + nv + t]; }
#pragma unroll
                for (int j = 0; j < PF; j++) if (k - j >= 0) { const size_t n = slot<CM>(d, s, k - j); const double sv = b[j] + rho; rho = sv + m[j]; sk[n * (nv + nx) + t] = sv; }
            }
            top[(size_t)s * (nv + 2 * nx) + t] = rho;
        } else {
            const int j0 = t - nv;
            double kap = 0, q = 0;
            for (int k = d.L - 1; k >= 0; k -= PF) {
                double av[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) av[j] = qa[slot<CM>(d, s, k - j >= 0 ? k - j : 0) * nx + j0];
#pragma unroll
                for (int j = 0; j < PF; j++) if (k - j >= 0) { const size_t n = slot<CM>(d, s, k - j); kap += q; sk[n * (nv + nx) + nv + j0] = kap; q += av[j]; }
            }
            top[(size_t)s * (nv + 2 * nx) + nv + j0] = kap; top[(size_t)s * (nv + 2 * nx) + nv + nx + j0] = q;
        }
    }
}
template <bool CM>
__global__ void __launch_bounds__(THREADS) k_down(Dims d, const double *__restrict__ lvb, const double *__restrict__ uhat, const double *__restrict__ eb, const double *__restrict__ dy, double *hx) {
    const int s = blockIdx.x, nx = d.nx, nu = d.nu, ny = d.ny, w = nu + nx;
    for (int t = threadIdx.x; t < w; t += THREADS) {
        if (t < nu) {
            double run = 0.5;
            for (int k = 0; k < d.L; k += PF) {
                double dv[PF], uh[PF], d0[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) { const int kk = k + j < d.L ? k + j : d.L - 1; const size_t n = slot<CM>(d, s, kk); dv[j] = lvb[n * w + t]; uh[j] = uhat[n * nu + t]; d0[j] = dy[(size_t)kk * ny + 2 * nx + t]; }
#pragma unroll
                for (int j = 0; j < PF; j++) if (k + j < d.L) { run += dv[j]; const double uv = uh[j] + run; hx[((size_t)(k + j) * d.K + s) * ny + 2 * nx + t] = d0[j] * uv; }
            }
        } else {
            const int j0 = t - nu;
            double bw = 0.1, xr = 0.2;
            for (int k = 0; k < d.L; k += PF) {
                double dv[PF], ev[PF], d0[PF], d1[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) { const int kk = k + j < d.L ? k + j : d.L - 1; const size_t n = slot<CM>(d, s, kk); dv[j] = lvb[n * w + nu + j0]; ev[j] = eb[n * nx + j0]; d0[j] = dy[(size_t)kk * ny + j0]; d1[j] = dy[(size_t)kk * ny + nx + j0]; }
#pragma unroll
                for (int j = 0; j < PF; j++) if (k + j < d.L) { bw += dv[j]; xr += ev[j] + bw; const size_t o = ((size_t)(k + j) * d.K + s) * ny; hx[o + j0] = d0[j] * xr; hx[o + nx + j0] = d1[j] * xr; }
            }
        }
    }
}
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k_sweep(const d2 *p, size_t n, double *sink) {
    double a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const d2 v = __builtin_nontemporal_load(p + i); a += v[0] + v[1]; }
    if (a == 1.2345e-300) *sink = a;
}
int main() {
    const Dims d{493, 22, 97, 63, 114, 240};
    const size_t nodes = (size_t)d.K * d.L;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto dalloc = [&](size_t n) { double *p = nullptr; if (hipMalloc(&p, n * sizeof(double)) != hipSuccess) return (double *)nullptr; hipMemset(p, 0, n * sizeof(double)); return p; };
    double *beta = dalloc(nodes * d.nv), *my = dalloc(nodes * 2 * d.nv), *qa = dalloc(nodes * d.nx), *sk = dalloc(nodes * (d.nv + d.nx)), *top = dalloc((size_t)d.K * (d.nv + 2 * d.nx));
    double *lvb = dalloc(nodes * (d.nu + d.nx)), *uhat = dalloc(nodes * d.nu), *eb = dalloc(nodes * d.nx), *dy = dalloc((size_t)d.L * d.ny), *hx = dalloc(nodes * d.ny), *sink = dalloc(1);
    const size_t big = (size_t)2 << 30;
    double *sweep = dalloc(big / 8);
    if (!beta || !my || !qa || !sk || !top || !lvb || !uhat || !eb || !dy || !hx || !sweep) { printf("allocation failed\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 60;
    for (int cm = 0; cm < 2; cm++) {
        std::vector<float> up, down;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, st, (const d2 *)sweep, big / 16, sink);
            CK(hipEventRecord(e0, st));
            if (cm) hipLaunchKernelGGL(k_up<true>, dim3(d.K), dim3(THREADS), 0, st, d, beta, my, qa, sk, top);
            else hipLaunchKernelGGL(k_up<false>, dim3(d.K), dim3(THREADS), 0, st, d, beta, my, qa, sk, top);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); up.push_back(ms * 1e3f);
            hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, st, (const d2 *)sweep, big / 16, sink);
            CK(hipEventRecord(e0, st));
            if (cm) hipLaunchKernelGGL(k_down<true>, dim3(d.K), dim3(THREADS), 0, st, d, lvb, uhat, eb, dy, hx);
            else hipLaunchKernelGGL(k_down<false>, dim3(d.K), dim3(THREADS), 0, st, d, lvb, uhat, eb, dy, hx);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1)); down.push_back(ms * 1e3f);
        }
        std::sort(up.begin(), up.end()); std::sort(down.begin(), down.end());
        printf("%s: up walk median %.2f us (min %.2f), down walk median %.2f us (min %.2f)   [hipEvents around the launch, behind a 2 GiB non-temporal sweep]\n",
               cm ? "chain-major [chain][stage][dim]" : "stage-major [stage][chain][dim]", up[reps / 2], up[0], down[reps / 2], down[0]);
    }
    // back to back without the sweep (warm caches): the launch floor plus the walk itself
    for (int cm = 0; cm < 2; cm++) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 200; r++) {
            if (cm) { hipLaunchKernelGGL(k_up<true>, dim3(d.K), dim3(THREADS), 0, st, d, beta, my, qa, sk, top); hipLaunchKernelGGL(k_down<true>, dim3(d.K), dim3(THREADS), 0, st, d, lvb, uhat, eb, dy, hx); }
            else { hipLaunchKernelGGL(k_up<false>, dim3(d.K), dim3(THREADS), 0, st, d, beta, my, qa, sk, top); hipLaunchKernelGGL(k_down<false>, dim3(d.K), dim3(THREADS), 0, st, d, lvb, uhat, eb, dy, hx); }
        }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: up + down back to back, warm caches: %.2f us per pair\n", cm ? "chain-major" : "stage-major", ms * 1e3f / 200);
    }
    return 0;
}
