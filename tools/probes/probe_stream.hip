// Write-side / mixed-stream HBM probes for the fused dual update (k_dual_fused): what do pure stores, copies and a
// "3 reads + 2 writes" elementwise kernel reach on this box, as a function of access shape, grid, cache policy and footprint?
//   hipcc --offload-arch=gfx950 -O3 -o probe_stream probe_stream.hip && ./probe_stream
// Every variant is timed with hipEvents over `reps` back-to-back launches; "evict" variants run a 4 GiB read sweep between the
// timed launches (the real iteration streams 4 GB of operator blocks between two dual updates, so nothing of the 21 MB vectors
// survives in L2 / Infinity Cache) and time each launch on its own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NT>
__global__ void __launch_bounds__(256) k_fill(d2 *dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        d2 v = {(double)i, 1.0};
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <int NT>
__global__ void __launch_bounds__(256) k_fill_chunk(d2 *dst, long long n) {   // one contiguous piece per workgroup
    const long long per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        d2 v = {(double)i, 1.0};
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <int NTL, int NTS>
__global__ void __launch_bounds__(256) k_copy(const d2 *src, d2 *dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        d2 v = NTL ? __builtin_nontemporal_load(src + i) : src[i];
        if (NTS) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <int NTL, int NTS, int U>
__global__ void __launch_bounds__(256) k_copy_u(const d2 *src, d2 *dst, long long n) {   // U loads in flight per lane
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const long long i = i0 + u * stride; v[u] = i < n ? (NTL ? __builtin_nontemporal_load(src + i) : src[i]) : d2{0, 0}; }
#pragma unroll
        for (int u = 0; u < U; u++) { const long long i = i0 + u * stride; if (i < n) { if (NTS) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
    }
}
// the dual update's stream mix: 3 reads, 2 writes, a few flops
template <int NTL, int NTS, int U>
__global__ void __launch_bounds__(256) k_mix32(const d2 *a, const d2 *b, const d2 *c, d2 *o1, d2 *o2, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i0 = (long long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += U * stride) {
        d2 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long i = i0 + u * stride < n ? i0 + u * stride : i0;
            x[u] = NTL ? __builtin_nontemporal_load(a + i) : a[i];
            y[u] = NTL ? __builtin_nontemporal_load(b + i) : b[i];
            z[u] = NTL ? __builtin_nontemporal_load(c + i) : c[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long i = i0 + u * stride;
            if (i < n) {
                const d2 r1 = x[u] + 0.5 * y[u], r2 = 1.5 * r1 - 0.5 * z[u];
                if (NTS) { __builtin_nontemporal_store(r1, o1 + i); __builtin_nontemporal_store(r2, o2 + i); } else { o1[i] = r1; o2[i] = r2; }
            }
        }
    }
}
template <int NTL, int NTS>
__global__ void __launch_bounds__(256) k_mix32_chunk(const d2 *a, const d2 *b, const d2 *c, d2 *o1, d2 *o2, long long n) {
    const long long per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const d2 x = NTL ? __builtin_nontemporal_load(a + i) : a[i], y = NTL ? __builtin_nontemporal_load(b + i) : b[i];
        const d2 z = NTL ? __builtin_nontemporal_load(c + i) : c[i];
        const d2 r1 = x + 0.5 * y, r2 = 1.5 * r1 - 0.5 * z;
        if (NTS) { __builtin_nontemporal_store(r1, o1 + i); __builtin_nontemporal_store(r2, o2 + i); } else { o1[i] = r1; o2[i] = r2; }
    }
}
__global__ void __launch_bounds__(256) k_read(const d2 *src, long long n, double *sink) {
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const d2 v = __builtin_nontemporal_load(src + i); acc += v[0] + v[1]; }
    if (acc == 1.2345e-300) sink[blockIdx.x & 1023] = acc;
}

static hipStream_t s;
static d2 *big; static double *sink; static const long long bigN = (4LL << 30) / 16;
static hipEvent_t e0, e1;
template <typename F>
static double time_us(F launch, int reps, bool evict) {
    launch(); CK(hipStreamSynchronize(s));
    if (!evict) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return 1e3 * ms / reps;
    }
    std::vector<double> t;
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, s, big, bigN, sink);
        CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(1e3 * ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMalloc(&big, bigN * 16)); CK(hipMemset(big, 0, bigN * 16));
    CK(hipMalloc(&sink, 1024 * 8));
    for (long long bytes : {20858880LL, 1LL << 30}) {   // one dual-shaped vector of the 493-scenario tree (10 864 x 240 x 8 B); 1 GiB
        const long long n = bytes / 16;
        d2 *v[5];
        for (int i = 0; i < 5; i++) { CK(hipMalloc(&v[i], bytes)); CK(hipMemset(v[i], 0, bytes)); }
        printf("=== vectors of %.1f MB ===\n", bytes / 1e6);
        for (int evict = 0; evict < 2; evict++) {
            if (evict && bytes > (64 << 20)) continue;
            const int reps = evict ? 9 : 20;
            printf("--- %s\n", evict ? "4 GiB read sweep before every timed launch (median of 9)" : "back to back");
            for (int grid : {1024, 2048, 4096, 8192}) {
                auto GB = [&](double us, int streams) { return streams * (double)bytes / us / 1e3; };
                double t;
                t = time_us([&] { hipLaunchKernelGGL(k_fill<0>, dim3(grid), dim3(256), 0, s, v[0], n); }, reps, evict); printf("grid %5d fill plain      %7.2f us %6.0f GB/s\n", grid, t, GB(t, 1));
                t = time_us([&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, s, v[0], n); }, reps, evict); printf("grid %5d fill nt         %7.2f us %6.0f GB/s\n", grid, t, GB(t, 1));
                t = time_us([&] { hipLaunchKernelGGL(k_fill_chunk<0>, dim3(grid), dim3(256), 0, s, v[0], n); }, reps, evict); printf("grid %5d fill chunk      %7.2f us %6.0f GB/s\n", grid, t, GB(t, 1));
                t = time_us([&] { hipLaunchKernelGGL((k_copy<0, 0>), dim3(grid), dim3(256), 0, s, v[0], v[1], n); }, reps, evict); printf("grid %5d copy plain      %7.2f us %6.0f GB/s\n", grid, t, GB(t, 2));
                t = time_us([&] { hipLaunchKernelGGL((k_copy<1, 1>), dim3(grid), dim3(256), 0, s, v[0], v[1], n); }, reps, evict); printf("grid %5d copy nt/nt      %7.2f us %6.0f GB/s\n", grid, t, GB(t, 2));
                t = time_us([&] { hipLaunchKernelGGL((k_copy<1, 0>), dim3(grid), dim3(256), 0, s, v[0], v[1], n); }, reps, evict); printf("grid %5d copy ntload     %7.2f us %6.0f GB/s\n", grid, t, GB(t, 2));
                t = time_us([&] { hipLaunchKernelGGL((k_copy_u<0, 0, 4>), dim3(grid), dim3(256), 0, s, v[0], v[1], n); }, reps, evict); printf("grid %5d copy plain u4   %7.2f us %6.0f GB/s\n", grid, t, GB(t, 2));
                t = time_us([&] { hipLaunchKernelGGL((k_mix32<0, 0, 1>), dim3(grid), dim3(256), 0, s, v[0], v[1], v[2], v[3], v[4], n); }, reps, evict); printf("grid %5d mix3r2w plain   %7.2f us %6.0f GB/s\n", grid, t, GB(t, 5));
                t = time_us([&] { hipLaunchKernelGGL((k_mix32<1, 1, 1>), dim3(grid), dim3(256), 0, s, v[0], v[1], v[2], v[3], v[4], n); }, reps, evict); printf("grid %5d mix3r2w nt/nt   %7.2f us %6.0f GB/s\n", grid, t, GB(t, 5));
                t = time_us([&] { hipLaunchKernelGGL((k_mix32<1, 0, 1>), dim3(grid), dim3(256), 0, s, v[0], v[1], v[2], v[3], v[4], n); }, reps, evict); printf("grid %5d mix3r2w ntload  %7.2f us %6.0f GB/s\n", grid, t, GB(t, 5));
                t = time_us([&] { hipLaunchKernelGGL((k_mix32<0, 0, 2>), dim3(grid), dim3(256), 0, s, v[0], v[1], v[2], v[3], v[4], n); }, reps, evict); printf("grid %5d mix3r2w plain u2 %6.2f us %6.0f GB/s\n", grid, t, GB(t, 5));
                t = time_us([&] { hipLaunchKernelGGL((k_mix32_chunk<0, 0>), dim3(grid), dim3(256), 0, s, v[0], v[1], v[2], v[3], v[4], n); }, reps, evict); printf("grid %5d mix3r2w chunk   %7.2f us %6.0f GB/s\n", grid, t, GB(t, 5));
            }
        }
        for (int i = 0; i < 5; i++) CK(hipFree(v[i]));
    }
    return 0;
}
