// What does it cost ONE workgroup to read tagged packets that the previous kernel wrote into an inbox?  (one-shot exchange,
// kernels.hpp peer_gather.)  Memory kinds: hipMalloc (coarse-grained), hipExtMallocWithFlags(hipDeviceMallocUncached),
// hipDeviceMallocFinegrained; read styles: 16-byte plain / nontemporal loads, 8-byte system-scope atomic loads; the writer uses
// plain stores or system-scope atomic stores.  Reports the reader's wall-clock time inside the kernel (100 MHz clock), the number
// of stale first looks, and the kernel duration between events.
// hipcc --offload-arch=gfx950 -O2 -o probe_uncached probe_uncached.hip && ./probe_uncached
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
__global__ void k_write(u64 *box, int n, unsigned seq, int atomicStores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 lo = (u64)(unsigned)i | ((u64)seq << 32), hi = (u64)(unsigned)(i * 7) | ((u64)seq << 32);
    if (atomicStores) {
        __hip_atomic_store(box + 2 * i, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(box + 2 * i + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else { box[2 * i] = lo; box[2 * i + 1] = hi; }
}
// mode 0: 16-byte plain loads; 1: 16-byte nontemporal loads; 2: 8-byte system-scope atomic loads
template <int MODE, int PER>
__global__ void k_read(const u64 *box, int n, unsigned seq, u64 *out, long long *stamps, int *stale) {
    const long long t0 = wall_clock64();
    u64 a[PER], b[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = threadIdx.x + k * blockDim.x;
        const int ii = i < n ? i : 0;
        if (MODE == 2) {
            a[k] = __hip_atomic_load(box + 2 * ii, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b[k] = __hip_atomic_load(box + 2 * ii + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if (MODE == 1) { const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(box) + ii); a[k] = v[0]; b[k] = v[1]; }
        else { const u64x2 v = reinterpret_cast<const u64x2 *>(box)[ii]; a[k] = v[0]; b[k] = v[1]; }
    }
    u64 acc = 0; int st = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = threadIdx.x + k * blockDim.x;
        if (i < n) { if ((unsigned)(a[k] >> 32) != seq || (unsigned)(b[k] >> 32) != seq) st++; acc += (a[k] & 0xffffffffull) + (b[k] & 0xffffffffull); }
    }
    out[threadIdx.x] = acc;
    if (st) atomicAdd(stale, st);
    __syncthreads();
    if (threadIdx.x == 0) { stamps[0] = t0; stamps[1] = wall_clock64(); }
}
// something big in between, like the streaming kernel: sweeps the caches
__global__ void k_sweep(const double *p, size_t n, double *sink) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(p + i);
    if (s == 1.2345) *sink = s;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 3795, threads = 384, PER = 10;
    u64 *out; long long *stamps; int *stale; double *big, *sink;
    CK(hipMalloc(&out, threads * 8)); CK(hipMalloc(&stamps, 16)); CK(hipMalloc(&stale, 4)); CK(hipMalloc(&sink, 8));
    const size_t bigN = (size_t)64 << 20; CK(hipMalloc(&big, bigN * 8)); CK(hipMemset(big, 0, bigN * 8));
    const char *kinds[3] = {"hipMalloc (coarse)", "uncached", "fine-grained"};
    for (int kind = 0; kind < 3; kind++) {
        u64 *box;
        if (kind == 0) CK(hipMalloc(&box, n * 16));
        else CK(hipExtMallocWithFlags((void **)&box, n * 16, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
        CK(hipMemset(box, 0, n * 16)); CK(hipDeviceSynchronize());
        for (int atomicStores = 0; atomicStores < 2; atomicStores++)
            for (int mode = 0; mode < 3; mode++) {
                double tin = 0, tev = 0; int staleTot = 0; const int reps = 20;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int r = 0; r < reps + 2; r++) {
                    const unsigned seq = 100 * (kind * 6 + atomicStores * 3 + mode) + r + 1;
                    CK(hipMemsetAsync(stale, 0, 4, s));
                    hipLaunchKernelGGL(k_sweep, dim3(1024), dim3(256), 0, s, big, bigN, sink);
                    hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, s, box, n, seq, atomicStores);
                    CK(hipEventRecord(e0, s));
                    if (mode == 0) hipLaunchKernelGGL((k_read<0, PER>), dim3(1), dim3(threads), 0, s, box, n, seq, out, stamps, stale);
                    else if (mode == 1) hipLaunchKernelGGL((k_read<1, PER>), dim3(1), dim3(threads), 0, s, box, n, seq, out, stamps, stale);
                    else hipLaunchKernelGGL((k_read<2, PER>), dim3(1), dim3(threads), 0, s, box, n, seq, out, stamps, stale);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    long long st[2]; int sl; float ms;
                    CK(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&sl, stale, 4, hipMemcpyDeviceToHost));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) { tin += (st[1] - st[0]) / 100.0; tev += ms * 1e3; staleTot += sl; }
                }
                printf("%-20s writer %-7s reader %-22s in-kernel %6.2f us  launch %6.2f us  stale looks %d of %d\n", kinds[kind], atomicStores ? "atomic" : "plain",
                       mode == 0 ? "16 B plain" : mode == 1 ? "16 B nontemporal" : "8 B system atomic", tin / reps, tev / reps, staleTot, 2 * 0 + reps * n);
            }
        CK(hipFree(box));
    }
    return 0;
}
