// fbe_kernels.hpp -- device kernels of the global-FBE and NAMA outer loops (SURVEY.md section 8(f) rank 3;
// SmpcController.cu:884-1476).  The expensive part of both loops is the Hessian oracle, which is the tree sweep of
// kernels.hpp run with zero affine terms; what is here is the vector algebra around it: deterministic multi-dot
// reductions, the L-BFGS two-loop updates driven by device-resident scalars (no host round trip inside the
// recursion), the line-search trial step and the FBE value terms.  All of it is HBM-bound streaming over the dual
// vectors in the y layout ([node][2nx | nu]); dot products are permutation invariant, so the layout
// difference to the reference's (all xi | all psi) vectors does not change any value.
#ifndef RAPIDNET_FBE_KERNELS_HPP_
#define RAPIDNET_FBE_KERNELS_HPP_

#include "common.hpp"
#include "k_dual.hpp"
#include "k_slab.hpp"

namespace rn {

constexpr int DOT_MAX = 6;
constexpr int FBE_SCALARS = 64;   // device scalar slots: [0, DOT_MAX) latest k_dots result, [16, 16 + m] L-BFGS alpha

template <typename T>
struct DotArgs {
    const T *a[DOT_MAX];
    const T *b[DOT_MAX];
    int cnt;
    long long n;
    double *partials;   // [blocks][DOT_MAX]
    long long first;    // sharded contexts: elements below `first` (the replicated crown) do not count on ranks other than 0
};
// up to DOT_MAX dot products <a_j, b_j> over the same index range in ONE pass; fp64 accumulation, fixed reduction
// order (a last-block-folds variant that saves the k_dots_finish launch was measured: the per-block agent-scope release
// makes k_dots 2.5x and k_value_terms 4x slower on this multi-XCD part -- two launches it stays);
// order (thread-strided partial sums -> wave shuffle -> LDS -> one partial per block), so a repeat is bitwise equal
// block tail of every kernel that produces dot partials: wave shuffle -> LDS -> one partial per block and dot, fixed order.
// Kernels that fuse a dot product into another pass (k_lbfgs_fused, k_lbfgs_diffs_dots, k_prox_res) walk the range exactly as
// k_dots does (same grid, same 16-byte vectors, same element order), so a fused dot is bitwise the k_dots value.
__device__ __forceinline__ void dots_block_reduce(double (&acc)[DOT_MAX], double *partials) {
    __shared__ double sh[ELT_THREADS / 64][DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) {
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < DOT_MAX) {
        double s = 0;
        for (int k = 0; k < ELT_THREADS / 64; k++) s += sh[k][threadIdx.x];
        partials[(size_t)blockIdx.x * DOT_MAX + threadIdx.x] = s;
    }
}
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_dots(DotArgs<T> g) {
    double acc[DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) acc[j] = 0;
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    unsigned long long misal = 0;   // columns of the L-BFGS buffers start at multiples of n values: 16-byte aligned only if n allows
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) if (j < g.cnt) misal |= (unsigned long long)g.a[j] | (unsigned long long)g.b[j];
    const long long nvec = (misal & 15ull) ? 0 : g.n / VN;
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * ELT_THREADS) {   // 16-byte loads
#pragma unroll
        for (int j = 0; j < DOT_MAX; j++)
            if (j < g.cnt) {
                const VT av = reinterpret_cast<const VT *>(g.a[j])[i], bv = reinterpret_cast<const VT *>(g.b[j])[i];
#pragma unroll
                for (int e = 0; e < VN; e++) if (i * VN + e >= g.first) acc[j] = fma_rn((double)av[e], (double)bv[e], acc[j]);
            }
    }
    for (long long i = nvec * VN + (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < g.n; i += (long long)gridDim.x * ELT_THREADS) {
#pragma unroll
        for (int j = 0; j < DOT_MAX; j++)
            if (j < g.cnt && i >= g.first) acc[j] = fma_rn((double)g.a[j][i], (double)g.b[j][i], acc[j]);
    }
    dots_block_reduce(acc, g.partials);
}
// second stage: out[j] = sum over blocks of partials[b][j], fixed order
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_dots_finish(const double *partials, int nblocks, int cnt, double *out) {
    __shared__ double sh[ELT_THREADS];
    for (int j = 0; j < cnt; j++) {
        double s = 0;
        for (int b = threadIdx.x; b < nblocks; b += ELT_THREADS) s += partials[(size_t)b * DOT_MAX + j];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int w = ELT_THREADS / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[j] = sh[0];
        __syncthreads();
    }
}

// The same fold by EVERY workgroup of a consumer kernel (k_lbfgs_fused): dot 0 of `nblocks` partials in exactly k_dots_finish's
// order (thread-strided sums, then the LDS tree), so the value is bitwise the one a k_dots_finish launch would have left in
// scal[0] -- and that launch (5 us of dependent-launch floor, 16 of them per two-loop recursion) is not needed.  All threads
// of the workgroup call; every thread gets the sum.
__device__ __forceinline__ double fold_dot0(const double *partials, int nblocks) {
    __shared__ double fsh[ELT_THREADS];
    double s = 0;
    for (int b = threadIdx.x; b < nblocks; b += ELT_THREADS) s += partials[(size_t)b * DOT_MAX];
    fsh[threadIdx.x] = s;
    __syncthreads();
    for (int w = ELT_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) fsh[threadIdx.x] += fsh[threadIdx.x + w];
        __syncthreads();
    }
    const double r = fsh[0];
    __syncthreads();
    return r;
}

// L-BFGS two-loop updates (SmpcController::twoLoopRecursionLbfgs, SmpcController.cu:1175-1229); scal[0] holds the
// dot product just reduced by k_dots_finish, alphaArr the first loop's coefficients.
//   mode 0: alpha_c = rho_c <S_c, dir> ; dir -= alpha_c Y_c         (vec = Y_c)
//   mode 1: beta   = rho_c <Y_c, dir> ; dir += (alpha_c - beta) S_c (vec = S_c)
template <typename T>
__global__ void k_lbfgs_axpy(T *dir, const T *vec, const double *scal, double rho, double *alphaArr, int c, int mode, long long n) {
    const T prod = (T)rho * (T)scal[0];
    const T coef = mode == 0 ? -prod : (T)alphaArr[c] - prod;
    if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) alphaArr[c] = (double)prod;   // nobody reads alphaArr in mode 0
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dir[i] += coef * vec[i];
}
// S = y - yPrev ; Y = g - gPrev   (SmpcController::updateLbfgsBuffer, SmpcController.cu:1119-1130)
template <typename T>
__global__ void k_lbfgs_diffs(T *S, T *Y, const T *y, const T *yPrev, const T *g, const T *gPrev, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        S[i] = y[i] - yPrev[i];
        Y[i] = g[i] - gPrev[i];
    }
}

// scale * (an update that has been rounded already): the product must not be contracted into the update's multiply-add
template <typename T>
__device__ __forceinline__ T lbfgs_scaled(T scale, T rounded) {
#pragma clang fp contract(off)
    return scale * rounded;
}
// One step of the two-loop recursion AND the dot product the next step needs, in one pass over the direction
// (SmpcController::twoLoopRecursionLbfgs, SmpcController.cu:1175-1229; the reference issues Sdot + Saxpy per step):
//   mode -1: dir = scale * src                          (start: src = gradient, scale = -1; between the loops: src = dir, scale = H0)
//   mode  0: alpha_c = rho_c scal[0]; dir -= alpha_c vec           (vec = Y_c)
//   mode  1: dir += (alpha_c - rho_c scal[0]) vec                  (vec = S_c)
//   next != nullptr: partials[block][0] = <next, dir_new> over the block's elements (k_dots order), for k_dots_finish
//   src != nullptr with mode 0 (the first loop's first step inside the loops): the direction is not read but formed, dir = srcScale * src
//   (= -gradient: the start of the recursion rides along; its dot product came out of k_lbfgs_diffs_dots)
//   post != 0 (with mode 0, the first loop's last step): the scaling between the loops rides along, dir = scale * (dir - alpha_c vec)
//   -- the update rounded as the step stores it, then the product: the two roundings of the two passes it replaces
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_lbfgs_fused(T *dir, const T *src, const T *vec, const double *scal, double rho, double *alphaArr, int c,
                                                             int mode, T scale, const T *next, double *partials, long long n, long long first,
                                                             const double *prevPartials, int nPrev, int post = 0, T srcScale = 0) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    T coef = 0;
    double dot0 = 0;
    if (mode >= 0) {
        // the dot product the previous step reduced: folded here from its partials (prevPartials), or read from scal[0] where a
        // k_dots_finish launch (and, on sharded contexts, the all-reduce) has left it
        dot0 = prevPartials ? fold_dot0(prevPartials, nPrev) : scal[0];
        const T prod = (T)rho * (T)dot0;
        coef = mode == 0 ? -prod : (T)alphaArr[c] - prod;
    }
    double acc[DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) acc[j] = 0;
    const unsigned long long misal = (unsigned long long)dir | (unsigned long long)(mode < 0 ? src : vec) | (unsigned long long)(next ? next : dir) | (unsigned long long)(src ? src : dir);
    const long long nvec = (misal & 15ull) ? 0 : n / VN;
    const long long stride = (long long)gridDim.x * ELT_THREADS, gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    for (long long i = gid; i < nvec; i += stride) {
        VT d;
        if (mode < 0) {
            const VT sv = reinterpret_cast<const VT *>(src)[i];
#pragma unroll
            for (int e = 0; e < VN; e++) d[e] = scale * sv[e];
        } else {
            if (src) {
                const VT sv = reinterpret_cast<const VT *>(src)[i];
#pragma unroll
                for (int e = 0; e < VN; e++) d[e] = srcScale * sv[e];
            } else d = reinterpret_cast<VT *>(dir)[i];
            const VT vv = reinterpret_cast<const VT *>(vec)[i];
#pragma unroll
            for (int e = 0; e < VN; e++) d[e] = fma_rn(coef, vv[e], d[e]);      // one rounding, spelled out (not left to the compiler's contraction choice)
            if (post) {
#pragma unroll
                for (int e = 0; e < VN; e++) d[e] = lbfgs_scaled(scale, d[e]);
            }
        }
        reinterpret_cast<VT *>(dir)[i] = d;
        if (next) {
            const VT nv = reinterpret_cast<const VT *>(next)[i];
#pragma unroll
            for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[0] = fma_rn((double)nv[e], (double)d[e], acc[0]);
        }
    }
    for (long long i = nvec * VN + gid; i < n; i += stride) {
        T d;
        if (mode < 0) d = scale * src[i]; else { d = fma_rn(coef, vec[i], src ? srcScale * src[i] : dir[i]); if (post) d = lbfgs_scaled(scale, d); }
        dir[i] = d;
        if (next && i >= first) acc[0] = fma_rn((double)next[i], (double)d, acc[0]);
    }
    // the coefficient of the first loop is kept for the second one; written after every block has read scal / alphaArr is not
    // required: nobody reads alphaArr[c] in mode 0, and the next launch is stream-ordered behind this one
    if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) alphaArr[c] = (double)((T)rho * (T)dot0);
    if (next) dots_block_reduce(acc, partials);
}
// S = y - yPrev, Y = g - gPrev (SmpcController::updateLbfgsBuffer, :1119-1130) and the four dot products its skip rule and
// H0 scaling need (<g,g>, <S,Y>, <Y,Y>, <S,S>, :1131-1156) in the same pass
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_lbfgs_diffs_dots(T *S, T *Y, const T *y, const T *yPrev, const T *g, const T *gPrev, double *partials, long long n,
                                                                  long long first, int extra = 0, const T *sPrev = nullptr) {
    // extra != 0 (inside the loops): also <S, g> (dot 4) and <sPrev, g> (dot 5; sPrev = the newest column so far, may be null) -- whichever
    // of the two is the newest column after the host's skip decision, its negative is the dot product the two-loop recursion starts with
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    double acc[DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) acc[j] = 0;
    // S and Y are L-BFGS columns after the first accepted pairs (columns are exchanged with the scratch pair, not copied): a column
    // starts at a multiple of n values, 16-byte aligned only if n allows -- otherwise the scalar walk, as in k_dots
    const unsigned long long misal = (unsigned long long)S | (unsigned long long)Y | (unsigned long long)y | (unsigned long long)yPrev | (unsigned long long)g | (unsigned long long)gPrev |
                                     (unsigned long long)(sPrev ? sPrev : g);
    const long long nvec = (misal & 15ull) ? 0 : n / VN;
    const long long stride = (long long)gridDim.x * ELT_THREADS, gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    for (long long i = gid; i < nvec; i += stride) {
        const VT yv = reinterpret_cast<const VT *>(y)[i], ypv = reinterpret_cast<const VT *>(yPrev)[i];
        const VT gv = reinterpret_cast<const VT *>(g)[i], gpv = reinterpret_cast<const VT *>(gPrev)[i];
        VT sv, dv;
#pragma unroll
        for (int e = 0; e < VN; e++) { sv[e] = yv[e] - ypv[e]; dv[e] = gv[e] - gpv[e]; }
        reinterpret_cast<VT *>(S)[i] = sv;
        reinterpret_cast<VT *>(Y)[i] = dv;
        if (i * VN + VN - 1 < first) continue;    // replicated crown elements: counted on rank 0 only (first = 0 there)
#pragma unroll
        for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[0] = fma_rn((double)gv[e], (double)gv[e], acc[0]);
#pragma unroll
        for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[1] = fma_rn((double)sv[e], (double)dv[e], acc[1]);
#pragma unroll
        for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[2] = fma_rn((double)dv[e], (double)dv[e], acc[2]);
#pragma unroll
        for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[3] = fma_rn((double)sv[e], (double)sv[e], acc[3]);
        if (extra) {
#pragma unroll
            for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[4] = fma_rn((double)sv[e], (double)gv[e], acc[4]);
            if (sPrev) {
                const VT pv = reinterpret_cast<const VT *>(sPrev)[i];
#pragma unroll
                for (int e = 0; e < VN; e++) if (i * VN + e >= first) acc[5] = fma_rn((double)pv[e], (double)gv[e], acc[5]);
            }
        }
    }
    for (long long i = nvec * VN + gid; i < n; i += stride) {
        const T sv = y[i] - yPrev[i], dv = g[i] - gPrev[i];
        S[i] = sv; Y[i] = dv;
        if (i < first) continue;
        acc[0] = fma_rn((double)g[i], (double)g[i], acc[0]); acc[1] = fma_rn((double)sv, (double)dv, acc[1]); acc[2] = fma_rn((double)dv, (double)dv, acc[2]); acc[3] = fma_rn((double)sv, (double)sv, acc[3]);
        if (extra) { acc[4] = fma_rn((double)sv, (double)g[i], acc[4]); if (sPrev) acc[5] = fma_rn((double)sPrev[i], (double)g[i], acc[5]); }
    }
    dots_block_reduce(acc, partials);
}
// prox + fixed-point residual (+ the negated residual the quasi-Newton loops start from) in one pass, with the partials of the
// two dot products of computeValueFbe (<w, res>, <res, res>, SmpcController.cu:1430-1441) and of the prox distances:
//   z = clamp(hx + w / lambda)   res = hx - z   [gneg = -res]          (:778-792, :839-850, :1077 / :1060)
// The soft-constraint correction, if the distances trip it, is applied afterwards by k_prox_soft_res.
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_prox_res(DualArgs<T> a, T *gneg, double *dotPartials) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    __shared__ double sx[ELT_THREADS / 64], ss[ELT_THREADS / 64];
    double acc[DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) acc[j] = 0;
    double d2x = 0, d2s = 0;
    const int nx = a.nx, ny = a.ny;
    const long long nvec = a.n / VN;
    const long long stride = (long long)gridDim.x * ELT_THREADS, gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    int c0 = (int)((gid * VN) % ny);
    const int cstep = (int)((stride * VN) % ny);
    for (long long i = gid; i < nvec; i += stride) {
        const VT hx = reinterpret_cast<const VT *>(a.hx)[i], w = reinterpret_cast<const VT *>(a.w)[i];
        const VT lo = reinterpret_cast<const VT *>(a.lo)[i], hi = reinterpret_cast<const VT *>(a.hi)[i];
        VT z, r, gn;
        int c = c0;
#pragma unroll
        for (int e = 0; e < VN; e++) {
            const T t = hx[e] + a.invLambda * w[e];
            z[e] = t < lo[e] ? lo[e] : (t > hi[e] ? hi[e] : t);
            r[e] = hx[e] - z[e];
            gn[e] = -r[e];
            const double diff = (a.countCrown || i * VN + e >= a.crownElems) ? (double)(t - z[e]) : 0.0;   // sharded: the replicated crown counts on rank 0 only
            if (c < nx) d2x = fma_rn(diff, diff, d2x); else if (c < 2 * nx) d2s = fma_rn(diff, diff, d2s);
            if (++c == ny) c = 0;
        }
        reinterpret_cast<VT *>(a.z)[i] = z;
        reinterpret_cast<VT *>(a.res)[i] = r;
        if (gneg) reinterpret_cast<VT *>(gneg)[i] = gn;
#pragma unroll
        for (int e = 0; e < VN; e++) if (a.countCrown || i * VN + e >= a.crownElems) acc[0] = fma_rn((double)w[e], (double)r[e], acc[0]);
#pragma unroll
        for (int e = 0; e < VN; e++) if (a.countCrown || i * VN + e >= a.crownElems) acc[1] = fma_rn((double)r[e], (double)r[e], acc[1]);
        c0 += cstep; if (c0 >= ny) c0 -= ny;
    }
    for (long long i = nvec * VN + gid; i < a.n; i += stride) {
        const int c = (int)(i % ny);
        const T t = a.hx[i] + a.invLambda * a.w[i];
        const T z = t < a.lo[i] ? a.lo[i] : (t > a.hi[i] ? a.hi[i] : t);
        const T r = a.hx[i] - z;
        a.z[i] = z; a.res[i] = r;
        if (gneg) gneg[i] = -r;
        if (!a.countCrown && i < a.crownElems) continue;
        const double diff = (double)(t - z);
        if (c < nx) d2x = fma_rn(diff, diff, d2x); else if (c < 2 * nx) d2s = fma_rn(diff, diff, d2s);
        acc[0] = fma_rn((double)a.w[i], (double)r, acc[0]); acc[1] = fma_rn((double)r, (double)r, acc[1]);
    }
    for (int off = 32; off > 0; off >>= 1) { d2x += __shfl_down(d2x, off); d2s += __shfl_down(d2s, off); }
    if ((threadIdx.x & 63) == 0) { sx[threadIdx.x >> 6] = d2x; ss[threadIdx.x >> 6] = d2s; }
    dots_block_reduce(acc, dotPartials);   // contains the barrier the two arrays above need
    if (threadIdx.x == 0) {
        Partial p{};
        for (int k = 0; k < ELT_THREADS / 64; k++) { p.d2x += sx[k]; p.d2s += ss[k]; }
        a.partials[blockIdx.x] = p;
    }
}
// second half of the fused prox (only does anything when k_decide found a distance above its threshold): the prox of
// gamma * dist(., C) on the tripped halves (:793-815), then residual, negated residual and the dot partials redone
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_prox_soft_res(DualArgs<T> a, T *gneg, double *dotPartials) {
    if (!a.st->tripped) return;
    const T scX = (T)a.st->scaleX, scS = (T)a.st->scaleS;
    constexpr int VN = VecOf<T>::N;
    double acc[DOT_MAX];
#pragma unroll
    for (int j = 0; j < DOT_MAX; j++) acc[j] = 0;
    const long long nvec = a.n / VN;
    const long long stride = (long long)gridDim.x * ELT_THREADS, gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    auto one = [&](long long i) -> T {
        const int c = (int)(i % a.ny);
        const T t = a.hx[i] + a.invLambda * a.w[i];
        T z = a.z[i];
        if (c < 2 * a.nx) { z = z + (c < a.nx ? scX : scS) * (t - z); a.z[i] = z; }
        const T r = a.hx[i] - z;
        a.res[i] = r;
        if (gneg) gneg[i] = -r;
        return r;
    };
    for (long long i = gid; i < nvec; i += stride) {       // element order of k_dots: the VN elements of vector i, then i + stride
        T r[VN];
#pragma unroll
        for (int e = 0; e < VN; e++) r[e] = one(i * VN + e);
#pragma unroll
        for (int e = 0; e < VN; e++) if (a.countCrown || i * VN + e >= a.crownElems) acc[0] = fma_rn((double)a.w[i * VN + e], (double)r[e], acc[0]);
#pragma unroll
        for (int e = 0; e < VN; e++) if (a.countCrown || i * VN + e >= a.crownElems) acc[1] = fma_rn((double)r[e], (double)r[e], acc[1]);
    }
    for (long long i = nvec * VN + gid; i < a.n; i += stride) {
        const T r = one(i);
        if (a.countCrown || i >= a.crownElems) { acc[0] = fma_rn((double)a.w[i], (double)r, acc[0]); acc[1] = fma_rn((double)r, (double)r, acc[1]); }
    }
    dots_block_reduce(acc, dotPartials);
}

// one line-search trial step (SmpcController.cu:1274-1283 / 1383-1392): x += tau xdir, u += tau udir, w += tau dir,
// Hx += tau HxDir in one launch over the four index ranges
template <typename T>
struct TrialArgs {
    T *x, *u, *w, *hx;
    const T *xdir, *udir, *dir, *hxdir;
    long long nX, nU, nY;
    T tau;
};
template <typename T>
__global__ void k_trial_step(TrialArgs<T> g) {
    const long long total = g.nX + g.nU + 2 * g.nY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (i < g.nY) g.w[i] += g.tau * g.dir[i];
        else if (i < 2 * g.nY) { const long long k = i - g.nY; g.hx[k] += g.tau * g.hxdir[k]; }
        else if (i < 2 * g.nY + g.nX) { const long long k = i - 2 * g.nY; g.x[k] += g.tau * g.xdir[k]; }
        else { const long long k = i - 2 * g.nY - g.nX; g.u[k] += g.tau * g.udir[k]; }
    }
}

// SmpcController::dualUpdate, FBE / NAMA branch (SmpcController.cu:866-880) in one pass:
//   gPrev = g ; yPrev = y ; y = w + lambda res ; w = y
template <typename T>
__global__ void k_fbe_dual_update(T *gPrev, const T *g, T *yPrev, T *y, T *w, const T *res, T lambda, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        gPrev[i] = g[i];
        yPrev[i] = y[i];
        const T yn = w[i] + lambda * res[i];
        y[i] = yn;
        w[i] = yn;
    }
}

// the two primal terms of SmpcController::computeValueFbe (SmpcController.cu:1451-1472):
//   quad = sum_i p_i du_i' W du_i  with du_i = u_i - u_anc(i) (root: u_0 - prevU)   and   lin = sum_i p_i u_i' alpha_i
// W (nu x nu, dense in general) is shared by all nodes: a workgroup takes VALUE_TILE nodes at a time, stages their du in
// LDS and lets thread t accumulate row t of W du for the whole tile, so every element of W fetched from L2 is used
// VALUE_TILE times (the first version, one node at a time, spent 176 us per call on W loads).  partials[block][0..1]
constexpr int VALUE_THREADS = 128;
constexpr int VALUE_TILE = 16;
#ifndef RN_VALUE_JB
#define RN_VALUE_JB 8
#endif
constexpr int VALUE_JB = RN_VALUE_JB;
constexpr int VALUE_RPW = VALUE_TILE / (VALUE_THREADS / 64);   // node rows per wave
constexpr int VALUE_TC = 2;                                      // 64-wide column chunks requested together
template <typename T>
__global__ void __launch_bounds__(VALUE_THREADS) k_value_terms(const T *u, const T *prevU, const int *parent, const T *prob, const T *W,
                                                              const T *alpha, int nu, int nodes, double *partials, int firstNode) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char fbe_smem[];
    T *du = reinterpret_cast<T *>(fbe_smem);            // [nu][VALUE_TILE]: the tile's values of one column are contiguous
    __shared__ double sq[VALUE_THREADS / 64], sl[VALUE_THREADS / 64];
    double quad = 0, lin = 0;
    const int tiles = (nodes + VALUE_TILE - 1) / VALUE_TILE;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int n0 = tile * VALUE_TILE;
        const int cnt = nodes - n0 < VALUE_TILE ? nodes - n0 : VALUE_TILE;
        {   // du of the tile -> LDS: every wave takes VALUE_RPW whole node rows; the parents of all its rows are requested
            // first, then every u / u_parent / alpha value of the rows in one batch (two dependent round trips per tile
            // instead of three per element as in the first version)
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            int nodeOf[VALUE_RPW], par[VALUE_RPW];
            T pb[VALUE_RPW];
#pragma unroll
            for (int r = 0; r < VALUE_RPW; r++) {
                const int n = wave * VALUE_RPW + r;
                nodeOf[r] = (n < cnt && n0 + n >= firstNode) ? n0 + n : -1;   // sharded: the replicated crown counts on rank 0 only
                const int nc = n < cnt ? n0 + n : n0;
                par[r] = parent[nc]; pb[r] = prob[nc];
            }
            for (int t0 = lane; t0 < nu; t0 += 64 * VALUE_TC) {
                T un[VALUE_TC][VALUE_RPW], up[VALUE_TC][VALUE_RPW], al[VALUE_TC][VALUE_RPW];
#pragma unroll
                for (int c = 0; c < VALUE_TC; c++) {
                    const int t = t0 + 64 * c < nu ? t0 + 64 * c : t0;
#pragma unroll
                    for (int r = 0; r < VALUE_RPW; r++) {
                        const int nc = nodeOf[r] >= 0 ? nodeOf[r] : n0;
                        un[c][r] = u[(size_t)nc * nu + t];
                        up[c][r] = par[r] < 0 ? prevU[t] : u[(size_t)par[r] * nu + t];
                        al[c][r] = alpha[(size_t)nc * nu + t];
                    }
                }
#pragma unroll
                for (int c = 0; c < VALUE_TC; c++) {
                    const int t = t0 + 64 * c;
                    if (t < nu) {
#pragma unroll
                        for (int r = 0; r < VALUE_RPW; r++) {
                            const bool live = nodeOf[r] >= 0;
                            du[t * VALUE_TILE + wave * VALUE_RPW + r] = live ? un[c][r] - up[c][r] : (T)0;
                            if (live) lin += (double)(pb[r] * un[c][r]) * (double)al[c][r];
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nu; t += VALUE_THREADS) {
            T wd[VALUE_TILE];
#pragma unroll
            for (int n = 0; n < VALUE_TILE; n++) wd[n] = 0;
            // VALUE_JB columns of W requested together: with one load per trip the loop is a chain of 114 L2 round trips
            // per tile (measured 100 us per call); the summation order over j is unchanged
            for (int j0 = 0; j0 < nu; j0 += VALUE_JB) {
                T wv[VALUE_JB];
#pragma unroll
                for (int jj = 0; jj < VALUE_JB; jj++) wv[jj] = W[t + (size_t)(j0 + jj < nu ? j0 + jj : nu - 1) * nu];
#pragma unroll
                for (int jj = 0; jj < VALUE_JB; jj++) {
                    if (j0 + jj < nu) {
                        // the tile's du values of column j are 16-byte aligned and contiguous: VALUE_TILE / VN wide LDS reads
                        const VT *dj = reinterpret_cast<const VT *>(du + (size_t)(j0 + jj) * VALUE_TILE);
#pragma unroll
                        for (int q = 0; q < VALUE_TILE / VN; q++) {
                            const VT dv = dj[q];
#pragma unroll
                            for (int e = 0; e < VN; e++) wd[q * VN + e] += wv[jj] * dv[e];
                        }
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < VALUE_TILE; n++)
                if (n < cnt) quad += (double)(prob[n0 + n] * du[t * VALUE_TILE + n]) * (double)wd[n];
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { quad += __shfl_down(quad, off); lin += __shfl_down(lin, off); }
    if ((threadIdx.x & 63) == 0) { sq[threadIdx.x >> 6] = quad; sl[threadIdx.x >> 6] = lin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int k = 0; k < VALUE_THREADS / 64; k++) { a += sq[k]; b += sl[k]; }
        partials[(size_t)blockIdx.x * DOT_MAX + 0] = a;
        partials[(size_t)blockIdx.x * DOT_MAX + 1] = b;
    }
}

// ------------------------------------------------------------------------------------------------------
// The line search, all candidates at once (SmpcController::computeLineSearchLbfgsUpdate / ...AmeLbfgsUpdate, SmpcController.cu:1272-1300,
// 1381-1409).  The reference tries tau = 1, then moves on by -1/2, -1/4, ... while the FBE value does NOT exceed the reference value:
// every trial is a cumulative update x += tau_k xdir, u += tau_k udir, w += tau_k dir, Hx += tau_k HxDir, a prox + residual pass and
// computeValueFbe -- per trial 8 launches and two blocking read-backs here before (4-5 trials per iteration on the feasible
// workloads: 0.7 of 3.0 ms).  The trials' states depend on nothing but the increments, so LS_K of them are evaluated by two passes
// WITHOUT touching the state -- each candidate's state spelled as the sequential trial would have produced it (the same chain of
// fused multiply-adds, k_trial_step's), each candidate's reductions accumulated in the element order of the sequential kernels --
// one fold, one read-back; the host walks the reference's accept / stop logic over the values, applies the accepted number of
// steps in one pass (k_trial_multi) and runs the ordinary prox + residual pass on the final state.  A candidate whose prox
// distances trip the soft-constraint branch (never with the shipped penalties) sends the whole search down the sequential path.
constexpr int LS_K = 6;          // candidates per evaluation batch at most (11 trials in all)
constexpr int LS_FIRST = 4;      // candidates of a search's first batch
constexpr int LS_EVAL = 4;       // partials per candidate of k_ls_eval: <w, res>, <res, res>, dist^2 box half, dist^2 safety half
constexpr int LS_SCAL = 6;       // scalars per candidate after the fold: the four above, quad, lin
template <typename T>
struct LsTaus { T tau[LS_K]; int n; };
template <typename T>
__device__ __forceinline__ T trial_elem(T v, T tau, T d) { return v + tau * d; }      // k_trial_step's update of one element
// the applied steps of a finished search, in ONE pass over the four index ranges (what `n` k_trial_step launches would leave)
template <typename T>
__global__ void k_trial_multi(TrialArgs<T> g, LsTaus<T> ts) {
    const long long total = g.nX + g.nU + 2 * g.nY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        T *dst; const T *dir; long long k;
        if (i < g.nY) { dst = g.w; dir = g.dir; k = i; }
        else if (i < 2 * g.nY) { dst = g.hx; dir = g.hxdir; k = i - g.nY; }
        else if (i < 2 * g.nY + g.nX) { dst = g.x; dir = g.xdir; k = i - 2 * g.nY; }
        else { dst = g.u; dir = g.udir; k = i - 2 * g.nY - g.nX; }
        T v = dst[k];
        const T dd = dir[k];
#pragma unroll
        for (int c = 0; c < LS_K; c++) if (c < ts.n) v = trial_elem(v, ts.tau[c], dd);
        dst[k] = v;
    }
}
// prox + residual of every candidate, reduced: partials[block][candidate][LS_EVAL] (walks the range exactly as k_prox_res does)
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_ls_eval(DualArgs<T> a, const T *dir, const T *hxdir, LsTaus<T> ts, double *partials) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    __shared__ double sh[ELT_THREADS / 64][LS_K * LS_EVAL];
    double acc[LS_K][LS_EVAL];
#pragma unroll
    for (int c = 0; c < LS_K; c++)
#pragma unroll
        for (int j = 0; j < LS_EVAL; j++) acc[c][j] = 0;
    const int nx = a.nx, ny = a.ny;
    const long long nvec = a.n / VN;
    const long long stride = (long long)gridDim.x * ELT_THREADS, gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    int c0 = (int)((gid * VN) % ny);
    const int cstep = (int)((stride * VN) % ny);
    auto one = [&](T hx, T w, T lo, T hi, T dh, T dw, int col, bool counted) {
#pragma unroll
        for (int c = 0; c < LS_K; c++) {
            if (c < ts.n) {
                w = trial_elem(w, ts.tau[c], dw);
                hx = trial_elem(hx, ts.tau[c], dh);
                const T t = hx + a.invLambda * w;
                const T z = t < lo ? lo : (t > hi ? hi : t);
                const T r = hx - z;
                if (counted) {
                    const double diff = (double)(t - z);
                    acc[c][0] = fma_rn((double)w, (double)r, acc[c][0]);
                    acc[c][1] = fma_rn((double)r, (double)r, acc[c][1]);
                    // (selects, not branches: `if (box) acc[c][2] ... else acc[c][3] ...` was compiled into ONE update of acc[c][2 + half], a
                    //  dynamically indexed array -- all 24 accumulators in scratch, 112 bytes per lane)
                    const double d2b = fma_rn(diff, diff, acc[c][2]), d2s = fma_rn(diff, diff, acc[c][3]);
                    acc[c][2] = col < nx ? d2b : acc[c][2];
                    acc[c][3] = (col >= nx && col < 2 * nx) ? d2s : acc[c][3];
                }
            }
        }
    };
    for (long long i = gid; i < nvec; i += stride) {
        const VT hx = reinterpret_cast<const VT *>(a.hx)[i], w = reinterpret_cast<const VT *>(a.w)[i];
        const VT lo = reinterpret_cast<const VT *>(a.lo)[i], hi = reinterpret_cast<const VT *>(a.hi)[i];
        const VT dh = reinterpret_cast<const VT *>(hxdir)[i], dw = reinterpret_cast<const VT *>(dir)[i];
        int c = c0;
#pragma unroll
        for (int e = 0; e < VN; e++) {
            one(hx[e], w[e], lo[e], hi[e], dh[e], dw[e], c, a.countCrown || i * VN + e >= a.crownElems);
            if (++c == ny) c = 0;
        }
        c0 += cstep; if (c0 >= ny) c0 -= ny;
    }
    for (long long i = nvec * VN + gid; i < a.n; i += stride)
        one(a.hx[i], a.w[i], a.lo[i], a.hi[i], hxdir[i], dir[i], (int)(i % ny), a.countCrown || i >= a.crownElems);
    // block fold: wave shuffle -> LDS -> one partial per block, candidate and quantity (dots_block_reduce's order)
#pragma unroll
    for (int c = 0; c < LS_K; c++)
#pragma unroll
        for (int j = 0; j < LS_EVAL; j++) {
            double v = acc[c][j];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c * LS_EVAL + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < LS_K * LS_EVAL) {
        double s = 0;
        for (int k = 0; k < ELT_THREADS / 64; k++) s += sh[k][threadIdx.x];
        partials[(size_t)blockIdx.x * (LS_K * LS_EVAL) + threadIdx.x] = s;
    }
}
// the two primal terms of computeValueFbe for every candidate: k_value_terms with the candidates' u spelled out.  A workgroup takes
// LS_TILE nodes at a time; their du of all candidates sit in LDS ([candidate][nu][LS_TILE]) and thread t accumulates row t of W du
// for all of them, so W is streamed from L2 once per tile, not once per tile and candidate.  partials[block][candidate][2]
constexpr int LS_TILE = 8;
constexpr int LS_RPW = LS_TILE / (VALUE_THREADS / 64);
template <typename T>
__global__ void __launch_bounds__(VALUE_THREADS) k_ls_value(const T *u, const T *udir, const T *prevU, const int *parent, const T *prob, const T *W, const T *alpha,
                                                           int nu, int nodes, LsTaus<T> ts, double *partials, int firstNode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fbe_smem[];
    T *du = reinterpret_cast<T *>(fbe_smem);            // [LS_K][nu][LS_TILE]
    __shared__ double sq[VALUE_THREADS / 64][LS_K], sl[VALUE_THREADS / 64][LS_K];
    double quad[LS_K], lin[LS_K];
#pragma unroll
    for (int c = 0; c < LS_K; c++) { quad[c] = 0; lin[c] = 0; }
    const size_t cstride = (size_t)nu * LS_TILE;
    const int tiles = (nodes + LS_TILE - 1) / LS_TILE;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int n0 = tile * LS_TILE;
        const int cnt = nodes - n0 < LS_TILE ? nodes - n0 : LS_TILE;
        {
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            int nodeOf[LS_RPW], par[LS_RPW];
            T pb[LS_RPW];
#pragma unroll
            for (int r = 0; r < LS_RPW; r++) {
                const int n = wave * LS_RPW + r;
                nodeOf[r] = (n < cnt && n0 + n >= firstNode) ? n0 + n : -1;
                const int nc = n < cnt ? n0 + n : n0;
                par[r] = parent[nc]; pb[r] = prob[nc];
            }
            for (int t = lane; t < nu; t += 64) {
                T un[LS_RPW], up[LS_RPW], dn[LS_RPW], dp[LS_RPW], al[LS_RPW];
#pragma unroll
                for (int r = 0; r < LS_RPW; r++) {
                    const int nc = nodeOf[r] >= 0 ? nodeOf[r] : n0;
                    un[r] = u[(size_t)nc * nu + t]; dn[r] = udir[(size_t)nc * nu + t];
                    up[r] = par[r] < 0 ? prevU[t] : u[(size_t)par[r] * nu + t];
                    dp[r] = par[r] < 0 ? (T)0 : udir[(size_t)par[r] * nu + t];
                    al[r] = alpha[(size_t)nc * nu + t];
                }
#pragma unroll
                for (int c = 0; c < LS_K; c++) {
                    if (c < ts.n) {
#pragma unroll
                        for (int r = 0; r < LS_RPW; r++) {
                            un[r] = trial_elem(un[r], ts.tau[c], dn[r]);
                            if (par[r] >= 0) up[r] = trial_elem(up[r], ts.tau[c], dp[r]);     // the root's predecessor is prevU: not part of the iterate
                            const bool live = nodeOf[r] >= 0;
                            du[c * cstride + (size_t)t * LS_TILE + wave * LS_RPW + r] = live ? un[r] - up[r] : (T)0;
                            if (live) lin[c] += (double)(pb[r] * un[r]) * (double)al[r];
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nu; t += VALUE_THREADS) {
            T wd[LS_K][LS_TILE];
#pragma unroll
            for (int c = 0; c < LS_K; c++)
#pragma unroll
                for (int n = 0; n < LS_TILE; n++) wd[c][n] = 0;
            for (int j0 = 0; j0 < nu; j0 += VALUE_JB) {      // the summation order over j is k_value_terms's
                T wv[VALUE_JB];
#pragma unroll
                for (int jj = 0; jj < VALUE_JB; jj++) wv[jj] = W[t + (size_t)(j0 + jj < nu ? j0 + jj : nu - 1) * nu];
#pragma unroll
                for (int jj = 0; jj < VALUE_JB; jj++) {
                    if (j0 + jj < nu) {
#pragma unroll
                        for (int c = 0; c < LS_K; c++) {
                            if (c < ts.n) {
                                const T *dj = du + c * cstride + (size_t)(j0 + jj) * LS_TILE;
#pragma unroll
                                for (int n = 0; n < LS_TILE; n++) wd[c][n] += wv[jj] * dj[n];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < LS_K; c++)
                if (c < ts.n) {
#pragma unroll
                    for (int n = 0; n < LS_TILE; n++)
                        if (n < cnt) quad[c] += (double)(prob[n0 + n] * du[c * cstride + (size_t)t * LS_TILE + n]) * (double)wd[c][n];
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < LS_K; c++) {
        double q = quad[c], l = lin[c];
        for (int off = 32; off > 0; off >>= 1) { q += __shfl_down(q, off); l += __shfl_down(l, off); }
        if ((threadIdx.x & 63) == 0) { sq[threadIdx.x >> 6][c] = q; sl[threadIdx.x >> 6][c] = l; }
    }
    __syncthreads();
    if (threadIdx.x < LS_K) {
        double a = 0, b = 0;
        for (int k = 0; k < VALUE_THREADS / 64; k++) { a += sq[k][threadIdx.x]; b += sl[k][threadIdx.x]; }
        partials[((size_t)blockIdx.x * LS_K + threadIdx.x) * 2 + 0] = a;
        partials[((size_t)blockIdx.x * LS_K + threadIdx.x) * 2 + 1] = b;
    }
}
// the fold of both partial arrays: workgroup j folds one (candidate, quantity) column in k_dots_finish's order
// out[candidate][LS_SCAL] = {<w, res>, <res, res>, dist^2 box, dist^2 safety, quad, lin}
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_ls_finish(const double *evalPartials, int nbEval, const double *valPartials, int nbVal, double *out) {
    __shared__ double sh[ELT_THREADS];
    const int c = blockIdx.x / LS_SCAL, q = blockIdx.x % LS_SCAL;
    double s = 0;
    if (q < LS_EVAL) { for (int b = threadIdx.x; b < nbEval; b += ELT_THREADS) s += evalPartials[(size_t)b * (LS_K * LS_EVAL) + c * LS_EVAL + q]; }
    else { for (int b = threadIdx.x; b < nbVal; b += ELT_THREADS) s += valPartials[((size_t)b * LS_K + c) * 2 + (q - LS_EVAL)]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = ELT_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c * LS_SCAL + q] = sh[0];
}

// The primal terms of computeValueFbe on the matrix cores: quad_c = sum_i p_i du_i' (W du_i) for up to LS_K candidate states (or the
// current state: plain != 0) -- W du for a slab of 16 nodes is one shared-operator product, the same MFMA loop as the sweep's
// (slab_mfma over W stored padded like every shared operator), and the dot with du rides in the epilogue.  k_value_terms / k_ls_value
// do this on the vector ALUs at a tenth of their peak (33 us for one state, 25 us per candidate on the 493-scenario tree: every W
// element crosses LDS-fed FMAs 16 or 8 nodes at a time); 10 864 x 114 x 114 is 0.3 GFLOP, 4 us of fp64 MFMA time.
// A workgroup owns slabs blockIdx, blockIdx + grid, ...; a thread keeps its elements of (u, udir) of the slab's nodes and of their
// parents in registers and advances them candidate by candidate with k_trial_step's update (trial_elem: the state the sequential
// trial would have), so the candidates share one round of loads.  Summation over j inside (W du)_t is the MFMA's, not the serial
// loop's: the values differ from k_value_terms's in the last bits; both searches (batched and trial by trial) and the reference
// value they are compared with come from THIS kernel whenever it is the one in use, so the accept / stop decisions are consistent.
// partials[block * blockStride + c * candStride + {0, 1}] = {quad_c, lin_c}
constexpr int VM_WAVES = 8;
constexpr int VM_MAXE = 6;            // elements of the 16 x nu slab per thread at most (nu <= 192)
constexpr int VM_KMAX = 32;           // k-steps (of 4) of W a wave keeps in registers at most
template <typename T>
__global__ void __launch_bounds__(64 * VM_WAVES) k_value_mfma(const T *u, const T *udir, const T *prevU, const int *parent, const T *prob, const T *Wp, int mp, int kp,
                                                              const T *alpha, int nu, int nodes, LsTaus<T> ts, int plain, double *partials, int blockStride,
                                                              int candStride, int firstNode, int SB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fbe_smem[];
    T *sB = reinterpret_cast<T *>(fbe_smem);            // [16][SB] du of the slab's nodes, zero beyond nu and for nodes that do not count
    __shared__ double sq[VM_WAVES][LS_K], sl[VM_WAVES][LS_K];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = (nu + 15) / 16, ksteps = kp / 4;
    const int nCand = plain ? 1 : ts.n;
    if (lane < LS_K) { sq[wave][lane] = 0; sl[wave][lane] = 0; }               // per-wave running sums of every candidate (lane 0 adds to them)
    // nu <= 128 (at most one 16-row tile of W per wave, K <= 128): the wave's A fragments of W -- the same for every slab and candidate --
    // are loaded ONCE into registers, so the products' inner loop reads LDS only (with the fragments fetched from L2 group by group the
    // launch was bound by that latency: one workgroup per CU, 65 us for four candidates on the 493-scenario tree)
    const bool wInRegs = tiles <= VM_WAVES && ksteps <= VM_KMAX;
    T wreg[VM_KMAX];
    if (wInRegs) {
        const T *Ap = Wp + (size_t)(wave < tiles ? wave : 0) * 16 + (lane & 15) + (size_t)(lane >> 4) * mp;
#pragma unroll
        for (int ks = 0; ks < VM_KMAX; ks++) wreg[ks] = Ap[(size_t)(ks < ksteps ? ks : 0) * 4 * mp];
    }
    for (int i = threadIdx.x; i < 16 * SB; i += 64 * VM_WAVES) sB[i] = (T)0;      // the padding columns stay zero
    const int slabs = (nodes + 15) / 16, perSlab = 16 * nu;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int node0 = slab * 16;
        T un[VM_MAXE], dn[VM_MAXE], up[VM_MAXE], dp[VM_MAXE], al[VM_MAXE], pb[VM_MAXE];
        int pos[VM_MAXE];                       // LDS position of the element, -1: none; bit 30: the node does not count (du = 0, no lin)
        bool rootPar[VM_MAXE];
#pragma unroll
        for (int e = 0; e < VM_MAXE; e++) {
            const int idx = (int)threadIdx.x + e * 64 * VM_WAVES;
            const bool has = idx < perSlab;
            const int r = has ? idx / nu : 0, k = has ? idx - r * nu : 0;
            const int node = node0 + r < nodes ? node0 + r : nodes - 1;
            const bool live = has && node0 + r < nodes && node >= firstNode;     // sharded: the replicated crown counts on rank 0 only
            const int par = parent[node];
            pos[e] = has ? (r * SB + k) | (live ? 0 : (1 << 30)) : -1;
            rootPar[e] = par < 0;
            pb[e] = prob[node];
            un[e] = u[(size_t)node * nu + k];
            dn[e] = plain ? (T)0 : udir[(size_t)node * nu + k];
            up[e] = par < 0 ? prevU[k] : u[(size_t)par * nu + k];
            dp[e] = (plain || par < 0) ? (T)0 : udir[(size_t)par * nu + k];
            al[e] = alpha[(size_t)node * nu + k];
        }
        for (int c = 0; c < nCand; c++) {
            __syncthreads();                    // the previous candidate's (slab's) products have read sB
            double q = 0, l = 0;
            const T tau = plain ? (T)0 : ts.tau[c];
#pragma unroll
            for (int e = 0; e < VM_MAXE; e++) {
                if (pos[e] < 0) continue;
                if (!plain) {
                    un[e] = trial_elem(un[e], tau, dn[e]);
                    if (!rootPar[e]) up[e] = trial_elem(up[e], tau, dp[e]);      // the root's predecessor is prevU: not part of the iterate
                }
                const bool live = !(pos[e] & (1 << 30));
                sB[pos[e] & ~(1 << 30)] = live ? un[e] - up[e] : (T)0;
                if (live) l += (double)(pb[e] * un[e]) * (double)al[e];
            }
            __syncthreads();
            const int col = lane & 15;
            const int nodeC = node0 + col < nodes ? node0 + col : nodes - 1;
            const T pc = prob[nodeC];
            if (wInRegs) {
                if (wave < tiles) {
                    typename Mfma16<T>::acc_t acc = {0, 0, 0, 0};
                    const T *Bp = sB + col * SB + (lane >> 4);
#pragma unroll
                    for (int g = 0; g < VM_KMAX / 4; g++) {
                        if (g * 4 < ksteps) {      // ksteps is a whole number of groups of 4 (pad_k)
                            T b[4];
#pragma unroll
                            for (int i = 0; i < 4; i++) b[i] = Bp[(g * 4 + i) * 4];
#pragma unroll
                            for (int i = 0; i < 4; i++) acc = Mfma16<T>::run(wreg[g * 4 + i], b[i], acc);
                        }
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int gr = wave * 16 + Mfma16<T>::row(lane, reg);
                        if (gr < nu) q += (double)(pc * sB[col * SB + gr]) * (double)acc[reg];
                    }
                }
            } else
            for (int t0 = wave; t0 < tiles; t0 += VM_WAVES) {
                typename Mfma16<T>::acc_t acc[1];
                slab_mfma<T, 1, RN_SLAB_KU, false>(acc, Wp, mp, t0, VM_WAVES, tiles, ksteps, sB, SB, lane);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t0 * 16 + Mfma16<T>::row(lane, reg);
                    if (gr < nu) q += (double)(pc * sB[col * SB + gr]) * (double)acc[0][reg];      // du = 0 for the nodes that do not count
                }
            }
            for (int off = 32; off > 0; off >>= 1) { q += __shfl_down(q, off); l += __shfl_down(l, off); }
            if (lane == 0) { sq[wave][c] += q; sl[wave][c] += l; }
        }
        __syncthreads();
    }
    __syncthreads();
    if ((int)threadIdx.x < nCand) {
        double a = 0, b = 0;
        for (int k = 0; k < VM_WAVES; k++) { a += sq[k][threadIdx.x]; b += sl[k][threadIdx.x]; }
        partials[(size_t)blockIdx.x * blockStride + (size_t)threadIdx.x * candStride + 0] = a;
        partials[(size_t)blockIdx.x * blockStride + (size_t)threadIdx.x * candStride + 1] = b;
    }
}

// updatePrimalInfeasibity inside the loops: fold k_absmax's partials (first-index ties, as the host fold of rn_update_primal_infeasibility)
// into out[0..4) = (v_xi, -v_xi, v_psi, -v_psi): the form a MAX all-reduce over the ranks can combine (largest magnitude and its sign)
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_inf_fold(const Partial *partials, int nblocks, double *out) {
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, true, tx2, ts2, p);
    if (threadIdx.x == 0) { out[0] = p.valXi; out[1] = -p.valXi; out[2] = p.valPsi; out[3] = -p.valPsi; }
}

}  // namespace rn
#endif
