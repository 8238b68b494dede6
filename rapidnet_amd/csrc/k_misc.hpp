// k_misc.hpp -- streaming probes, the factor step and the affine terms on the device, small utility kernels
// (part of the kernel sources of librapidnet_hip.so; kernels.hpp includes every family header, the translation units k_*.hip instantiate them)
#pragma once
#include "common.hpp"

namespace rn {

// ------------------------------------------------------------------------------------------------------
// HBM ceiling probes for bench.py (rn_measure_hbm): what kernels that do nothing but stream reach on THIS box -- the practical
// denominator next to the 8 TB/s spec.  16 B per lane per load, non-temporal, flat: the whole grid sweeps one region, thread-interleaved.
// (The probes in the solver's access shapes -- chunk per workgroup, lockstep pieces -- live in tools/probes/probe_stream.hip.)
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(256) k_bw_read(const nat_d2 *src, long long n, double *sink) {
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const nat_d2 v = __builtin_nontemporal_load(src + i);
        acc += v[0] + v[1];
    }
    if (acc == 1.2345e-300) sink[blockIdx.x & 65535] = acc;   // keeps the loads alive without a store stream
}
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(256) k_bw_copy(const nat_d2 *src, nat_d2 *dst, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);   // non-temporal both ways: the fastest copy variant of probe_stream.hip
}

// ------------------------------------------------------------------------------------------------------
// Factor step on the device (Engine::factorStep Engine.cu:671-774 + preconditioning Utilities.cu:33-58,
// 360-405): expands the per-node blocks from the shared factors computed on the host in fp64,
//   Phi_i(:,c) = -T1(:,j) d_c / (2 sqrt p_i),  D_i(:,c)    = Bbt(:,j) sqrt(p_i) d_c     c = xi column of tank j
//   Psi_i(:,c) = -T2(:,j) d_c / (2 sqrt p_i),  Ftil_i(:,c) = Lt(:,j)  sqrt(p_i) d_c     c = psi column of input j
// with T1 = Rinv Bbt, T2 = Rinv L'.  One workgroup per (node, column); pure streaming store.
template <typename T>
struct ExpandArgs {
    TreeDev<T> tr;
    int nx, nu, nv, ny, LD, nodes;
    size_t strideA;
    const T *T1, *T2, *Bbt, *Lt;
    T *A;
    // scaled bounds in y order
    int skipBlocks;       // structured operator mode: only the scaled bounds are produced
    const T *blo, *bhi;   // [ny] unscaled: xmin|xsafe|umin and xmax|+BIG|umax
    T *lo, *hi;           // [node][ny]
};
template <typename T>
__global__ void k_expand_operators(ExpandArgs<T> a) {
    const int node = blockIdx.x;
    const int stage = a.tr.stageOf[node];
    const T sp = a.tr.sqrtp[node];
    const T *dy = a.tr.dy + (size_t)stage * a.ny;
    for (int c = blockIdx.y; c < a.ny; c += gridDim.y) {
        const T d = dy[c];
        const T s1 = (T)(-0.5) * d / sp, s2 = sp * d;
        const T *m1, *m2;
        if (c < 2 * a.nx) { const int j = c % a.nx; m1 = a.T1 + (size_t)j * a.nv; m2 = a.Bbt + (size_t)j * a.nv; }
        else { const int j = c - 2 * a.nx; m1 = a.T2 + (size_t)j * a.nv; m2 = a.Lt + (size_t)j * a.nv; }
        if (!a.skipBlocks) {
            T *col = a.A + (size_t)node * a.strideA + (size_t)c * a.LD;
            for (int r = threadIdx.x; r < a.LD; r += blockDim.x)
                col[r] = r < a.nv ? s1 * m1[r] : (r < 2 * a.nv ? s2 * m2[r - a.nv] : (T)0);
        }
        if (threadIdx.x == 0) {
            // bound scaling: preconditionConstraintX/U.  "+BIG" stays +BIG (no upper bound on the safety half)
            const T k = sp * d;
            a.lo[(size_t)node * a.ny + c] = k * a.blo[c];
            const bool safety = (c >= a.nx && c < 2 * a.nx);
            a.hi[(size_t)node * a.ny + c] = safety ? a.bhi[c] : k * a.bhi[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Per-control-step affine terms (Engine::eliminateInputDistubanceCoupling Engine.cu:1147-1298), two kernels,
// one workgroup per node:
//   k_affine_demand: d_i = errD_i + dhat[stage]; e_i = Gd d_i; uhat_i = Lhat d_i;
//                    alpha_i = w_e (errP_i + ahat[stage] + alpha1)
//   k_affine_beta:   zeta_i = p_i (uhat_i - uhat_anc) - sum_c p_c (uhat_c - uhat_i)   (Utilities.cu:69-131)
//                    beta_i = 2 (W L)' zeta_i + p_i L' alpha_i
template <typename T>
struct AffineArgs {
    TreeDev<T> tr;
    int nx, nu, nv, nd;
    const T *Gd, *Lhat, *WLt, *Lt;      // WLt = (W L)' (nv x nu), Lt = L' (nv x nu)
    const T *errD, *errP, *dhat, *ahat, *alpha1, *prevUhat;
    T wEco; int useErrD, useErrP;
    T *e, *uhat, *alpha, *beta;
    // multi-GPU: children moments of the cut parents (whole tree): momE [parents][nd] = sum_c p_c errD_c, momP = sum_c p_c
    const T *momE, *momP; int cutStage;
};
constexpr int AFF_THREADS = 128;
template <typename T>
__global__ void __launch_bounds__(AFF_THREADS) k_affine_demand(AffineArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh_d = reinterpret_cast<T *>(smem_raw);                 // nd
    T *sh_o = sh_d + ((a.nd + 3) & ~3);                        // max(nx, nu)
    T *sh_scr = sh_o + ((max(a.nx, a.nu) + 3) & ~3);
    const int node = blockIdx.x, tid = threadIdx.x;
    const int stage = a.tr.stageOf[node];
    for (int t = tid; t < a.nd; t += AFF_THREADS)
        sh_d[t] = (a.useErrD ? a.errD[(size_t)node * a.nd + t] : (T)0) + a.dhat[(size_t)stage * a.nd + t];
    __syncthreads();
    block_gemv_shared<T>(a.Gd, a.nx, a.nd, sh_d, sh_o, sh_scr, AFF_THREADS);
    for (int t = tid; t < a.nx; t += AFF_THREADS) a.e[(size_t)node * a.nx + t] = sh_o[t];
    __syncthreads();
    block_gemv_shared<T>(a.Lhat, a.nu, a.nd, sh_d, sh_o, sh_scr, AFF_THREADS);
    for (int t = tid; t < a.nu; t += AFF_THREADS) {
        a.uhat[(size_t)node * a.nu + t] = sh_o[t];
        const T ep = a.useErrP ? a.errP[(size_t)node * a.nu + t] : (T)0;
        a.alpha[(size_t)node * a.nu + t] = a.wEco * (ep + (a.ahat[(size_t)stage * a.nu + t] + a.alpha1[t]));
    }
}
template <typename T>
__global__ void __launch_bounds__(AFF_THREADS) k_affine_beta(AffineArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh_z = reinterpret_cast<T *>(smem_raw);                 // nu  zeta
    T *sh_a = sh_z + ((a.nu + 3) & ~3);                        // nu  alpha
    T *sh_o = sh_a + ((a.nu + 3) & ~3);                        // max(nv, nu)
    T *sh_o2 = sh_o + ((max(a.nv, a.nu) + 3) & ~3);            // nv
    T *sh_d = sh_o2 + ((a.nv + 3) & ~3);                       // nd
    T *sh_scr = sh_d + ((a.nd + 3) & ~3);
    const int node = blockIdx.x, tid = threadIdx.x, nu = a.nu;
    const int par = a.tr.parent[node];
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const T p = a.tr.prob[node];
    const int stage = a.tr.stageOf[node];
    const bool presummed = a.momE != nullptr && stage == a.cutStage - 1;
    if (presummed) {   // sum_c p_c uhat_c = Lhat (E_i + P_i dhat[stage+1]) over ALL children, local or not
        const int pos = node - a.tr.stageCum[stage];
        const T P = a.momP[pos];
        for (int t = tid; t < a.nd; t += AFF_THREADS)
            sh_d[t] = (a.useErrD ? a.momE[(size_t)pos * a.nd + t] : (T)0) + P * a.dhat[(size_t)(stage + 1) * a.nd + t];
        __syncthreads();
        block_gemv_shared<T>(a.Lhat, nu, a.nd, sh_d, sh_o, sh_scr, AFF_THREADS);
    }
    for (int t = tid; t < nu; t += AFF_THREADS) {
        const T ui = a.uhat[(size_t)node * nu + t];
        const T ua = par < 0 ? a.prevUhat[t] : a.uhat[(size_t)par * nu + t];
        T z = p * (ui - ua);
        if (presummed) z -= sh_o[t] - a.momP[node - a.tr.stageCum[stage]] * ui;
        else for (int c = 0; c < nc; c++) z -= a.tr.prob[c0 + c] * (a.uhat[(size_t)(c0 + c) * nu + t] - ui);
        sh_z[t] = z;
        sh_a[t] = a.alpha[(size_t)node * nu + t];
    }
    __syncthreads();
    block_gemv_shared<T>(a.WLt, a.nv, nu, sh_z, sh_o, sh_scr, AFF_THREADS);
    block_gemv_shared<T>(a.Lt, a.nv, nu, sh_a, sh_o2, sh_scr, AFF_THREADS);
    for (int t = tid; t < a.nv; t += AFF_THREADS) a.beta[(size_t)node * a.nv + t] = (T)2 * sh_o[t] + p * sh_o2[t];
}

// small utilities ------------------------------------------------------------------------------------
// (one workgroup each, once per control step; the columns are requested eight at a time -- one at a time the loop was a chain of
//  `cols` dependent round trips, 35-50 us for a 63 x 114 matrix -- and added in the same order as before)
template <typename T>
__global__ void k_bw0(const T *__restrict__ B, int nx, int nu, const T *__restrict__ prevU, const T *__restrict__ prevUhat, T *__restrict__ bw0) {   // bw0 = B (prevU - prevUhat)
    for (int r = threadIdx.x; r < nx; r += blockDim.x) {
        T s = 0;
        for (int j0 = 0; j0 < nu; j0 += 8) {
            T m[8], d[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const int j = j0 + i < nu ? j0 + i : nu - 1; m[i] = B[r + (size_t)j * nx]; d[i] = prevU[j] - prevUhat[j]; }
#pragma unroll
            for (int i = 0; i < 8; i++) if (j0 + i < nu) s += m[i] * d[i];
        }
        bw0[r] = s;
    }
}
template <typename T>
__global__ void k_gemv_small(const T *__restrict__ M, int rows, int cols, const T *__restrict__ x, T *__restrict__ y) {   // y = M x, one block
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        T s = 0;
        for (int j0 = 0; j0 < cols; j0 += 8) {
            T m[8], v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const int j = j0 + i < cols ? j0 + i : cols - 1; m[i] = M[r + (size_t)j * rows]; v[i] = x[j]; }
#pragma unroll
            for (int i = 0; i < 8; i++) if (j0 + i < cols) s += m[i] * v[i];
        }
        y[r] = s;
    }
}
// y-layout <-> reference layout ([node][2nx] xi arrays and [node][nu] psi arrays)
template <typename T>
__global__ void k_pack(T *y, T *part, int ny, int off, int dim, long long nodes, int toY) {
    const long long n = nodes * dim;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long node = i / dim; const int t = (int)(i % dim);
        if (toY) y[node * ny + off + t] = part[i]; else part[i] = y[node * ny + off + t];
    }
}
template <typename T>
__global__ void k_clamp_vec(T *u, const T *lo, const T *hi, int n) {   // projectionBox<<<1,nu>>> SmpcController.cu:1649
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const T v = u[i]; u[i] = v < lo[i] ? lo[i] : (v > hi[i] ? hi[i] : v); }
}


// Structured mode, composite operator (k_gemm_comp): the forward walk's affine terms join the product's constant operand, so that the walk's running
// sums of [L v_i ; B L v_i] are u_i and x_i - x_anc themselves and the walk requests neither uhat nor eb (SweepArgs::lin bit 2):
//   lvconst_i[0 .. nu) += uhat_i - uhat_anc   (root: - prevUhat) ;   lvconst_i[nu .. nu + nx) += eb_i - eb_anc   (root: eb_0)
template <typename T>
__global__ void k_fold_affine(T *lvconst, const T *uhat, const T *eb, const T *prevUhat, const int *parent, int nodes, int nu, int nx) {
    const int w = nu + nx;
    const long long n = (long long)nodes * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int node = (int)(i / w), t = (int)(i % w), par = parent[node];
        T add;
        if (t < nu) add = uhat[(size_t)node * nu + t] - (par < 0 ? prevUhat[t] : uhat[(size_t)par * nu + t]);
        else { const int j = t - nu; add = eb[(size_t)node * nx + j] - (par < 0 ? (T)0 : eb[(size_t)par * nx + j]); }
        lvconst[i] += add;
    }
}

}  // namespace rn
