// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the APG solve path.
//
// Data layout in HBM (T = float | double), chosen for coalesced 16-byte-per-lane streaming:
//   A      [node][ny][LD]   per-node operator block: column c (c indexes y = [xi_box | xi_safe | psi]) holds
//                           rows 0..nv-1 = [Phi_i | Psi_i](:,c)  and rows nv..2nv-1 = [D_i | Ftil_i](:,c),
//                           zero padded to LD = roundup(2nv, 4).  One pass over A_i yields both mat-vecs of
//                           SmpcController::solveStep's backward sweep (SmpcController.cu:617-638).
//   y-like [node][ny]       dual vectors, ny = 2nx+nu: xi (2nx) and psi (nu) of a node are adjacent.
//   x,u,v,q,rt,beta,uhat,e  [node][dim], nodes numbered breadth-first so a stage is one contiguous slab.
// Shared small operators (L2-resident): Rinv = (L'WL)^-1 (nv x nv), Bbt = (BL)' (nv x nx), L (nu x nv),
// B (nx x nu); the reference keeps K identical copies of these (Engine.cu:306-308) and per-position
// Omega_i = Rinv/p_i, Theta_i = -Rinv*Bbt/(2 p_i) (Engine.cu:707-737); here they are applied as
// (shared matrix) x (per-node scalar).
#pragma once
#include <hip/hip_runtime.h>

namespace rn {

constexpr int RPL = 4;          // rows of A per lane (16 B fp32 / 32 B fp64 per lane per column)
constexpr int BWD_THREADS = 256;
constexpr int BWD_WAVES = BWD_THREADS / 64;
constexpr int ELT_THREADS = 256;
constexpr int ELT_MAX_BLOCKS = 1024;

template <typename T> struct Vec4 { T v[4]; };

template <typename T>
struct TreeDev {
    const int *stageCum;    // [N+1]
    const int *parent;      // [nodes] 0-based, -1 for the root
    const int *childStart;  // [nodes] first child (0-based), children are contiguous
    const int *childCount;  // [nodes]
    const int *stageOf;     // [nodes]
    const T *sqrtp;         // [nodes] sqrt(p_i)
    const T *prob;          // [nodes]
    const T *dy;            // [N][ny] preconditioner diagonal in y order: d_x | d_xs | d_u
};

template <typename T>
struct SweepArgs {
    TreeDev<T> tr;
    int nx, nu, nv, ny, LD;
    const T *A;
    const T *Rinv, *Bbt, *L, *B;
    const T *beta, *uhat, *e;
    const T *curX, *prevU, *prevUhat;
    const T *w;       // accelerated dual the sweep is evaluated at, [node][ny]
    T *v, *rt, *q;    // rt_i = r_i + Bbt q_i (what the parent consumes), q_i
    T *x, *u, *hx;
    const T *cutSums; // multi-GPU: [cutParents][nv+nx] all-reduced children sums, or nullptr
    int cutStage;     // stage whose parents take cutSums instead of summing their local children (-1: none)
};

// ------------------------------------------------------------------------------------------------------
// wide loads of RPL consecutive rows
template <typename T> __device__ __forceinline__ void load_rows(const T *p, T (&a)[RPL]);
typedef double nat_d2 __attribute__((ext_vector_type(2)));
typedef float nat_f4 __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ void load_rows<double>(const double *p, double (&a)[RPL]) {
    const nat_d2 lo = __builtin_nontemporal_load(reinterpret_cast<const nat_d2 *>(p));
    const nat_d2 hi = __builtin_nontemporal_load(reinterpret_cast<const nat_d2 *>(p) + 1);
    a[0] = lo.x; a[1] = lo.y; a[2] = hi.x; a[3] = hi.y;
}
template <> __device__ __forceinline__ void load_rows<float>(const float *p, float (&a)[RPL]) {
    const nat_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nat_f4 *>(p));
    a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
}

// out[r] = sum_j M[r + j*rows] * vec[j] for r < rows, computed by the whole block: thread (h, r) with
// r = tid % RB, h = tid / RB sums columns j == h (mod H); partials are combined through `scratch` (>= H*RB).
// M is a shared, L2-resident matrix.  Result is left in out[] (LDS) after the trailing barrier.
template <typename T>
__device__ __forceinline__ void block_gemv_shared(const T *__restrict__ M, int rows, int cols, const T *vec,
                                                  T *out, T *scratch, int nthreads) {
    int RB = 64;
    while (RB < rows && RB < nthreads) RB <<= 1;
    const int H = nthreads / RB;  // nthreads and RB are powers of two times 64
    const int tid = threadIdx.x;
    const int r = tid % RB, h = tid / RB;
    if (rows <= RB) {
        T s = 0;
        if (r < rows && h < H) {
            int j = h;
            for (; j + 3 * H < cols; j += 4 * H) {
                const T m0 = M[r + (size_t)j * rows], m1 = M[r + (size_t)(j + H) * rows];
                const T m2 = M[r + (size_t)(j + 2 * H) * rows], m3 = M[r + (size_t)(j + 3 * H) * rows];
                s += m0 * vec[j] + m1 * vec[j + H] + m2 * vec[j + 2 * H] + m3 * vec[j + 3 * H];
            }
            for (; j < cols; j += H) s += M[r + (size_t)j * rows] * vec[j];
        }
        if (h < H) scratch[h * RB + r] = s;
        __syncthreads();
        if (tid < rows) {
            T t = 0;
            for (int k = 0; k < H; k++) t += scratch[k * RB + tid];
            out[tid] = t;
        }
        __syncthreads();
    } else {  // more rows than threads: plain row loop
        for (int rr = tid; rr < rows; rr += nthreads) {
            T s = 0;
            for (int j = 0; j < cols; j++) s += M[rr + (size_t)j * rows] * vec[j];
            out[rr] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// Backward sweep, one stage (SmpcController.cu:593-673 + solveSumChildren Utilities.cu:168-201), one
// workgroup per node i of the stage:
//   q_i  = F_i' xi_i + sum_children q_c                                  (:651/656, :665)
//   s_i  = beta_i + sum_children rt_c        ( = sigma_i + Gtil q_in,   :599, :644, :667 )
//   [m1; m2] = A_i y_i                       ( = Phi xi + Psi psi ; D xi + Ftil psi,  :617-638 )
//   v_i  = -1/(2 p_i) Rinv s_i + m1          ( = -1/2 Omega sigma + Theta q_in + ...,  :604-623 )
//   rt_i = s_i + m2 + Bbt q_i                ( r_i of the reference plus the Gtil q term its parent adds )
// HBM traffic per node: LD*ny*sizeof(T) for A_i (read once, non-temporal) + O(ny + nv + nx) vectors.
template <typename T>
__global__ void __launch_bounds__(BWD_THREADS) k_backward_stage(SweepArgs<T> a, int stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh_y = reinterpret_cast<T *>(smem_raw);          // ny
    T *sh_s = sh_y + ((a.ny + 3) & ~3);                 // nv
    T *sh_q = sh_s + ((a.nv + 3) & ~3);                 // nx
    T *sh_g = sh_q + ((a.nx + 3) & ~3);                 // nv   (Rinv s, then reused for Bbt q)
    T *sh_red = sh_g + ((a.nv + 3) & ~3);               // max(BWD_WAVES*LDp, BWD_THREADS) scratch
    const int tid = threadIdx.x;
    const int node = a.tr.stageCum[stage] + blockIdx.x;
    const int nx = a.nx, nu = a.nu, nv = a.nv, ny = a.ny, LD = a.LD;
    const T *dy = a.tr.dy + (size_t)stage * ny;
    const T sp = a.tr.sqrtp[node];
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const bool presummed = (a.cutSums != nullptr) && (stage == a.cutStage - 1);

    for (int c = tid; c < ny; c += BWD_THREADS) sh_y[c] = a.w[(size_t)node * ny + c];
    __syncthreads();
    for (int t = tid; t < nx; t += BWD_THREADS) {
        T qv = sp * (dy[t] * sh_y[t] + dy[nx + t] * sh_y[nx + t]);
        if (presummed) qv += a.cutSums[(size_t)blockIdx.x * (nv + nx) + nv + t];
        else for (int c = 0; c < nc; c++) qv += a.q[(size_t)(c0 + c) * nx + t];
        sh_q[t] = qv;
        a.q[(size_t)node * nx + t] = qv;
    }
    for (int t = tid; t < nv; t += BWD_THREADS) {
        T s = a.beta[(size_t)node * nv + t];
        if (presummed) s += a.cutSums[(size_t)blockIdx.x * (nv + nx) + t];
        else for (int c = 0; c < nc; c++) s += a.rt[(size_t)(c0 + c) * nv + t];
        sh_s[t] = s;
    }
    __syncthreads();

    // ---- stream A_i: wave w takes (row block rb, column phase cp); lane l owns rows rb*256 + 4l .. +3
    const int wave = tid >> 6, lane = tid & 63;
    const int nRB = (LD + 64 * RPL - 1) / (64 * RPL);     // row blocks of 256 rows
    const int nCP = BWD_WAVES / nRB > 0 ? BWD_WAVES / nRB : 1;
    for (int rb0 = 0; rb0 < nRB; rb0 += BWD_WAVES) {       // nRB > BWD_WAVES only for very large nv
        const int rb = rb0 + (wave % (nRB < BWD_WAVES ? nRB : BWD_WAVES));
        const int cp = wave / (nRB < BWD_WAVES ? nRB : BWD_WAVES);
        const int row = rb * 64 * RPL + lane * RPL;
        T part[RPL] = {0, 0, 0, 0};
        if (rb < nRB && cp < nCP && row < LD) {
            const T *Ab = a.A + (size_t)node * ny * LD + row;
            int c = cp;
            for (; c + 7 * nCP < ny; c += 8 * nCP) {     // 8 columns in flight per lane
                T m[8][RPL];
#pragma unroll
                for (int k = 0; k < 8; k++) load_rows<T>(Ab + (size_t)(c + k * nCP) * LD, m[k]);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const T yc = sh_y[c + k * nCP];
#pragma unroll
                    for (int r = 0; r < RPL; r++) part[r] += m[k][r] * yc;
                }
            }
            for (; c < ny; c += nCP) {
                T m[RPL];
                load_rows<T>(Ab + (size_t)c * LD, m);
                const T yc = sh_y[c];
#pragma unroll
                for (int r = 0; r < RPL; r++) part[r] += m[r] * yc;
            }
        }
        // combine the column phases of this row block through LDS: sh_red[cp][row]
        const int LDp = (LD + 3) & ~3;
        if (rb < nRB && cp < nCP && row < LD) {
#pragma unroll
            for (int r = 0; r < RPL; r++) sh_red[(size_t)cp * LDp + row + r] = part[r];
        }
    }
    __syncthreads();
    // sh_red now holds nCP partial copies of A_i y_i; fold them into copy 0
    {
        const int LDp = (LD + 3) & ~3;
        for (int r = tid; r < 2 * nv; r += BWD_THREADS) {
            T s = sh_red[r];
            for (int k = 1; k < nCP; k++) s += sh_red[(size_t)k * LDp + r];
            sh_red[r] = s;
        }
    }
    __syncthreads();
    T *sh_scr = sh_red + BWD_WAVES * ((LD + 3) & ~3);      // scratch for block_gemv_shared
    // g = Rinv s
    block_gemv_shared<T>(a.Rinv, nv, nv, sh_s, sh_g, sh_scr, BWD_THREADS);
    const T invp2 = (T)(-0.5) / a.tr.prob[node];
    for (int t = tid; t < nv; t += BWD_THREADS) a.v[(size_t)node * nv + t] = invp2 * sh_g[t] + sh_red[t];
    __syncthreads();
    // g = Bbt q_i
    block_gemv_shared<T>(a.Bbt, nv, nx, sh_q, sh_g, sh_scr, BWD_THREADS);
    for (int t = tid; t < nv; t += BWD_THREADS) a.rt[(size_t)node * nv + t] = sh_s[t] + sh_red[nv + t] + sh_g[t];
}

// multi-GPU: partial children sums of the cut parents, [parent][rt(nv) | q(nx)] (the vector that is all-reduced)
template <typename T>
__global__ void k_cut_partial_sums(SweepArgs<T> a, T *out) {
    const int parentStage = a.cutStage - 1;
    const int node = a.tr.stageCum[parentStage] + blockIdx.x;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    for (int t = threadIdx.x; t < a.nv + a.nx; t += blockDim.x) {
        T s = 0;
        if (t < a.nv) for (int c = 0; c < nc; c++) s += a.rt[(size_t)(c0 + c) * a.nv + t];
        else for (int c = 0; c < nc; c++) s += a.q[(size_t)(c0 + c) * a.nx + (t - a.nv)];
        out[(size_t)blockIdx.x * (a.nv + a.nx) + t] = s;
    }
}

// ------------------------------------------------------------------------------------------------------
// Forward sweep, one stage (SmpcController.cu:676-741 + solveChildNodesUpdate Utilities.cu:142-155) and the
// Hx products (:744-747, F_i and G_i are diagonal), one workgroup per node:
//   u_i = uhat_i + L v_i + (u_anc - uhat_anc)        root: (prevU - prevUhat)
//   x_i = x_anc + e_i + B u_i                        root: currentX
//   Hx_i = sqrt(p_i) [d_x o x_i ; d_xs o x_i ; d_u o u_i]
constexpr int FWD_THREADS = 128;
template <typename T>
__global__ void __launch_bounds__(FWD_THREADS) k_forward_stage(SweepArgs<T> a, int stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh_v = reinterpret_cast<T *>(smem_raw);     // nv
    T *sh_u = sh_v + ((a.nv + 3) & ~3);            // nu
    T *sh_o = sh_u + ((a.nu + 3) & ~3);            // max(nu, nx)
    T *sh_scr = sh_o + ((max(a.nu, a.nx) + 3) & ~3);
    const int tid = threadIdx.x;
    const int node = a.tr.stageCum[stage] + blockIdx.x;
    const int nx = a.nx, nu = a.nu, nv = a.nv, ny = a.ny;
    const int par = a.tr.parent[node];
    const T *dy = a.tr.dy + (size_t)stage * ny;
    const T sp = a.tr.sqrtp[node];
    for (int t = tid; t < nv; t += FWD_THREADS) sh_v[t] = a.v[(size_t)node * nv + t];
    __syncthreads();
    block_gemv_shared<T>(a.L, nu, nv, sh_v, sh_o, sh_scr, FWD_THREADS);
    for (int t = tid; t < nu; t += FWD_THREADS) {
        const T wanc = (par < 0) ? (a.prevU[t] - a.prevUhat[t]) : (a.u[(size_t)par * nu + t] - a.uhat[(size_t)par * nu + t]);
        const T uv = a.uhat[(size_t)node * nu + t] + wanc + sh_o[t];
        sh_u[t] = uv;
        a.u[(size_t)node * nu + t] = uv;
        a.hx[(size_t)node * ny + 2 * nx + t] = sp * dy[2 * nx + t] * uv;
    }
    __syncthreads();
    block_gemv_shared<T>(a.B, nx, nu, sh_u, sh_o, sh_scr, FWD_THREADS);
    for (int t = tid; t < nx; t += FWD_THREADS) {
        const T xanc = (par < 0) ? a.curX[t] : a.x[(size_t)par * nx + t];
        const T xv = xanc + a.e[(size_t)node * nx + t] + sh_o[t];
        a.x[(size_t)node * nx + t] = xv;
        a.hx[(size_t)node * ny + t] = sp * dy[t] * xv;
        a.hx[(size_t)node * ny + nx + t] = sp * dy[nx + t] * xv;
    }
}

// ------------------------------------------------------------------------------------------------------
// Fused dual update: prox (SmpcController.cu:759-835), fixed-point residual (:839-850), dual update
// (:859-864), primal-infeasibility arg-max (:1480-1496) of iteration t and the extrapolation (:535-557) of
// iteration t+1, in ONE pass:  reads hx, w, yprev, lo, hi; writes ynew, wnext (7 streams of n = nodes*ny
// elements) [+ z, res when MATERIALIZE].
//   t = hx + w/lambda ; z = clamp(t, lo, hi) [+ sc_half (t - clamp) when the soft-constraint branch trips]
//   res = hx - z ; ynew = w + lambda res ; wnext = (1 + ln) ynew - ln yprev
struct IterState {         // device-resident scalars of the APG loop
    int it;                // iteration counter (incremented by k_finalize)
    int tripped;           // soft-constraint branch taken in this iteration
    double scaleX, scaleS; // 1 - gamma/(lambda dist) for the two halves (0 when not tripped)
    double distX, distS;   // tree-global distances of this iteration
};
struct Partial {           // per-block partial reductions of the fused kernel
    double d2x, d2s;       // sum (t - clamp)^2 over the box / safety halves
    double absXi, valXi;   // max |res| over xi entries and the signed entry there
    double absPsi, valPsi;
    long long idxXi, idxPsi;
};

template <typename T>
struct DualArgs {
    const T *hx, *w, *yprev, *lo, *hi;
    T *ynew, *wnext, *z, *res;
    long long n;           // nodes * ny
    int nx, ny;
    T lambda, invLambda;
    const double *lamNext; // extrapolation parameter table indexed by iteration
    double thrX, thrS;     // gamma_x / lambda, gamma_s / lambda
    IterState *st;
    Partial *partials;     // [gridDim.x]
    int crownElems;        // multi-GPU: leading elements replicated on every rank (counted once, on rank 0)
    int countCrown;
};

__device__ __forceinline__ void better(double &a, double &v, long long &i, double a2, double v2, long long i2) {
    if (a2 > a || (a2 == a && i2 < i)) { a = a2; v = v2; i = i2; }
}

template <typename T, bool MATERIALIZE, bool FIXUP>
__global__ void __launch_bounds__(ELT_THREADS) k_dual_fused(DualArgs<T> a) {
    __shared__ Partial sh_p[ELT_THREADS / 64];
    T scX = 0, scS = 0;
    if (FIXUP) {
        if (!a.st->tripped) return;   // common case: nothing to redo
        scX = (T)a.st->scaleX; scS = (T)a.st->scaleS;
    }
    const T ln = (T)a.lamNext[a.st->it + 1];
    const int nx = a.nx, ny = a.ny;
    double d2x = 0, d2s = 0, absXi = -1, valXi = 0, absPsi = -1, valPsi = 0;
    long long idxXi = 0x7fffffffffffffffLL, idxPsi = 0x7fffffffffffffffLL;
    const long long stride = (long long)gridDim.x * ELT_THREADS;
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < a.n; i += stride) {
        const int c = (int)(i % ny);
        const T hx = a.hx[i], w = a.w[i], lo = a.lo[i], hi = a.hi[i], yp = a.yprev[i];
        const T t = hx + a.invLambda * w;
        T z = t < lo ? lo : (t > hi ? hi : t);
        const T diff = t - z;
        const bool counted = a.countCrown || i >= a.crownElems;
        if (c < nx) { if (counted) d2x += (double)diff * (double)diff; if (FIXUP) z += scX * diff; }
        else if (c < 2 * nx) { if (counted) d2s += (double)diff * (double)diff; if (FIXUP) z += scS * diff; }
        const T res = hx - z;
        const T yn = w + a.lambda * res;
        a.ynew[i] = yn;
        a.wnext[i] = ((T)1 + ln) * yn - ln * yp;
        if (MATERIALIZE) { a.z[i] = z; a.res[i] = res; }
        const double ar = fabs((double)res);
        if (c < 2 * nx) better(absXi, valXi, idxXi, ar, (double)res, i);
        else better(absPsi, valPsi, idxPsi, ar, (double)res, i);
    }
    // wave reduction (64 lanes), then across the block's waves
    for (int off = 32; off > 0; off >>= 1) {
        d2x += __shfl_down(d2x, off); d2s += __shfl_down(d2s, off);
        const double a2 = __shfl_down(absXi, off), v2 = __shfl_down(valXi, off);
        const long long i2 = __shfl_down(idxXi, off);
        better(absXi, valXi, idxXi, a2, v2, i2);
        const double a3 = __shfl_down(absPsi, off), v3 = __shfl_down(valPsi, off);
        const long long i3 = __shfl_down(idxPsi, off);
        better(absPsi, valPsi, idxPsi, a3, v3, i3);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh_p[wave] = Partial{d2x, d2s, absXi, valXi, absPsi, valPsi, idxXi, idxPsi};
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial p = sh_p[0];
        for (int k = 1; k < ELT_THREADS / 64; k++) {
            p.d2x += sh_p[k].d2x; p.d2s += sh_p[k].d2s;
            better(p.absXi, p.valXi, p.idxXi, sh_p[k].absXi, sh_p[k].valXi, sh_p[k].idxXi);
            better(p.absPsi, p.valPsi, p.idxPsi, sh_p[k].absPsi, sh_p[k].valPsi, sh_p[k].idxPsi);
        }
        a.partials[blockIdx.x] = p;
    }
}

// one workgroup: fold the block partials; decide whether the soft-constraint branch trips
// (dist > gamma/lambda, SmpcController.cu:793, :811)
__global__ void __launch_bounds__(ELT_THREADS) k_decide(const Partial *partials, int nblocks, IterState *st, double thrX,
                                                        double thrS) {
    __shared__ double sx[ELT_THREADS], ss[ELT_THREADS];
    double d2x = 0, d2s = 0;
    for (int b = threadIdx.x; b < nblocks; b += ELT_THREADS) { d2x += partials[b].d2x; d2s += partials[b].d2s; }
    sx[threadIdx.x] = d2x; ss[threadIdx.x] = d2s;
    __syncthreads();
    for (int off = ELT_THREADS / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) { sx[threadIdx.x] += sx[threadIdx.x + off]; ss[threadIdx.x] += ss[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double dX = sqrt(sx[0]), dS = sqrt(ss[0]);
        st->distX = dX; st->distS = dS;
        const bool tx = dX > thrX, ts = dS > thrS;
        st->tripped = (tx || ts) ? 1 : 0;
        st->scaleX = tx ? 1.0 - thrX / dX : 0.0;
        st->scaleS = ts ? 1.0 - thrS / dS : 0.0;
    }
}

// one workgroup: primal infeasibility of this iteration (max of the signed entries at the two arg-max |.|
// positions -- the reference's quirk) into hist[it]; advance the iteration counter.
__global__ void __launch_bounds__(ELT_THREADS) k_finalize(const Partial *partials, int nblocks, IterState *st, double *hist,
                                                          double *histParts, int histCap) {
    __shared__ Partial sh[ELT_THREADS / 64];
    double absXi = -1, valXi = 0, absPsi = -1, valPsi = 0;
    long long idxXi = 0x7fffffffffffffffLL, idxPsi = 0x7fffffffffffffffLL;
    for (int b = threadIdx.x; b < nblocks; b += ELT_THREADS) {
        better(absXi, valXi, idxXi, partials[b].absXi, partials[b].valXi, partials[b].idxXi);
        better(absPsi, valPsi, idxPsi, partials[b].absPsi, partials[b].valPsi, partials[b].idxPsi);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a2 = __shfl_down(absXi, off), v2 = __shfl_down(valXi, off);
        const long long i2 = __shfl_down(idxXi, off);
        better(absXi, valXi, idxXi, a2, v2, i2);
        const double a3 = __shfl_down(absPsi, off), v3 = __shfl_down(valPsi, off);
        const long long i3 = __shfl_down(idxPsi, off);
        better(absPsi, valPsi, idxPsi, a3, v3, i3);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = Partial{0, 0, absXi, valXi, absPsi, valPsi, idxXi, idxPsi};
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial p = sh[0];
        for (int k = 1; k < ELT_THREADS / 64; k++) {
            better(p.absXi, p.valXi, p.idxXi, sh[k].absXi, sh[k].valXi, sh[k].idxXi);
            better(p.absPsi, p.valPsi, p.idxPsi, sh[k].absPsi, sh[k].valPsi, sh[k].idxPsi);
        }
        const int it = st->it;
        if (it < histCap) {
            hist[it] = p.valXi > p.valPsi ? p.valXi : p.valPsi;
            histParts[4 * (size_t)it + 0] = p.absXi; histParts[4 * (size_t)it + 1] = p.valXi;
            histParts[4 * (size_t)it + 2] = p.absPsi; histParts[4 * (size_t)it + 3] = p.valPsi;
        }
        st->it = it + 1;
    }
}

// ------------------------------------------------------------------------------------------------------
// step-wise elementwise kernels (known-answer test API; same arithmetic as the fused kernel)
template <typename T>
__global__ void k_extrapolate(T *acc, T *xi, const T *upd, T lambda, long long n) {   // SmpcController.cu:535-557
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const T y1 = upd[i];
        acc[i] = ((T)1 + lambda) * y1 - lambda * xi[i];
        xi[i] = y1;
    }
}
// prox phase 1: z = clamp(hx + w/lambda); per-block dist^2 partials          (SmpcController.cu:778-792, :810)
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_prox_clamp(DualArgs<T> a) {
    __shared__ double sx[ELT_THREADS / 64], ss[ELT_THREADS / 64];
    double d2x = 0, d2s = 0;
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < a.n; i += (long long)gridDim.x * ELT_THREADS) {
        const int c = (int)(i % a.ny);
        const T t = a.hx[i] + a.invLambda * a.w[i];
        const T lo = a.lo[i], hi = a.hi[i];
        const T z = t < lo ? lo : (t > hi ? hi : t);
        a.z[i] = z;
        const double diff = (double)(t - z);
        if (c < a.nx) d2x += diff * diff; else if (c < 2 * a.nx) d2s += diff * diff;
    }
    for (int off = 32; off > 0; off >>= 1) { d2x += __shfl_down(d2x, off); d2s += __shfl_down(d2s, off); }
    if ((threadIdx.x & 63) == 0) { sx[threadIdx.x >> 6] = d2x; ss[threadIdx.x >> 6] = d2s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial p{};
        for (int k = 0; k < ELT_THREADS / 64; k++) { p.d2x += sx[k]; p.d2s += ss[k]; }
        a.partials[blockIdx.x] = p;
    }
}
// prox phase 2 (only when tripped): z += sc (t - z) on the tripped halves     (SmpcController.cu:793-797, :811-815)
template <typename T>
__global__ void k_prox_soft(DualArgs<T> a) {
    if (!a.st->tripped) return;
    const T scX = (T)a.st->scaleX, scS = (T)a.st->scaleS;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.ny);
        if (c >= 2 * a.nx) continue;
        const T t = a.hx[i] + a.invLambda * a.w[i];
        const T z = a.z[i];
        a.z[i] = z + (c < a.nx ? scX : scS) * (t - z);
    }
}
template <typename T>
__global__ void k_axpby(T *out, const T *x, const T *y, T alpha, T beta, long long n) {   // out = alpha x + beta y
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = alpha * x[i] + beta * y[i];
}
// arg-max |res| partials for rn_update_primal_infeasibility (SmpcController.cu:1480-1496)
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_absmax(const T *res, long long n, int nx, int ny, Partial *partials) {
    __shared__ Partial sh_p[ELT_THREADS / 64];
    double absXi = -1, valXi = 0, absPsi = -1, valPsi = 0;
    long long idxXi = 0x7fffffffffffffffLL, idxPsi = 0x7fffffffffffffffLL;
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * ELT_THREADS) {
        const int c = (int)(i % ny);
        const double r = (double)res[i];
        if (c < 2 * nx) better(absXi, valXi, idxXi, fabs(r), r, i); else better(absPsi, valPsi, idxPsi, fabs(r), r, i);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double a2 = __shfl_down(absXi, off), v2 = __shfl_down(valXi, off);
        const long long i2 = __shfl_down(idxXi, off);
        better(absXi, valXi, idxXi, a2, v2, i2);
        const double a3 = __shfl_down(absPsi, off), v3 = __shfl_down(valPsi, off);
        const long long i3 = __shfl_down(idxPsi, off);
        better(absPsi, valPsi, idxPsi, a3, v3, i3);
    }
    if ((threadIdx.x & 63) == 0) sh_p[threadIdx.x >> 6] = Partial{0, 0, absXi, valXi, absPsi, valPsi, idxXi, idxPsi};
    __syncthreads();
    if (threadIdx.x == 0) {
        Partial p = sh_p[0];
        for (int k = 1; k < ELT_THREADS / 64; k++) {
            better(p.absXi, p.valXi, p.idxXi, sh_p[k].absXi, sh_p[k].valXi, sh_p[k].idxXi);
            better(p.absPsi, p.valPsi, p.idxPsi, sh_p[k].absPsi, sh_p[k].valPsi, sh_p[k].idxPsi);
        }
        partials[blockIdx.x] = p;
    }
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
    const T *T1, *T2, *Bbt, *Lt;
    T *A;
    // scaled bounds in y order
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
        T *col = a.A + ((size_t)node * a.ny + c) * a.LD;
        for (int r = threadIdx.x; r < a.LD; r += blockDim.x)
            col[r] = r < a.nv ? s1 * m1[r] : (r < 2 * a.nv ? s2 * m2[r - a.nv] : (T)0);
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
    T *sh_o = sh_a + ((a.nu + 3) & ~3);                        // nv
    T *sh_o2 = sh_o + ((a.nv + 3) & ~3);                       // nv
    T *sh_scr = sh_o2 + ((a.nv + 3) & ~3);
    const int node = blockIdx.x, tid = threadIdx.x, nu = a.nu;
    const int par = a.tr.parent[node];
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const T p = a.tr.prob[node];
    for (int t = tid; t < nu; t += AFF_THREADS) {
        const T ui = a.uhat[(size_t)node * nu + t];
        const T ua = par < 0 ? a.prevUhat[t] : a.uhat[(size_t)par * nu + t];
        T z = p * (ui - ua);
        for (int c = 0; c < nc; c++) z -= a.tr.prob[c0 + c] * (a.uhat[(size_t)(c0 + c) * nu + t] - ui);
        sh_z[t] = z;
        sh_a[t] = a.alpha[(size_t)node * nu + t];
    }
    __syncthreads();
    block_gemv_shared<T>(a.WLt, a.nv, nu, sh_z, sh_o, sh_scr, AFF_THREADS);
    block_gemv_shared<T>(a.Lt, a.nv, nu, sh_a, sh_o2, sh_scr, AFF_THREADS);
    for (int t = tid; t < a.nv; t += AFF_THREADS) a.beta[(size_t)node * a.nv + t] = (T)2 * sh_o[t] + p * sh_o2[t];
}

// small utilities ------------------------------------------------------------------------------------
template <typename T>
__global__ void k_gemv_small(const T *M, int rows, int cols, const T *x, T *y) {   // y = M x, one block
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        T s = 0;
        for (int j = 0; j < cols; j++) s += M[r + (size_t)j * rows] * x[j];
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

}  // namespace rn
