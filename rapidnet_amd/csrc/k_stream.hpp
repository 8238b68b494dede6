// k_stream.hpp -- the dominant kernel: all per-node mat-vecs of the backward sweep in one streaming launch (k_stream_gemv), and the structured mode's elementwise part
// (part of the kernel sources of librapidnet_hip.so; kernels.hpp includes every family header, the translation units k_*.hip instantiate them)
#pragma once
#include "common.hpp"

namespace rn {

// ------------------------------------------------------------------------------------------------------
// The dominant kernel.  Batched per-node mat-vec  [m1_i; m2_i] = A_i y_i  for ALL nodes of the tree in one
// launch (SmpcController.cu:617-638 issues these as 4 cublasSgemmBatched per stage inside the sequential sweep;
// they do not depend on the recursion, only the vector sums do -- see k_up_*).  One workgroup per node.
//
// Access shape (decided by measurement, tools/probe_hbm.py): a do-nothing reader with one workgroup per 376 KB node
// block reaches 6.7-6.8 TB/s on MI355X when every wave-load is 64 lanes x 16 B = 1 KB CONTIGUOUS, and only 5.9 TB/s
// when a lane fetches 32 adjacent bytes as two loads (each wave-load then touches 16 cache lines and uses half of
// each) -- which is what "4 fp64 rows per lane" amounts to.  So the block is walked in 16-byte SLOTS: a column of A_i
// (LD values) is SPC = LD*sizeof(T)/16 slots, a SPAN is G consecutive columns, and thread t owns slots t, t+512, ...
// (NL of them) of every span: consecutive threads read consecutive 16 B, across column boundaries, and each thread always
// meets the same rows, so its partial sums stay in registers.  G is chosen on the host (Ctx::stream_shape) so that a
// span is a whole number of 128-byte lines: with spans that end inside a line (first version: G = 5, 490 of 512 slots
// busy) the two wave-loads sharing that line are issued a group apart and the non-temporal stream fetches it twice --
// FETCH_SIZE showed 4.35 GB per launch against 4.09 GB algorithmic; with G = 8 (784 slots = 98 lines, NL = 2, 77 % of
// the lanes busy) it is 4.14 GB and the kernel 6 % faster.  Loads are non-temporal (A is read once per iteration and is
// far larger than the 256 MiB Infinity Cache) and double-buffered in groups of D spans (one 8-wave workgroup per CU
// with 10 x 16 B per lane in flight measured best: 512 threads, D = 5).  Also emits a_i = F_i' xi_i (F_i is diagonal:
// Utilities.cu:33-58).
// HBM bytes per node: LD*ny*sizeof(T) + (ny + 2nv + nx)*sizeof(T).
// SPLIT = false is the kernel as it always was (span0 = 0, every block one workgroup): the instantiation the unsplit launches run --
// the whole 493-scenario tree among them, where the split gains nothing and the second code path would cost (same-box A/B against
// the round-3 kernel: 605 instead of 590 us with one shared instantiation, whose register allocation let two workgroups share a CU)
// NR = 2: TWO right-hand sides in one pass over the blocks (the quasi-Newton loops' pairs of independent Hessian sweeps: the launch is
// bound by the blocks' bytes, so the second product rides for free): the second vector r2.w, its results in r2.my / r2.qa.  Each
// right-hand side's sums are formed exactly as the one-vector kernel forms them (same order over the columns): bitwise the results
// of two launches.  Unsplit launches with w in memory only.
template <typename T>
struct StreamRhs2 { const T *w; T *my; T *qa; };
template <typename T, int NL, bool SPLIT, int NR = 1>
__global__ void __launch_bounds__(STREAM_THREADS, RN_STREAM_MINW) k_stream_gemv(SweepArgs<T> a, int G, int node0, StreamSplit<T> sp, StreamRhs2<T> r2) {
    static_assert(NR == 1 || !SPLIT, "two right-hand sides: unsplit launches only");
    typedef typename Slot<T>::type VT;
    // (two right-hand sides in fp32: 4 values per slot x 2 accumulator sets -- a group one span shorter (NL = 4: one span per group)
    //  keeps the kernel inside its registers: 150-172 instead of 256 + 20-28 bytes of scratch per lane with the full depth.  The
    //  order in which a thread meets its columns does not depend on the depth: same sums, bit for bit)
    constexpr int VPL = Slot<T>::N, D0 = NL <= 2 ? RN_STREAM_D : RN_STREAM_D_WIDE, D = (NR == 2 && sizeof(T) == 4) ? (NL == 4 ? 1 : D0 - 1) : D0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh_y = reinterpret_cast<T *>(smem_raw);          // ny (+ G of zero padding is not needed: guarded reads)
    T *sh_red = sh_y + ((a.ny + 3) & ~3);               // G * LD
    T *sh_y2 = sh_red + (size_t)G * a.LD;               // NR = 2: the same pair again for the second right-hand side
    T *sh_red2 = sh_y2 + ((a.ny + 3) & ~3);
    const int tid = threadIdx.x;
    // blocks [0, first): one workgroup each; from there on two workgroups per block (first half, second half)
    const bool split = SPLIT && (int)blockIdx.x >= sp.first;
    const int node = split ? sp.first + (((int)blockIdx.x - sp.first) >> 1) : (int)blockIdx.x;
    const bool second = split && ((((int)blockIdx.x - sp.first) & 1) != 0);
    const int nx = a.nx, nv = a.nv, ny = a.ny, LD = a.LD;
    const int SPC = LD / VPL, spanSlots = G * SPC;
    const long long blockSlots = (long long)ny * SPC;
    const VT *__restrict__ Ab = reinterpret_cast<const VT *>(a.A + (size_t)node * a.strideA);
    // this workgroup's spans: [span0, span1) of the block's ceil(ny / G)
    const int spansAll = (ny + G - 1) / G;
    const int span0 = SPLIT ? (second ? sp.spanHalf : 0) : 0, span1 = (split && !second) ? sp.spanHalf : spansAll;
    int off[NL], cj[NL];
    T msk[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int q = tid + STREAM_THREADS * j;
        const bool ok = q < spanSlots;
        off[j] = ok ? q : spanSlots - 1;          // idle slots re-read the last slot of the span and multiply by zero
        cj[j] = off[j] / SPC;
        msk[j] = ok ? (T)1 : (T)0;
    }
    T part[NL][VPL], part2[NR == 2 ? NL : 1][VPL];
#pragma unroll
    for (int j = 0; j < NL; j++)
#pragma unroll
        for (int e = 0; e < VPL; e++) { part[j][e] = 0; if (NR == 2) part2[j][e] = 0; }
    const int nFull = (ny / G < span1 ? ny / G : span1) - span0;   // spans of this workgroup made of G whole columns
    const int nGroups = nFull / D;
    VT bufA[D][NL], bufB[D][NL];
#define RN_LOADG(buf, g_)                                                                                              \
    _Pragma("unroll") for (int d = 0; d < D; d++)                                                                      \
        _Pragma("unroll") for (int j = 0; j < NL; j++)                                                                 \
            buf[d][j] = __builtin_nontemporal_load(Ab + (size_t)(span0 + (g_) * D + d) * spanSlots + off[j]);
#define RN_USEG(buf, g_)                                                                                               \
    _Pragma("unroll") for (int d = 0; d < D; d++)                                                                      \
        _Pragma("unroll") for (int j = 0; j < NL; j++) {                                                               \
            const T yc = sh_y[(span0 + (g_) * D + d) * G + cj[j]] * msk[j];                                            \
            _Pragma("unroll") for (int e = 0; e < VPL; e++) part[j][e] += buf[d][j][e] * yc;                           \
            if (NR == 2) {                                                                                             \
                const T yc2 = sh_y2[(span0 + (g_) * D + d) * G + cj[j]] * msk[j];                                      \
                _Pragma("unroll") for (int e = 0; e < VPL; e++) part2[j][e] += buf[d][j][e] * yc2;                     \
            }                                                                                                          \
        }
    // Prologue.  A wave's loads return in order, so everything the prologue needs is requested FIRST and the first group of
    // A_i (which does not depend on y) right behind it: the prologue's arithmetic then runs while that group streams in, and
    // nothing after the barrier has to queue a small load behind the stream.  The first group is requested unconditionally
    // (clamped into the block when the block is shorter than a group): a branch around it makes the compiler's wait-count
    // bookkeeping merge two paths and fall back to "wait for everything" in front of the prologue's arithmetic.
    const bool has0 = tid < ny;
    const size_t i0 = (size_t)node * ny + (has0 ? tid : 0);
    // stage of the node by arithmetic in the chain region (every stage >= chainStage has K nodes: no table load in front of
    // the preconditioner row's address); a table load for the few crown nodes before it
    const int stage = node >= node0 ? a.chainStage + (node - node0) / a.K : a.tr.stageOf[node];
    const T *dyRow = a.tr.dy + (size_t)stage * ny;
    const T spn = a.tr.sqrtp[node];
    const int tq = tid < nx ? tid : 0;
    const T dq0 = dyRow[tq], dq1 = dyRow[nx + tq];    // for a_i below
    const T w0a = a.w[i0];      // the y column: the accelerated dual the sweep is evaluated at
    T w20 = 0;
    if (NR == 2) w20 = r2.w[i0];
    asm volatile("" ::: "memory");   // keep the request order: the compiler otherwise hoists the group's loads above the small ones
    {
        const int lastSlot = (int)blockSlots - 1;
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const int sl = (span0 + d) * spanSlots + off[j];
                bufA[d][j] = __builtin_nontemporal_load(Ab + (sl < lastSlot ? sl : lastSlot));
            }
    }
    asm volatile("" ::: "memory");
    // a_i is STORED AT THE END of the kernel: stores count in the same in-order counter as the loads, so a store issued
    // here has to be acknowledged before the wave may consume any group of A_i requested after it
    T qa0 = 0, qa02 = 0;
    if (has0) sh_y[tid] = w0a;
    for (int c = tid + STREAM_THREADS; c < ny; c += STREAM_THREADS) sh_y[c] = a.w[(size_t)node * ny + c];
    if (NR == 2) {
        if (has0) sh_y2[tid] = w20;
        for (int c = tid + STREAM_THREADS; c < ny; c += STREAM_THREADS) sh_y2[c] = r2.w[(size_t)node * ny + c];
    }
    __syncthreads();
    // a_i = F_i' xi_i = sqrt(p_i) (d_x o xi_box + d_xs o xi_safe)      (a split block: written by its first half)
    if (tid < nx) qa0 = stream_qa_elem(spn, dq0, sh_y[tid], dq1, sh_y[nx + tid]);
    if (!second) for (int t = tid + STREAM_THREADS; t < nx; t += STREAM_THREADS)
        a.qa[(size_t)node * nx + t] = stream_qa_elem(spn, dyRow[t], sh_y[t], dyRow[nx + t], sh_y[nx + t]);
    if (NR == 2) {
        if (tid < nx) qa02 = stream_qa_elem(spn, dq0, sh_y2[tid], dq1, sh_y2[nx + tid]);
        for (int t = tid + STREAM_THREADS; t < nx; t += STREAM_THREADS)
            r2.qa[(size_t)node * nx + t] = stream_qa_elem(spn, dyRow[t], sh_y2[t], dyRow[nx + t], sh_y2[nx + t]);
    }
    if (nGroups > 0) {
        int g = 0;
        // steady state has no branch inside, so the compiler's vmcnt waits are exact: while group g is consumed, group
        // g+1 (and then g+2) is in flight
        for (; g + 2 < nGroups; g += 2) {
            RN_LOADG(bufB, g + 1)
            RN_USEG(bufA, g)
            RN_LOADG(bufA, g + 2)
            RN_USEG(bufB, g + 1)
        }
        if (g + 1 < nGroups) {
            RN_LOADG(bufB, g + 1)
            RN_USEG(bufA, g)
            RN_USEG(bufB, g + 1)
        } else {
            RN_USEG(bufA, g)
        }
    }
#undef RN_LOADG
#undef RN_USEG
    // remaining whole spans and the last, partial one (ny % G columns): guarded
    for (int s = span0 + nGroups * D; s < span1; s++) {
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int c = s * G + cj[j];
            const long long slot = (long long)s * spanSlots + off[j];
            const bool live = msk[j] != (T)0 && c < ny && slot < blockSlots;
            const VT v = __builtin_nontemporal_load(Ab + (live ? slot : 0));
            const T yc = live ? sh_y[c] : (T)0;
#pragma unroll
            for (int e = 0; e < VPL; e++) part[j][e] += v[e] * yc;
            if (NR == 2) {
                const T yc2 = live ? sh_y2[c] : (T)0;
#pragma unroll
                for (int e = 0; e < VPL; e++) part2[j][e] += v[e] * yc2;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NL; j++)
        if (msk[j] != (T)0) {
#pragma unroll
            for (int e = 0; e < VPL; e++) { sh_red[(size_t)off[j] * VPL + e] = part[j][e]; if (NR == 2) sh_red2[(size_t)off[j] * VPL + e] = part2[j][e]; }
        }
    if (tid < nx && !second) stream_out(qa0, a.qa + (size_t)node * nx + tid);
    if (NR == 2 && tid < nx) stream_out(qa02, r2.qa + (size_t)node * nx + tid);
    __syncthreads();
    T *const myOut = second ? sp.my2 + (size_t)(node - sp.first) * 2 * nv : a.my + (size_t)node * 2 * nv;
    if (NR == 2) {      // both right-hand sides' partials folded in one walk (each in the order of the one-vector kernel): the LDS round trips overlap
        T *const myOut2 = r2.my + (size_t)node * 2 * nv;
        for (int r = tid; r < 2 * nv; r += STREAM_THREADS) {
            T s = sh_red[r], s2 = sh_red2[r];
            for (int k = 1; k < G; k++) { s += sh_red[(size_t)k * LD + r]; s2 += sh_red2[(size_t)k * LD + r]; }
            stream_out(s, myOut + r);
            stream_out(s2, myOut2 + r);
        }
        return;
    }
    for (int r = tid; r < 2 * nv; r += STREAM_THREADS) {    // slot q of a span = column q / SPC, rows (q % SPC) * VPL ...
        T s = sh_red[r];
        for (int k = 1; k < G; k++) s += sh_red[(size_t)k * LD + r];
        stream_out(s, myOut + r);
    }
}

// Structured operator mode ("shared-operator model"): every per-node block of the factor step
// is (shared matrix) x (stage diagonal) x (power of p_i)  --  D_i = Bbt F_i', Ftil_i = L' G_i', Phi_i = -Omega_i D_i / 2,
// Psi_i = -Omega_i Ftil_i / 2 (Engine.cu:721-745) with F_i, G_i diagonal (Utilities.cu:33-58).  Hence
//   m2_i = D_i xi_i + Ftil_i psi_i = [Bbt | L'] [a_i; b_i],   a_i = F_i' xi_i,  b_i = G_i' psi_i     (elementwise + one GEMM)
//   m1_i = -Rinv m2_i / (2 p_i)   is folded into  v_i = -(Rinv rho_i + Rinv Bbt kappa_i) / (2 p_i)
// and no per-node block is ever stored or read.  This kernel is the elementwise part.
template <typename T>
__global__ void k_struct_prep(SweepArgs<T> a) {
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nx + nu;
    const long long n = (long long)a.nodes * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int node = (int)(i / w), t = (int)(i % w);
        const T *dy = a.tr.dy + (size_t)a.tr.stageOf[node] * ny;
        const size_t y = (size_t)node * ny;
        const T sp = a.tr.sqrtp[node];
        T val;
        if (t < nx) { val = sp * (dy[t] * a.w[y + t] + dy[nx + t] * a.w[y + nx + t]); a.qa[(size_t)node * nx + t] = val; }
        else { const int j = t - nx; val = sp * dy[2 * nx + j] * a.w[y + 2 * nx + j]; }
        a.ab[i] = val;
    }
}


}  // namespace rn
