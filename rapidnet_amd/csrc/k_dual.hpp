// k_dual.hpp -- the fused dual update (k_dual_fused, k_dual_stage), its bookkeeping kernels, the step-wise prox / residual / extrapolation kernels
// (part of the kernel sources of librapidnet_hip.so; kernels.hpp includes every family header, the translation units k_*.hip instantiate them)
#pragma once
#include "common.hpp"

namespace rn {

// ------------------------------------------------------------------------------------------------------
// Fused dual update: prox (SmpcController.cu:759-835), fixed-point residual (:839-850), dual update
// (:859-864), primal-infeasibility arg-max (:1480-1496) of iteration t and the extrapolation (:535-557) of
// iteration t+1, in ONE pass:  reads hx, w, yprev, lo, hi; writes ynew, wnext (7 streams of n = nodes*ny
// elements) [+ z, res when MATERIALIZE].
//   t = hx + w/lambda ; z = clamp(t, lo, hi) [+ sc_half (t - clamp) when the soft-constraint branch trips]
//   res = hx - z ; ynew = w + lambda res ; wnext = (1 + ln) ynew - ln yprev
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
    int finalizedEarly;    // 1: k_decide_finalize already wrote hist[it] and advanced it; the fix-up must redo hist[it-1]
    double *hist, *histParts; int histCap;
    // regen != 0: the scaled bounds are not read (2 of the 7 streams) but rebuilt as in k_expand_operators,
    // lo = (sqrt(p_i) d_c) blo_c, hi = (sqrt(p_i) d_c) bhi_c (safety half: bhi_c), from tables that live in L1/L2
    int regen;
    const int *stageOf; const T *sqrtp, *dy, *blo, *bhi;
    // decideHere (fix-up launch only): the trip decision and the bookkeeping of the iteration are done by THIS launch
    // instead of a k_decide_finalize launch of their own: every workgroup folds the main pass's dist^2 partials itself
    // (same order everywhere => same decision), workgroup 0 also folds the arg-max, writes the history entry and advances
    // the iteration counter.  itHost = iteration index (the host's count; st->it is not read), mainPartials / nMain = the
    // partials of the main pass; this launch writes its own partials to `partials`.
    int decideHere, itHost, nMain;
    const Partial *mainPartials;
};



// Fold of per-workgroup partials by ONE workgroup of ELT_THREADS threads (every thread calls; the result is valid in thread
// 0): tx2 / ts2 = sums of the dist^2 partials, `out` = arg-max pairs (only when wantArgmax).  Fixed association, so every
// caller that folds the same partials gets the same bits (the fix-up launch relies on that: all its workgroups take the same
// trip decision).
__device__ __forceinline__ void fold_partials(const Partial *partials, int nblocks, bool wantArgmax, double &tx2, double &ts2, Partial &out) {
    __shared__ double f_sx[ELT_THREADS / 64], f_ss[ELT_THREADS / 64];
    __shared__ Partial f_sh[ELT_THREADS / 64];
    double d2x = 0, d2s = 0, absXi = -1, valXi = 0, absPsi = -1, valPsi = 0;
    long long idxXi = 0x7fffffffffffffffLL, idxPsi = 0x7fffffffffffffffLL;
    // four partials requested per round trip (one per node when the streaming kernel hosts the update: 10 864 of them on the
    // 493-scenario tree, 42 per thread -- one at a time that was ~10 us of dependent L2 latency); same fold order as a plain loop
    constexpr int FU = 4;
    for (int b0 = threadIdx.x; b0 < nblocks; b0 += FU * ELT_THREADS) {
        Partial q[FU];
#pragma unroll
        for (int u = 0; u < FU; u++) { const int b = b0 + u * ELT_THREADS; q[u] = partials[b < nblocks ? b : b0]; }
#pragma unroll
        for (int u = 0; u < FU; u++) {
            if (b0 + u * ELT_THREADS >= nblocks) break;
            d2x += q[u].d2x; d2s += q[u].d2s;
            if (wantArgmax) { better(absXi, valXi, idxXi, q[u].absXi, q[u].valXi, q[u].idxXi); better(absPsi, valPsi, idxPsi, q[u].absPsi, q[u].valPsi, q[u].idxPsi); }
        }
    }
    d2x = wave_sum_f64(d2x); d2s = wave_sum_f64(d2s);
    if (wantArgmax) { wave_argmax(absXi, valXi, idxXi); wave_argmax(absPsi, valPsi, idxPsi); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { f_sx[wave] = d2x; f_ss[wave] = d2s; f_sh[wave] = Partial{0, 0, absXi, valXi, absPsi, valPsi, idxXi, idxPsi}; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out = f_sh[0];
        tx2 = f_sx[0]; ts2 = f_ss[0];
        for (int k = 1; k < ELT_THREADS / 64; k++) {
            tx2 += f_sx[k]; ts2 += f_ss[k];
            better(out.absXi, out.valXi, out.idxXi, f_sh[k].absXi, f_sh[k].valXi, f_sh[k].idxXi);
            better(out.absPsi, out.valPsi, out.idxPsi, f_sh[k].absPsi, f_sh[k].valPsi, f_sh[k].idxPsi);
        }
    }
    __syncthreads();
}
// primal infeasibility of one iteration from the folded arg-max pairs: the larger of the two SIGNED entries (the reference's
// updatePrimalInfeasibity quirk, SmpcController.cu:1480-1496) into the history, with the four parts kept for sharded callers
__device__ __forceinline__ void write_history(const Partial &p, int it, double *hist, double *histParts, int histCap) {
    if (it < 0 || it >= histCap) return;
    hist[it] = p.valXi > p.valPsi ? p.valXi : p.valPsi;
    histParts[4 * (size_t)it + 0] = p.absXi; histParts[4 * (size_t)it + 1] = p.valXi;
    histParts[4 * (size_t)it + 2] = p.absPsi; histParts[4 * (size_t)it + 3] = p.valPsi;
}


#ifndef RN_DUAL_U
#define RN_DUAL_U 1
#endif
#ifndef RN_DUAL_NT
#define RN_DUAL_NT 0
#endif
constexpr int DUAL_U = RN_DUAL_U;
template <typename T, bool MATERIALIZE, bool FIXUP>
__global__ void __launch_bounds__(ELT_THREADS) k_dual_fused(DualArgs<T> a) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    __shared__ Partial sh_p[ELT_THREADS / 64];
    T scX = 0, scS = 0;
    if (FIXUP && a.decideHere) {
        __shared__ double dec[3];   // tripped, scaleX, scaleS
        double tx2 = 0, ts2 = 0;
        Partial p;
        fold_partials(a.mainPartials, a.nMain, blockIdx.x == 0, tx2, ts2, p);
        if (threadIdx.x == 0) {
            const double dX = sqrt(tx2), dS = sqrt(ts2);
            const bool trX = dX > a.thrX, trS = dS > a.thrS;
            dec[0] = (trX || trS) ? 1.0 : 0.0;
            dec[1] = trX ? 1.0 - a.thrX / dX : 0.0;
            dec[2] = trS ? 1.0 - a.thrS / dS : 0.0;
            if (blockIdx.x == 0) {
                a.st->distX = dX; a.st->distS = dS;
                a.st->tripped = (trX || trS) ? 1 : 0;
                a.st->scaleX = dec[1]; a.st->scaleS = dec[2];
                write_history(p, a.itHost, a.hist, a.histParts, a.histCap);
                a.st->it = a.itHost + 1;
            }
        }
        __syncthreads();
        if (dec[0] == 0.0) return;   // common case: nothing to redo
        scX = (T)dec[1]; scS = (T)dec[2];
    } else if (FIXUP) {
        if (!a.st->tripped) return;   // common case: nothing to redo
        scX = (T)a.st->scaleX; scS = (T)a.st->scaleS;
    }
    // the fix-up runs after the iteration counter has been advanced (k_decide_finalize / the decision block above)
    const int itIdx = a.decideHere ? a.itHost : ((FIXUP && a.finalizedEarly) ? a.st->it - 1 : a.st->it);
    const T ln = (T)a.lamNext[itIdx + 1];
    const T lambda = a.lambda, invLambda = a.invLambda;
    const int nx = a.nx, ny = a.ny;
    double d2x = 0, d2s = 0, absXi = -1, valXi = 0, absPsi = -1, valPsi = 0;
    long long idxXi = 0x7fffffffffffffffLL, idxPsi = 0x7fffffffffffffffLL;
    const long long stride = (long long)gridDim.x * ELT_THREADS;
    const long long gid = (long long)blockIdx.x * ELT_THREADS + threadIdx.x;
    const long long nvec = a.n / VN;
    // column index of the first element of this thread's vector, advanced incrementally (no division in the loop)
    int c0 = (int)((gid * VN) % ny);
    const int cstep = (int)((stride * VN) % ny);
    // node of the first element of this thread's vector, advanced together with the column (regen only)
    int nd0 = (int)((gid * VN) / ny);
    const int nstep = (int)((stride * VN) / ny);
    // 16 bytes per lane per stream; DUAL_U grid-stride positions per trip with all their loads requested up front
    for (long long iv0 = gid; iv0 < nvec; iv0 += (long long)DUAL_U * stride) {
        VT hxv[DUAL_U], wv[DUAL_U], ypv[DUAL_U];
#pragma unroll
        for (int u = 0; u < DUAL_U; u++) {
            const long long ivu = iv0 + u * stride;
            const long long ix = ivu < nvec ? ivu : iv0;
#if RN_DUAL_NT
            hxv[u] = __builtin_nontemporal_load(reinterpret_cast<const VT *>(a.hx) + ix);
            wv[u] = __builtin_nontemporal_load(reinterpret_cast<const VT *>(a.w) + ix);
            ypv[u] = __builtin_nontemporal_load(reinterpret_cast<const VT *>(a.yprev) + ix);
#else
            hxv[u] = reinterpret_cast<const VT *>(a.hx)[ix]; wv[u] = reinterpret_cast<const VT *>(a.w)[ix];
            ypv[u] = reinterpret_cast<const VT *>(a.yprev)[ix];
#endif
        }
#pragma unroll
        for (int u = 0; u < DUAL_U; u++) {
            const long long iv = iv0 + u * stride;
            if (iv < nvec) {
                const VT hx = hxv[u], w = wv[u], yp = ypv[u];
                VT lo, hi;
                if (a.regen) {
                    int cc = c0, nn = nd0;
#pragma unroll
                    for (int e = 0; e < VN; e++) {
                        const T k = a.sqrtp[nn] * a.dy[(size_t)a.stageOf[nn] * ny + cc];
                        lo[e] = k * a.blo[cc];
                        hi[e] = (cc >= nx && cc < 2 * nx) ? a.bhi[cc] : k * a.bhi[cc];
                        if (++cc == ny) { cc = 0; nn++; }
                    }
                } else { lo = reinterpret_cast<const VT *>(a.lo)[iv]; hi = reinterpret_cast<const VT *>(a.hi)[iv]; }
                VT yn, wn, z, res;
                int c = c0;
#pragma unroll
                for (int e = 0; e < VN; e++) {
                    const bool isBox = c < nx, isXi = c < 2 * nx;
                    const T sc = FIXUP ? (isBox ? scX : (isXi ? scS : (T)0)) : (T)0;
                    const DualOut<T> o = dual_elem<T, FIXUP>(hx[e], w[e], lo[e], hi[e], yp[e], lambda, invLambda, ln, sc);
                    yn[e] = o.yn; wn[e] = o.wn; z[e] = o.z; res[e] = o.res;
                    const long long i = iv * VN + e;
                    const double dd = (a.countCrown || i >= a.crownElems) ? (double)o.diff * (double)o.diff : 0.0;
                    d2x += isBox ? dd : 0.0;
                    d2s += (isXi && !isBox) ? dd : 0.0;
                    const double ar = fabs((double)o.res);
                    if (isXi) { if (ar > absXi) { absXi = ar; valXi = (double)o.res; idxXi = i; } }
                    else { if (ar > absPsi) { absPsi = ar; valPsi = (double)o.res; idxPsi = i; } }
                    if (++c == ny) c = 0;
                }
                // plain (cached) stores: non-temporal ones make this kernel no faster and the next kernel, which re-reads w, slower
                reinterpret_cast<VT *>(a.ynew)[iv] = yn;
                reinterpret_cast<VT *>(a.wnext)[iv] = wn;
                if (MATERIALIZE) { reinterpret_cast<VT *>(a.z)[iv] = z; reinterpret_cast<VT *>(a.res)[iv] = res; }
            }
            c0 += cstep; nd0 += nstep;
            if (c0 >= ny) { c0 -= ny; nd0++; }
        }
    }
    for (long long i = nvec * VN + gid; i < a.n; i += stride) {   // at most VN-1 tail elements
        const int c = (int)(i % ny);
        const bool isBox = c < nx, isXi = c < 2 * nx;
        const T sc = FIXUP ? (isBox ? scX : (isXi ? scS : (T)0)) : (T)0;
        const DualOut<T> o = dual_elem<T, FIXUP>(a.hx[i], a.w[i], a.lo[i], a.hi[i], a.yprev[i], lambda, invLambda, ln, sc);
        a.ynew[i] = o.yn; a.wnext[i] = o.wn;
        if (MATERIALIZE) { a.z[i] = o.z; a.res[i] = o.res; }
        const double dd = (a.countCrown || i >= a.crownElems) ? (double)o.diff * (double)o.diff : 0.0;
        d2x += isBox ? dd : 0.0;
        d2s += (isXi && !isBox) ? dd : 0.0;
        const double ar = fabs((double)o.res);
        if (isXi) { if (ar > absXi) { absXi = ar; valXi = (double)o.res; idxXi = i; } }
        else { if (ar > absPsi) { absPsi = ar; valPsi = (double)o.res; idxPsi = i; } }
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
        if (FIXUP && a.finalizedEarly) {
            // rare path: the residual changed, so the history entry written by k_decide_finalize must be redone by
            // the last block to arrive (release -> ticket -> acquire, Guideline 16)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int t = atomicAdd(&a.st->ticket, 1u);
            if (t == gridDim.x - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                double aX = -1, vX = 0, aP = -1, vP = 0;
                long long iX = 0x7fffffffffffffffLL, iP = 0x7fffffffffffffffLL;
                for (unsigned int b = 0; b < gridDim.x; b++) {
                    const Partial q = a.partials[b];
                    better(aX, vX, iX, q.absXi, q.valXi, q.idxXi);
                    better(aP, vP, iP, q.absPsi, q.valPsi, q.idxPsi);
                }
                Partial pr{0, 0, aX, vX, aP, vP, iX, iP};
                write_history(pr, a.decideHere ? a.itHost : a.st->it - 1, a.hist, a.histParts, a.histCap);
                a.st->ticket = 0;
            }
        }
    }
}

// Main pass of the fused dual update, stage-tiled (the default whenever ny is a whole number of 16-byte vectors).
// k_dual_fused above walks the flat element range with a grid stride; to rebuild the scaled bounds it needs, per 16 bytes of
// payload, ten 8-byte table gathers, two of them (stageOf -> dy) a dependent L2 round trip behind the in-order vmcnt of the
// HBM loads: 23-25 us for 104 MB where a bare "3 reads + 2 writes" kernel takes 18 us on the same box
// (tools/probes/probe_stream.hip).  Here a workgroup owns a tile of consecutive 16-byte vectors INSIDE ONE STAGE, found by
// arithmetic on the block index (stages >= cs all have K nodes: the chain region of the tree; the few crown nodes in front
// of it are handled by the first `crownBlocks` workgroups, which look the stage up per vector).  The stage is then
// wave-uniform, the preconditioner row dy[stage] and the bounds are three 16-byte vector loads at addresses known up front,
// sqrt(p_i) one more, and all seven loads of a vector are requested together: one memory round trip, no dependent chain.
// Same arithmetic, element by element, as k_dual_fused (dual_elem); the partials (one per workgroup) are folded by the same
// bookkeeping code.
#ifndef RN_DUAL_ABL
#define RN_DUAL_ABL 0   // timing ablations of k_dual_stage (tools/sweep_variants.sh; results are WRONG when set): bit 0 = no table loads,
#endif                  // bit 1 = no reductions / partials, bit 2 = no stores, bit 3 = no store of the extrapolated dual w (4 streams instead of 5)
struct DualStageShape {
    int cs, K, node0;        // first regular stage, nodes per regular stage, first node of stage cs
    int bps, crownBlocks;    // workgroups per regular stage; leading workgroups that cover the nodes [0, node0)
    int vpn;                 // 16-byte vectors per node (ny / VN)
    unsigned int vpnMagic;   // floor(2^32 / vpn) + 1: j / vpn == umulhi(j, vpnMagic) for j < 2^32 / vpn
    int trips;               // vectors per thread; a tile is ELT_THREADS * trips vectors
    double lnNext;           // extrapolation parameter of the NEXT iteration, by value: no st->it -> lamNext[] load chain in front of
                             // the streams (every workgroup would pay those two dependent scalar round trips before its first load)
};
// one 16-byte vector of the tile with everything its update needs (all seven loads are independent)
template <typename T>
struct DualSlot {
    typename VecOf<T>::type hx, w, yp, blo, bhi, dy;
    T sp;
    int c;            // column of the vector's first element
    long long iv;     // global vector index
    bool on;
};
template <typename T>
struct DualAcc {      // per-thread running reductions; arg-max keeps the signed entry (|.| is recomputed in the compare) and
    double d2x = 0, d2s = 0, valXi = 0, valPsi = 0;   // the 32-bit element index of its first occurrence (strict >, ascending walk)
    unsigned int idxXi = 0xffffffffu, idxPsi = 0xffffffffu;
};
template <typename T>
__device__ __forceinline__ void dual_slot_load(DualSlot<T> &s, const DualArgs<T> &a, const DualStageShape &g, int trip, int jbase, int cnt,
                                               int nodeFirst, int stageU, bool crownBlock) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    const int off = trip * ELT_THREADS + (int)threadIdx.x;
    s.on = off < cnt;
    const unsigned int J = (unsigned int)(jbase + (s.on ? off : 0));
    const int q = (int)__umulhi(J, g.vpnMagic);
    const int node = nodeFirst + q;
    s.c = ((int)J - q * g.vpn) * VN;
    s.iv = (long long)nodeFirst * g.vpn + J;
    int stage = stageU;
    if (crownBlock) stage = a.stageOf[node];
    s.hx = reinterpret_cast<const VT *>(a.hx)[s.iv]; s.w = reinterpret_cast<const VT *>(a.w)[s.iv];
    s.yp = reinterpret_cast<const VT *>(a.yprev)[s.iv];
#if RN_DUAL_ABL & 1
    s.sp = (T)1; for (int e = 0; e < VN; e++) { s.dy[e] = (T)1; s.blo[e] = (T)-1; s.bhi[e] = (T)stage; }
#else
    s.sp = a.sqrtp[node];
    s.dy = *reinterpret_cast<const VT *>(a.dy + (size_t)stage * a.ny + s.c);
    s.blo = *reinterpret_cast<const VT *>(a.blo + s.c);
    s.bhi = *reinterpret_cast<const VT *>(a.bhi + s.c);
#endif
}
template <typename T, bool MATERIALIZE, bool SCALE = false>
__device__ __forceinline__ void dual_slot_use(const DualSlot<T> &s, const DualArgs<T> &a, T ln, DualAcc<T> &r) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    if (!s.on) return;
    VT yn, wn, z, res;
    const long long i0 = s.iv * VN;
    const bool counted = a.countCrown || i0 >= a.crownElems;
#pragma unroll
    for (int e = 0; e < VN; e++) {
        const int c = s.c + e;
        const bool isBox = c < a.nx, isXi = c < 2 * a.nx;
        const T k = s.sp * s.dy[e];
        const T lo = k * s.blo[e];
        const T hi = (isXi && !isBox) ? s.bhi[e] : k * s.bhi[e];
        // SCALE: k_down_chain<T, true> left the primal values (every node)
        const T hxv = SCALE ? k * s.hx[e] : s.hx[e];
        const DualOut<T> o = dual_elem<T, false>(hxv, s.w[e], lo, hi, s.yp[e], a.lambda, a.invLambda, ln, (T)0);
        yn[e] = o.yn; wn[e] = o.wn; z[e] = o.z; res[e] = o.res;
#if !(RN_DUAL_ABL & 2)
        const double dd = counted ? (double)o.diff * (double)o.diff : 0.0;
        r.d2x += isBox ? dd : 0.0;
        r.d2s += (isXi && !isBox) ? dd : 0.0;
        const double rv = (double)o.res;
        const unsigned int ie = (unsigned int)i0 + (unsigned int)e;
        const bool upX = isXi && (fabs(rv) > fabs(r.valXi) || r.idxXi == 0xffffffffu);
        const bool upP = !isXi && (fabs(rv) > fabs(r.valPsi) || r.idxPsi == 0xffffffffu);
        r.valXi = upX ? rv : r.valXi; r.idxXi = upX ? ie : r.idxXi;
        r.valPsi = upP ? rv : r.valPsi; r.idxPsi = upP ? ie : r.idxPsi;
#endif
    }
#if RN_DUAL_ABL & 4
    if (yn[0] == (T)1.2345e-30) reinterpret_cast<VT *>(a.ynew)[s.iv] = wn;
#else
    reinterpret_cast<VT *>(a.ynew)[s.iv] = yn;
#if RN_DUAL_ABL & 8
    if (yn[0] == (T)1.2345e-30)
#endif
    reinterpret_cast<VT *>(a.wnext)[s.iv] = wn;
    if (MATERIALIZE) { reinterpret_cast<VT *>(a.z)[s.iv] = z; reinterpret_cast<VT *>(a.res)[s.iv] = res; }
#endif
    (void)counted;
}
// PIPE = 1: one vector at a time;  PIPE = 2: double-buffered -- the loads of trip t+1 are requested before trip t is consumed, so
// a wave always has a trip in flight (the kernel lives on memory-level parallelism: its VALU phase is a gap in the streams)
template <typename T, bool MATERIALIZE, int PIPE, bool SCALE = false>
__global__ void __launch_bounds__(ELT_THREADS) k_dual_stage(DualArgs<T> a, DualStageShape g) {
    __shared__ Partial sh_p[ELT_THREADS / 64];
    const T ln = (T)g.lnNext;
    const int vpn = g.vpn, tile = ELT_THREADS * g.trips;
    const bool crownBlock = (int)blockIdx.x < g.crownBlocks;
    int stageU = 0, nodeFirst = 0, jbase, cnt;
    if (crownBlock) {
        jbase = (int)blockIdx.x * tile;
        cnt = g.node0 * vpn - jbase;
    } else {
        const int rb = (int)blockIdx.x - g.crownBlocks, sIdx = rb / g.bps, lb = rb - sIdx * g.bps;
        stageU = g.cs + sIdx; nodeFirst = g.node0 + sIdx * g.K;
        jbase = lb * tile;
        cnt = g.K * vpn - jbase;
    }
    cnt = cnt < tile ? cnt : tile;
    DualAcc<T> r;
    if (PIPE == 1) {
        for (int t = 0; t < g.trips; t++) {
            DualSlot<T> s;
            dual_slot_load<T>(s, a, g, t, jbase, cnt, nodeFirst, stageU, crownBlock);
            dual_slot_use<T, MATERIALIZE, SCALE>(s, a, ln, r);
        }
    } else {
        DualSlot<T> sA, sB;
        dual_slot_load<T>(sA, a, g, 0, jbase, cnt, nodeFirst, stageU, crownBlock);
        for (int t = 0; t < g.trips; t += 2) {
            const bool hasB = t + 1 < g.trips;
            if (hasB) dual_slot_load<T>(sB, a, g, t + 1, jbase, cnt, nodeFirst, stageU, crownBlock);
            dual_slot_use<T, MATERIALIZE, SCALE>(sA, a, ln, r);
            if (t + 2 < g.trips) dual_slot_load<T>(sA, a, g, t + 2, jbase, cnt, nodeFirst, stageU, crownBlock);
            if (hasB) dual_slot_use<T, MATERIALIZE, SCALE>(sB, a, ln, r);
        }
    }
#if RN_DUAL_ABL & 2
    if (r.d2x == 1.2345e-30) a.partials[blockIdx.x] = Partial{r.d2x, r.d2s, 0, r.valXi, 0, r.valPsi, r.idxXi, r.idxPsi};
    return;
#endif
    double valXi = r.valXi, valPsi = r.valPsi;
    long long idxXi = r.idxXi == 0xffffffffu ? 0x7fffffffffffffffLL : (long long)r.idxXi;
    long long idxPsi = r.idxPsi == 0xffffffffu ? 0x7fffffffffffffffLL : (long long)r.idxPsi;
    double absXi = r.idxXi == 0xffffffffu ? -1.0 : fabs(valXi), absPsi = r.idxPsi == 0xffffffffu ? -1.0 : fabs(valPsi);
    const double d2x = wave_sum_f64(r.d2x), d2s = wave_sum_f64(r.d2s);
    wave_argmax(absXi, valXi, idxXi);
    wave_argmax(absPsi, valPsi, idxPsi);
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
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_decide(const Partial *partials, int nblocks, IterState *st, double thrX,
                                                        double thrS) {
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, false, tx2, ts2, p);
    if (threadIdx.x == 0) {
        const double dX = sqrt(tx2), dS = sqrt(ts2);
        st->distX = dX; st->distS = dS;
        const bool tx = dX > thrX, ts = dS > thrS;
        st->tripped = (tx || ts) ? 1 : 0;
        st->scaleX = tx ? 1.0 - thrX / dX : 0.0;
        st->scaleS = ts ? 1.0 - thrS / dS : 0.0;
    }
}

// Multi-GPU optimistic bookkeeping (one collective per iteration).  The fused kernel runs without the soft-constraint
// correction; this kernel folds the block partials, stores the rank-local dist^2 of THIS iteration in the tail of the
// cut payload (it rides on the NEXT iteration's all-reduce), writes the rank-local history entry and advances `it`.
template <typename T>
__device__ void finalize_optimistic_body(const FinArgs &fin, const PeerTable *peer, unsigned int peerSeq, unsigned int tailIdx) {
    const Partial *partials = fin.partials;
    const int nblocks = fin.nblocks, histCap = fin.histCap;
    IterState *st = fin.st;
    T *tail = reinterpret_cast<T *>(fin.tail);
    double *hist = fin.hist, *histParts = fin.histParts;
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, true, tx2, ts2, p);
    if (threadIdx.x == 0) {
        if (tail) { tail[0] = (T)tx2; tail[1] = (T)ts2; }
        if (tail && peer) { peer_push(*peer, peerSeq, tailIdx, (T)tx2); peer_push(*peer, peerSeq, tailIdx + 1, (T)ts2); }   // one-shot exchange: the tail travels too
        if (fin.thrX >= 0) {   // single GPU: the distances are complete -- verify the projection-only prox right here
            const double dX = sqrt(tx2), dS = sqrt(ts2);
            st->distX = dX; st->distS = dS;
            if (dX > fin.thrX || dS > fin.thrS) st->violated = 1;
        }
        const int it = st->it;
        write_history(p, it, hist, histParts, histCap);
        st->it = it + 1;
    }
}
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_finalize_optimistic(const Partial *partials, int nblocks, IterState *st, T *tail,
                                                                     double *hist, double *histParts, int histCap, double thrX, double thrS, int *hostVerdict = nullptr) {
    finalize_optimistic_body<T>(FinArgs{partials, nblocks, st, (void *)tail, hist, histParts, histCap, thrX, thrS});
    // single GPU: the batch's verdict goes straight into a host-mapped word (thread 0 did the bookkeeping above: program order), so the host
    // reads it behind its stream synchronisation without a device-to-host copy of its own
    if (hostVerdict != nullptr && threadIdx.x == 0) __hip_atomic_store(hostVerdict, st->violated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Sharded APG, end of a batch (SmpcController::updatePrimalInfeasibity, SmpcController.cu:1480-1496, records ONE tree-global value
// per iteration in vecPrimalInfs, :1521): the ranks' history entries are made tree-global by one MAX all-reduce per BATCH, which
// also carries the ranks' verdicts.  pack: out[0] = this rank's verdict (tail != nullptr: the all-reduced dist^2 of the batch's
// last iteration is checked first against the thresholds), out[1 + 4 i ...] = (v_xi, -v_xi, v_psi, -v_psi) of iteration first + i,
// v = the signed entry at the rank's arg-max |.|: the maxima over the ranks give the largest magnitude of either sign, hence the
// entry at the tree-global arg-max (equal magnitudes of opposite sign on two ranks: the positive one).  out[1 + 4 n] = this rank's
// commFail flag (one-shot exchange): after the MAX every rank knows that some reader gave up, and all of them fail the batch.
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_batch_close_pack(const T *tail, IterState *st, double thrX, double thrS, const double *histParts,
                                                                  int first, int n, double *out) {
    if (threadIdx.x == 0) {
        if (tail) {
            const double dX = sqrt((double)tail[0]), dS = sqrt((double)tail[1]);
            st->distX = dX; st->distS = dS;
            if (dX > thrX || dS > thrS) st->violated = 1;
            out[0] = st->violated ? 1.0 : 0.0;
        } else out[0] = 0.0;
        out[1 + 4 * (size_t)n] = st->commFail ? 1.0 : 0.0;
    }
    for (int i = threadIdx.x; i < n; i += ELT_THREADS) {
        const double vx = histParts[4 * (size_t)(first + i) + 1], vp = histParts[4 * (size_t)(first + i) + 3];
        out[1 + 4 * i] = vx; out[2 + 4 * i] = -vx; out[3 + 4 * i] = vp; out[4 + 4 * i] = -vp;
    }
}
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_batch_close_unpack(const double *in, double *hist, int first, int n, IterState *st) {
    if (threadIdx.x == 0 && in[1 + 4 * (size_t)n] > 0.0) st->commFail = 1;      // a reader gave up on SOME rank: the batch is invalid on every rank
    for (int i = threadIdx.x; i < n; i += ELT_THREADS) {
        const double gx = in[1 + 4 * i] >= in[2 + 4 * i] ? in[1 + 4 * i] : -in[2 + 4 * i];
        const double gp = in[3 + 4 * i] >= in[4 + 4 * i] ? in[3 + 4 * i] : -in[4 + 4 * i];
        hist[first + i] = gx > gp ? gx : gp;
    }
}

// multi-GPU variant of k_decide: fold the local partials to (d2x, d2s), all-reduce those two numbers, then decide
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_reduce_dist(const Partial *partials, int nblocks, double *out2) {
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, false, tx2, ts2, p);
    if (threadIdx.x == 0) { out2[0] = tx2; out2[1] = ts2; }
}
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void k_decide_from(const double *d2, IterState *st, double thrX, double thrS) {
    const double dX = sqrt(d2[0]), dS = sqrt(d2[1]);
    st->distX = dX; st->distS = dS;
    const bool tx = dX > thrX, ts = dS > thrS;
    st->tripped = (tx || ts) ? 1 : 0;
    st->scaleX = tx ? 1.0 - thrX / dX : 0.0;
    st->scaleS = ts ? 1.0 - thrS / dS : 0.0;
}

// one workgroup: primal infeasibility of this iteration (max of the signed entries at the two arg-max |.|
// positions -- the reference's quirk) into hist[it]; advance the iteration counter.
template <int PLAIN = 0>   // (a template so that one translation unit owns its code: instantiations/*.inc)
__global__ void __launch_bounds__(ELT_THREADS) k_finalize(const Partial *partials, int nblocks, IterState *st, double *hist,
                                                          double *histParts, int histCap) {
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, true, tx2, ts2, p);
    if (threadIdx.x == 0) {
        const int it = st->it;
        write_history(p, it, hist, histParts, histCap);
        st->it = it + 1;
    }
}

// ------------------------------------------------------------------------------------------------------
// step-wise elementwise kernels (known-answer test API; same arithmetic as the fused kernel)
template <typename T>
__global__ void k_extrapolate(T *acc, T *xi, const T *upd, T lambda, long long n) {   // SmpcController.cu:535-557
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const T y1 = upd[i];
        acc[i] = extrap_elem(y1, xi[i], lambda);   // the roundings of the fused dual update's w_next
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

// Hx from the primal values k_down_chain<T, true> left in its place (every node): hx = (sqrt(p_i) d_k) * value -- the expression of
// k_dual_stage<..., SCALE>; only for a consumer of Hx other than that kernel behind an unscaled walk (does not happen in the batch loops as they are).
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_hx_scale(T *hx, const T *sqrtp, const T *dy, const int *stageOf, int ny, long long total) {
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * ELT_THREADS) {
        const long long node = i / ny;
        const int r = (int)(i - node * ny);
        hx[i] = (sqrtp[node] * dy[(size_t)stageOf[node] * ny + r]) * hx[i];
    }
}

// Opening of an optimistic batch in ONE launch: the checkpoint of (y, y+, w) the exact replay would start from, the verdict flag cleared, and -- sharded
// contexts -- the 2-element dist^2 tail of the cut payload zeroed (three device-to-device copies, a fill and, sharded, one more fill before: per batch
// of 20 iterations that was 4-5 launches of 5-6 us each; the copies of the small trees are launch floors).
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_batch_open(const T *y0, const T *y1, const T *w, T *c0, T *c1, T *c2, long long n, IterState *st, T *tail) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    if (blockIdx.x == 0 && threadIdx.x == 0) { st->violated = 0; if (tail) { tail[0] = (T)0; tail[1] = (T)0; } }
    const long long nv = n / VN;
    const bool aligned = ((((size_t)y0 | (size_t)y1 | (size_t)w | (size_t)c0 | (size_t)c1 | (size_t)c2) & 15) == 0);
    if (aligned) {
        for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < nv; i += (long long)gridDim.x * ELT_THREADS) {
            const VT a = reinterpret_cast<const VT *>(y0)[i], b = reinterpret_cast<const VT *>(y1)[i], c = reinterpret_cast<const VT *>(w)[i];
            reinterpret_cast<VT *>(c0)[i] = a; reinterpret_cast<VT *>(c1)[i] = b; reinterpret_cast<VT *>(c2)[i] = c;
        }
        for (long long i = nv * VN + (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * ELT_THREADS) { c0[i] = y0[i]; c1[i] = y1[i]; c2[i] = w[i]; }
    } else {
        for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * ELT_THREADS) { c0[i] = y0[i]; c1[i] = y1[i]; c2[i] = w[i]; }
    }
}


}  // namespace rn
