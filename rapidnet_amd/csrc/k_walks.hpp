// k_walks.hpp -- the vector recursions of the tree sweep: chain walks (k_up_chain, k_up_chain_cut, k_down_chain, k_down_chain_dual), crown steps, the cut parents' partial sums
// (part of the kernel sources of librapidnet_hip.so; kernels.hpp includes every family header, the translation units k_*.hip instantiate them)
#pragma once
#include "common.hpp"
#include "k_dual.hpp"

namespace rn {

// ------------------------------------------------------------------------------------------------------
// Leaf-to-root recursion of the backward sweep (SmpcController.cu:593-673 + solveSumChildren
// Utilities.cu:168-201), re-associated so that no matrix product sits on the sequential path:
//   rho_i   = beta_i + m2_i + sum_c rho_c        kappa_i = sum_c (kappa_c + q_c)        q_i = a_i + sum_c q_c
// with r_i = rho_i + Bbt kappa_i  (the reference's r_j of :629-646; Bbt = Gtil) and
//   s_i = beta_i + sum_c rho_c  so that  sigma_i + Gtil q_in = s_i + Bbt kappa_i     (:599, :644)
// The products with the shared Rinv and Rinv*Bbt are applied afterwards to all nodes at once (k_gemm_shared).
// Chain region (stages >= c*, every node has exactly one child at the same position): one workgroup per
// scenario chain, thread t owns one component and walks from the leaf to the chain top.
constexpr int CHAIN_THREADS = 256;
static_assert(CHAIN_THREADS == ELT_THREADS, "the bookkeeping workgroup of k_up_chain folds with ELT_THREADS threads");
#ifndef RN_CHAIN_PF
#define RN_CHAIN_PF 12
#endif
constexpr int CHAIN_PF = RN_CHAIN_PF;
constexpr int CROWN_THREADS = 1024;   // stages prefetched per round trip (the recursion itself is a running sum)
// (the walk is spelled out twice, here and in k_up_chain_cut: shared through a device function it measured 8 us slower on the
//  493-scenario tree -- 20.1 instead of 11.8 us)
// SPLIT: the instantiation that adds the second partial m2 of k_stream_gemv's split last round (the other one is the walk as it always was)
#ifndef RN_UP_PF
#define RN_UP_PF 24
#endif
constexpr int UP_PF = RN_UP_PF;     // stages per batch of loads of k_up_chain: the 22 stages of a Barcelona chain in ONE batch (88 -> see the resource report; the walk down keeps CHAIN_PF: four arrays per stage)
template <typename T, bool SPLIT = false>
__global__ void __launch_bounds__(CHAIN_THREADS) k_up_chain(SweepArgs<T> a, FinArgs fin) {
    if ((int)blockIdx.x >= a.K) { finalize_optimistic_body<T>(fin); return; }   // CHAIN_THREADS == ELT_THREADS
    const int s = blockIdx.x;                      // chain = position within the stage
    const int nv = a.nv, nx = a.nx;
    const int top = a.chainStage;
    const T *__restrict__ beta = a.beta;
    const T *__restrict__ my = a.my;
    const T *__restrict__ qa = a.qa;
    // every stage >= c* has K nodes: the node of stage k in chain s is nodeTop + (k - c*) K -- no stage-table load per step
    const size_t nodeTop = (size_t)a.chain0 + s;      // (by value: no stage-table load in front of the first requests)
    for (int t = threadIdx.x; t < nv + nx; t += CHAIN_THREADS) {
        if (t < nv) {
            T rho = 0;
            // the second partial m2 of the chain's last stages (k_stream_gemv's split last round; zero where a node was not split)
            T mx[STREAM_SPLIT_STAGES];
#pragma unroll
            for (int j = 0; j < STREAM_SPLIT_STAGES; j++) {
                const int kk = a.N - 1 - j >= top ? a.N - 1 - j : top;
                const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                const bool has = SPLIT && a.N - 1 - j >= top && node >= (size_t)a.splitFirst;
                mx[j] = has ? a.my2[(node - (has ? (size_t)a.splitFirst : 0)) * 2 * nv + nv + t] : (T)0;
            }
            for (int k = a.N - 1; k >= top; k -= UP_PF) {
                T b[UP_PF], m[UP_PF];
#pragma unroll
                for (int j = 0; j < UP_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    b[j] = beta[node * nv + t];
                    m[j] = my[node * 2 * nv + nv + t];
                }
                if (SPLIT && k == a.N - 1) {
#pragma unroll
                    for (int j = 0; j < STREAM_SPLIT_STAGES && j < UP_PF; j++) m[j] += mx[j];      // (first half) + (second half)
                }
#pragma unroll
                for (int j = 0; j < UP_PF; j++) {
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        const T sv = b[j] + rho;                   // s_i
                        rho = sv + m[j];
                        a.sk[node * (nv + nx) + t] = a.structured ? rho : sv;
                    }
                }
            }
            a.rkq[nodeTop * (nv + 2 * nx) + t] = rho;
        } else {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int k = a.N - 1; k >= top; k -= UP_PF) {
                T av[UP_PF];
#pragma unroll
                for (int j = 0; j < UP_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    av[j] = qa[(nodeTop + (size_t)(kk - top) * a.K) * nx + j0];
                }
#pragma unroll
                for (int j = 0; j < UP_PF; j++) {
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        kap += q;                                  // kappa_i = kappa_c + q_c
                        a.sk[node * (nv + nx) + nv + j0] = kap;
                        q += av[j];                                // q_i = a_i + q_c
                    }
                }
            }
            const size_t ntop = nodeTop;
            a.rkq[ntop * (nv + 2 * nx) + nv + j0] = kap;
            a.rkq[ntop * (nv + 2 * nx) + nv + nx + j0] = q;
        }
    }
}
// Sharded runs whose cut lies right above the chains, few local chains per cut parent (an 8-way split of the 17 x 29 tree: 3 or 4):
// ONE workgroup per CUT PARENT walks all its local chains side by side and sums their tops through LDS -- the all-reduce
// payload [parent][rho | kappa | q] comes out of this launch and k_cut_partial_sums (a dependent launch of ~5 us that reads
// 62 x 223 values back) disappears.  Children are added in ascending order, as k_cut_partial_sums does.  One more workgroup
// (blockIdx = nParents, when fin.partials != nullptr) does the bookkeeping of the previous iteration's dual update.
constexpr int UPCUT_THREADS = 1024;
// GATHER (one-shot exchange): every workgroup takes the other ranks' packets for ITS parent out of the inbox right behind its own pushes
// and leaves the all-rank sums in `out` -- what the collective would have left there -- so the exchange is gathered by as many
// workgroups as there are cut parents (223 values x ranks each) instead of by the one critical workgroup of the v / Lv launch
// (3 791 x ranks: +5 us there), and every later kernel is the RCCL path's.  The dist^2 tail is gathered by the bookkeeping workgroup.
template <typename T, bool SPLIT = false, bool GATHER = false>
__global__ void __launch_bounds__(UPCUT_THREADS) k_up_chain_cut(SweepArgs<T> a, T *out, int nParents, int lanesPer, FinArgs fin) {
    if ((int)blockIdx.x >= nParents) {
        if (threadIdx.x >= ELT_THREADS) return;      // the bookkeeping is written for ELT_THREADS threads
        const unsigned int tailIdx = (unsigned int)nParents * (unsigned int)(a.nv + 2 * a.nx);
        finalize_optimistic_body<T>(fin, a.peer.nranks > 0 ? &a.peer : nullptr, a.peerSeq, tailIdx);
        if (GATHER && a.peer.nranks > 0 && a.peerTail && threadIdx.x < 64)      // (the wave of thread 0, which pushed the local tail)
            peer_gather_small<T>(a.peer, a.peerSeq, out, (int)tailIdx, (int)tailIdx + 2, threadIdx.x, 64, reinterpret_cast<IterState *>(a.iterState));
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh = reinterpret_cast<T *>(smem_raw);         // [slots][nv + 2 nx]
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx;
    const int node = a.cut0 + blockIdx.x;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    // lanesPer is a whole number of waves: the chain a wave works on is wave-uniform, and telling the compiler so keeps the node
    // indices and the stage-table loads on the scalar unit, as in k_up_chain
    const int slot = __builtin_amdgcn_readfirstlane((int)threadIdx.x / lanesPer), t = (int)threadIdx.x - slot * lanesPer;
    if (slot < nc && t < nv + nx) {
        const int top = a.chainStage;
        const T *__restrict__ beta = a.beta;
        const T *__restrict__ my = a.my;
        const T *__restrict__ qa = a.qa;
        const size_t nodeTop = (size_t)(c0 + slot);   // the chain's top node; every stage >= c* has K nodes
        if (t < nv) {
            T rho = 0;
            // the second partial m2 of the chain's last stages (k_stream_gemv's split last round; zero where a node was not split)
            T mx[STREAM_SPLIT_STAGES];
#pragma unroll
            for (int j = 0; j < STREAM_SPLIT_STAGES; j++) {
                const int kk = a.N - 1 - j >= top ? a.N - 1 - j : top;
                const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                const bool has = SPLIT && a.N - 1 - j >= top && node >= (size_t)a.splitFirst;
                mx[j] = has ? a.my2[(node - (has ? (size_t)a.splitFirst : 0)) * 2 * nv + nv + t] : (T)0;
            }
            for (int k = a.N - 1; k >= top; k -= CHAIN_PF) {
                T b[CHAIN_PF], m[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    b[j] = beta[node * nv + t];
                    m[j] = my[node * 2 * nv + nv + t];
                }
                if (SPLIT && k == a.N - 1) {
#pragma unroll
                    for (int j = 0; j < STREAM_SPLIT_STAGES && j < CHAIN_PF; j++) m[j] += mx[j];      // (first half) + (second half)
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        const T sv = b[j] + rho;                   // s_i
                        rho = sv + m[j];
                        a.sk[node * (nv + nx) + t] = a.structured ? rho : sv;
                    }
                }
            }
            a.rkq[nodeTop * w + t] = rho;
            sh[slot * w + t] = rho;
        } else {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int k = a.N - 1; k >= top; k -= CHAIN_PF) {
                T av[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    av[j] = qa[(nodeTop + (size_t)(kk - top) * a.K) * nx + j0];
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        kap += q;                                  // kappa_i = kappa_c + q_c
                        a.sk[node * (nv + nx) + nv + j0] = kap;
                        q += av[j];                                // q_i = a_i + q_c
                    }
                }
            }
            const size_t ntop = nodeTop;
            a.rkq[ntop * w + nv + j0] = kap;
            a.rkq[ntop * w + nv + nx + j0] = q;
            sh[slot * w + nv + j0] = kap;
            sh[slot * w + nv + nx + j0] = q;
        }
    }
    __syncthreads();
    for (int tt = threadIdx.x; tt < w; tt += UPCUT_THREADS) {
        T sum = 0;
        for (int c = 0; c < nc; c++) sum += sh[c * w + tt];
        out[(size_t)blockIdx.x * w + tt] = sum;
        if (a.peer.nranks > 0) peer_push(a.peer, a.peerSeq, (unsigned int)blockIdx.x * (unsigned int)w + (unsigned int)tt, sum);   // one-shot exchange: straight to every peer
    }
    if (GATHER && a.peer.nranks > 0)      // (thread tt gathers the very elements it pushed)
        peer_gather_small<T>(a.peer, a.peerSeq, out, (int)blockIdx.x * w, ((int)blockIdx.x + 1) * w, threadIdx.x, UPCUT_THREADS, reinterpret_cast<IterState *>(a.iterState));
}
// Crown region (stages < c*), one node: children are summed explicitly (loads batched CHAIN_PF at a time).
template <typename T>
__device__ __forceinline__ void up_crown_node(const SweepArgs<T> &a, int stage, int pos, int tid, int nthreads) {
    const int node = a.tr.stageCum[stage] + pos;
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const bool presummed = (a.cutSums != nullptr) && (stage == a.cutStage - 1);
    const T *rk = a.rkq;
    for (int t = tid; t < nv + nx; t += nthreads) {
        if (t < nv) {
            T sum = 0;
            if (presummed) sum = a.cutSums[(size_t)pos * w + t];
            else for (int c = 0; c < nc; c += CHAIN_PF) {
                T r[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) r[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * w + t] : (T)0;
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) sum += r[j];
            }
            const T sv = a.beta[(size_t)node * nv + t] + sum;
            const T rho = sv + a.my[(size_t)node * 2 * nv + nv + t];
            a.sk[(size_t)node * (nv + nx) + t] = a.structured ? rho : sv;
            a.rkq[(size_t)node * w + t] = rho;
        } else {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            if (presummed) { kap = a.cutSums[(size_t)pos * w + nv + j0]; q = a.cutSums[(size_t)pos * w + nv + nx + j0]; kap += q; }
            else for (int c = 0; c < nc; c += CHAIN_PF) {
                T kc[CHAIN_PF], qc[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    kc[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * w + nv + j0] : (T)0;
                    qc[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * w + nv + nx + j0] : (T)0;
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) { kap += kc[j] + qc[j]; q += qc[j]; }
            }
            a.sk[(size_t)node * (nv + nx) + nv + j0] = kap;
            a.rkq[(size_t)node * w + nv + j0] = kap;
            a.rkq[(size_t)node * w + nv + nx + j0] = q + a.qa[(size_t)node * nx + j0];
        }
    }
}
// Exchange stage of a sharded run, nodes [lo, hi) of `stage` (= cutStage - 1): the children sums are the all-reduced
// payload, so a node is 'beta + payload' -- (node, component) pairs are dealt flat to the threads and the loads of UP_FLAT
// pairs are requested together (one or two round trips for the whole stage instead of one per node).
constexpr int UP_FLAT = 6;
template <typename T>
__device__ __forceinline__ void up_crown_presummed_flat(const SweepArgs<T> &a, int stage, int lo, int hi, int tid, int nthreads) {
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx, per = nv + nx;
    const int s0 = a.tr.stageCum[stage];
    const int total = (hi - lo) * per;
    for (int i0 = tid; i0 < total; i0 += nthreads * UP_FLAT) {
        T c0[UP_FLAT], c1[UP_FLAT], b0[UP_FLAT], b1[UP_FLAT];
#pragma unroll
        for (int u = 0; u < UP_FLAT; u++) {
            const int i = i0 + u * nthreads;
            const bool on = i < total;
            const int node = lo + (on ? i / per : 0), t = on ? i % per : 0, pos = node - s0;
            if (t < nv) {
                c0[u] = a.cutSums[(size_t)pos * w + t]; c1[u] = 0;
                b0[u] = a.beta[(size_t)node * nv + t]; b1[u] = a.my[(size_t)node * 2 * nv + nv + t];
            } else {
                const int j0 = t - nv;
                c0[u] = a.cutSums[(size_t)pos * w + nv + j0]; c1[u] = a.cutSums[(size_t)pos * w + nv + nx + j0];
                b0[u] = a.qa[(size_t)node * nx + j0]; b1[u] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < UP_FLAT; u++) {
            const int i = i0 + u * nthreads;
            if (i < total) {
                const int node = lo + i / per, t = i % per;
                if (t < nv) {                                        // same association as up_crown_node
                    const T sv = b0[u] + c0[u];
                    const T rho = sv + b1[u];
                    a.sk[(size_t)node * per + t] = a.structured ? rho : sv;
                    a.rkq[(size_t)node * w + t] = rho;
                } else {
                    const int j0 = t - nv;
                    const T kap = c0[u] + c1[u];
                    a.sk[(size_t)node * per + nv + j0] = kap;
                    a.rkq[(size_t)node * w + nv + j0] = kap;
                    a.rkq[(size_t)node * w + nv + nx + j0] = c1[u] + b0[u];
                }
            }
        }
    }
}
// Sharded runs with a two-stage crown (root + exchange stage): the ROOT's step straight from the exchange stage's inputs,
// rho_c = (beta_c + payload_c) + m2_c etc. recomputed per child instead of read back -- bitwise the values
// up_crown_presummed_flat stores, summed in up_crown_node's order -- so that the root does not wait for a store -> barrier
// -> load round trip behind the exchange stage (both steps cost one batch of independent loads).
template <typename T>
__device__ __forceinline__ void up_root_from_presummed(const SweepArgs<T> &a, int tid, int nthreads) {
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx, per = nv + nx;
    const int c0 = a.rootC0, nc = a.rootNc, s1 = a.s1;
    for (int t = tid; t < per; t += nthreads) {
        if (t < nv) {
            T sum = 0;
            for (int c = 0; c < nc; c += CHAIN_PF) {
                T cs[CHAIN_PF], bs[CHAIN_PF], ms[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int ch = c0 + (c + j < nc ? c + j : 0);
                    cs[j] = a.cutSums[(size_t)(ch - s1) * w + t]; bs[j] = a.beta[(size_t)ch * nv + t]; ms[j] = a.my[(size_t)ch * 2 * nv + nv + t];
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) if (c + j < nc) sum += (bs[j] + cs[j]) + ms[j];
            }
            const T sv = a.beta[t] + sum;
            const T rho = sv + a.my[nv + t];
            a.sk[t] = a.structured ? rho : sv;
            a.rkq[t] = rho;
        } else {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int c = 0; c < nc; c += CHAIN_PF) {
                T ck[CHAIN_PF], cq[CHAIN_PF], qs[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int ch = c0 + (c + j < nc ? c + j : 0);
                    ck[j] = a.cutSums[(size_t)(ch - s1) * w + nv + j0]; cq[j] = a.cutSums[(size_t)(ch - s1) * w + nv + nx + j0];
                    qs[j] = a.qa[(size_t)ch * nx + j0];
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++)
                    if (c + j < nc) { const T kc = ck[j] + cq[j], qc = cq[j] + qs[j]; kap += kc + qc; q += qc; }
            }
            a.sk[nv + j0] = kap;
            a.rkq[nv + j0] = kap;
            a.rkq[nv + nx + j0] = q + a.qa[j0];
        }
    }
}
// The same two steps (exchange stage + root) by ONE workgroup in one batch of loads: the (child, component) pairs of the
// root's children are dealt flat to all threads, every child's rho / kappa / q goes to global memory (what
// up_crown_presummed_flat stores) and to LDS, and after one barrier thread t folds the children's values in ascending
// order (up_crown_node's association).  sh: nc * (nv + 2 nx) reals.
// slab != nullptr: the [s; kappa] columns of the nodes 0 .. 15 (the workgroup's own slab of the v product: the root and the first
// stage-1 nodes) are also written straight into the slab buffer ([16][SB], zeroed by the caller), so that the workgroup neither
// waits for its stores nor reads them back from global memory
template <typename T>
__device__ __forceinline__ void up_crown2_wg0(const SweepArgs<T> &a, T *sh, int tid, int nthreads, T *slab = nullptr, int SB = 0) {
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx, per = nv + nx;
    const int c0n = a.rootC0, nc = a.rootNc, s1 = a.s1;
    const int total = nc * per;
    // the root's own terms, requested with the first batch
    T rb = 0, rm = 0;
    if (tid < nv) { rb = a.beta[tid]; rm = a.my[nv + tid]; } else if (tid < per) rb = a.qa[tid - nv];
    for (int i0 = tid; i0 < total; i0 += nthreads * UP_FLAT) {
        T c0[UP_FLAT], c1[UP_FLAT], b0[UP_FLAT], b1[UP_FLAT];
#pragma unroll
        for (int u = 0; u < UP_FLAT; u++) {
            const int i = i0 + u * nthreads;
            const bool on = i < total;
            const int c = on ? i / per : 0, t = on ? i % per : 0, node = c0n + c, pos = node - s1;
            if (t < nv) {
                c0[u] = a.cutSums[(size_t)pos * w + t]; c1[u] = 0;
                b0[u] = a.beta[(size_t)node * nv + t]; b1[u] = a.my[(size_t)node * 2 * nv + nv + t];
            } else {
                const int j0 = t - nv;
                c0[u] = a.cutSums[(size_t)pos * w + nv + j0]; c1[u] = a.cutSums[(size_t)pos * w + nv + nx + j0];
                b0[u] = a.qa[(size_t)node * nx + j0]; b1[u] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < UP_FLAT; u++) {
            const int i = i0 + u * nthreads;
            if (i < total) {
                const int c = i / per, t = i % per, node = c0n + c;
                if (t < nv) {
                    const T sv = b0[u] + c0[u];
                    const T rho = sv + b1[u];
                    a.sk[(size_t)node * per + t] = a.structured ? rho : sv;
                    a.rkq[(size_t)node * w + t] = rho;
                    sh[(size_t)c * w + t] = rho;
                    if (slab && node < 16) slab[node * SB + t] = a.structured ? rho : sv;
                } else {
                    const int j0 = t - nv;
                    const T kap = c0[u] + c1[u], q = c1[u] + b0[u];
                    a.sk[(size_t)node * per + nv + j0] = kap;
                    a.rkq[(size_t)node * w + nv + j0] = kap;
                    a.rkq[(size_t)node * w + nv + nx + j0] = q;
                    sh[(size_t)c * w + nv + j0] = kap + q;
                    sh[(size_t)c * w + nv + nx + j0] = q;
                    if (slab && node < 16) slab[node * SB + nv + j0] = kap;
                }
            }
        }
    }
    __syncthreads();
    if (tid < nv) {
        T sum = 0;
        for (int c = 0; c < nc; c++) sum += sh[(size_t)c * w + tid];
        const T sv = rb + sum;
        const T rho = sv + rm;
        a.sk[tid] = a.structured ? rho : sv;
        a.rkq[tid] = rho;
        if (slab) slab[tid] = a.structured ? rho : sv;
    } else if (tid < per) {
        const int j0 = tid - nv;
        T kap = 0, q = 0;
        for (int c = 0; c < nc; c++) { kap += sh[(size_t)c * w + nv + j0]; q += sh[(size_t)c * w + nv + nx + j0]; }
        a.sk[nv + j0] = kap;
        a.rkq[nv + j0] = kap;
        a.rkq[nv + nx + j0] = q + rb;
        if (slab) slab[nv + j0] = kap;
    }
}
// one launch per stage, one workgroup per node; the children are split over `parts` thread groups so that all the
// loads of a node are in flight at once, partial sums are folded through LDS
template <typename T>
__global__ void __launch_bounds__(CROWN_THREADS) k_up_crown(SweepArgs<T> a, int stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh = reinterpret_cast<T *>(smem_raw);            // parts x w
    const int pos = blockIdx.x;
    const int node = a.tr.stageCum[stage] + pos;
    const int nv = a.nv, nx = a.nx, w = nv + 2 * nx;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const bool presummed = (a.cutSums != nullptr) && (stage == a.cutStage - 1);
    if (presummed && a.distTail != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        IterState *st = reinterpret_cast<IterState *>(a.iterState);
        const double dX = sqrt((double)a.distTail[0]), dS = sqrt((double)a.distTail[1]);
        st->distX = dX; st->distS = dS;
        if (dX > a.thrX || dS > a.thrS) st->violated = 1;
    }
    const int wp = (w + 63) / 64 * 64;
    const int parts = CROWN_THREADS / wp > 0 ? CROWN_THREADS / wp : 1;
    const int part = threadIdx.x / wp;
    // vectors wider than the workgroup (w > CROWN_THREADS: parts == 1, part == 0 everywhere): a thread owns components
    // t, t + CROWN_THREADS, ...; otherwise one pass (the second trip starts at t >= wp >= w)
    const int tstep = wp < CROWN_THREADS ? wp : CROWN_THREADS;
    if (part < parts) for (int t = threadIdx.x % wp; t < w; t += tstep) {
        T sum = 0;
        if (!presummed) {
            const T *rk = a.rkq;
            for (int c = part; c < nc; c += parts * CHAIN_PF) {
                T r[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) r[j] = (c + j * parts < nc) ? rk[(size_t)(c0 + c + j * parts) * w + t] : (T)0;
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) sum += r[j];
            }
        } else if (part == 0) sum = a.cutSums[(size_t)pos * w + t];
        sh[part * w + t] = sum;
    }
    __syncthreads();
    // fold: rhoSum (nv) | kappaSum (nx) | qSum (nx)
    for (int tt = threadIdx.x; tt < nv + nx; tt += CROWN_THREADS) {
        if (tt < nv) {
            T sum = 0;
            for (int p = 0; p < parts; p++) sum += sh[p * w + tt];
            const T sv = a.beta[(size_t)node * nv + tt] + sum;
            const T rho = sv + a.my[(size_t)node * 2 * nv + nv + tt];
            a.sk[(size_t)node * (nv + nx) + tt] = a.structured ? rho : sv;
            a.rkq[(size_t)node * w + tt] = rho;
        } else {
            const int j0 = tt - nv;
            T ks = 0, qs = 0;
            for (int p = 0; p < parts; p++) { ks += sh[p * w + nv + j0]; qs += sh[p * w + nv + nx + j0]; }
            const T kap = ks + qs;                                   // kappa_i = sum_c (kappa_c + q_c)
            a.sk[(size_t)node * (nv + nx) + nv + j0] = kap;
            a.rkq[(size_t)node * w + nv + j0] = kap;
            a.rkq[(size_t)node * w + nv + nx + j0] = qs + a.qa[(size_t)node * nx + j0];
        }
    }
}
// ------------------------------------------------------------------------------------------------------
// Structured operator mode, LINEAR form of the leaf-to-root recursion (round 6; unsharded contexts).
// In structured mode every per-node product is (shared matrix) x (diagonal) x (power of p_i):  m2_i = Bbt a_i + L' b_i  with
// a_i = F_i' xi_i, b_i = G_i' psi_i element-wise functions of the dual, and the recursion only ever ADDS such terms up the tree:
//   rho_i = beta_i + m2_i + sum_c rho_c  =  Bs_i + Bbt q_i + L' Bu_i,     Bs_i = beta_i + sum_c Bs_c,  q_i = a_i + sum_c q_c,  Bu_i = b_i + sum_c Bu_c
// (q_i is the recursion's own q; SmpcController.cu:629-672).  So the shared matrices can be applied AFTER the sums:
//   v_i = -(Rinv rho_i + T1 kappa_i) / (2 p_i) = -[Rinv | T1 | T2] [Bs_i ; q_i + kappa_i ; Bu_i] / (2 p_i),      T1 = Rinv Bbt, T2 = Rinv L'
// and the product m2 = [Bbt | L'] [a; b] of ALL nodes -- a launch of its own in front of the chain walks (k_gemm_prep_m2, 21 us on the
// 493-scenario tree) -- disappears: the walk forms a_i, b_i from the dual as it goes, and the v product's K grows from nv + nx to
// nv + nx + nu.  Same sums in another association: iterates agree with the dense form to rounding, as the structured mode always has.
// k_up_chain_lin: one workgroup per scenario chain; component t of [Bs (nv) | kappa, q (nx) | Bu (nu)] belongs to thread t (the few
// components beyond the workgroup's 256 threads to a second trip).  One more workgroup does the previous iteration's bookkeeping.
constexpr int LIN_PF = 12;     // stages per batch of loads of the branches that need the dual and the preconditioner row per stage (registers)
// the walk of chain s: W = a.w (global memory: node-major [node][ny]) or, LDS != nullptr, the chain's rows of the NEXT accelerated dual as the
// fused walk + dual update has just left them in its LDS tile ([row = stage - c*][ny]); the same sums in the same order either way
template <typename T>
__device__ __forceinline__ void up_chain_lin_walk(const SweepArgs<T> &a, int s, const T *LDS, int tid, int nthreads) {
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny, top = a.chainStage;
    const int W = nv + nx + nu, W2 = nv + 2 * nx + nu;
    const size_t nodeTop = (size_t)a.chain0 + s;      // (by value: no stage-table load in front of the first requests)
    const T *__restrict__ dy = a.tr.dy;
    const T sp = a.tr.sqrtp[nodeTop];                  // (a chain does not branch: one probability from its top to its leaf)
    auto wAt = [&](int kk, size_t node, int c) -> T { return LDS ? LDS[(size_t)(kk - top) * ny + c] : a.w[node * ny + c]; };
    // Components are dealt to the threads in WAVE-ALIGNED ranges -- [kappa, q | Bu | Bs], each padded to whole waves -- so that a wave runs one kind
    // of walk (a wave that straddles two kinds runs both, one after the other); the components beyond the workgroup's threads (the tail of Bs,
    // the cheapest kind: one array, one batch of loads) take a second trip.
    // (a.lin & 2: Bs_i -- the subtree sums of beta, which no iteration changes -- is not walked: its term of v_i is a constant of the control step,
    //  SweepArgs::lin)
    const int kqW = (nx + 63) / 64 * 64, buW = (nu + 63) / 64 * 64, bsW = (a.lin & 2) ? 0 : (nv + 63) / 64 * 64;
    for (int slot = tid; slot < kqW + buW + bsW; slot += nthreads) {
        int t;                                          // component in the order [Bs (nv) | kappa, q (nx) | Bu (nu)] of sk2
        if (slot < kqW) { if (slot >= nx) continue; t = nv + slot; }
        else if (slot < kqW + buW) { if (slot - kqW >= nu) continue; t = nv + nx + (slot - kqW); }
        else { if (slot - kqW - buW >= nv) continue; t = slot - kqW - buW; }
        if (t < nv) {                                   // Bs_i = beta_i + Bs_child
            T acc = 0;
            for (int k = a.N - 1; k >= top; k -= UP_PF) {
                T b[UP_PF];
#pragma unroll
                for (int j = 0; j < UP_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    b[j] = a.beta[(nodeTop + (size_t)(kk - top) * a.K) * nv + t];
                }
#pragma unroll
                for (int j = 0; j < UP_PF; j++)
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        acc += b[j];
                        a.sk2[node * W + t] = acc;
                    }
            }
            a.rkq2[nodeTop * W2 + t] = acc;
        } else if (t < nv + nx) {                       // kappa_i = kappa_c + q_c ; q_i = a_i + q_c ; the product's input is q_i + kappa_i
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int k = a.N - 1; k >= top; k -= LIN_PF) {
                T w0[LIN_PF], w1[LIN_PF], d0[LIN_PF], d1[LIN_PF];
#pragma unroll
                for (int j = 0; j < LIN_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    w0[j] = wAt(kk, node, j0); w1[j] = wAt(kk, node, nx + j0);
                    d0[j] = dy[(size_t)kk * ny + j0]; d1[j] = dy[(size_t)kk * ny + nx + j0];
                }
#pragma unroll
                for (int j = 0; j < LIN_PF; j++)
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        kap += q;
                        q += stream_qa_elem(sp, d0[j], w0[j], d1[j], w1[j]);      // a_i, with the roundings every other form of a_i has
                        a.sk2[node * W + t] = q + kap;
                    }
            }
            a.rkq2[nodeTop * W2 + nv + j0] = kap;
            a.rkq2[nodeTop * W2 + nv + nx + j0] = q;
        } else {                                        // Bu_i = b_i + Bu_child
            const int j0 = t - nv - nx;
            T acc = 0;
            for (int k = a.N - 1; k >= top; k -= LIN_PF) {
                T w0[LIN_PF], d0[LIN_PF];
#pragma unroll
                for (int j = 0; j < LIN_PF; j++) {
                    const int kk = k - j >= top ? k - j : top;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    w0[j] = wAt(kk, node, 2 * nx + j0); d0[j] = dy[(size_t)kk * ny + 2 * nx + j0];
                }
#pragma unroll
                for (int j = 0; j < LIN_PF; j++)
                    if (k - j >= top) {
                        const size_t node = nodeTop + (size_t)(k - j - top) * a.K;
                        acc += lin_b_elem(sp, d0[j], w0[j]);
                        a.sk2[node * W + t] = acc;
                    }
            }
            a.rkq2[nodeTop * W2 + nv + 2 * nx + j0] = acc;
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(CHAIN_THREADS) k_up_chain_lin(SweepArgs<T> a, FinArgs fin) {
    if ((int)blockIdx.x >= a.K) { finalize_optimistic_body<T>(fin); return; }
    up_chain_lin_walk<T>(a, (int)blockIdx.x, nullptr, (int)threadIdx.x, CHAIN_THREADS);
}
// one crown node in the linear form: its children's (Bs | kappa | q | Bu) summed in ascending order, its own beta / a / b added
template <typename T>
__device__ __forceinline__ void up_crown_node_lin(const SweepArgs<T> &a, int stage, int pos, int tid, int nthreads) {
    const int node = a.tr.stageCum[stage] + pos;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny, W = nv + nx + nu, W2 = nv + 2 * nx + nu;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const T *rk = a.rkq2;
    const T sp = a.tr.sqrtp[node];
    const T *dy = a.tr.dy + (size_t)stage * ny;
    const T *wn = a.w + (size_t)node * ny;
    for (int t = tid + ((a.lin & 2) ? nv : 0); t < W; t += nthreads) {     // (a.lin & 2: the Bs columns are constants of the control step)
        if (t < nv || t >= nv + nx) {                   // Bs or Bu: one children sum
            const int col = t < nv ? t : t + nx;        // column in rkq2
            T sum = 0;
            for (int c = 0; c < nc; c += CHAIN_PF) {
                T r[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) r[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * W2 + col] : (T)0;
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) sum += r[j];
            }
            T own;
            if (t < nv) own = a.beta[(size_t)node * nv + t];
            else { const int j0 = t - nv - nx; own = lin_b_elem(sp, dy[2 * nx + j0], wn[2 * nx + j0]); }
            const T val = own + sum;
            a.sk2[(size_t)node * W + t] = val;
            a.rkq2[(size_t)node * W2 + col] = val;
        } else {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int c = 0; c < nc; c += CHAIN_PF) {
                T kc[CHAIN_PF], qc[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    kc[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * W2 + nv + j0] : (T)0;
                    qc[j] = (c + j < nc) ? rk[(size_t)(c0 + c + j) * W2 + nv + nx + j0] : (T)0;
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) { kap += kc[j] + qc[j]; q += qc[j]; }
            }
            const T qi = q + stream_qa_elem(sp, dy[j0], wn[j0], dy[nx + j0], wn[nx + j0]);
            a.sk2[(size_t)node * W + t] = qi + kap;
            a.rkq2[(size_t)node * W2 + nv + j0] = kap;
            a.rkq2[(size_t)node * W2 + nv + nx + j0] = qi;
        }
    }
}
// one launch per crown stage, one workgroup per node (the root's step rides in workgroup 0 of the v / Lv launch: up_crown_node_lin there)
// One launch per crown stage, one workgroup per node.  The children's rows (Bs | kappa | q | Bu) are summed by `parts` groups of threads side by
// side -- every load of the node in flight at once (up to CROWN_LIN_PF children per thread) -- and folded through LDS in part order (fixed
// association).  (One more workgroup -- blockIdx = nodes of the stage, when fin.partials != nullptr -- does the previous iteration's bookkeeping:
// the chain walk that otherwise hosts it rode in the previous iteration's fused walk + dual update, k_down_chain_dual UPLIN.)
constexpr int CROWN_LIN_PF = 16;
template <typename T>
__global__ void __launch_bounds__(CROWN_THREADS) k_up_crown_lin(SweepArgs<T> a, int stage, int nNodes, FinArgs fin) {
    if ((int)blockIdx.x >= nNodes) { if (threadIdx.x < ELT_THREADS) finalize_optimistic_body<T>(fin); return; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sh = reinterpret_cast<T *>(smem_raw);            // [parts][W2]
    const int node = a.tr.stageCum[stage] + (int)blockIdx.x;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny, W = nv + nx + nu, W2 = nv + 2 * nx + nu;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    const int cLo = (a.lin & 2) ? nv : 0;               // first column that is summed (a.lin & 2: the Bs columns are constants of the control step)
    const int wp = (W2 - cLo + 63) / 64 * 64;
    const int parts = CROWN_THREADS / wp > 0 ? CROWN_THREADS / wp : 1;
    const int part = threadIdx.x / wp;
    const int tstep = wp < CROWN_THREADS ? wp : CROWN_THREADS;
    // this thread's own terms, requested with the children's rows
    const T sp = a.tr.sqrtp[node];
    const T *dy = a.tr.dy + (size_t)stage * ny;
    const T *wn = a.w + (size_t)node * ny;
    if (part < parts) for (int t = cLo + threadIdx.x % wp; t < W2; t += tstep) {
        T sum = 0;
        for (int c = part; c < nc; c += parts * CROWN_LIN_PF) {
            T r[CROWN_LIN_PF];
#pragma unroll
            for (int j = 0; j < CROWN_LIN_PF; j++) r[j] = (c + j * parts < nc) ? a.rkq2[(size_t)(c0 + c + j * parts) * W2 + t] : (T)0;
#pragma unroll
            for (int j = 0; j < CROWN_LIN_PF; j++) sum += r[j];
        }
        sh[part * W2 + t] = sum;
    }
    __syncthreads();
    for (int t = cLo + threadIdx.x; t < W; t += CROWN_THREADS) {
        if (t < nv || t >= nv + nx) {
            const int col = t < nv ? t : t + nx;
            T sum = 0;
            for (int p = 0; p < parts; p++) sum += sh[p * W2 + col];
            T own;
            if (t < nv) own = a.beta[(size_t)node * nv + t];
            else { const int j0 = t - nv - nx; own = lin_b_elem(sp, dy[2 * nx + j0], wn[2 * nx + j0]); }
            const T val = own + sum;
            a.sk2[(size_t)node * W + t] = val;
            a.rkq2[(size_t)node * W2 + col] = val;
        } else {
            const int j0 = t - nv;
            T ks = 0, qs = 0;
            for (int p = 0; p < parts; p++) { ks += sh[p * W2 + nv + j0]; qs += sh[p * W2 + nv + nx + j0]; }
            const T kap = ks + qs;                       // kappa_i = sum_c (kappa_c + q_c)
            const T qi = qs + stream_qa_elem(sp, dy[j0], wn[j0], dy[nx + j0], wn[nx + j0]);
            a.sk2[(size_t)node * W + t] = qi + kap;
            a.rkq2[(size_t)node * W2 + nv + j0] = kap;
            a.rkq2[(size_t)node * W2 + nv + nx + j0] = qi;
        }
    }
}

// multi-GPU: partial children sums of the cut parents, [parent][rho(nv) | kappa(nx) | q(nx)] (the all-reduce payload).
// Optimistic exchange: one more workgroup (blockIdx = number of cut parents, when fin.partials != nullptr) does the
// bookkeeping of the PREVIOUS iteration's fused dual update -- folds its partials, writes the history entry, advances the
// iteration counter and puts the rank-local dist^2 into the payload's tail -- which would otherwise be a launch of its own
// (k_finalize_optimistic) on the critical path of every iteration.
constexpr int CUT_THREADS = 256;   // = ELT_THREADS (the bookkeeping block's reduction is written for it)
template <typename T, bool GATHER = false>
__global__ void __launch_bounds__(CUT_THREADS) k_cut_partial_sums(SweepArgs<T> a, T *out, int nParents, FinArgs fin) {
    if ((int)blockIdx.x >= nParents) {
        const unsigned int tailIdx = (unsigned int)nParents * (unsigned int)(a.nv + 2 * a.nx);
        finalize_optimistic_body<T>(fin, a.peer.nranks > 0 ? &a.peer : nullptr, a.peerSeq, tailIdx);
        if (GATHER && a.peer.nranks > 0 && a.peerTail && threadIdx.x < 64)      // one-shot exchange gathered here (see k_up_chain_cut)
            peer_gather_small<T>(a.peer, a.peerSeq, out, (int)tailIdx, (int)tailIdx + 2, threadIdx.x, 64, reinterpret_cast<IterState *>(a.iterState));
        return;
    }
    const int node = a.cut0 + blockIdx.x;
    const int w = a.nv + 2 * a.nx;
    const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
    for (int t = threadIdx.x; t < w; t += blockDim.x) {
        T s = 0;
        for (int c = 0; c < nc; c += CHAIN_PF) {            // same summation order as one child after the other
            T r[CHAIN_PF];
#pragma unroll
            for (int j = 0; j < CHAIN_PF; j++) r[j] = (c + j < nc) ? a.rkq[(size_t)(c0 + c + j) * w + t] : (T)0;
#pragma unroll
            for (int j = 0; j < CHAIN_PF; j++) if (c + j < nc) s += r[j];
        }
        out[(size_t)blockIdx.x * w + t] = s;
        if (a.peer.nranks > 0) peer_push(a.peer, a.peerSeq, (unsigned int)blockIdx.x * (unsigned int)w + (unsigned int)t, s);
    }
    if (GATHER && a.peer.nranks > 0)
        peer_gather_small<T>(a.peer, a.peerSeq, out, (int)blockIdx.x * w, ((int)blockIdx.x + 1) * w, threadIdx.x, blockDim.x, reinterpret_cast<IterState *>(a.iterState));
}

// ------------------------------------------------------------------------------------------------------
// Root-to-leaf recursions of the forward sweep (SmpcController.cu:676-741 + solveChildNodesUpdate
// Utilities.cu:142-155) and the diagonal Hx products (:744-747):
//   u_i = uhat_i + (u_anc - uhat_anc) + L v_i        root: (prevU - prevUhat)
//   x_i = x_anc + (e_i + B u_i)                       root: currentX
//   Hx_i = sqrt(p_i) [d_x o x_i ; d_xs o x_i ; d_u o u_i]
// With w_i = u_i - uhat_i = w_anc + L v_i  the state recursion x_i = x_anc + e_i + B u_i becomes
//   x_i = x_anc + (e_i + B uhat_i) + bw_i,   bw_i = B w_i = bw_anc + (B L) v_i
// so ONE GEMM gives [L v_i ; B L v_i] and ONE pass over the tree does both recursions (eb_i = e_i + B uhat_i is
// iteration-invariant, computed with the affine terms).
// foldCrown: the chain workgroup also walks the crown path above its chain (root -> ... -> parent of the chain top; the
// crown nodes' inputs lvb / uhat / eb are all available, so this is a handful of independent loads and adds, no
// dependent round trips) instead of reading u / x / bw of its parent from a crown launch of its own.  The workgroup
// whose chain is the first descendant of a crown node writes that node's u, x, Hx (foldCrown = 1).  Sharded runs
// (foldCrown = 2): a replicated crown node may have no chain on this rank, so the crown nodes are dealt round-robin
// to the workgroups, each of which walks root -> its node once more and writes it; the chain's own path walk writes
// nothing.  (First version: workgroup 0 wrote all of them stage by stage -- 18 dependent passes, 42 us instead of 20.)
constexpr int CROWN_MAX_DEPTH = 8;
template <typename T>
__device__ __forceinline__ void down_crown_node(const SweepArgs<T> &a, int stage, int pos, int tid, int nthreads);
// UNSC (inner iterations of a device-resident batch whose dual update is k_dual_stage<..., HXM = 2>): the Hx buffer receives the PRIMAL values
// (x_i | x_i | u_i) and the dual update applies the scaling sqrt(p_i) d_k -- it has that factor in registers for the bounds anyway, the product is
// the same two roundings -- so the walk does not request the preconditioner table at all (a third to a half of its load instructions)
template <typename T, bool UNSC = false, int PF = CHAIN_PF>
__global__ void __launch_bounds__(CHAIN_THREADS) k_down_chain(SweepArgs<T> a, int foldCrown) {
    // foldCrown = 2 (sharded runs): the grid has one more workgroup per crown node behind the K chain workgroups; it writes that
    // node (root -> node walk at the end of this kernel) while the chain workgroups walk their chains, instead of 18 of the 62
    // chain workgroups doing it after their own chain (15.2 -> see DESIGN.md section 6 for the measured effect)
    // Request order (round 6): the first batch of the chain's rows needs nothing but the chain's number, so it is requested FIRST; the crown path
    // comes from a per-chain table (SweepArgs::chainAnc: one 32-byte scalar load instead of a parent -> parent chase of dependent table loads),
    // its rows are requested right behind the batch -- one round trip for both instead of three.
    const bool crownWriter = (int)blockIdx.x >= a.K;
    const int s = crownWriter ? 0 : (int)blockIdx.x;
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nu + nx;
    const int top = a.chainStage;
    const int ntop = a.chain0 + s;
    const size_t nodeTop = (size_t)ntop;   // every stage >= c* has K nodes: node of stage k in this chain = nodeTop + (k - c*) K
    const T sp = a.tr.sqrtp[ntop];   // p is constant along a chain
    const T *__restrict__ lvb = a.lvb;
    const T *__restrict__ uhat = a.uhat;
    const T *__restrict__ eb = a.eb;
    const T *__restrict__ dyAll = a.tr.dy;
    const int *__restrict__ cum = a.tr.stageCum;
    // crown path, leaf-most first: anc[0] = parent of the chain top (stage top-1) ... anc[top-1] = root
    int anc[CROWN_MAX_DEPTH];
    bool writer[CROWN_MAX_DEPTH];
    if (foldCrown) {
        const int *ca = a.chainAnc + (size_t)s * CROWN_MAX_DEPTH;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) {
            const int e = ca[dd];
            anc[dd] = e & 0x3fffffff; writer[dd] = dd < top && (e >> 30) != 0 && foldCrown == 1;
        }
    }
    const int par = foldCrown ? 0 : a.tr.parent[ntop];
    // a.lin & 4 (structured mode, composite operator): the affine terms uhat_i - uhat_anc, eb_i - eb_anc are part of the product's constant operand
    // (Ctx::lin_const_refresh), so lvb's running sums ARE u_i and B u_i + e-sums: the walk requests neither uhat nor eb
    const bool AF = (a.lin & 4) != 0;
    for (int t = crownWriter ? w : (int)threadIdx.x; t < w; t += CHAIN_THREADS) {
        if (t < nu) {
            T dv[PF], uh[PF], d0[PF];
            auto request = [&](int k) {
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + t];
                    uh[j] = AF ? (T)0 : uhat[node * nu + t];
                    d0[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + 2 * nx + t];
                }
            };
            request(top);
            T run;
            if (foldCrown) {
                run = AF ? a.prevU[t] : a.prevU[t] - a.prevUhat[t];
                T lv[CROWN_MAX_DEPTH], uhc[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + t]; uhc[dd] = AF ? (T)0 : uhat[(size_t)anc[dd] * nu + t]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd < top) {
                        const int k = top - 1 - dd;                       // stage of anc[dd]
                        const T uv = uhc[dd] + run + lv[dd];              // same association as down_crown_node
                        run = uv - uhc[dd];                                // what a child reads back: u_par - uhat_par
                        if (writer[dd]) {
                            const T spc = a.tr.sqrtp[anc[dd]];
                            if (a.writePrimal) a.u[(size_t)anc[dd] * nu + t] = uv;
                            a.hx[(size_t)anc[dd] * ny + 2 * nx + t] = UNSC ? uv : spc * dyAll[(size_t)k * ny + 2 * nx + t] * uv;
                        }
                    }
            } else if (AF) run = par < 0 ? a.prevU[t] : a.u[(size_t)par * nu + t];
            else run = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : (a.u[(size_t)par * nu + t] - a.uhat[(size_t)par * nu + t]);
            for (int k = top; k < a.N; k += PF) {
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    if (k + j < a.N) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        run += dv[j];
                        const T uv = uh[j] + run;
                        if (a.writePrimal) a.u[node * nu + t] = uv;
                        a.hx[node * ny + 2 * nx + t] = UNSC ? uv : sp * d0[j] * uv;
                    }
                }
                if (k + PF < a.N) request(k + PF);
            }
        } else {
            const int j0 = t - nu;
            T dv[PF], ev[PF], d0[PF], d1[PF];
            auto request = [&](int k) {
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + nu + j0];
                    ev[j] = AF ? (T)0 : eb[node * nx + j0];
                    d0[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + j0];
                    d1[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + nx + j0];
                }
            };
            request(top);
            T bw, xr;
            if (foldCrown) {
                bw = a.bw0[j0]; xr = a.curX[j0];
                T lv[CROWN_MAX_DEPTH], evc[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + nu + j0]; evc[dd] = AF ? (T)0 : eb[(size_t)anc[dd] * nx + j0]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd < top) {
                        const int k = top - 1 - dd;
                        bw = bw + lv[dd];
                        xr = xr + evc[dd] + bw;
                        if (writer[dd]) {
                            const T spc = a.tr.sqrtp[anc[dd]];
                            a.bw[(size_t)anc[dd] * nx + j0] = bw;
                            if (a.writePrimal) a.x[(size_t)anc[dd] * nx + j0] = xr;
                            a.hx[(size_t)anc[dd] * ny + j0] = UNSC ? xr : spc * dyAll[(size_t)k * ny + j0] * xr;
                            a.hx[(size_t)anc[dd] * ny + nx + j0] = UNSC ? xr : spc * dyAll[(size_t)k * ny + nx + j0] * xr;
                        }
                    }
            } else {
                bw = par < 0 ? a.bw0[j0] : a.bw[(size_t)par * nx + j0];
                xr = par < 0 ? a.curX[j0] : a.x[(size_t)par * nx + j0];
            }
            for (int k = top; k < a.N; k += PF) {
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    if (k + j < a.N) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        bw += dv[j];
                        xr += ev[j] + bw;
                        if (a.writePrimal) a.x[node * nx + j0] = xr;
                        a.hx[node * ny + j0] = UNSC ? xr : sp * d0[j] * xr;
                        a.hx[node * ny + nx + j0] = UNSC ? xr : sp * d1[j] * xr;
                    }
                }
                if (k + PF < a.N) request(k + PF);
            }
        }
    }
    if (foldCrown == 2) {
        // sharded runs: crown node j is written by workgroup j mod gridDim, which walks root -> j itself (every input
        // of the path is already there: independent loads, then a short running sum with down_crown_node's association)
        const int nCrown = cum[top];
        for (int j = crownWriter ? (int)blockIdx.x - a.K : nCrown; j < nCrown; j += nCrown) {
            const int kj = a.tr.stageOf[j];
            int pth[CROWN_MAX_DEPTH];              // pth[0] = j, pth[kj] = root
            {
                int n = j;
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) { pth[dd] = n; if (dd < kj) n = a.tr.parent[n]; }
            }
            const T spj = a.tr.sqrtp[j];
            for (int t = threadIdx.x; t < w; t += CHAIN_THREADS) {
                if (t < nu) {
                    T run = AF ? a.prevU[t] : a.prevU[t] - a.prevUhat[t];
                    T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
                    for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                        if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + t]; uh[dd] = AF ? (T)0 : uhat[(size_t)pth[dd] * nu + t]; }
                    T uv = 0;
#pragma unroll
                    for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                        if (dd <= kj) { uv = uh[dd] + run + lv[dd]; run = uv - uh[dd]; }
                    if (a.writePrimal) a.u[(size_t)j * nu + t] = uv;
                    a.hx[(size_t)j * ny + 2 * nx + t] = UNSC ? uv : spj * dyAll[(size_t)kj * ny + 2 * nx + t] * uv;
                } else {
                    const int j0 = t - nu;
                    T bw = a.bw0[j0], xr = a.curX[j0];
                    T lv[CROWN_MAX_DEPTH], ev[CROWN_MAX_DEPTH];
#pragma unroll
                    for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                        if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + nu + j0]; ev[dd] = AF ? (T)0 : eb[(size_t)pth[dd] * nx + j0]; }
#pragma unroll
                    for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                        if (dd <= kj) { bw = bw + lv[dd]; xr = xr + ev[dd] + bw; }
                    a.bw[(size_t)j * nx + j0] = bw;
                    if (a.writePrimal) a.x[(size_t)j * nx + j0] = xr;
                    a.hx[(size_t)j * ny + j0] = UNSC ? xr : spj * dyAll[(size_t)kj * ny + j0] * xr;
                    a.hx[(size_t)j * ny + nx + j0] = UNSC ? xr : spj * dyAll[(size_t)kj * ny + nx + j0] * xr;
                }
            }
        }
    }
}
template <typename T>
__device__ __forceinline__ void down_crown_node(const SweepArgs<T> &a, int stage, int pos, int tid, int nthreads) {
    const int node = a.tr.stageCum[stage] + pos;
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nu + nx;
    const int par = a.tr.parent[node];
    const T sp = a.tr.sqrtp[node];
    const T *dy = a.tr.dy + (size_t)stage * ny;
    const bool AF = (a.lin & 4) != 0;       // the affine terms ride in lvb (k_down_chain)
    for (int t = tid; t < w; t += nthreads) {
        if (t < nu) {
            const T wanc = AF ? (par < 0 ? a.prevU[t] : a.u[(size_t)par * nu + t])
                              : (par < 0 ? (a.prevU[t] - a.prevUhat[t]) : (a.u[(size_t)par * nu + t] - a.uhat[(size_t)par * nu + t]));
            const T uv = (AF ? (T)0 : a.uhat[(size_t)node * nu + t]) + wanc + a.lvb[(size_t)node * w + t];
            a.u[(size_t)node * nu + t] = uv;
            a.hx[(size_t)node * ny + 2 * nx + t] = sp * dy[2 * nx + t] * uv;
        } else {
            const int j0 = t - nu;
            const T bw = (par < 0 ? a.bw0[j0] : a.bw[(size_t)par * nx + j0]) + a.lvb[(size_t)node * w + nu + j0];
            const T xv = (par < 0 ? a.curX[j0] : a.x[(size_t)par * nx + j0]) + (AF ? (T)0 : a.eb[(size_t)node * nx + j0]) + bw;
            a.bw[(size_t)node * nx + j0] = bw;
            a.x[(size_t)node * nx + j0] = xv;
            a.hx[(size_t)node * ny + j0] = sp * dy[j0] * xv;
            a.hx[(size_t)node * ny + nx + j0] = sp * dy[nx + j0] * xv;
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(CHAIN_THREADS) k_down_crown(SweepArgs<T> a, int stage) {
    down_crown_node<T>(a, stage, blockIdx.x, threadIdx.x, CHAIN_THREADS);
}
template <typename T>
__global__ void __launch_bounds__(CROWN_THREADS) k_down_crown_all(SweepArgs<T> a, int nStages) {
    const int per = a.nx + a.nu;
    for (int k = 0; k < nStages; k++) {
        const int nk = a.tr.stageCum[k + 1] - a.tr.stageCum[k];
        const int lanesPerNode = per < CROWN_THREADS ? ((per + 63) / 64) * 64 : CROWN_THREADS;
        const int nodesPerPass = CROWN_THREADS / lanesPerNode;
        for (int p0 = 0; p0 < nk; p0 += nodesPerPass) {
            const int pos = p0 + threadIdx.x / lanesPerNode;
            if (pos < nk && threadIdx.x / lanesPerNode < nodesPerPass) down_crown_node<T>(a, k, pos, threadIdx.x % lanesPerNode, lanesPerNode);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// The forward walk AND the dual update of the nodes it has just walked, in one launch (round 5, opt-in: RAPIDNET_FUSE_DOWN_DUAL).
// k_down_chain produces Hx of a chain's nodes (and of the crown nodes it writes); the fused dual update of exactly those elements needs
// nothing else from the sweep, so the same workgroup can do it: phase A is k_down_chain with Hx kept in LDS ([rows][ny]; global memory
// only when the primal iterates are stored), phase B walks the rows' 16-byte vectors like a k_dual_stage tile (dual_elem: the same
// arithmetic element by element).  One dependent launch and the 42 MB round trip of Hx less per iteration.  The arg-max keeps the
// reference's tie rule by comparing indices on equal magnitudes (a thread does not meet its elements in ascending order here).
// UPLIN (structured mode, linear form; inner iterations of a batch): the chain's leaf-to-top running sums of the NEXT iteration (k_up_chain_lin)
// ride here as phase C -- the next accelerated dual of the chain's rows stays in the LDS tile in place of Hx, so the next sweep starts at its
// crown launch: one dependent launch less per iteration (sk2 / rkq2 / beta are the same arrays in every sweep of the context).
#ifndef RN_FUSE_U
#define RN_FUSE_U 3     // 16-byte vectors per thread and pass of phase B
#endif
template <typename T, bool MATERIALIZE, bool UPLIN = false>
__global__ void __launch_bounds__(CHAIN_THREADS, 2) k_down_chain_dual(SweepArgs<T> a, int foldCrown, DualArgs<T> da, double lnNext, int P) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *shx = reinterpret_cast<T *>(smem_raw);      // [L + top][ny]: rows 0 .. L-1 the chain's nodes (stage top + r), rows L + dd the crown nodes this workgroup writes
    __shared__ Partial sh_p[CHAIN_THREADS / 64];
    __shared__ int sh_rowNode[CROWN_MAX_DEPTH];     // crown rows: node (or -1)
    // P workgroups per chain (trees and shards with fewer chains than CUs): workgroup (s, part) walks the chain from its top as far as its own
    // rows [r0, r1) reach -- the walk is a running sum, the loads of the rows above are L2 hits shared with the chain's other workgroups -- and
    // updates the dual of those rows only.  The last part walks the whole chain (it stores the primal iterates when they are asked for), part 0
    // also takes the crown rows.  UPLIN: P = 1 (phase C needs the whole chain's rows in one tile).
    const bool crownWriter = (int)blockIdx.x >= a.K * P;
    const int s = crownWriter ? 0 : (int)blockIdx.x % a.K;
    const int part = crownWriter ? 0 : (int)blockIdx.x / a.K;
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nu + nx;
    const int top = a.chainStage, L = a.N - top;
    const int r0 = crownWriter ? 0 : (int)((long long)part * L / P), r1 = crownWriter ? 0 : (int)((long long)(part + 1) * L / P);
    const int kEnd = top + r1;                         // stages [top, kEnd) are walked
    const bool primalWg = part == P - 1, crownRowsWg = part == 0;
    const bool AF = (a.lin & 4) != 0;       // the affine terms ride in lvb: neither uhat nor eb is requested (k_down_chain)
    const int ntop = a.chain0 + s;
    const size_t nodeTop = (size_t)ntop;
    const T sp = a.tr.sqrtp[ntop];
    const T *__restrict__ lvb = a.lvb;
    const T *__restrict__ uhat = a.uhat;
    const T *__restrict__ eb = a.eb;
    const T *__restrict__ dyAll = a.tr.dy;
    int anc[CROWN_MAX_DEPTH];
    bool writer[CROWN_MAX_DEPTH];
    {      // the crown path from the per-chain table (k_down_chain: one scalar load instead of a chase of dependent ones)
        const int *ca = a.chainAnc + (size_t)s * CROWN_MAX_DEPTH;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) {
            const int e = ca[dd];
            anc[dd] = e & 0x3fffffff; writer[dd] = dd < top && (e >> 30) != 0 && foldCrown == 1 && !crownWriter && crownRowsWg;
        }
    }
    if (threadIdx.x < CROWN_MAX_DEPTH) {
        int nd = -1;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) if ((int)threadIdx.x == dd && writer[dd]) nd = anc[dd];
        sh_rowNode[threadIdx.x] = nd;
    }
    // ---- phase A: the walk (k_down_chain, foldCrown 1 / 2), the PRIMAL values (x | x | u) of the rows into LDS: phase B applies the scaling
    // sqrt(p_i) d_k -- it has that factor in registers for the bounds anyway, the product is the same two roundings (k_down_chain UNSC) -- so the walk
    // requests no preconditioner entries, and Hx, when it is asked for (writePrimal), is stored by the workgroup that updates the row
    auto put = [&](int row, int c, T val) { shx[(size_t)row * ny + c] = val; };
    for (int t = crownWriter ? w : (int)threadIdx.x; t < w; t += CHAIN_THREADS) {
        if (t < nu) {
            T dv[CHAIN_PF], uh2[CHAIN_PF];
            auto request = [&](int k) {
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k + j < kEnd ? k + j : kEnd - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + t];
                    uh2[j] = AF ? (T)0 : uhat[node * nu + t];
                }
            };
            if (kEnd > top) request(top);          // needs nothing but the chain's number: requested first (k_down_chain)
            T run = AF ? a.prevU[t] : a.prevU[t] - a.prevUhat[t];
            T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
            for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + t]; uh[dd] = AF ? (T)0 : uhat[(size_t)anc[dd] * nu + t]; }
#pragma unroll
            for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                if (dd < top) {
                    const int k = top - 1 - dd;
                    const T uv = uh[dd] + run + lv[dd];
                    run = uv - uh[dd];
                    if (writer[dd]) {
                        if (a.writePrimal) a.u[(size_t)anc[dd] * nu + t] = uv;
                        put(L + dd, 2 * nx + t, uv);
                    }
                }
            for (int k = top; k < kEnd; k += CHAIN_PF) {
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k + j < kEnd) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        run += dv[j];
                        const T uv = uh2[j] + run;
                        if (a.writePrimal && primalWg) a.u[node * nu + t] = uv;
                        put(k + j - top, 2 * nx + t, uv);
                    }
                }
                if (k + CHAIN_PF < kEnd) request(k + CHAIN_PF);
            }
        } else {
            const int j0 = t - nu;
            T dv[CHAIN_PF], ev[CHAIN_PF];
            auto request = [&](int k) {
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k + j < kEnd ? k + j : kEnd - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + nu + j0];
                    ev[j] = AF ? (T)0 : eb[node * nx + j0];
                }
            };
            if (kEnd > top) request(top);
            T bw = a.bw0[j0], xr = a.curX[j0];
            T lv[CROWN_MAX_DEPTH], ev0[CROWN_MAX_DEPTH];
#pragma unroll
            for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + nu + j0]; ev0[dd] = AF ? (T)0 : eb[(size_t)anc[dd] * nx + j0]; }
#pragma unroll
            for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                if (dd < top) {
                    const int k = top - 1 - dd;
                    bw = bw + lv[dd];
                    xr = xr + ev0[dd] + bw;
                    if (writer[dd]) {
                        a.bw[(size_t)anc[dd] * nx + j0] = bw;
                        if (a.writePrimal) a.x[(size_t)anc[dd] * nx + j0] = xr;
                        put(L + dd, j0, xr);
                        put(L + dd, nx + j0, xr);
                    }
                }
            for (int k = top; k < kEnd; k += CHAIN_PF) {
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k + j < kEnd) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        bw += dv[j];
                        xr += ev[j] + bw;
                        if (a.writePrimal && primalWg) a.x[node * nx + j0] = xr;
                        put(k + j - top, j0, xr);
                        put(k + j - top, nx + j0, xr);
                    }
                }
                if (k + CHAIN_PF < kEnd) request(k + CHAIN_PF);
            }
        }
    }
    int cwNode = -1, cwStage = 0;
    if (crownWriter) {      // sharded runs (foldCrown = 2): this workgroup writes crown node j (root -> j walk), row 0
        const int j = (int)blockIdx.x - a.K * P;
        cwNode = j; cwStage = a.tr.stageOf[j];
        const int kj = cwStage;
        int pth[CROWN_MAX_DEPTH];
        {
            int n = j;
#pragma unroll
            for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) { pth[dd] = n; if (dd < kj) n = a.tr.parent[n]; }
        }
        const T spj = a.tr.sqrtp[j];
        for (int t = threadIdx.x; t < w; t += CHAIN_THREADS) {
            if (t < nu) {
                T run = AF ? a.prevU[t] : a.prevU[t] - a.prevUhat[t];
                T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + t]; uh[dd] = AF ? (T)0 : uhat[(size_t)pth[dd] * nu + t]; }
                T uv = 0;
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd <= kj) { uv = uh[dd] + run + lv[dd]; run = uv - uh[dd]; }
                if (a.writePrimal) a.u[(size_t)j * nu + t] = uv;
                put(0, 2 * nx + t, uv);
            } else {
                const int j0 = t - nu;
                T bw = a.bw0[j0], xr = a.curX[j0];
                T lv[CROWN_MAX_DEPTH], ev[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + nu + j0]; ev[dd] = AF ? (T)0 : eb[(size_t)pth[dd] * nx + j0]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd <= kj) { bw = bw + lv[dd]; xr = xr + ev[dd] + bw; }
                a.bw[(size_t)j * nx + j0] = bw;
                if (a.writePrimal) a.x[(size_t)j * nx + j0] = xr;
                put(0, j0, xr);
                put(0, nx + j0, xr);
            }
        }
    }
    __syncthreads();
    // ---- phase B: the dual update of the rows' elements (dual_slot_use's arithmetic; LAZY = 0)
    const int vpn = ny / VN;
    const int nOwn = r1 - r0;
    const int nRows = crownWriter ? 1 : nOwn + (crownRowsWg ? top : 0);     // this workgroup's rows: its slice of the chain, then (part 0) the crown rows
    const T ln = (T)lnNext;
    DualAcc<T> r;
    constexpr int U = RN_FUSE_U;
    for (int v0 = threadIdx.x; v0 < nRows * vpn; v0 += U * CHAIN_THREADS) {
        VT hxv[U], wv[U], ypv[U], dyv[U], blov[U], bhiv[U];
        T spv[U];
        long long ivv[U];
        int cv[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int v = v0 + u * CHAIN_THREADS;
            const bool in = v < nRows * vpn;
            const int lr = in ? v / vpn : 0, j = in ? v - lr * vpn : 0;
            const int row = crownWriter ? 0 : (lr < nOwn ? r0 + lr : L + (lr - nOwn));
            int node, stage;
            T spn;
            if (crownWriter) { node = cwNode; stage = cwStage; spn = a.tr.sqrtp[cwNode]; }
            else if (row < L) { node = (int)nodeTop + row * a.K; stage = top + row; spn = sp; }
            else { node = sh_rowNode[row - L]; stage = top - 1 - (row - L); spn = node >= 0 ? a.tr.sqrtp[node] : (T)0; }
            on[u] = in && node >= 0;
            const int nd = on[u] ? node : 0;
            cv[u] = j * VN;
            ivv[u] = (long long)nd * vpn + j;
            spv[u] = spn;
            hxv[u] = *reinterpret_cast<const VT *>(shx + (size_t)row * ny + cv[u]);
            wv[u] = reinterpret_cast<const VT *>(da.w)[ivv[u]];
            ypv[u] = reinterpret_cast<const VT *>(da.yprev)[ivv[u]];
            dyv[u] = *reinterpret_cast<const VT *>(da.dy + (size_t)(on[u] ? stage : 0) * ny + cv[u]);
            blov[u] = *reinterpret_cast<const VT *>(da.blo + cv[u]);
            bhiv[u] = *reinterpret_cast<const VT *>(da.bhi + cv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!on[u]) continue;
            VT yn, wn, z, res;
            const long long i0 = ivv[u] * VN;
            const bool counted = da.countCrown || i0 >= da.crownElems;
#pragma unroll
            for (int e = 0; e < VN; e++) {
                const int c = cv[u] + e;
                const bool isBox = c < da.nx, isXi = c < 2 * da.nx;
                const T k = spv[u] * dyv[u][e];
                const T lo = k * blov[u][e];
                const T hi = (isXi && !isBox) ? bhiv[u][e] : k * bhiv[u][e];
                hxv[u][e] = k * hxv[u][e];                 // Hx = sqrt(p_i) d_k (x | x | u): the walk left the primal value
                const DualOut<T> o = dual_elem<T, false>(hxv[u][e], wv[u][e], lo, hi, ypv[u][e], da.lambda, da.invLambda, ln, (T)0);
                yn[e] = o.yn; wn[e] = o.wn; z[e] = o.z; res[e] = o.res;
                const double dd = counted ? (double)o.diff * (double)o.diff : 0.0;
                r.d2x += isBox ? dd : 0.0;
                r.d2s += (isXi && !isBox) ? dd : 0.0;
                const double rv = (double)o.res;
                const unsigned int ie = (unsigned int)i0 + (unsigned int)e;
                const bool upX = isXi && (r.idxXi == 0xffffffffu || fabs(rv) > fabs(r.valXi) || (fabs(rv) == fabs(r.valXi) && ie < r.idxXi));
                const bool upP = !isXi && (r.idxPsi == 0xffffffffu || fabs(rv) > fabs(r.valPsi) || (fabs(rv) == fabs(r.valPsi) && ie < r.idxPsi));
                r.valXi = upX ? rv : r.valXi; r.idxXi = upX ? ie : r.idxXi;
                r.valPsi = upP ? rv : r.valPsi; r.idxPsi = upP ? ie : r.idxPsi;
            }
            reinterpret_cast<VT *>(da.ynew)[ivv[u]] = yn;
            reinterpret_cast<VT *>(da.wnext)[ivv[u]] = wn;
            if (a.writePrimal) reinterpret_cast<VT *>(a.hx)[ivv[u]] = hxv[u];
            if (MATERIALIZE) { reinterpret_cast<VT *>(da.z)[ivv[u]] = z; reinterpret_cast<VT *>(da.res)[ivv[u]] = res; }
            if (UPLIN) {      // the next accelerated dual of this row, in place of its Hx (read above, by this thread only)
                const int v = v0 + u * CHAIN_THREADS, row = v / vpn;      // (P = 1: local row = row)
                *reinterpret_cast<VT *>(shx + (size_t)row * ny + cv[u]) = wn;
            }
        }
    }
    if (UPLIN && !crownWriter) {      // ---- phase C: the next sweep's chain walk, from the LDS tile
        __syncthreads();
        up_chain_lin_walk<T>(a, s, shx, (int)threadIdx.x, CHAIN_THREADS);
    }
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
        for (int k = 1; k < CHAIN_THREADS / 64; k++) {
            p.d2x += sh_p[k].d2x; p.d2s += sh_p[k].d2s;
            better(p.absXi, p.valXi, p.idxXi, sh_p[k].absXi, sh_p[k].valXi, sh_p[k].idxXi);
            better(p.absPsi, p.valPsi, p.idxPsi, sh_p[k].absPsi, sh_p[k].valPsi, sh_p[k].idxPsi);
        }
        da.partials[blockIdx.x] = p;
    }
}


}  // namespace rn
