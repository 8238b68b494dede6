// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the APG solve path.
//
// Data layout in HBM (T = float | double), chosen for coalesced 16-byte-per-lane streaming:
//   A      [node][ny][LD]   per-node operator block (node stride padded to whole 128-byte lines): column c (c indexes y = [xi_box | xi_safe | psi]) holds
//                           rows 0..nv-1 = [Phi_i | Psi_i](:,c)  and rows nv..2nv-1 = [D_i | Ftil_i](:,c),
//                           zero padded to whole 16-byte slots (fp64: LD = 2nv).  One pass over A_i yields both mat-vecs of
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

constexpr int ELT_THREADS = 256;
#ifndef RN_ELT_MAX_BLOCKS
#define RN_ELT_MAX_BLOCKS 1024
#endif
constexpr int ELT_MAX_BLOCKS = RN_ELT_MAX_BLOCKS;

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

struct IterState {         // device-resident scalars of the APG loop
    int it;                // iteration counter (advanced by whichever kernel does the iteration's bookkeeping)
    unsigned int ticket;   // arrival counter of the fix-up kernel's blocks (rare path only)
    int violated;          // multi-GPU optimistic mode: a tree-global distance exceeded its threshold (sticky)
    int tripped;           // soft-constraint branch taken in this iteration
    double scaleX, scaleS; // 1 - gamma/(lambda dist) for the two halves (0 when not tripped)
    double distX, distS;   // tree-global distances of this iteration
    int commFail;          // one-shot exchange: a reader gave up waiting for a peer's packets (sticky; the host turns it into RN_E_COMM)
};

// ---- one-shot exchange at the cut (opt-in transport, rn_set_exchange_transport) -----------------------
// Instead of an all-reduce launch between the chain walks and the crown, every rank WRITES its partial children sums straight
// into an inbox on every peer (xGMI peer mappings; its own inbox included) and the crown workgroups READ the n contributions
// and add them in rank order -- the same bits on every rank.  No fence, no flag: an element travels as self-validating 8-byte
// packets {32 payload bits, 32-bit sequence tag} (a double = two packets), written with system-scope relaxed atomic stores and
// polled with system-scope atomic loads, so no cache can hold either side back and a torn element is recognised by its tags.
// Two buffers alternate by the parity of the sequence number: a rank can only start exchange s + 2 after it has read every
// peer's packets of s + 1, which every peer wrote after it had finished reading s.  The reader's spin is bounded by the wall
// clock; on time-out it raises IterState::commFail and carries on with what it has (the grid always drains).
constexpr int PEER_MAX = 16;
struct PeerTable {                         // travels BY VALUE in the kernel arguments (a table in memory would put two dependent
                                           // round trips in front of the first packet load of the launch's critical workgroup)
    unsigned long long *inbox[PEER_MAX];   // every rank's inbox as mapped into THIS process (own rank: the local allocation)
    unsigned long long *own;               // = inbox[rank]
    int nranks, rank;                      // nranks == 0: the one-shot exchange is off for this launch
    unsigned int slots;                    // elements per source rank and buffer: cut parents x (nv + 2 nx) + 2 (the dist^2 tail)
    unsigned long long timeoutTicks;       // bound of a reader's wait, in ticks of the 100 MHz wall clock
};
template <typename T> struct PeerPk;
template <> struct PeerPk<double> { static constexpr int N = 2; };
template <> struct PeerPk<float> { static constexpr int N = 1; };
template <typename T>
__device__ __forceinline__ size_t peer_word(int nranks, unsigned int slots, unsigned int seq, int src, unsigned int idx) {
    return (((size_t)(seq & 1u) * (size_t)nranks + (size_t)src) * slots + idx) * PeerPk<T>::N;
}
template <typename T>
__device__ __forceinline__ size_t peer_word(const PeerTable &p, unsigned int seq, int src, unsigned int idx) {
    return peer_word<T>(p.nranks, p.slots, seq, src, idx);
}
__device__ __forceinline__ void peer_push(const PeerTable &p, unsigned int seq, unsigned int idx, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = (unsigned long long)seq << 32;
    const unsigned long long lo = (bits & 0xffffffffull) | tag, hi = (bits >> 32) | tag;
    const size_t wd = peer_word<double>(p, seq, p.rank, idx);
#pragma unroll
    for (int r = 0; r < PEER_MAX; r++)      // static indices: the table lives in the kernel-argument registers
        if (r < p.nranks) {
            __hip_atomic_store(p.inbox[r] + wd, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(p.inbox[r] + wd + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
}
__device__ __forceinline__ void peer_push(const PeerTable &p, unsigned int seq, unsigned int idx, float v) {
    const unsigned long long pk = (unsigned long long)__float_as_uint(v) | ((unsigned long long)seq << 32);
    const size_t wd = peer_word<float>(p, seq, p.rank, idx);
#pragma unroll
    for (int r = 0; r < PEER_MAX; r++)
        if (r < p.nranks) __hip_atomic_store(p.inbox[r] + wd, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// dst[i] = sum over the ranks, in ascending rank order, of element i of exchange `seq`, for i in [i0, i1): called by a whole
// workgroup (tid / nthreads) right behind its own pushes -- the cut parent's workgroup of k_up_chain_cut / k_cut_partial_sums gathers
// the parent's own 223 values x ranks, so the exchange is spread over as many workgroups as there are cut parents.  Four ranks of an
// element at a time with system-scope atomic loads (always a fresh look), a packet that has not arrived is polled on the spot; the
// wait is bounded by the wall clock (IterState::commFail on time-out: the grid always drains).  (Round 4's first form gathered the
// whole payload in workgroup 0 of the v / Lv launch: +5 us on that launch's critical workgroup; removed in round 6.)
template <typename T>
__device__ __forceinline__ void peer_gather_small(const PeerTable &pt, unsigned int seq, T *dst, int i0, int i1, int tid, int nthreads, IterState *st) {
    constexpr int N = PeerPk<T>::N;
    const int R = pt.nranks;
    const unsigned int slotsN = pt.slots * N;
    const unsigned long long limit = pt.timeoutTicks;
    const unsigned long long *base = pt.own + (size_t)(seq & 1u) * (size_t)R * slotsN;
    const long long t0 = wall_clock64();
    bool ok = true;
    for (int i = i0 + tid; i < i1; i += nthreads) {
        T s = 0;
        for (int r0 = 0; r0 < R; r0 += 4) {
            unsigned long long pk[4][N];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + u < R ? r0 + u : R - 1;
                const unsigned long long *w = base + (size_t)r * slotsN + (size_t)i * N;
#pragma unroll
                for (int h = 0; h < N; h++) pk[u][h] = __hip_atomic_load(w + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (r0 + u < R) {
                    const unsigned long long *w = base + (size_t)(r0 + u) * slotsN + (size_t)i * N;
#pragma unroll
                    for (int h = 0; h < N; h++)
                        while ((unsigned int)(pk[u][h] >> 32) != seq) {
                            if ((unsigned long long)(wall_clock64() - t0) > limit) { ok = false; break; }
                            __builtin_amdgcn_s_sleep(4);
                            pk[u][h] = __hip_atomic_load(w + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    T v;
                    if (N == 2) v = (T)__hiloint2double((int)(unsigned int)pk[u][N - 1], (int)(unsigned int)pk[u][0]);
                    else v = (T)__uint_as_float((unsigned int)pk[u][0]);
                    s = (r0 + u == 0) ? v : s + v;
                }
            }
        }
        dst[i] = s;
    }
    if (!ok) st->commFail = 1;
}
template <typename T>
struct SweepArgs {
    TreeDev<T> tr;
    int nx, nu, nv, ny, LD, N, nodes;
    size_t strideA;   // values between consecutive nodes' blocks in A
    int chainStage;   // c*: first stage from which the tree is K parallel chains (no branching at or after it)
    int K;            // nodes per stage in the chain region
    const T *A;
    const T *RT;      // [Rinv | Rinv*Bbt]  nv x (nv+nx)
    const T *L, *B;   // nu x nv, nx x nu
    const T *beta, *uhat, *e;
    const T *curX, *prevU, *prevUhat;
    const T *w;       // accelerated dual the sweep is evaluated at, [node][ny]
    int structured;   // 1: no per-node blocks; m2_i comes from a shared-operator GEMM, m1_i is folded into the v GEMM
    T *ab;            // structured: [node][nx+nu]  a_i = F_i' xi_i ; b_i = G_i' psi_i
    T *my;            // [node][2nv]  m1_i = Phi xi + Psi psi ; m2_i = D xi + Ftil psi
    // k_stream_gemv's split last round (StreamSplit): the nodes >= splitFirst -- all in the last STREAM_SPLIT_STAGES stages of the chain
    // region -- have a second partial [m1; m2] in my2[node - splitFirst] that their consumers add (splitFirst = nodes: none)
    const T *my2; int splitFirst;
    T *qa;            // [node][nx]   a_i = F_i' xi_i
    T *sk;            // [node][nv+nx] s_i = beta_i + sum_children rho_c ; kappa_i
    T *rkq;           // [node][nv+2nx] rho_i, kappa_i, q_i (kept for chain tops and crown nodes)
    // structured mode, LINEAR form of the leaf-to-root recursion (lin != 0; k_up_chain_lin, k_walks.hpp): the running sums are taken of the
    // INPUTS of the shared-operator products instead of their outputs, so that no product sits in front of the recursion at all
    int lin;          // 0: off; bit 0: the linear form; bit 1: Bs_i is not walked (a constant of the control step: Ctx::lin_const_refresh); bit 2: lvb carries the forward walk's affine terms (no uhat / eb requests)
    T *sk2;           // [node][nv+nx+nu]  Bs_i | q_i + kappa_i | Bu_i : the v product's input ([Rinv | T1 | T2] applied to it)
    T *rkq2;          // [node][nv+2nx+nu] Bs_i | kappa_i | q_i | Bu_i  (chain tops and crown nodes: what a parent sums)
    T *v, *lvb;       // [node][nv] ; [node][nu+nx] = [L v_i ; B L v_i]
    const T *eb;      // [node][nx] e_i + B uhat_i (per control step)
    const T *bw0;     // [nx] B (prevU - prevUhat)
    T *bw;            // [node][nx] B (u_i - uhat_i), kept for the crown nodes (parents of the chain tops)
    T *x, *u, *hx;
    const T *cutSums; // multi-GPU: [cutParents][nv+2nx] all-reduced children sums, or nullptr
    int cutStage;     // stage whose parents take cutSums instead of summing their local children (-1: none)
    // optimistic exchange: the all-reduced payload ends with the tree-global dist^2 of the previous iteration; the first
    // crown kernel after the all-reduce checks it against the thresholds (no extra launch)
    const T *distTail; double thrX, thrS; void *iterState;
    // one-shot exchange (nullptr: the payload is all-reduced by a collective between the launches): the kernels that produce the
    // cut parents' local sums push them to every peer under sequence number peerSeq, the crown kernels gather and add them;
    // peerTail: the payload's 2-element dist^2 tail travels with this exchange (the previous iteration's bookkeeping rode along)
    PeerTable peer; unsigned int peerSeq; int peerTail;
    // 0: the primal iterates x, u, v are not stored by this sweep (inner iterations of a device-resident batch: only Hx feeds
    // the dual update; the last iteration of every batch and every step-wise call store them)
    int writePrimal;
    // the few tree-table entries the crown steps start from, by value (a table load in front of the first batch of requests is one
    // more dependent round trip on the critical workgroup of the v / Lv launch): stageCum[1], stageCum[2], childStart[0], childCount[0]
    int s1, e1, rootC0, rootNc;
    // the forward chain walks' crown paths as a per-chain table (Ctx::ensure_chain_anc): [K][CROWN_MAX_DEPTH = 8] ancestors of the chain's top, leaf-most
    // first, bit 30 set where the chain is the node's first descendant chain (the one that writes the node); chain0 = first node of stage chainStage
    const int *chainAnc; int chain0;
    int cut0;         // first node of stage cutStage - 1 (the cut parents), by value
};

// ------------------------------------------------------------------------------------------------------
typedef double nat_d2 __attribute__((ext_vector_type(2)));
typedef float nat_f4 __attribute__((ext_vector_type(4)));

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

#ifndef RN_STREAM_THREADS
#define RN_STREAM_THREADS 512
#endif
#ifndef RN_STREAM_D
#define RN_STREAM_D 5
#endif
#ifndef RN_STREAM_D_WIDE
#define RN_STREAM_D_WIDE 3     // spans of a group when a thread owns 3 or 4 slots per span (register budget)
#endif
#ifndef RN_STREAM_MINW
#define RN_STREAM_MINW 2
#endif
constexpr int STREAM_THREADS = RN_STREAM_THREADS;
constexpr int STREAM_SPLIT_STAGES = 3;   // the split round of k_stream_gemv lies within the last this-many stages (the host checks)
constexpr int STREAM_NLMAX = 4;   // slots per thread and span

// ------------------------------------------------------------------------------------------------------
// Pieces of the fused dual update shared by the kernels further down: the per-workgroup partial reductions, the wave-level
// reductions and the update of one element.
struct Partial {           // per-block partial reductions of the fused kernel
    double d2x, d2s;       // sum (t - clamp)^2 over the box / safety halves
    double absXi, valXi;   // max |res| over xi entries and the signed entry there
    double absPsi, valPsi;
    long long idxXi, idxPsi;
};

__device__ __forceinline__ void better(double &a, double &v, long long &i, double a2, double v2, long long i2) {
    if (a2 > a || (a2 == a && i2 < i)) { a = a2; v = v2; i = i2; }
}

// Wave64 reductions on the VALU: DPP row shifts inside the 16-lane rows, then the gfx9 row broadcasts (row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3); the wave's result ends up in lane 63 and is read back with v_readlane.
// __shfl_down compiles to ds_bpermute_b32 -- two per double, through the CU's ONE LDS pipe: the six-step arg-max fold of the
// fused dual update was 96 of them per wave, ~4 us of LDS time per CU when all 27 resident waves reach their tail together
// (measured: the kernel without its reductions ran 3.7 us faster; nothing else in it touches the LDS pipe).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double old, double x) {
    const int rl = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, ROWMASK, 0xf, false);
    const int rh = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(rh, rl);
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {   // lane must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ long long readlane_i64(long long x, int lane) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(x & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), lane);
    return ((long long)hi << 32) | (long long)lo;
}
__device__ __forceinline__ double wave_sum_f64(double x) {   // fixed association => bitwise repeatable
    x += dpp_f64<0x111, 0xf>(0.0, x);   // row_shr:1
    x += dpp_f64<0x112, 0xf>(0.0, x);   // row_shr:2
    x += dpp_f64<0x114, 0xf>(0.0, x);   // row_shr:4
    x += dpp_f64<0x118, 0xf>(0.0, x);   // row_shr:8  -> lane 15 of every row holds the row's sum
    x += dpp_f64<0x142, 0xa>(0.0, x);   // row_bcast:15 -> rows 1, 3
    x += dpp_f64<0x143, 0xc>(0.0, x);   // row_bcast:31 -> rows 2, 3
    return readlane_f64(x, 63);
}
__device__ __forceinline__ double wave_max_f64(double x, double identity) {
    x = fmax(x, dpp_f64<0x111, 0xf>(identity, x));
    x = fmax(x, dpp_f64<0x112, 0xf>(identity, x));
    x = fmax(x, dpp_f64<0x114, 0xf>(identity, x));
    x = fmax(x, dpp_f64<0x118, 0xf>(identity, x));
    x = fmax(x, dpp_f64<0x142, 0xa>(identity, x));
    x = fmax(x, dpp_f64<0x143, 0xc>(identity, x));
    return readlane_f64(x, 63);
}
// wave-wide arg-max of |.| with the reference's tie rule (cublasIsamax: the FIRST index of the largest magnitude,
// SmpcController.cu:1487-1494): max by DPP, then the lane holding it -- almost always exactly one -- is read back; ties are
// resolved by index in a (wave-uniform) loop over the tied lanes.  absV < 0 marks "no entry".  Result in every lane.
__device__ __forceinline__ void wave_argmax(double &absV, double &val, long long &idx) {
    const double m = wave_max_f64(absV, -1.0);
    if (m < 0.0) { absV = -1.0; val = 0.0; idx = 0x7fffffffffffffffLL; return; }
    unsigned long long tie = __ballot(absV == m);
    int src = (int)__ffsll((long long)tie) - 1;
    if (tie & (tie - 1)) {
        long long best = readlane_i64(idx, src);
        for (unsigned long long t = tie & (tie - 1); t; t &= t - 1) {
            const int l = (int)__ffsll((long long)t) - 1;
            const long long il = readlane_i64(idx, l);
            if (il < best) { best = il; src = l; }
        }
    }
    absV = m; val = readlane_f64(val, src); idx = readlane_i64(idx, src);
}

template <typename T> struct VecOf;
template <> struct VecOf<double> { typedef nat_d2 type; static constexpr int N = 2; };
template <> struct VecOf<float> { typedef nat_f4 type; static constexpr int N = 4; };

template <typename T> struct DualOut { T yn, wn, z, res, diff; };
__device__ __forceinline__ double fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// The update of one element.  Several kernels inline this (k_dual_fused with and without the soft-constraint fix-up,
// k_dual_stage) and a batch mixes them, so the roundings are spelled out -- explicit fused multiply-adds, contraction of
// everything else off -- instead of left to each call site's instruction selection (in fp32 the compiler was seen to
// contract w_next differently in two instances).
template <typename T, bool FIXUP>
__device__ __forceinline__ DualOut<T> dual_elem(T hx, T w, T lo, T hi, T yp, T lambda, T invLambda, T ln, T sc) {
#pragma clang fp contract(off)
    DualOut<T> o;
    const T t = fma_rn(invLambda, w, hx);
    T z = t < lo ? lo : (t > hi ? hi : t);
    o.diff = t - z;
    if (FIXUP) z = fma_rn(sc, o.diff, z);      // sc = 0 on the psi part and on halves that did not trip
    o.z = z;
    o.res = hx - z;
    o.yn = fma_rn(lambda, o.res, w);
    const T a = ((T)1 + ln) * o.yn;
    o.wn = fma_rn(-ln, yp, a);
    return o;
}
// w = (1 + ln) y1 - ln y0 with exactly the roundings of dual_elem's wn: whoever derives the extrapolated dual from the two
// iterates instead of reading it gets the bits the dual update would have stored (ln = 0, y0 = y1: returns y1 unchanged)
template <typename T>
__device__ __forceinline__ T extrap_elem(T y1, T y0, T ln) {
#pragma clang fp contract(off)
    const T a = ((T)1 + ln) * y1;
    return fma_rn(-ln, y0, a);
}

template <typename T> struct Slot;
template <> struct Slot<double> { typedef nat_d2 type; static constexpr int N = 2; };
template <> struct Slot<float> { typedef nat_f4 type; static constexpr int N = 4; };

// Results of the streaming kernel leave with write-through stores (system-scope relaxed atomic stores: `sc0 sc1` on gfx950).
// One workgroup is resident per CU and it cannot retire before its last stores are acknowledged, and lines left dirty in L2
// compete with the read stream when they are written back: measured on the 493-scenario tree, the kernel takes 572 us
// without its stores, 619 us with plain stores (write-back through L2), 595 us with non-temporal ones -- whose consumers
// (k_up_chain, k_gemm_vlv) then read them 3 us slower -- and 582-590 us with write-through stores, which the consumers read
// as fast as plain ones.  Interleaved same-box A/B over 6 rounds (tools/ab_rounds.sh), ms per iteration: plain 0.687,
// non-temporal 0.676, write-through 0.666.
#ifndef RN_STREAM_OUT_POLICY
#define RN_STREAM_OUT_POLICY 2   // 0 plain, 1 non-temporal, 2 write-through
#endif
template <typename T>
__device__ __forceinline__ void store_policy(T v, T *dst, int policy) {   // policy is a compile-time constant at every call site
    if (policy == 2) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (policy == 1) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}
template <typename T>
__device__ __forceinline__ void stream_out(T v, T *dst) { store_policy(v, dst, RN_STREAM_OUT_POLICY); }
template <typename T>
__device__ __forceinline__ T lin_b_elem(T sp, T d, T y) {   // b_i = G_i' psi_i, one element: sqrt(p_i) (d_u y), as k_gemm_prep_m2 forms it
#pragma clang fp contract(off)
    return sp * d * y;
}
template <typename T>
__device__ __forceinline__ T stream_qa_elem(T sp, T d0, T y0, T d1, T y1) {   // roundings spelled out (fp32: contraction is otherwise the compiler's choice)
#pragma clang fp contract(off)
    const T p = d1 * y1;
    return sp * fma_rn(d0, y0, p);
}
// The LAST, partial round of the launch is split by COLUMNS (StreamSplit): one workgroup alone can only pull what it has in flight
// per memory round trip (~40 GB/s: 9.3 us for a 372 KB block however empty the machine is), and the launch times on 700 ... 1 844
// blocks show it -- every started round costs ~9 us at once, then 28 ns per further block (tools/stream_vs_nodes.sh): 5.4 rounds =
// 5 x 14.5 us + 11 us.  So the blocks of the last round (when it is at most half full) are each dealt to TWO workgroups: the first
// takes the spans [0, spanHalf) -- whole columns, the contiguous first part of the block -- and stores its partial [m1; m2] where
// it always goes, the second takes the rest and stores its partial in my2[block - first]; the consumers of the last stages add the
// two (k_up_chain / k_up_chain_cut: m2; the v product's epilogue: m1).  (A split by ROWS -- disjoint outputs, no consumer
// change -- was measured in round 3 and was slower: it reads 400-780-byte runs of every column.  More bytes in flight per
// workgroup do not help either: spans of 10 instead of 5 per group spill and measured 89 -> 95 us.)
template <typename T>
struct StreamSplit {
    int first;        // first block of the split round (= the number of blocks: no split)
    int spanHalf;     // spans of the first half (a whole number of groups)
    T *my2;           // [blocks - first][2 nv] partial sums of the second halves
};

// Bookkeeping of the PREVIOUS iteration's fused dual update (optimistic modes: no decision launch of its own): folds the
// partials, writes the history entry, advances the iteration counter, and either puts the rank-local dist^2 into the
// all-reduce payload's tail (sharded: checked after the collective) or checks it against the thresholds right away
// (thrX >= 0: single GPU).  Rides as one extra workgroup in k_cut_partial_sums (sharded) or k_up_chain (single GPU).
struct FinArgs { const Partial *partials; int nblocks; IterState *st; void *tail; double *hist, *histParts; int histCap; double thrX, thrS; };
template <typename T>
__device__ void finalize_optimistic_body(const FinArgs &fin, const PeerTable *peer = nullptr, unsigned int peerSeq = 0, unsigned int tailIdx = 0);

}  // namespace rn
