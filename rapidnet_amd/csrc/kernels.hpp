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
    // wy1 != nullptr: w is NOT in memory (inner iterations of a device-resident batch, whose dual update does not store it):
    // the consumers of w form  w = (1 + wLn) wy1 - wLn wy0  (extrap_elem: the very roundings of the dual update) from the two
    // dual iterates y_t = wy1, y_{t-1} = wy0 on the fly
    const T *wy1, *wy0; T wLn;
    int structured;   // 1: no per-node blocks; m2_i comes from a shared-operator GEMM, m1_i is folded into the v GEMM
    T *ab;            // structured: [node][nx+nu]  a_i = F_i' xi_i ; b_i = G_i' psi_i
    T *my;            // [node][2nv]  m1_i = Phi xi + Psi psi ; m2_i = D xi + Ftil psi
    // k_stream_gemv's split last round (StreamSplit): the nodes >= splitFirst -- all in the last STREAM_SPLIT_STAGES stages of the chain
    // region -- have a second partial [m1; m2] in my2[node - splitFirst] that their consumers add (splitFirst = nodes: none)
    const T *my2; int splitFirst;
    T *qa;            // [node][nx]   a_i = F_i' xi_i
    T *sk;            // [node][nv+nx] s_i = beta_i + sum_children rho_c ; kappa_i
    T *rkq;           // [node][nv+2nx] rho_i, kappa_i, q_i (kept for chain tops and crown nodes)
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
    // the y column: w itself, or -- when the dual update did not store it -- the two dual iterates it is made of.  Branch-free:
    // without the iterates both loads hit the same w element and extrap_elem(w, w, 0) returns w unchanged
    const T *wA = a.wy1 ? a.wy1 : a.w, *wB = a.wy1 ? a.wy0 : a.w;
    const T wLn = a.wy1 ? a.wLn : (T)0;
    const T w0a = wA[i0], w0b = wB[i0];
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
    if (has0) sh_y[tid] = extrap_elem(w0a, w0b, wLn);
    for (int c = tid + STREAM_THREADS; c < ny; c += STREAM_THREADS) sh_y[c] = extrap_elem(wA[(size_t)node * ny + c], wB[(size_t)node * ny + c], wLn);
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
// element i of the accelerated dual the sweep is evaluated at (SweepArgs::w, or derived from the two iterates)
template <typename T>
__device__ __forceinline__ T sweep_w(const SweepArgs<T> &a, size_t i) {
    return a.wy1 ? extrap_elem(a.wy1[i], a.wy0[i], a.wLn) : a.w[i];
}
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
        if (t < nx) { val = sp * (dy[t] * sweep_w(a, y + t) + dy[nx + t] * sweep_w(a, y + nx + t)); a.qa[(size_t)node * nx + t] = val; }
        else { const int j = t - nx; val = sp * dy[2 * nx + j] * sweep_w(a, y + 2 * nx + j); }
        a.ab[i] = val;
    }
}

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
// Bookkeeping of the PREVIOUS iteration's fused dual update (optimistic modes: no decision launch of its own): folds the
// partials, writes the history entry, advances the iteration counter, and either puts the rank-local dist^2 into the
// all-reduce payload's tail (sharded: checked after the collective) or checks it against the thresholds right away
// (thrX >= 0: single GPU).  Rides as one extra workgroup in k_cut_partial_sums (sharded) or k_up_chain (single GPU).
struct FinArgs { const Partial *partials; int nblocks; IterState *st; void *tail; double *hist, *histParts; int histCap; double thrX, thrS; };
template <typename T>
__device__ void finalize_optimistic_body(const FinArgs &fin, const PeerTable *peer = nullptr, unsigned int peerSeq = 0, unsigned int tailIdx = 0);
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
    const size_t nodeTop = (size_t)a.tr.stageCum[top] + s;
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
    const int node = a.tr.stageCum[a.cutStage - 1] + blockIdx.x;
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
    const int node = a.tr.stageCum[a.cutStage - 1] + blockIdx.x;
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
// Batched product with a SHARED small matrix:  out_i = epi( M * in_i )  for all nodes i, as a tiled GEMM
// [m x k] * [k x nodes] on the matrix cores (MFMA 16x16x4, one wave = 64 rows x 16 nodes, operands straight
// from L2-resident M and the node-major vectors; these products are genuine dense contractions).
// The reference issues these as per-stage cublasSgemm / per-node SgemmBatched calls on K identical copies of the
// matrices (SmpcController.cu:604-611, :692-736).  Epilogues:
//   EPI_V : v_i  = m1_i - acc / (2 p_i)        M = [Rinv | Rinv Bbt], in = [s_i; kappa_i]     (:604-623)
//   EPI_LV: lv_i = acc                          M = L,  in = v_i                                (:692,:701,:727)
//   EPI_Z : z_i  = e_i + acc                    M = B,  in = u_i                                (:695,:715,:736)
enum { EPI_V = 0, EPI_LV = 1, EPI_Z = 2 };   // EPI_LV is also used for the structured m2_i = [Bbt | L'] [a_i; b_i]
template <typename T>
struct GemmArgs {
    const T *M; int m, k;        // logical m x k; stored zero-padded, col-major, mp x kp with mp % 64 == 0, kp % (4 * RN_SLAB_KU) == 0
    int mp, kp;
    const T *in; int ldin;       // in_i = in + i*ldin (k entries)
    T *out; int ldout;           // out_i = out + i*ldout (m entries)
    const T *aux; int ldaux;     // EPI_V: my (m1 at aux + i*ldaux) ; EPI_Z: e
    const T *prob;
    int nodes;
    // EPI_V: the nodes >= auxSplit have a second partial m1 in aux2[(i - auxSplit) * ldaux] (k_stream_gemv's split last round)
    const T *aux2; int auxSplit;    // the same operator in MFMA FRAGMENT ORDER (nullptr: not used): [16-row tile][pair of k-steps][lane][2] -- lane (row = lane & 15, kq = lane >> 4)
    // of tile t finds its A operands of the k-steps 2p and 2p + 1 side by side at ((t * kp / 8 + p) * 64 + lane) * 2, so a wave requests ONE contiguous
    // 16-byte-per-lane kilobyte (fp32: 512 bytes) where the column-major copy takes two requests of four 128-byte lines each
    const T *Mf;
};
// the auxiliary operand of node i, row r (EPI_V: m1, both partials of a split node added; EPI_Z: e)
template <typename T, int EPI>
__device__ __forceinline__ T gemm_aux(const GemmArgs<T> &g, int i, int r) {
    if (g.aux == nullptr) return (T)0;            // structured operator mode: m1 is folded into the v product, nothing to add (uniform branch)
    T v = g.aux[(size_t)i * g.ldaux + r];
    if (EPI == EPI_V && g.aux2 != nullptr && i >= g.auxSplit) v += g.aux2[(size_t)(i - g.auxSplit) * g.ldaux + r];
    return v;
}
// MFMA 16x16x4 wrappers.  fp64: v_mfma_f64_16x16x4_f64, C/D row = (lane>>4) + 4*reg;  fp32: v_mfma_f32_16x16x4_f32,
// C/D row = 4*(lane>>4) + reg;  both: A[row = lane&15][k = lane>>4], B[k = lane>>4][col = lane&15], col = lane&15.
template <typename T> struct Mfma16;
template <> struct Mfma16<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma16<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_WAVES = GEMM_THREADS / 64;
constexpr int GEMM_RT = 4;       // 16-row tiles per workgroup tile (64 output rows x 16 nodes)
// One workgroup = one 64 x 16 output tile; its 4 waves split K (contiguous quarters, multiples of 4) so that 4x more
// waves are in flight (the loop is latency-, not MFMA-bound), partial tiles are summed through LDS by wave 0.
template <typename T, int EPI, int KS>
__global__ void __launch_bounds__(GEMM_THREADS) k_gemm_shared(GemmArgs<T> g) {
    typedef typename Mfma16<T>::acc_t acc_t;
    __shared__ T sh_acc[GEMM_WAVES - 1][GEMM_RT * 4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rowGroups = g.mp / (16 * GEMM_RT);
    const int unit = blockIdx.x;
    const int nodeTile = unit / rowGroups, rg = unit % rowGroups;
    const int r0 = rg * 16 * GEMM_RT;
    const int col = lane & 15, kq = lane >> 4;
    const int node = nodeTile * 16 + col;
    const bool nodeOk = node < g.nodes;
    // K range of this wave
    const int ksteps = g.kp / 4;
    const int per = (ksteps + GEMM_WAVES - 1) / GEMM_WAVES;
    const int kBeg = 4 * (wave * per < ksteps ? wave * per : ksteps);
    const int kEnd = 4 * ((wave + 1) * per < ksteps ? (wave + 1) * per : ksteps);
    // lanes of nodes past the end read a valid node's data and simply do not write the result
    const T *__restrict__ inp = g.in + (size_t)(nodeOk ? node : g.nodes - 1) * g.ldin + kq;
    const T *__restrict__ Mp = g.M + r0 + col + (size_t)(kBeg + kq) * g.mp;   // mp % 64 == 0: all tiles in bounds
    acc_t acc[GEMM_RT];
#pragma unroll
    for (int t = 0; t < GEMM_RT; t++) acc[t] = acc_t{0, 0, 0, 0};
    // every operand of this wave's K range is requested before the first MFMA (KS k-steps: 5 loads each), so the L2
    // latency is paid once per wave instead of once per k-step pair; steps past the range get a zero B operand
    const int nsteps = (kEnd - kBeg) / 4;
    for (int s0 = 0; s0 < nsteps; s0 += KS) {
        T bv[KS], av[KS][GEMM_RT];
#pragma unroll
        for (int i = 0; i < KS; i++) {
            const int k0 = kBeg + 4 * (s0 + i);
            const bool on = (s0 + i < nsteps);
            const int kc = on ? k0 : kBeg;                       // in-bounds address for idle steps
            const bool live = on && (k0 + kq < g.k);
            bv[i] = live ? inp[live ? k0 : 0] : (T)0;          // inp already carries +kq; index 0 is always in bounds
#pragma unroll
            for (int t = 0; t < GEMM_RT; t++) av[i][t] = Mp[t * 16 + (size_t)(kc - kBeg) * g.mp];
        }
#pragma unroll
        for (int i = 0; i < KS; i++)
#pragma unroll
            for (int t = 0; t < GEMM_RT; t++) acc[t] = Mfma16<T>::run(av[i][t], bv[i], acc[t]);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < GEMM_RT; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sh_acc[wave - 1][t * 4 + reg][lane] = acc[t][reg];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < GEMM_WAVES - 1; w++)
#pragma unroll
        for (int t = 0; t < GEMM_RT; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) acc[t][reg] += sh_acc[w][t * 4 + reg][lane];
    if (!nodeOk) return;
    T scale = 0;
    if (EPI == EPI_V) scale = (T)(-0.5) / g.prob[node];
    // epilogue: all auxiliary loads first (independent, clamped in-bounds), then the stores
    T auxv[GEMM_RT][4];
    if (EPI != EPI_LV) {
#pragma unroll
        for (int t = 0; t < GEMM_RT; t++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int gr = r0 + t * 16 + Mfma16<T>::row(lane, reg);
                auxv[t][reg] = gemm_aux<T, EPI>(g, node, gr < g.m ? gr : g.m - 1);
            }
    }
#pragma unroll
    for (int t = 0; t < GEMM_RT; t++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int gr = r0 + t * 16 + Mfma16<T>::row(lane, reg);
            T r = acc[t][reg];
            if (EPI == EPI_V) r = auxv[t][reg] + scale * r;
            if (EPI == EPI_Z) r = auxv[t][reg] + r;
            if (gr < g.m) g.out[(size_t)node * g.ldout + gr] = r;
        }
    }
}

// Slab variant of the same product (the default).  k_gemm_shared re-reads its A fragments per 16-node tile and K
// quarter and gathers its B operand in 32-byte pieces: the vector-memory pipe of the CU, not HBM or the MFMA pipe,
// bounds it.  Here one workgroup owns a slab of 16 consecutive nodes: the slab's inputs (contiguous in memory,
// in = [node][k]) are copied to LDS with fully coalesced loads, every wave owns whole 16-row tiles over the full K (no
// split, no LDS reduction), B operands come from LDS (row stride = 4 mod 64 elements: conflict-free for ds_read_b32 and
// ds_read_b64 in the MFMA operand pattern), only ceil(m/16) row tiles are computed (M stays stored 64-row padded), and
// the workgroup has as many waves as divide the tile count evenly (6 for the 88- and 177-row operators).
// fp64 note: v_mfma_f64_16x16x4 runs at 1/64 per cycle per SIMD (78 TFLOP/s chip-wide), so these products have an
// MFMA floor of the same order as their HBM floor (4-5 us each on the 493-scenario tree).
#ifndef RN_CROWN2_FALLBACK
#define RN_CROWN2_FALLBACK 0
#endif
#ifndef RN_SLAB_MAX_WAVES
#define RN_SLAB_MAX_WAVES 8
#endif
constexpr int SLAB_MAX_WAVES = RN_SLAB_MAX_WAVES;
#ifndef RN_SLAB_ROTATE
#define RN_SLAB_ROTATE 1
#endif
// 1 (ablation builds; measured slower): ALL slab products multiply TRANSPOSED -- the MFMA's A operand is the slab (16 nodes x 4 k,
// from LDS), its B operand the operator's fragment (4 k x 16 rows), so a lane's four accumulator entries are ONE operator row for FOUR nodes
// instead of four rows of one node: for a fixed entry the 64 lanes of a wave touch 4 nodes x 16 consecutive rows = four whole 128-byte pieces of
// the node-major vectors (m1 loads, v / [Lv; BLv] stores) where the untransposed form touches sixteen 32-byte pieces.  Same operand values, same
// k order: the same bits (checked against the LDS-staged and register-resident forms).  But the epilogue then holds four nodes' scales and
// indices per lane: k_gemm_vlv 152 -> 173 registers = two waves per SIMD instead of three, 28.3 -> 32.7 us on the 493-scenario tree (k_down_chain,
// which reads the whole lines, 18.0 -> 17.6); capped at 168 registers it spills.
// 2: only the products without epilogue operands are transposed (EPI_LV: [L v; B L v] and the structured mode's m2) -- their stores leave as
// whole lines and no register is added.
#ifndef RN_SLAB_T
#define RN_SLAB_T 2
#endif
#define RN_SLAB_TR(EPI) (RN_SLAB_T == 1 || (RN_SLAB_T == 2 && (EPI) == EPI_LV))
#ifndef RN_SLAB_KU
#define RN_SLAB_KU 4   // k-steps per group of operands; the operators' K is stored padded to whole groups (host: pad_k)
#endif
template <typename T, int SLAB_LD = 8>
__device__ __forceinline__ void slab_load(T *sB, int SB, const T *in, int ldin, int k, int kp, int node0, int cnt, int wave, int nw, int lane, int rows = 16) {
    // 64 consecutive elements of one row per wave-instruction, SLAB_LD requests in flight per wave; zero fill up to kp / `rows` rows
    const int cpr = (kp + 63) / 64;            // 64-element chunks per row: cover [0, kp) (kp <= SB; beyond k: zeros)
    const int nChunks = rows * cpr;
    for (int c0 = wave; c0 < nChunks; c0 += nw * SLAB_LD) {
        T v[SLAB_LD];
        int dst[SLAB_LD];
#pragma unroll
        for (int u = 0; u < SLAB_LD; u++) {
            const int c = c0 + nw * u;
            const int r = c / cpr, kk = (c - r * cpr) * 64 + lane;
            const bool live = c < nChunks && r < cnt && kk < k;
            v[u] = live ? in[(size_t)(node0 + r) * ldin + kk] : (T)0;
            dst[u] = (c < nChunks && kk < SB) ? r * SB + kk : -1;
        }
#pragma unroll
        for (int u = 0; u < SLAB_LD; u++) if (dst[u] >= 0) sB[dst[u]] = v[u];
    }
}
// The MFMA loop of the slab products: acc[j][c] += A_tile_j (16 x K) * B_coltile_c (K x 16) over G groups of KU k-steps.
// Software-pipelined by hand -- the compiler does not do it, and a loop that requests a group, waits for it and then issues
// its MFMAs runs at a third of the matrix pipe's rate (round 3: 81 ns per v_mfma_f64_16x16x4_f64 and wave where the pipe
// issues one per 27 ns, tools/probes/probe_mfma_f64.hip): the A fragments (global memory: the shared operator, L2-resident)
// are requested TWO groups ahead, the B fragments (LDS) one group ahead, and the loop body has no guard of any kind: the
// operators are stored with K padded by zero columns to a whole number of groups (host: pad_k), the LDS slab is zero beyond
// k, and the prefetches past the last group re-read the last group (a scalar min on the group index, no branch).
// Same order of accumulation over k as every earlier version of these kernels.
// FRAG: Ap[j] points at the tile's data in the operator's fragment-ordered copy (GemmArgs::Mf: this lane's pair of k-steps 2p, 2p + 1 at
// Ap[j] + p * 128): one 16-byte request per pair instead of two strided 8-byte ones; aStep is not used.  Same operands, same order of MFMAs.
template <typename T, int TG, int CT, int KU, bool FRAG = false, bool TR = false>
__device__ __forceinline__ void slab_mfma_pipe(typename Mfma16<T>::acc_t (&acc)[TG][CT], const T *(&Ap)[TG], size_t aStep, const T *Bp,
                                               int bTile, int G) {
    typedef T frag2 __attribute__((ext_vector_type(2)));
    static_assert(KU % 2 == 0, "pairs of k-steps");
    T a0[KU][TG], a1[KU][TG], a2[KU][TG], b0[KU][CT], b1[KU][CT], b2[KU][CT];   // three rotating sets: no register copies of in-flight loads
#define RN_LOAD_A(dst, g_)                                                                                             \
    if (FRAG) {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < KU; i += 2)                                                              \
            _Pragma("unroll") for (int j = 0; j < TG; j++) {                                                           \
                const frag2 v_ = reinterpret_cast<const frag2 *>(Ap[j])[((size_t)(g_) * (KU / 2) + i / 2) * 64];         \
                dst[i][j] = v_[0]; dst[i + 1][j] = v_[1];                                                              \
            }                                                                                                          \
    } else                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < KU; i++)                                                                     \
        _Pragma("unroll") for (int j = 0; j < TG; j++) dst[i][j] = Ap[j][((size_t)(g_) * KU + i) * aStep];
#define RN_LOAD_B(dst, g_)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < KU; i++)                                                                     \
        _Pragma("unroll") for (int c = 0; c < CT; c++) dst[i][c] = Bp[c * bTile + ((g_) * KU + i) * 4];
    // one group: request A of group g + 2 and B of group g + 1, multiply group g, then PIN the operands of group g + 1 (an empty
    // asm that uses them: they are registers here) -- without the pins the compiler sinks every request down to its first use
    // in a later step and the loop is "request, wait, multiply" again; the scheduling barriers keep the requests in front of
    // the MFMAs.  The A requests of group g + 2 stay in flight across the pin.
#define RN_STEP(cur, nxt, far, bcur, bnxt, g_)                                                                         \
    {                                                                                                                  \
        const int gf_ = (g_) + 2 < gl ? (g_) + 2 : gl, gn_ = (g_) + 1 < gl ? (g_) + 1 : gl;                            \
        RN_LOAD_A(far, gf_)                                                                                            \
        RN_LOAD_B(bnxt, gn_)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < KU; i++)                                                                 \
            _Pragma("unroll") for (int j = 0; j < TG; j++)                                                             \
                _Pragma("unroll") for (int c = 0; c < CT; c++) acc[j][c] = TR ? Mfma16<T>::run(bcur[i][c], cur[i][j], acc[j][c]) : Mfma16<T>::run(cur[i][j], bcur[i][c], acc[j][c]); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < KU; i++) {                                                               \
            _Pragma("unroll") for (int j = 0; j < TG; j++) asm volatile("" ::"v"(nxt[i][j]));                          \
            _Pragma("unroll") for (int c = 0; c < CT; c++) asm volatile("" ::"v"(bnxt[i][c]));                         \
        }                                                                                                              \
    }
    const int gl = G - 1;
    RN_LOAD_A(a0, 0)
    RN_LOAD_A(a1, (1 < gl ? 1 : gl))
    RN_LOAD_B(b0, 0)
    int g = 0;
    for (; g + 3 <= G; g += 3) {
        RN_STEP(a0, a1, a2, b0, b1, g)
        RN_STEP(a1, a2, a0, b1, b2, g + 1)
        RN_STEP(a2, a0, a1, b2, b0, g + 2)
    }
    if (g < G) RN_STEP(a0, a1, a2, b0, b1, g)
    if (g + 1 < G) RN_STEP(a1, a2, a0, b1, b2, g + 1)
#undef RN_STEP
#undef RN_LOAD_A
#undef RN_LOAD_B
}
// acc[j] = M[tile t0 + j*ts] * slab  for j < TG (tiles past `tiles` recompute tile t0; the caller drops them).
// PIPE: the software-pipelined loop above (220 VGPRs: for launches with at most one workgroup per CU, where nothing else hides
// the operand latency -- small and sharded trees); otherwise the lean loop (request a group, multiply it; 100 VGPRs), which
// leaves the latency hiding to the three workgroups that share a CU when there are more slabs than CUs.
template <typename T, int TG, int KU, bool PIPE, bool TR = false>
__device__ __forceinline__ void slab_mfma(typename Mfma16<T>::acc_t (&acc)[TG], const T *M, int mp, int t0, int ts, int tiles, int ksteps,
                                          const T *sB, int SB, int lane, const T *Mf = nullptr) {
    typedef typename Mfma16<T>::acc_t acc_t;
    const int col = lane & 15, kq = lane >> 4;
    const T *Ap[TG];
    const T *Bp = sB + col * SB + kq;
    if (PIPE) {
        acc_t a2[TG][1];
#pragma unroll
        for (int j = 0; j < TG; j++) {
            const int t = t0 + ts * j;
            Ap[j] = Mf ? Mf + ((size_t)(t < tiles ? t : t0) * (ksteps / 2) * 64 + lane) * 2 : M + (size_t)(t < tiles ? t : t0) * 16 + col + (size_t)kq * mp;
            a2[j][0] = acc_t{0, 0, 0, 0};
        }
        if (Mf) slab_mfma_pipe<T, TG, 1, KU, true, TR>(a2, Ap, 0, Bp, 0, ksteps / KU);
        else
        slab_mfma_pipe<T, TG, 1, KU, false, TR>(a2, Ap, (size_t)4 * mp, Bp, 0, ksteps / KU);
#pragma unroll
        for (int j = 0; j < TG; j++) acc[j] = a2[j][0];
        return;
    }
#pragma unroll
    for (int j = 0; j < TG; j++) {
        const int t = t0 + ts * j;
        Ap[j] = M + (size_t)(t < tiles ? t : t0) * 16 + col + (size_t)kq * mp;
        acc[j] = acc_t{0, 0, 0, 0};
    }
    if (Mf != nullptr) {                          // A operands from the fragment-ordered copy: one 16-byte request per lane and pair of k-steps
        typedef T frag2 __attribute__((ext_vector_type(2)));
        static_assert(KU % 2 == 0, "pairs of k-steps");
        const frag2 *Af[TG];
#pragma unroll
        for (int j = 0; j < TG; j++) {
            const int t = t0 + ts * j;
            Af[j] = reinterpret_cast<const frag2 *>(Mf) + (size_t)(t < tiles ? t : t0) * (ksteps / 2) * 64 + lane;
        }
        for (int ks = 0; ks < ksteps; ks += KU) {
            T av[KU][TG], bv[KU];
#pragma unroll
            for (int i = 0; i < KU; i += 2) {
#pragma unroll
                for (int j = 0; j < TG; j++) { const frag2 v = Af[j][(size_t)((ks + i) / 2) * 64]; av[i][j] = v[0]; av[i + 1][j] = v[1]; }
                bv[i] = Bp[(ks + i) * 4];
                bv[i + 1] = Bp[(ks + i + 1) * 4];
            }
#pragma unroll
            for (int i = 0; i < KU; i++)
#pragma unroll
                for (int j = 0; j < TG; j++) acc[j] = TR ? Mfma16<T>::run(bv[i], av[i][j], acc[j]) : Mfma16<T>::run(av[i][j], bv[i], acc[j]);
        }
        return;
    }
    for (int ks = 0; ks < ksteps; ks += KU) {     // ksteps is a whole number of groups (K padded by the host)
        T av[KU][TG], bv[KU];
#pragma unroll
        for (int i = 0; i < KU; i++) {
#pragma unroll
            for (int j = 0; j < TG; j++) av[i][j] = Ap[j][(size_t)(ks + i) * 4 * mp];
            bv[i] = Bp[(ks + i) * 4];
        }
#pragma unroll
        for (int i = 0; i < KU; i++)
#pragma unroll
            for (int j = 0; j < TG; j++) acc[j] = TR ? Mfma16<T>::run(bv[i], av[i][j], acc[j]) : Mfma16<T>::run(av[i][j], bv[i], acc[j]);
    }
}
// auxiliary operands of the epilogue (m1_i or e_i), requested BEFORE the MFMA loop so that their latency hides behind it
template <typename T, int EPI, int TG>
__device__ __forceinline__ void slab_aux(T (&auxv)[TG][4], T (&scale)[4], const GemmArgs<T> &g, int t0, int ts, int node0, int lane) {
    if (RN_SLAB_TR(EPI)) {
        // accumulator entry `reg` of tile j: node node0 + row(lane, reg), operator row 16 (t0 + ts j) + (lane & 15)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int node = node0 + Mfma16<T>::row(lane, reg);
            const int nodeC = node < g.nodes ? node : g.nodes - 1;
            scale[reg] = (EPI == EPI_V) ? (T)(-0.5) / g.prob[nodeC] : (T)0;
#pragma unroll
            for (int j = 0; j < TG; j++) {
                const int gr = (t0 + ts * j) * 16 + (lane & 15);
                auxv[j][reg] = (EPI != EPI_LV) ? gemm_aux<T, EPI>(g, nodeC, gr < g.m ? gr : g.m - 1) : (T)0;
            }
        }
        return;
    }
    const int node = node0 + (lane & 15);
    const int nodeC = node < g.nodes ? node : g.nodes - 1;
#pragma unroll
    for (int reg = 0; reg < 4; reg++) scale[reg] = (EPI == EPI_V) ? (T)(-0.5) / g.prob[nodeC] : (T)0;
#pragma unroll
    for (int j = 0; j < TG; j++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int gr = (t0 + ts * j) * 16 + Mfma16<T>::row(lane, reg);
            auxv[j][reg] = (EPI != EPI_LV) ? gemm_aux<T, EPI>(g, nodeC, gr < g.m ? gr : g.m - 1) : (T)0;
        }
}
// epilogue of one pass; sOut != nullptr also keeps the results in LDS ([16][SO], the B operand of a following product)
template <typename T, int EPI, int TG>
__device__ __forceinline__ void slab_store(const typename Mfma16<T>::acc_t (&acc)[TG], const T (&auxv)[TG][4], const T (&scale)[4], const GemmArgs<T> &g,
                                           int t0, int ts, int tiles, int node0, int lane, T *sOut, int SO) {
#pragma unroll
    for (int j = 0; j < TG; j++) {
        const int t = t0 + ts * j;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int ln = RN_SLAB_TR(EPI) ? Mfma16<T>::row(lane, reg) : (lane & 15);                    // node of the entry, within the slab
            const int gr = t * 16 + (RN_SLAB_TR(EPI) ? (lane & 15) : Mfma16<T>::row(lane, reg));         // operator row of the entry
            const int node = node0 + ln;
            const bool nodeOk = node < g.nodes;
            T r = acc[j][reg];
            if (EPI == EPI_V) r = auxv[j][reg] + scale[reg] * r;
            if (EPI == EPI_Z) r = auxv[j][reg] + r;
            const bool live = t < tiles && gr < g.m;
            if (g.out && live && nodeOk) g.out[(size_t)node * g.ldout + gr] = r;   // out == nullptr: the result only lives in sOut
            if (sOut && live) sOut[ln * SO + gr] = nodeOk ? r : (T)0;
        }
    }
}
// RN_KTIMING builds (tools/ktiming.py): phase stamps of a few workgroups, 100 MHz wall clock, read back by rn_debug_ktiming
#ifdef RN_KTIMING
__device__ unsigned long long g_ktiming[8 * 16];
#define RN_KT(slot) do { if (threadIdx.x == 0) { const int b_ = blockIdx.x == gridDim.x - 1 ? 3 : (int)blockIdx.x; if (b_ < 4) g_ktiming[b_ * 16 + (slot)] = wall_clock64(); } } while (0)
#else
#define RN_KT(slot) do { } while (0)
#endif
template <typename T, int EPI, int TG, int KU, bool PIPE>
__device__ __forceinline__ void slab_pass(const GemmArgs<T> &g, const T *sB, int SB, int node0, int t0, int nw, int tiles, int ksteps, int lane,
                                          T *sOut, int SO) {
    typename Mfma16<T>::acc_t acc[TG];
    T auxv[TG][4], scale[4];
    RN_KT(EPI == EPI_V ? 6 : 10);
    slab_aux<T, EPI, TG>(auxv, scale, g, t0, nw, node0, lane);
    RN_KT(EPI == EPI_V ? 7 : 11);
    slab_mfma<T, TG, KU, PIPE, RN_SLAB_TR(EPI)>(acc, g.M, g.mp, t0, nw, tiles, ksteps, sB, SB, lane, g.Mf);
    RN_KT(EPI == EPI_V ? 8 : 12);
    slab_store<T, EPI, TG>(acc, auxv, scale, g, t0, nw, tiles, node0, lane, sOut, SO);
    RN_KT(EPI == EPI_V ? 9 : 13);
}
template <typename T, int EPI, bool PIPE>
__device__ __forceinline__ void slab_product(const GemmArgs<T> &g, const T *sB, int SB, int node0, int wave, int nw, int lane, T *sOut, int SO) {
    const int tiles = (g.m + 15) / 16, ksteps = g.kp / 4;
    const int per = (tiles + nw - 1) / nw;               // tiles per wave
    const int tg = per >= 3 ? 3 : per;
    // when the tile count is not a multiple of the wave count some waves (= SIMDs) carry one tile more: rotate the
    // assignment from workgroup to workgroup so that the workgroups sharing a CU do not load the same SIMDs
#if RN_SLAB_ROTATE
    const int b = blockIdx.x;
    const int owner = (wave + b + (b >> 3) + (b >> 8)) % nw;
#else
    const int owner = wave;
#endif
    constexpr int KU = RN_SLAB_KU;
    for (int t0 = owner; t0 < tiles; t0 += nw * tg) {
        // tiles this wave really has in this pass (a wave whose last tile would lie past the end does not multiply a dummy: on a
        // 12-tile operator and 8 waves that dummy was a third of the SIMDs' MFMA time)
        const int have = (tiles - t0 + nw - 1) / nw;
        const int now = have < tg ? have : tg;
        if (now == 3) slab_pass<T, EPI, 3, KU, PIPE>(g, sB, SB, node0, t0, nw, tiles, ksteps, lane, sOut, SO);
        else if (now == 2) slab_pass<T, EPI, 2, KU, PIPE>(g, sB, SB, node0, t0, nw, tiles, ksteps, lane, sOut, SO);
        else slab_pass<T, EPI, 1, KU, PIPE>(g, sB, SB, node0, t0, nw, tiles, ksteps, lane, sOut, SO);
    }
}
template <typename T, int EPI, bool PIPE>
__global__ void __launch_bounds__(64 * SLAB_MAX_WAVES) k_gemm_slab(GemmArgs<T> g, int SB) {
    extern __shared__ unsigned char gemm_smem[];
    T *sB = reinterpret_cast<T *>(gemm_smem);   // [16][SB]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int node0 = blockIdx.x * 16;
    const int cnt = g.nodes - node0 < 16 ? g.nodes - node0 : 16;
    slab_load<T>(sB, SB, g.in, g.ldin, g.k, g.kp, node0, cnt, wave, nw, lane);
    __syncthreads();
    slab_product<T, EPI, PIPE>(g, sB, SB, node0, wave, nw, lane, nullptr, 0);
}
// Structured operator mode, first product of the sweep: m2_i = [Bbt | L'] [a_i; b_i] with a_i = F_i' xi_i, b_i = G_i' psi_i
// (F_i, G_i diagonal).  The slab of [a; b] is built in LDS straight from the duals (what k_struct_prep + a slab load
// would do in two launches and one HBM round trip); a_i is also written out (the q recursion of k_up_chain needs it).
template <typename T, bool PIPE>
__global__ void __launch_bounds__(64 * SLAB_MAX_WAVES) k_gemm_prep_m2(GemmArgs<T> g, SweepArgs<T> a, int SB) {
    extern __shared__ unsigned char gemm_smem[];
    T *sB = reinterpret_cast<T *>(gemm_smem);   // [16][SB]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int node0 = blockIdx.x * 16;
    const int cnt = g.nodes - node0 < 16 ? g.nodes - node0 : 16;
    const int nx = a.nx, ny = a.ny, k = g.k;     // k = nx + nu
    for (int r = wave; r < 16; r += nw) {         // one slab row (node) per wave and pass
        const int node = node0 + (r < cnt ? r : 0);
        const T *dy = a.tr.dy + (size_t)a.tr.stageOf[node] * ny;
        const size_t y = (size_t)node * ny;
        const T sp = a.tr.sqrtp[node];
        for (int t = lane; t < SB; t += 64) {
            T val = 0;
            if (r < cnt && t < k) {
                if (t < nx) { val = sp * (dy[t] * sweep_w(a, y + t) + dy[nx + t] * sweep_w(a, y + nx + t)); a.qa[(size_t)node * nx + t] = val; }
                else { const int j = t - nx; val = sp * dy[2 * nx + j] * sweep_w(a, y + 2 * nx + j); }
            }
            sB[r * SB + t] = val;
        }
    }
    __syncthreads();
    slab_product<T, EPI_LV, PIPE>(g, sB, SB, node0, wave, nw, lane, nullptr, 0);
}
// v_i = m1_i - RT [s_i; kappa_i] / (2 p_i)  and  [L v_i ; B L v_i]  in ONE launch: the v tile stays in LDS as the B operand
// of the second product (gL.in is ignored; gL.k must equal gV.m)
// foldRoot: the leaf-to-root recursion of the ROOT node (stage 0: its children sums) is done here by workgroup 0, which
// owns the root's slab, instead of in a launch of its own -- the other workgroups do not wait for it.
template <typename T, bool PIPE>
__global__ void __launch_bounds__(64 * SLAB_MAX_WAVES) k_gemm_vlv(GemmArgs<T> gV, GemmArgs<T> gL, int SB, int SV, SweepArgs<T> a, int foldRoot, int crownScratch) {
    extern __shared__ unsigned char gemm_smem[];
    T *sB = reinterpret_cast<T *>(gemm_smem);   // [16][SB] slab of [s; kappa]
    T *sV = sB + 16 * SB;                        // [16][SV] v of the slab, zero beyond gV.m
    bool slabReady = false;                      // workgroup 0, foldRoot = 2: its slab has been filled in LDS by the crown step
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // foldRoot = 2 (sharded, crown = root + the exchange stage): the exchange stage's step is just "beta + all-reduced
    // children sums" per node (presummed); every workgroup does it for the stage-1 nodes of its own slab, and workgroup 0
    // derives the root's step from the same inputs (up_root_from_presummed).
    RN_KT(0);
    if (foldRoot == 2) {
        const int s1 = a.s1, e1 = a.e1;
        const int lo = s1 > (int)blockIdx.x * 16 ? s1 : (int)blockIdx.x * 16;
        const int hi = e1 < (int)blockIdx.x * 16 + 16 ? e1 : (int)blockIdx.x * 16 + 16;
        // workgroup 0 does both steps in one batch of loads when the children's values fit in the (still unused) slab
        // buffers and it has a thread per component; every dependent batch costs 1-2.5 us right after the streaming kernel
        // has swept the caches and TLBs
        // (RN_CROWN2_FALLBACK builds force the two-function path below, which otherwise only very wide crowns take)
        if (!RN_CROWN2_FALLBACK && blockIdx.x == 0 && crownScratch > 0 && a.e1 >= 16 && (int)blockDim.x >= a.nv + a.nx) {
            // all 16 nodes of this workgroup's slab are crown nodes: their [s; kappa] columns go straight into the slab buffer (the
            // children's values through a scratch area BEHIND the slab buffers, sized by the host), so the launch's critical
            // workgroup neither drains its stores nor reads its slab back from global memory
            up_crown2_wg0<T>(a, sB + 16 * (SB + SV), threadIdx.x, blockDim.x, sB, SB);
            {   // the crown step wrote the nv + nx live columns of all 16 rows; the padding columns are zeroed here (disjoint: no barrier)
                const int per = a.nv + a.nx, padc = SB - per;
                for (int i = threadIdx.x; i < 16 * padc; i += blockDim.x) sB[(i / padc) * SB + per + i % padc] = (T)0;
            }
            slabReady = true;
        } else if (!RN_CROWN2_FALLBACK && blockIdx.x == 0 && a.rootNc * (a.nv + 2 * a.nx) <= 16 * (SB + SV) && (int)blockDim.x >= a.nv + a.nx) {
            up_crown2_wg0<T>(a, sB, threadIdx.x, blockDim.x);
        } else {
            if (blockIdx.x == 0) up_root_from_presummed<T>(a, threadIdx.x, blockDim.x);
            if (lo < hi) up_crown_presummed_flat<T>(a, 1, lo, hi, threadIdx.x, blockDim.x);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.distTail != nullptr) {   // optimistic exchange: dist^2 of the previous iteration
            IterState *st = reinterpret_cast<IterState *>(a.iterState);
            const double dX = sqrt((double)a.distTail[0]), dS = sqrt((double)a.distTail[1]);
            st->distX = dX; st->distS = dS;
            if (dX > a.thrX || dS > a.thrS) st->violated = 1;
        }
        if ((lo < hi || blockIdx.x == 0) && !slabReady) __threadfence_block();   // the workgroup reads its own sk rows back below (same CU, same L1)
        __syncthreads();
    }
    RN_KT(1);
    if (foldRoot == 1 && blockIdx.x == 0) {
        up_crown_node<T>(a, 0, 0, threadIdx.x, blockDim.x);
        __threadfence_block();                   // same workgroup reads sk of node 0 back below (same CU, same L1)
        __syncthreads();
    }
    RN_KT(2);
    const int node0 = blockIdx.x * 16;
    const int cnt = gV.nodes - node0 < 16 ? gV.nodes - node0 : 16;
    if (!slabReady) slab_load<T>(sB, SB, gV.in, gV.ldin, gV.k, gV.kp, node0, cnt, wave, nw, lane);
    for (int i = threadIdx.x; i < 16 * SV; i += blockDim.x) sV[i] = (T)0;
    __syncthreads();
    RN_KT(3);
    slab_product<T, EPI_V, PIPE>(gV, sB, SB, node0, wave, nw, lane, sV, SV);
    __syncthreads();
    RN_KT(4);
    slab_product<T, EPI_LV, PIPE>(gL, sV, SV, node0, wave, nw, lane, nullptr, 0);
#ifdef RN_KTIMING
    __syncthreads();
    RN_KT(5);
#endif
}

// The same two products for trees with MORE slabs than the chip has CUs (the 493-scenario tree: 679 slabs on 256 CUs).  There
// k_gemm_vlv puts three workgroups on most CUs, and each of them streams the shared operators (RT: 124 KB, [L; BL]: 137 KB)
// from L2 through the CU's one vector-memory pipe: 890 KB of A fragments per CU and launch, as long a stream as the fp64 MFMAs of
// the three slabs themselves -- and the two overlap imperfectly (section 3 of DESIGN.md: 31 us against a 12.4 us matrix-pipe floor).
// Here ONE workgroup per CU owns CT consecutive slabs (CT * 16 nodes): a wave owns a 16-row tile of the operator over the full K
// and keeps CT accumulators, so every A fragment it loads feeds CT MFMAs (B fragments: LDS, as before) -- a third of the
// operand stream at CT = 3, with the next group of A fragments requested before the current group's MFMAs are issued.
// Every output element is the same chain of MFMAs over k as in k_gemm_vlv: bitwise the same results.
#ifndef RN_WIDE_KU
#define RN_WIDE_KU RN_SLAB_KU
#endif
#ifndef RN_WIDE_THREADS
#define RN_WIDE_THREADS 512
#endif
#ifndef RN_WIDE_LD
#define RN_WIDE_LD 18     // slab rows x 64-element chunks a wave requests at once (48 rows x 3 chunks over 8 waves: one round trip)
#endif
template <typename T, int CT, int KU, bool TR = false>
__device__ __forceinline__ void wide_mfma(typename Mfma16<T>::acc_t (&acc)[CT], const T *M, int mp, int t, int ksteps, const T *sB, int SB, int lane, const T *Mf = nullptr) {
    typedef typename Mfma16<T>::acc_t acc_t;
    const int col = lane & 15, kq = lane >> 4;
    const T *Ap[1] = {Mf ? Mf + ((size_t)t * (ksteps / 2) * 64 + lane) * 2 : M + (size_t)t * 16 + col + (size_t)kq * mp};
    acc_t a2[1][CT];
#pragma unroll
    for (int c = 0; c < CT; c++) a2[0][c] = acc_t{0, 0, 0, 0};
    if (Mf) slab_mfma_pipe<T, 1, CT, KU, true, TR>(a2, Ap, 0, sB + col * SB + kq, 16 * SB, ksteps / KU);
    else
    slab_mfma_pipe<T, 1, CT, KU, false, TR>(a2, Ap, (size_t)4 * mp, sB + col * SB + kq, 16 * SB, ksteps / KU);
#pragma unroll
    for (int c = 0; c < CT; c++) acc[c] = a2[0][c];
}
template <typename T, int EPI, int CT>
__device__ __forceinline__ void wide_product(const GemmArgs<T> &g, const T *sB, int SB, int node0, int wave, int nw, int lane, T *sOut, int SO) {
    typedef typename Mfma16<T>::acc_t acc_t;
    const int tiles = (g.m + 15) / 16, ksteps = g.kp / 4;
    for (int t = wave; t < tiles; t += nw) {
        acc_t acc[CT];
        T auxv[CT][4], scale[CT][4];
        RN_KT(EPI == EPI_V ? 6 : 10);
        // the epilogue's operands (m1_i) are requested before the MFMA loop: their latency hides behind it
        // (entry `reg` of slab c: node c * 16 + ln, operator row 16 t + lr -- which of the lane's two coordinates is which: RN_SLAB_T, see slab_store)
#pragma unroll
        for (int c = 0; c < CT; c++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int ln = RN_SLAB_TR(EPI) ? Mfma16<T>::row(lane, reg) : (lane & 15);
                const int gr = t * 16 + (RN_SLAB_TR(EPI) ? (lane & 15) : Mfma16<T>::row(lane, reg));
                const int node = node0 + c * 16 + ln;
                const int nodeC = node < g.nodes ? node : g.nodes - 1;
                scale[c][reg] = (EPI == EPI_V) ? (T)(-0.5) / g.prob[nodeC] : (T)0;
                auxv[c][reg] = (EPI != EPI_LV) ? gemm_aux<T, EPI>(g, nodeC, gr < g.m ? gr : g.m - 1) : (T)0;
            }
        }
        RN_KT(EPI == EPI_V ? 7 : 11);
        wide_mfma<T, CT, RN_WIDE_KU, RN_SLAB_TR(EPI)>(acc, g.M, g.mp, t, ksteps, sB, SB, lane, g.Mf);
        RN_KT(EPI == EPI_V ? 8 : 12);
#pragma unroll
        for (int c = 0; c < CT; c++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int ln = RN_SLAB_TR(EPI) ? Mfma16<T>::row(lane, reg) : (lane & 15);
                const int gr = t * 16 + (RN_SLAB_TR(EPI) ? (lane & 15) : Mfma16<T>::row(lane, reg));
                const int node = node0 + c * 16 + ln;
                const bool nodeOk = node < g.nodes;
                T r = acc[c][reg];
                if (EPI == EPI_V) r = auxv[c][reg] + scale[c][reg] * r;
                if (EPI == EPI_Z) r = auxv[c][reg] + r;
                const bool live = gr < g.m;
                if (g.out && live && nodeOk) g.out[(size_t)node * g.ldout + gr] = r;
                if (sOut && live) sOut[(c * 16 + ln) * SO + gr] = nodeOk ? r : (T)0;
            }
        }
        RN_KT(EPI == EPI_V ? 9 : 13);
    }
}
template <typename T, int CT>
__global__ void __launch_bounds__(RN_WIDE_THREADS) k_gemm_vlv_wide(GemmArgs<T> gV, GemmArgs<T> gL, int SB, int SV, SweepArgs<T> a, int foldRoot) {
    extern __shared__ unsigned char gemm_smem[];
    T *sB = reinterpret_cast<T *>(gemm_smem);   // [CT * 16][SB] slabs of [s; kappa]
    T *sV = sB + CT * 16 * SB;                   // [CT * 16][SV] v of the slabs, zero beyond gV.m
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    RN_KT(0);
    RN_KT(1);
    if (foldRoot == 1 && blockIdx.x == 0) {     // the root's leaf-to-root step (see k_gemm_vlv); foldRoot = 2 never comes here
        up_crown_node<T>(a, 0, 0, threadIdx.x, blockDim.x);
        __threadfence_block();
        __syncthreads();
    }
    RN_KT(2);
    const int node0 = blockIdx.x * 16 * CT;
    const int cnt = gV.nodes - node0 < 16 * CT ? gV.nodes - node0 : 16 * CT;
    slab_load<T, RN_WIDE_LD>(sB, SB, gV.in, gV.ldin, gV.k, gV.kp, node0, cnt, wave, nw, lane, 16 * CT);
    for (int i = threadIdx.x; i < CT * 16 * SV; i += blockDim.x) sV[i] = (T)0;
    __syncthreads();
    RN_KT(3);
    wide_product<T, EPI_V, CT>(gV, sB, SB, node0, wave, nw, lane, sV, SV);
    __syncthreads();
    RN_KT(4);
    wide_product<T, EPI_LV, CT>(gL, sV, SV, node0, wave, nw, lane, nullptr, 0);
#ifdef RN_KTIMING
    __syncthreads();
    RN_KT(5);
#endif
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
    const bool crownWriter = (int)blockIdx.x >= a.K;
    const int s = crownWriter ? 0 : (int)blockIdx.x;
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nu + nx;
    const int top = a.chainStage;
    const int ntop = a.tr.stageCum[top] + s;
    const size_t nodeTop = (size_t)ntop;   // every stage >= c* has K nodes: node of stage k in this chain = nodeTop + (k - c*) K
    const int par = a.tr.parent[ntop];
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
        int n = ntop;
        bool first = true;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) {
            if (dd < top) {
                const int p = a.tr.parent[n];
                first = first && (a.tr.childStart[p] == n);
                anc[dd] = p; writer[dd] = first && foldCrown == 1;
                n = p;
            } else { anc[dd] = 0; writer[dd] = false; }
        }
    }
    for (int t = crownWriter ? w : (int)threadIdx.x; t < w; t += CHAIN_THREADS) {
        if (t < nu) {
            T run;
            if (foldCrown) {
                run = a.prevU[t] - a.prevUhat[t];
                T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + t]; uh[dd] = uhat[(size_t)anc[dd] * nu + t]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd < top) {
                        const int k = top - 1 - dd;                       // stage of anc[dd]
                        const T uv = uh[dd] + run + lv[dd];               // same association as down_crown_node
                        run = uv - uh[dd];                                 // what a child reads back: u_par - uhat_par
                        if (writer[dd]) {
                            const T spc = a.tr.sqrtp[anc[dd]];
                            if (a.writePrimal) a.u[(size_t)anc[dd] * nu + t] = uv;
                            a.hx[(size_t)anc[dd] * ny + 2 * nx + t] = UNSC ? uv : spc * dyAll[(size_t)k * ny + 2 * nx + t] * uv;
                        }
                    }
            } else run = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : (a.u[(size_t)par * nu + t] - a.uhat[(size_t)par * nu + t]);
            for (int k = top; k < a.N; k += PF) {
                T dv[PF], uh[PF], d0[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + t];
                    uh[j] = uhat[node * nu + t];
                    d0[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + 2 * nx + t];
                }
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
            }
        } else {
            const int j0 = t - nu;
            T bw, xr;
            if (foldCrown) {
                bw = a.bw0[j0]; xr = a.curX[j0];
                T lv[CROWN_MAX_DEPTH], ev[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + nu + j0]; ev[dd] = eb[(size_t)anc[dd] * nx + j0]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd < top) {
                        const int k = top - 1 - dd;
                        bw = bw + lv[dd];
                        xr = xr + ev[dd] + bw;
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
                T dv[PF], ev[PF], d0[PF], d1[PF];
#pragma unroll
                for (int j = 0; j < PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + nu + j0];
                    ev[j] = eb[node * nx + j0];
                    d0[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + j0];
                    d1[j] = UNSC ? (T)0 : dyAll[(size_t)kk * ny + nx + j0];
                }
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
                    T run = a.prevU[t] - a.prevUhat[t];
                    T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
                    for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                        if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + t]; uh[dd] = uhat[(size_t)pth[dd] * nu + t]; }
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
                        if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + nu + j0]; ev[dd] = eb[(size_t)pth[dd] * nx + j0]; }
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
    for (int t = tid; t < w; t += nthreads) {
        if (t < nu) {
            const T wanc = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : (a.u[(size_t)par * nu + t] - a.uhat[(size_t)par * nu + t]);
            const T uv = a.uhat[(size_t)node * nu + t] + wanc + a.lvb[(size_t)node * w + t];
            a.u[(size_t)node * nu + t] = uv;
            a.hx[(size_t)node * ny + 2 * nx + t] = sp * dy[2 * nx + t] * uv;
        } else {
            const int j0 = t - nu;
            const T bw = (par < 0 ? a.bw0[j0] : a.bw[(size_t)par * nx + j0]) + a.lvb[(size_t)node * w + nu + j0];
            const T xv = (par < 0 ? a.curX[j0] : a.x[(size_t)par * nx + j0]) + a.eb[(size_t)node * nx + j0] + bw;
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
    // k_dual_stage with LAZY: `w` points at y_{t-1} (the buffer ynew overwrites, element by element, after reading it) and the
    // accelerated dual of this iteration is formed on the fly, w_t = (1 + lnCur) yprev - lnCur y_{t-1} (extrap_elem: the bits a
    // stored w_t would have); wview (last iteration of a batch only): where w_t is stored for the getters
    T lnCur; T *wview;
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
// LAZY = 0: w is read, y+ and w_next are stored.  Device-resident batches of more than one iteration keep the accelerated dual
// out of memory between their iterations -- the next sweep and the next dual update derive it from the two iterates (4 streams
// instead of 5, and 21 MB of dirty lines less in front of the streaming kernel): LAZY = 3 (first iteration): w is read, w_next not
// stored; LAZY = 1 (inner iterations): w derived, w_next not stored; LAZY = 2 (last iteration): w derived, and both w_t (wview)
// and w_next stored, so that the state a caller can observe is what it always was.
template <typename T, bool MATERIALIZE, int LAZY, bool SCALE = false>
__device__ __forceinline__ void dual_slot_use(const DualSlot<T> &s, const DualArgs<T> &a, T ln, DualAcc<T> &r) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    if (!s.on) return;
    VT yn, wn, z, res, wcur;
    const long long i0 = s.iv * VN;
    const bool counted = a.countCrown || i0 >= a.crownElems;
#pragma unroll
    for (int e = 0; e < VN; e++) {
        const int c = s.c + e;
        const bool isBox = c < a.nx, isXi = c < 2 * a.nx;
        const T k = s.sp * s.dy[e];
        const T lo = k * s.blo[e];
        const T hi = (isXi && !isBox) ? s.bhi[e] : k * s.bhi[e];
        wcur[e] = (LAZY == 1 || LAZY == 2) ? extrap_elem(s.yp[e], s.w[e], a.lnCur) : s.w[e];
        // SCALE: k_down_chain<T, true> left the primal values (every node)
        const T hxv = SCALE ? k * s.hx[e] : s.hx[e];
        const DualOut<T> o = dual_elem<T, false>(hxv, wcur[e], lo, hi, s.yp[e], a.lambda, a.invLambda, ln, (T)0);
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
    if (LAZY == 0 || LAZY == 2) reinterpret_cast<VT *>(a.wnext)[s.iv] = wn;
    if (LAZY == 2) reinterpret_cast<VT *>(a.wview)[s.iv] = wcur;
    if (MATERIALIZE) { reinterpret_cast<VT *>(a.z)[s.iv] = z; reinterpret_cast<VT *>(a.res)[s.iv] = res; }
#endif
    (void)counted;
}
// PIPE = 1: one vector at a time;  PIPE = 2: double-buffered -- the loads of trip t+1 are requested before trip t is consumed, so
// a wave always has a trip in flight (the kernel lives on memory-level parallelism: its VALU phase is a gap in the streams)
template <typename T, bool MATERIALIZE, int PIPE, int LAZY, bool SCALE = false>
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
            dual_slot_use<T, MATERIALIZE, LAZY, SCALE>(s, a, ln, r);
        }
    } else {
        DualSlot<T> sA, sB;
        dual_slot_load<T>(sA, a, g, 0, jbase, cnt, nodeFirst, stageU, crownBlock);
        for (int t = 0; t < g.trips; t += 2) {
            const bool hasB = t + 1 < g.trips;
            if (hasB) dual_slot_load<T>(sB, a, g, t + 1, jbase, cnt, nodeFirst, stageU, crownBlock);
            dual_slot_use<T, MATERIALIZE, LAZY, SCALE>(sA, a, ln, r);
            if (t + 2 < g.trips) dual_slot_load<T>(sA, a, g, t + 2, jbase, cnt, nodeFirst, stageU, crownBlock);
            if (hasB) dual_slot_use<T, MATERIALIZE, LAZY, SCALE>(sB, a, ln, r);
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

// ------------------------------------------------------------------------------------------------------
// The forward walk AND the dual update of the nodes it has just walked, in one launch (round 5, opt-in: RAPIDNET_FUSE_DOWN_DUAL).
// k_down_chain produces Hx of a chain's nodes (and of the crown nodes it writes); the fused dual update of exactly those elements needs
// nothing else from the sweep, so the same workgroup can do it: phase A is k_down_chain with Hx kept in LDS ([rows][ny]; global memory
// only when the primal iterates are stored), phase B walks the rows' 16-byte vectors like a k_dual_stage tile (dual_elem: the same
// arithmetic element by element).  One dependent launch and the 42 MB round trip of Hx less per iteration.  The arg-max keeps the
// reference's tie rule by comparing indices on equal magnitudes (a thread does not meet its elements in ascending order here).
template <typename T, bool MATERIALIZE>
__global__ void __launch_bounds__(CHAIN_THREADS) k_down_chain_dual(SweepArgs<T> a, int foldCrown, DualArgs<T> da, double lnNext) {
    typedef typename VecOf<T>::type VT;
    constexpr int VN = VecOf<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *shx = reinterpret_cast<T *>(smem_raw);      // [L + top][ny]: rows 0 .. L-1 the chain's nodes (stage top + r), rows L + dd the crown nodes this workgroup writes
    __shared__ Partial sh_p[CHAIN_THREADS / 64];
    __shared__ int sh_rowNode[CROWN_MAX_DEPTH];     // crown rows: node (or -1)
    const bool crownWriter = (int)blockIdx.x >= a.K;
    const int s = crownWriter ? 0 : (int)blockIdx.x;
    const int nx = a.nx, nu = a.nu, ny = a.ny, w = nu + nx;
    const int top = a.chainStage, L = a.N - top;
    const int ntop = a.tr.stageCum[top] + s;
    const size_t nodeTop = (size_t)ntop;
    const T sp = a.tr.sqrtp[ntop];
    const T *__restrict__ lvb = a.lvb;
    const T *__restrict__ uhat = a.uhat;
    const T *__restrict__ eb = a.eb;
    const T *__restrict__ dyAll = a.tr.dy;
    int anc[CROWN_MAX_DEPTH];
    bool writer[CROWN_MAX_DEPTH];
    {
        int n = ntop;
        bool first = true;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) {
            if (dd < top) {
                const int p = a.tr.parent[n];
                first = first && (a.tr.childStart[p] == n);
                anc[dd] = p; writer[dd] = first && foldCrown == 1 && !crownWriter;
                n = p;
            } else { anc[dd] = 0; writer[dd] = false; }
        }
    }
    if (threadIdx.x < CROWN_MAX_DEPTH) {
        int nd = -1;
#pragma unroll
        for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++) if ((int)threadIdx.x == dd && writer[dd]) nd = anc[dd];
        sh_rowNode[threadIdx.x] = nd;
    }
    // ---- phase A: the walk (k_down_chain, foldCrown 1 / 2), Hx into LDS
    auto put = [&](int row, size_t node, int c, T val) {
        shx[(size_t)row * ny + c] = val;
        if (a.writePrimal) a.hx[node * ny + c] = val;
    };
    for (int t = crownWriter ? w : (int)threadIdx.x; t < w; t += CHAIN_THREADS) {
        if (t < nu) {
            T run = a.prevU[t] - a.prevUhat[t];
            T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
            for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + t]; uh[dd] = uhat[(size_t)anc[dd] * nu + t]; }
#pragma unroll
            for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                if (dd < top) {
                    const int k = top - 1 - dd;
                    const T uv = uh[dd] + run + lv[dd];
                    run = uv - uh[dd];
                    if (writer[dd]) {
                        const T spc = a.tr.sqrtp[anc[dd]];
                        if (a.writePrimal) a.u[(size_t)anc[dd] * nu + t] = uv;
                        put(L + dd, (size_t)anc[dd], 2 * nx + t, spc * dyAll[(size_t)k * ny + 2 * nx + t] * uv);
                    }
                }
            for (int k = top; k < a.N; k += CHAIN_PF) {
                T dv[CHAIN_PF], uh2[CHAIN_PF], d0[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + t];
                    uh2[j] = uhat[node * nu + t];
                    d0[j] = dyAll[(size_t)kk * ny + 2 * nx + t];
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k + j < a.N) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        run += dv[j];
                        const T uv = uh2[j] + run;
                        if (a.writePrimal) a.u[node * nu + t] = uv;
                        put(k + j - top, node, 2 * nx + t, sp * d0[j] * uv);
                    }
                }
            }
        } else {
            const int j0 = t - nu;
            T bw = a.bw0[j0], xr = a.curX[j0];
            T lv[CROWN_MAX_DEPTH], ev0[CROWN_MAX_DEPTH];
#pragma unroll
            for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                if (dd < top) { lv[dd] = lvb[(size_t)anc[dd] * w + nu + j0]; ev0[dd] = eb[(size_t)anc[dd] * nx + j0]; }
#pragma unroll
            for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                if (dd < top) {
                    const int k = top - 1 - dd;
                    bw = bw + lv[dd];
                    xr = xr + ev0[dd] + bw;
                    if (writer[dd]) {
                        const T spc = a.tr.sqrtp[anc[dd]];
                        a.bw[(size_t)anc[dd] * nx + j0] = bw;
                        if (a.writePrimal) a.x[(size_t)anc[dd] * nx + j0] = xr;
                        put(L + dd, (size_t)anc[dd], j0, spc * dyAll[(size_t)k * ny + j0] * xr);
                        put(L + dd, (size_t)anc[dd], nx + j0, spc * dyAll[(size_t)k * ny + nx + j0] * xr);
                    }
                }
            for (int k = top; k < a.N; k += CHAIN_PF) {
                T dv[CHAIN_PF], ev[CHAIN_PF], d0[CHAIN_PF], d1[CHAIN_PF];
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    const int kk = k + j < a.N ? k + j : a.N - 1;
                    const size_t node = nodeTop + (size_t)(kk - top) * a.K;
                    dv[j] = lvb[node * w + nu + j0];
                    ev[j] = eb[node * nx + j0];
                    d0[j] = dyAll[(size_t)kk * ny + j0];
                    d1[j] = dyAll[(size_t)kk * ny + nx + j0];
                }
#pragma unroll
                for (int j = 0; j < CHAIN_PF; j++) {
                    if (k + j < a.N) {
                        const size_t node = nodeTop + (size_t)(k + j - top) * a.K;
                        bw += dv[j];
                        xr += ev[j] + bw;
                        if (a.writePrimal) a.x[node * nx + j0] = xr;
                        put(k + j - top, node, j0, sp * d0[j] * xr);
                        put(k + j - top, node, nx + j0, sp * d1[j] * xr);
                    }
                }
            }
        }
    }
    int cwNode = -1, cwStage = 0;
    if (crownWriter) {      // sharded runs (foldCrown = 2): this workgroup writes crown node j (root -> j walk), row 0
        const int j = (int)blockIdx.x - a.K;
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
                T run = a.prevU[t] - a.prevUhat[t];
                T lv[CROWN_MAX_DEPTH], uh[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + t]; uh[dd] = uhat[(size_t)pth[dd] * nu + t]; }
                T uv = 0;
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd <= kj) { uv = uh[dd] + run + lv[dd]; run = uv - uh[dd]; }
                if (a.writePrimal) a.u[(size_t)j * nu + t] = uv;
                put(0, (size_t)j, 2 * nx + t, spj * dyAll[(size_t)kj * ny + 2 * nx + t] * uv);
            } else {
                const int j0 = t - nu;
                T bw = a.bw0[j0], xr = a.curX[j0];
                T lv[CROWN_MAX_DEPTH], ev[CROWN_MAX_DEPTH];
#pragma unroll
                for (int dd = 0; dd < CROWN_MAX_DEPTH; dd++)
                    if (dd <= kj) { lv[dd] = lvb[(size_t)pth[dd] * w + nu + j0]; ev[dd] = eb[(size_t)pth[dd] * nx + j0]; }
#pragma unroll
                for (int dd = CROWN_MAX_DEPTH - 1; dd >= 0; dd--)
                    if (dd <= kj) { bw = bw + lv[dd]; xr = xr + ev[dd] + bw; }
                a.bw[(size_t)j * nx + j0] = bw;
                if (a.writePrimal) a.x[(size_t)j * nx + j0] = xr;
                put(0, (size_t)j, j0, spj * dyAll[(size_t)kj * ny + j0] * xr);
                put(0, (size_t)j, nx + j0, spj * dyAll[(size_t)kj * ny + nx + j0] * xr);
            }
        }
    }
    __syncthreads();
    // ---- phase B: the dual update of the rows' elements (dual_slot_use's arithmetic; LAZY = 0)
    const int vpn = ny / VN;
    const int nRows = crownWriter ? 1 : L + top;
    const T ln = (T)lnNext;
    DualAcc<T> r;
    constexpr int U = 3;
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
            const int row = in ? v / vpn : 0, j = in ? v - row * vpn : 0;
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
            if (MATERIALIZE) { reinterpret_cast<VT *>(da.z)[ivv[u]] = z; reinterpret_cast<VT *>(da.res)[ivv[u]] = res; }
        }
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

// one workgroup: fold the block partials; decide whether the soft-constraint branch trips
// (dist > gamma/lambda, SmpcController.cu:793, :811)
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
__global__ void __launch_bounds__(ELT_THREADS) k_batch_close_unpack(const double *in, double *hist, int first, int n, IterState *st) {
    if (threadIdx.x == 0 && in[1 + 4 * (size_t)n] > 0.0) st->commFail = 1;      // a reader gave up on SOME rank: the batch is invalid on every rank
    for (int i = threadIdx.x; i < n; i += ELT_THREADS) {
        const double gx = in[1 + 4 * i] >= in[2 + 4 * i] ? in[1 + 4 * i] : -in[2 + 4 * i];
        const double gp = in[3 + 4 * i] >= in[4 + 4 * i] ? in[3 + 4 * i] : -in[4 + 4 * i];
        hist[first + i] = gx > gp ? gx : gp;
    }
}

// multi-GPU variant of k_decide: fold the local partials to (d2x, d2s), all-reduce those two numbers, then decide
__global__ void __launch_bounds__(ELT_THREADS) k_reduce_dist(const Partial *partials, int nblocks, double *out2) {
    double tx2 = 0, ts2 = 0;
    Partial p;
    fold_partials(partials, nblocks, false, tx2, ts2, p);
    if (threadIdx.x == 0) { out2[0] = tx2; out2[1] = ts2; }
}
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

// ------------------------------------------------------------------------------------------------------
// HBM ceiling probes for bench.py / tools/probe_hbm.py (rn_measure_hbm*): what kernels that do nothing but stream reach
// on THIS box -- the practical denominator next to the 8 TB/s spec.  16 B per lane per load, non-temporal.
//   k_bw_read        flat: the whole grid sweeps one region, thread-interleaved (grid-stride)
//   k_bw_read_chunks the streaming kernel's shape: one workgroup per contiguous chunk ("node"), 32 B per lane per step
//   k_bw_read_lockstep  persistent workgroups; in step s of batch b workgroup w reads piece ((b*S + s)*W + w): all
//                    workgroups together sweep one contiguous W*P-byte window per step
#ifndef RN_PROBE_PAIRS
#define RN_PROBE_PAIRS 0
#endif
__global__ void __launch_bounds__(256) k_bw_read(const nat_d2 *src, long long n, double *sink) {
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const nat_d2 v = __builtin_nontemporal_load(src + i);
        acc += v[0] + v[1];
    }
    if (acc == 1.2345e-300) sink[blockIdx.x & 65535] = acc;   // keeps the loads alive without a store stream
}
template <int UNR>
__global__ void __launch_bounds__(256) k_bw_read_chunks(const nat_d2 *src, long long chunk16, long long stride16, double *sink) {
    const nat_d2 *p = src + (long long)blockIdx.x * stride16;
    double acc = 0;
    long long i = threadIdx.x;
#if RN_PROBE_PAIRS   // lane reads 32 adjacent bytes as two loads (each wave-load touches 16 lines, half of every line)
    i = 2 * threadIdx.x;
    for (; i + 512 * (UNR - 1) + 1 < chunk16; i += 512 * UNR) {
        nat_d2 v[UNR][2];
#pragma unroll
        for (int u = 0; u < UNR; u++) { v[u][0] = __builtin_nontemporal_load(p + i + 512 * u); v[u][1] = __builtin_nontemporal_load(p + i + 512 * u + 1); }
#pragma unroll
        for (int u = 0; u < UNR; u++) acc += v[u][0][0] + v[u][0][1] + v[u][1][0] + v[u][1][1];
    }
    for (; i + 1 < chunk16; i += 512) { const nat_d2 a0 = __builtin_nontemporal_load(p + i), a1 = __builtin_nontemporal_load(p + i + 1); acc += a0[0] + a0[1] + a1[0] + a1[1]; }
#else                // every wave-load is 1 KB contiguous (8 full lines); two loads per step, 4 KB apart
    for (; i + 256 * (2 * UNR - 1) < chunk16; i += 512 * UNR) {
        nat_d2 v[UNR][2];
#pragma unroll
        for (int u = 0; u < UNR; u++) { v[u][0] = __builtin_nontemporal_load(p + i + 512 * u); v[u][1] = __builtin_nontemporal_load(p + i + 512 * u + 256); }
#pragma unroll
        for (int u = 0; u < UNR; u++) acc += v[u][0][0] + v[u][0][1] + v[u][1][0] + v[u][1][1];
    }
    for (; i < chunk16; i += 256) { const nat_d2 a0 = __builtin_nontemporal_load(p + i); acc += a0[0] + a0[1]; }
#endif
    if (acc == 1.2345e-300) sink[blockIdx.x & 65535] = acc;
}
template <int UNR>
__global__ void __launch_bounds__(256) k_bw_read_lockstep(const nat_d2 *src, int piece16, int steps, int batches, double *sink) {
    const long long W = gridDim.x;
    double acc = 0;
    const bool on = 2 * (int)threadIdx.x + 1 < piece16;
    for (int b = 0; b < batches; b++) {
        for (int s0 = 0; s0 < steps; s0 += UNR) {
            nat_d2 v[UNR][2];
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const int s = s0 + u < steps ? s0 + u : steps - 1;
                const nat_d2 *p = src + (((long long)b * steps + s) * W + blockIdx.x) * piece16 + 2 * threadIdx.x;
                if (on) { v[u][0] = __builtin_nontemporal_load(p); v[u][1] = __builtin_nontemporal_load(p + 1); }
                else { v[u][0] = nat_d2{0, 0}; v[u][1] = nat_d2{0, 0}; }
            }
#pragma unroll
            for (int u = 0; u < UNR; u++) acc += v[u][0][0] + v[u][0][1] + v[u][1][0] + v[u][1][1];
        }
    }
    if (acc == 1.2345e-300) sink[blockIdx.x & 65535] = acc;
}
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
