// chain_kernels.hpp -- the CHAIN-FUSED form of the tree sweep's helper path (round 5).
//
// The six-launch form (kernels.hpp) walks the K chains of the tree twice -- k_up_chain before the shared-operator
// products, k_down_chain behind them -- and runs the products of ALL nodes in a launch of their own in between: three dependent
// launches that hand [s; kappa] (14 MB), v and [Lv; BLv] (15 MB) to each other through global memory, each with its launch floor and
// its own dependent round trips.  But below the crown nothing couples two chains (SmpcController.cu:587-747: a non-branching node
// has ONE child, so solveSumChildren / solveChildNodesUpdate are copies there), and with the re-association of LAB_NOTEBOOK section 3
//     v_i = m1_i - RT [s_i; kappa_i] / (2 p_i),   s_i = beta_i + rho_child,   rho_i = s_i + m2_i,   kappa_i = kappa_child + q_child
// v_i of a chain node needs only the running sums of its OWN chain, and the forward recursion (:676-741)
//     u_i - uhat_i = (u_anc - uhat_anc) + L v_i,     x_i = x_anc + (e_i + B uhat_i) + B (u_i - uhat_i)
// is, along a chain, a LOCAL prefix sum plus an offset that is constant (u) or linear in the stage count (x) and only depends on the
// chain's parent P in the crown:
//     u_i = [uhat_i + sum_{j<=i} L v_j] + (u_P - uhat_P)
//     x_i = [sum_{j<=i} (eb_j + sum_{l<=j} B L v_l)] + x_P + (k_i - c* + 1) * bw_P,        bw_P = B (u_P - uhat_P)
// So ONE workgroup per chain does everything of its chain in one launch, in LDS: the leaf-to-top running sums, both MFMA products
// (the chain's N - c* nodes are the columns of two 16-column tiles), the local top-to-leaf prefix sums; it leaves the bracketed
// partial primal P_i in the Hx buffer and the chain top's (rho, kappa, q) for the crown.  The crown (a few nodes) is then ONE workgroup
// (k_crown_small) behind the cut parents' children sums (k_cut_partial_sums, the launch the sharded path already has), and the dual
// update adds the offsets while it forms Hx = sqrt(p_i) d_k (P_i + offset) (k_dual_stage OFFS; k_hx_finish for every other consumer).
//     six-launch form:   stream | up_chain, up_crown, gemm_vlv, down_chain | dual          (5 helper launches)
//     chain-fused form:  stream | chain_sweep, cut_partial_sums, crown_small | dual         (4, and 44 MB less through global memory)
// Same algebra, different association of the sums along a chain (local prefix + offset instead of one running sum from the root):
// the iterates agree with the six-launch form to rounding and are checked against the oracle like every other form.
#ifndef RAPIDNET_CHAIN_KERNELS_HPP_
#define RAPIDNET_CHAIN_KERNELS_HPP_

#include "kernels.hpp"

namespace rn {

constexpr int CF_THREADS = 512;       // 8 waves: two per SIMD; two workgroups per CU (<= 128 registers, <= 80 KB of LDS each)
constexpr int CF_MAX_LC = 32;         // nodes of a chain = columns of two 16-column MFMA tiles
constexpr int CF_PF = 12;             // stages whose loads a thread of the running sums requests at once (24: scratch)
constexpr int CF_KU = 4;              // k-steps per group of MFMA operands (= RN_SLAB_KU: the operators' K is padded to whole groups)
static_assert(CF_KU == RN_SLAB_KU, "the shared operators are stored with K padded to whole groups of RN_SLAB_KU k-steps");

template <typename T>
struct ChainArgs {
    const T *MV; int mV, kV, mpV, kpV;      // [Rinv | Rinv Bbt]: nv x (nv + nx), zero-padded col-major mpV x kpV
    const T *ML; int mL, kL, mpL, kpL;      // [L ; B L]: (nu + nx) x nv
    int SB, SV, SO;                          // LDS strides of the [s; kappa] columns, the v columns, the [Lv; BLv] columns
    int Lc;                                  // nodes of a chain: N - c*
    T *p;                                    // [node][ny]: P_i = [x_loc | x_loc | u_loc] (the Hx buffer)
};

// acc[c] += A_tile (16 x K, global memory: the shared operator, L2-resident) * B_c (K x 16, LDS) for two column tiles whose B
// pointers are given per tile (a chain has fewer than 32 nodes: the lanes of the missing columns re-read the last one), software-
// pipelined like slab_mfma_pipe: A fragments requested two groups ahead (three rotating register sets), B fragments one group ahead
// (two sets: 128 registers is what two workgroups per CU leave a wave).
template <typename T, int KU>
__device__ __forceinline__ void cf_mfma(typename Mfma16<T>::acc_t (&acc)[2], const T *Ap, size_t aStep, const T *Bp0, const T *Bp1, int G) {
    T a0[KU], a1[KU], a2[KU], b0[KU][2], b1[KU][2];
#define CF_LOAD_A(dst, g_) _Pragma("unroll") for (int i = 0; i < KU; i++) dst[i] = Ap[((size_t)(g_) * KU + i) * aStep];
#define CF_LOAD_B(dst, g_) _Pragma("unroll") for (int i = 0; i < KU; i++) { dst[i][0] = Bp0[((g_) * KU + i) * 4]; dst[i][1] = Bp1[((g_) * KU + i) * 4]; }
#define CF_STEP(cur, nxt, far, bcur, bnxt, g_)                                                                         \
    {                                                                                                                  \
        const int gf_ = (g_) + 2 < gl ? (g_) + 2 : gl, gn_ = (g_) + 1 < gl ? (g_) + 1 : gl;                            \
        CF_LOAD_A(far, gf_)                                                                                            \
        CF_LOAD_B(bnxt, gn_)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < KU; i++) {                                                               \
            acc[0] = Mfma16<T>::run(cur[i], bcur[i][0], acc[0]);                                                       \
            acc[1] = Mfma16<T>::run(cur[i], bcur[i][1], acc[1]);                                                       \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int i = 0; i < KU; i++) {                                                               \
            asm volatile("" ::"v"(nxt[i]));                                                                            \
            asm volatile("" ::"v"(bnxt[i][0]));                                                                        \
            asm volatile("" ::"v"(bnxt[i][1]));                                                                        \
        }                                                                                                              \
    }
    const int gl = G - 1;
    CF_LOAD_A(a0, 0)
    CF_LOAD_A(a1, (1 < gl ? 1 : gl))
    CF_LOAD_B(b0, 0)
    int g = 0;
    for (; g + 6 <= G; g += 6) {
        CF_STEP(a0, a1, a2, b0, b1, g)
        CF_STEP(a1, a2, a0, b1, b0, g + 1)
        CF_STEP(a2, a0, a1, b0, b1, g + 2)
        CF_STEP(a0, a1, a2, b1, b0, g + 3)
        CF_STEP(a1, a2, a0, b0, b1, g + 4)
        CF_STEP(a2, a0, a1, b1, b0, g + 5)
    }
    // the tail: at most five more groups, in the same rotation
    if (g < G) CF_STEP(a0, a1, a2, b0, b1, g)
    if (g + 1 < G) CF_STEP(a1, a2, a0, b1, b0, g + 1)
    if (g + 2 < G) CF_STEP(a2, a0, a1, b0, b1, g + 2)
    if (g + 3 < G) CF_STEP(a0, a1, a2, b1, b0, g + 3)
    if (g + 4 < G) CF_STEP(a1, a2, a0, b0, b1, g + 4)
#undef CF_STEP
#undef CF_LOAD_A
#undef CF_LOAD_B
}

// One workgroup per chain (blockIdx.x = position of the chain within a stage); see the head of this file.
//   phase A  running sums leaf -> top (k_up_chain's association), the columns [s_i; kappa_i] go to LDS, the top's (rho, kappa, q) to rkq
//   phase B  v = m1 - RT [s; kappa] / (2 p): MFMA, the v columns stay in LDS (and go to a.v when the primal iterates are stored)
//   phase C  [L v ; B L v]: MFMA, into LDS (over the [s; kappa] columns, which are dead by then)
//   phase D  prefix sums top -> leaf, P_i = [x_loc | x_loc | u_loc] to c.p
template <typename T>
__global__ void __launch_bounds__(CF_THREADS, 4) k_chain_sweep(SweepArgs<T> a, ChainArgs<T> c) {
    typedef typename Mfma16<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    const int Lc = c.Lc, SB = c.SB, SV = c.SV, SO = c.SO;
    T *sB = reinterpret_cast<T *>(cf_smem);                  // [Lc][SB]; phase C on: [Lc][SO]
    T *sV = sB + (size_t)Lc * (SB > SO ? SB : SO);           // [Lc][SV]
    T *sO = sB;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = CF_THREADS / 64;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny;
    const int top = a.chainStage, K = a.K;
    const size_t nodeTop = (size_t)a.tr.stageCum[top] + blockIdx.x;    // node of stage k in this chain: nodeTop + (k - c*) K
    // p_i is the same for every node of a chain, so the factor -1 / (2 p_i) of the v product goes onto its columns [s; kappa] here and the
    // accumulators of the product start from m1_i: no epilogue operands alive across the MFMA loop
    const T scale = (T)(-0.5) / a.tr.prob[nodeTop];
    // ---- phase A
    {   // (the host checks nv + nx <= CF_THREADS and nu + nx <= CF_THREADS: a thread per component)
        const int t = tid;
        if (t < nv) {
            T rho = 0;
            // the second partial m2 of the chain's last stages (k_stream_gemv's split last round lies within the last STREAM_SPLIT_STAGES stages; zero
            // where a node was not split) -- as in k_up_chain: requested up front, no branch inside the batches of loads
            T mx[STREAM_SPLIT_STAGES];
#pragma unroll
            for (int j = 0; j < STREAM_SPLIT_STAGES; j++) {
                const int kk = Lc - 1 - j >= 0 ? Lc - 1 - j : 0;
                const size_t node = nodeTop + (size_t)kk * K;
                const bool has = Lc - 1 - j >= 0 && node >= (size_t)a.splitFirst;
                mx[j] = has ? a.my2[(node - (has ? (size_t)a.splitFirst : 0)) * 2 * nv + nv + t] : (T)0;
            }
            for (int k0 = Lc - 1; k0 >= 0; k0 -= CF_PF) {
                T b[CF_PF], m[CF_PF];
#pragma unroll
                for (int j = 0; j < CF_PF; j++) {
                    const int kk = k0 - j >= 0 ? k0 - j : 0;
                    const size_t node = nodeTop + (size_t)kk * K;
                    b[j] = a.beta[node * nv + t];
                    m[j] = a.my[node * 2 * nv + nv + t];
                }
                if (k0 == Lc - 1) {
#pragma unroll
                    for (int j = 0; j < STREAM_SPLIT_STAGES && j < CF_PF; j++) m[j] += mx[j];      // (first half) + (second half)
                }
#pragma unroll
                for (int j = 0; j < CF_PF; j++) {
                    if (k0 - j >= 0) {
                        const T sv = b[j] + rho;                   // s_i
                        rho = sv + m[j];
                        sB[(k0 - j) * SB + t] = scale * (a.structured ? rho : sv);
                    }
                }
            }
            a.rkq[nodeTop * (nv + 2 * nx) + t] = rho;
        } else if (t < nv + nx) {
            const int j0 = t - nv;
            T kap = 0, q = 0;
            for (int k0 = Lc - 1; k0 >= 0; k0 -= CF_PF) {
                T av[CF_PF];
#pragma unroll
                for (int j = 0; j < CF_PF; j++) {
                    const int kk = k0 - j >= 0 ? k0 - j : 0;
                    av[j] = a.qa[(nodeTop + (size_t)kk * K) * nx + j0];
                }
#pragma unroll
                for (int j = 0; j < CF_PF; j++) {
                    if (k0 - j >= 0) {
                        kap += q;                                  // kappa_i = kappa_c + q_c
                        sB[(k0 - j) * SB + nv + j0] = scale * kap;
                        q += av[j];                                // q_i = a_i + q_c
                    }
                }
            }
            a.rkq[nodeTop * (nv + 2 * nx) + nv + j0] = kap;
            a.rkq[nodeTop * (nv + 2 * nx) + nv + nx + j0] = q;
        }
    }
    {   // the padding columns of the [s; kappa] slab and the whole v slab: zero (K of the operators is stored zero-padded, 0 * garbage is not 0)
        const int padB = SB - (nv + nx);
        for (int i = tid; i < Lc * padB; i += CF_THREADS) sB[(i / padB) * SB + nv + nx + i % padB] = (T)0;
        for (int i = tid; i < Lc * SV; i += CF_THREADS) sV[i] = (T)0;
    }
    __syncthreads();
    // the chain's node behind column c * 16 + col of the two tiles (columns past the chain's end: its last node; results dropped)
    // (registers are what limits this kernel -- 128 for two workgroups per CU: the lane's column data are derived afresh in each phase from
    //  an opaque copy of the thread index instead of living across the MFMA loops, and a node index is recomputed where it is used)
#define CF_LANE_COLUMNS                                                                                                \
    int lane_ = (int)threadIdx.x & 63;                                                                                 \
    asm volatile("" : "+v"(lane_));                                                                                    \
    const int lane = lane_, col = lane & 15, kq = lane >> 4;                                                           \
    const int colN[2] = {col < Lc ? col : Lc - 1, 16 + col < Lc ? 16 + col : Lc - 1};                                  \
    const bool live[2] = {col < Lc, 16 + col < Lc};
#define CF_NODE(ct) (nodeTop + (size_t)colN[ct] * K)
    // ---- phase B
    {
        CF_LANE_COLUMNS
        const int tiles = (c.mV + 15) / 16, G = c.kpV / (4 * CF_KU);
        for (int t = wave; t < tiles; t += NW) {
            acc_t acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++)         // m1_i: the accumulators' initial values
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg), grc = gr < c.mV ? gr : c.mV - 1;
                    acc[ct][reg] = a.my[CF_NODE(ct) * 2 * nv + grc];
                }
            if (a.splitFirst < a.nodes) {          // (uniform) the second partial m1 of the nodes of k_stream_gemv's split last round
#pragma unroll
                for (int ct = 0; ct < 2; ct++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int gr = t * 16 + Mfma16<T>::row(lane, reg), grc = gr < c.mV ? gr : c.mV - 1;
                        const bool has = CF_NODE(ct) >= (size_t)a.splitFirst;
                        acc[ct][reg] += has ? a.my2[(CF_NODE(ct) - (has ? (size_t)a.splitFirst : 0)) * 2 * nv + grc] : (T)0;
                    }
            }
            cf_mfma<T, CF_KU>(acc, c.MV + (size_t)t * 16 + col + (size_t)kq * c.mpV, (size_t)4 * c.mpV, sB + colN[0] * SB + kq, sB + colN[1] * SB + kq, G);
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                    if (live[ct] && gr < c.mV) {
                        sV[colN[ct] * SV + gr] = acc[ct][reg];
                        if (a.writePrimal) a.v[CF_NODE(ct) * nv + gr] = acc[ct][reg];
                    }
                }
        }
    }
    __syncthreads();
    // ---- phase C
    {
        CF_LANE_COLUMNS
        const int tiles = (c.mL + 15) / 16, G = c.kpL / (4 * CF_KU);
        for (int t = wave; t < tiles; t += NW) {
            acc_t acc[2] = {acc_t{0, 0, 0, 0}, acc_t{0, 0, 0, 0}};
            cf_mfma<T, CF_KU>(acc, c.ML + (size_t)t * 16 + col + (size_t)kq * c.mpL, (size_t)4 * c.mpL, sV + colN[0] * SV + kq, sV + colN[1] * SV + kq, G);
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                    if (live[ct] && gr < c.mL) sO[colN[ct] * SO + gr] = acc[ct][reg];
                }
        }
    }
#undef CF_NODE
#undef CF_LANE_COLUMNS
    // (the [s; kappa] columns under sO were last read in phase B, in front of the barrier above)
    // ---- phase D: the constants of the prefix sums (uhat_i, eb_i = e_i + B uhat_i) are requested in front of the barrier
    {
        const int w = nu + nx;
        int tid_ = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const bool on = tid_ < w;
        const int tc = on ? tid_ : 0;
        const bool isU = tc < nu;
        const int j0 = isU ? 0 : tc - nu;
        T run = 0, xr = 0;                 // u threads: sum of L v ; x threads: run = bw (sum of B L v), xr = x_loc
        for (int k0 = 0; k0 < Lc; k0 += CF_PF) {
            T cst[CF_PF];
#pragma unroll
            for (int j = 0; j < CF_PF; j++) {
                const int kk = k0 + j < Lc ? k0 + j : Lc - 1;
                const size_t node = nodeTop + (size_t)kk * K;
                cst[j] = isU ? a.uhat[node * nu + tc] : a.eb[node * nx + j0];
            }
            if (k0 == 0) __syncthreads();  // (uniform: every thread of the workgroup passes here once) the products' results are in LDS
            if (on) {
#pragma unroll
                for (int j = 0; j < CF_PF; j++) {
                    if (k0 + j < Lc) {
                        const size_t o = (nodeTop + (size_t)(k0 + j) * K) * ny;
                        run += sO[(k0 + j) * SO + tc];
                        if (isU) c.p[o + 2 * nx + tc] = cst[j] + run;
                        else {
                            xr += cst[j] + run;
                            c.p[o + j0] = xr;
                            c.p[o + nx + j0] = xr;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The crown in ONE workgroup: leaf-to-root steps of the stages < c* (the stage above the chains takes its children sums from
// a.cutSums, what k_cut_partial_sums leaves; the stages above it sum their children out of LDS), both shared-operator products for
// the crown's nodes (they are the nodes 0 .. nCrown - 1: wide_product with node0 = 0), the root-to-leaf pass, Hx of the crown nodes
// and the offset tables of the cut parents for the chain nodes:
//     off0[pos][ny] = [x_P | x_P | u_P - uhat_P],   off1[pos][ny] = [bw_P | bw_P | 0]          (pos = position of P in stage c* - 1)
// CT = 16-column tiles that hold the crown (1 or 2).  Associations: up_crown_node's and down_crown_node's.
template <typename T>
struct CrownArgs {
    GemmArgs<T> gV, gL;
    int SB, SV, SO;
    int nCrown;
    T *off0, *off1;
};
template <typename T, int CT>
__global__ void __launch_bounds__(CF_THREADS) k_crown_small(SweepArgs<T> a, CrownArgs<T> c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    const int SB = c.SB, SV = c.SV, SO = c.SO;
    T *sB = reinterpret_cast<T *>(cf_smem);        // [CT * 16][SB]
    T *sV = sB + (size_t)CT * 16 * SB;             // [CT * 16][SV]
    T *sO = sV + (size_t)CT * 16 * SV;             // [CT * 16][SO]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = CF_THREADS / 64;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny, w3 = nv + 2 * nx, per = nv + nx;
    const int cs = a.cutStage;                     // = c*: the chains' first stage
    T *rk = sV;                                    // [nC][w3] (rho, kappa, q) of the crown nodes during the upward steps: over sV and sO, which are
                                                   // unused until the products (the host checks SV + SO >= nv + 2 nx)
    const int RS = SV + SO;
    for (int i = tid; i < CT * 16 * SB; i += CF_THREADS) sB[i] = (T)0;
    __syncthreads();
    // ---- leaf-to-root: stage c* - 1 from the children sums ...
    {
        const int s0 = a.tr.stageCum[cs - 1], n1 = a.tr.stageCum[cs] - s0;
        for (int i = tid; i < n1 * per; i += CF_THREADS) {
            const int pos = i / per, t = i % per, node = s0 + pos;
            if (t < nv) {
                const T sv = a.beta[(size_t)node * nv + t] + a.cutSums[(size_t)pos * w3 + t];
                const T rho = sv + a.my[(size_t)node * 2 * nv + nv + t];
                sB[node * SB + t] = a.structured ? rho : sv;
                rk[node * RS + t] = rho;
            } else {
                const int j0 = t - nv;
                const T qs = a.cutSums[(size_t)pos * w3 + nv + nx + j0];
                const T kap = a.cutSums[(size_t)pos * w3 + nv + j0] + qs;
                sB[node * SB + nv + j0] = kap;
                rk[node * RS + nv + j0] = kap;
                rk[node * RS + nv + nx + j0] = qs + a.qa[(size_t)node * nx + j0];
            }
        }
    }
    // ... the stages above it from their children (ascending order)
    for (int k = cs - 2; k >= 0; k--) {
        __syncthreads();
        const int s0 = a.tr.stageCum[k], nk = a.tr.stageCum[k + 1] - s0;
        for (int i = tid; i < nk * per; i += CF_THREADS) {
            const int node = s0 + i / per, t = i % per;
            const int c0 = a.tr.childStart[node], nc = a.tr.childCount[node];
            if (t < nv) {
                T sum = 0;
                for (int ch = 0; ch < nc; ch++) sum += rk[(c0 + ch) * RS + t];
                const T sv = a.beta[(size_t)node * nv + t] + sum;
                const T rho = sv + a.my[(size_t)node * 2 * nv + nv + t];
                sB[node * SB + t] = a.structured ? rho : sv;
                rk[node * RS + t] = rho;
            } else {
                const int j0 = t - nv;
                T kap = 0, q = 0;
                for (int ch = 0; ch < nc; ch++) { const T qc = rk[(c0 + ch) * RS + nv + nx + j0]; kap += rk[(c0 + ch) * RS + nv + j0] + qc; q += qc; }
                sB[node * SB + nv + j0] = kap;
                rk[node * RS + nv + j0] = kap;
                rk[node * RS + nv + nx + j0] = q + a.qa[(size_t)node * nx + j0];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < CT * 16 * SV; i += CF_THREADS) sV[i] = (T)0;      // (rk is dead: the v slab must be zero beyond nv)
    __syncthreads();
    // ---- the two products
    wide_product<T, EPI_V, CT>(c.gV, sB, SB, 0, wave, NW, lane, sV, SV);
    __syncthreads();
    wide_product<T, EPI_LV, CT>(c.gL, sV, SV, 0, wave, NW, lane, sO, SO);
    __syncthreads();
    // ---- root-to-leaf over the crown; du, bw, x of the crown nodes are kept in LDS ([node][nu + 2 nx] over sB and sV, both dead by now:
    //      the host checks SB + SV >= nu + 2 nx)
    T *dn = sB;
    const int wd = nu + nx, DS = SB + SV;
    for (int k = 0; k < cs; k++) {
        const int s0 = a.tr.stageCum[k], nk = a.tr.stageCum[k + 1] - s0;
        const T *dy = a.tr.dy + (size_t)k * ny;
        for (int i = tid; i < nk * wd; i += CF_THREADS) {
            const int node = s0 + i / wd, t = i % wd;
            const int par = a.tr.parent[node];
            const T sp = a.tr.sqrtp[node];
            if (t < nu) {
                const T wanc = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : dn[par * DS + t];
                const T uh = a.uhat[(size_t)node * nu + t];
                const T uv = uh + wanc + sO[node * SO + t];
                dn[node * DS + t] = uv - uh;
                a.u[(size_t)node * nu + t] = uv;
                a.hx[(size_t)node * ny + 2 * nx + t] = sp * dy[2 * nx + t] * uv;
            } else {
                const int j0 = t - nu;
                const T bw = (par < 0 ? a.bw0[j0] : dn[par * DS + nu + j0]) + sO[node * SO + nu + j0];
                const T xv = (par < 0 ? a.curX[j0] : dn[par * DS + nu + nx + j0]) + a.eb[(size_t)node * nx + j0] + bw;
                dn[node * DS + nu + j0] = bw;
                dn[node * DS + nu + nx + j0] = xv;
                a.bw[(size_t)node * nx + j0] = bw;
                a.x[(size_t)node * nx + j0] = xv;
                a.hx[(size_t)node * ny + j0] = sp * dy[j0] * xv;
                a.hx[(size_t)node * ny + nx + j0] = sp * dy[nx + j0] * xv;
            }
        }
        __syncthreads();
    }
    // ---- the offsets of the chain nodes, per cut parent
    {
        const int s0 = a.tr.stageCum[cs - 1], n1 = a.tr.stageCum[cs] - s0;
        for (int i = tid; i < n1 * ny; i += CF_THREADS) {
            const int pos = i / ny, r = i % ny, node = s0 + pos;
            T o0, o1;
            if (r < 2 * nx) { const int j0 = r < nx ? r : r - nx; o0 = dn[node * DS + nu + nx + j0]; o1 = dn[node * DS + nu + j0]; }
            else { o0 = dn[node * DS + (r - 2 * nx)]; o1 = (T)0; }
            c.off0[(size_t)pos * ny + r] = o0;
            c.off1[(size_t)pos * ny + r] = o1;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// REGISTER-RESIDENT forms of the two kernels above, for operators of at most CFR_KSV / CFR_KSL k-steps (the Barcelona shapes: exactly 40
// and 28) and at most 8 / 16 row tiles.  Measured with the forms above on the 493-scenario tree: k_chain_sweep 41 us, k_crown_small 35 us --
// every group of four k-steps of a wave's MFMA loop waits for its A fragments' round trip to L2 (~1 us with so few waves in flight), ten
// and seven times per product.  Here a wave requests ALL A fragments of its row tile with one batch of loads -- the v product's at the very
// start of the kernel, so that they arrive during the running sums, the second product's when the first one's registers are free, in front
// of its epilogue and the barrier -- and the MFMA loops read only registers and LDS.  One workgroup per CU (up to 256 registers per lane).
// k_chain_sweep_reg<T, CPW>: CPW = 2 chains per workgroup when there are more chains than CUs (44 columns = three 16-column tiles: 92 % of
// the MFMA columns live instead of 69 %, every A fragment used for two chains, ONE round of workgroups on the 493-scenario tree), 1 otherwise.
constexpr int CFR_KSV = 40, CFR_KSL = 28;
// RN_KTIMING builds (tools/ktiming_cf.py): phase stamps (100 MHz wall clock) of workgroups 0, 1, 2 and the last one, read back by rn_debug_ktiming:
// rows 0-3 = k_chain_sweep_reg, row 4 = k_crown_small_reg (kernels.hpp keeps rows 0-3 for the slab kernels: one kind of launch per measurement)
#ifdef RN_KTIMING
#define CF_KT(slot) do { if (threadIdx.x == 0) { const int b_ = blockIdx.x == gridDim.x - 1 ? 3 : (int)blockIdx.x; if (b_ < 4) g_ktiming[b_ * 16 + (slot)] = wall_clock64(); } } while (0)
#define CR_KT(slot) do { if (threadIdx.x == 0) g_ktiming[4 * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define CF_KT(slot) do { } while (0)
#define CR_KT(slot) do { } while (0)
#endif
template <typename T, int KS>
__device__ __forceinline__ void cfr_load_a(T (&A)[KS], const T *p, size_t step, int ks) {
#pragma unroll
    for (int s = 0; s < KS; s++) A[s] = p[(size_t)(s < ks ? s : ks - 1) * step];
}
// acc[c] += A (16 x 4 G, registers) * B_c (LDS), G groups of four k-steps; the B fragments of group g + 1 are requested in front of the MFMAs of group g
template <typename T, int KS, int CT>
__device__ __forceinline__ void cfr_mfma(typename Mfma16<T>::acc_t (&acc)[CT], const T (&A)[KS], const T *const (&Bp)[CT], int G) {
    static_assert(KS % 4 == 0, "whole groups of four k-steps");
    T b[2][4][CT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < CT; c++) b[0][i][c] = Bp[c][i * 4];
#pragma unroll
    for (int g = 0; g < KS / 4; g++) {
        if (g < G) {
            if (g + 1 < G) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int c = 0; c < CT; c++) b[(g + 1) & 1][i][c] = Bp[c][((g + 1) * 4 + i) * 4];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int c = 0; c < CT; c++) acc[c] = Mfma16<T>::run(A[g * 4 + i], b[g & 1][i][c], acc[c]);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < G) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int c = 0; c < CT; c++) asm volatile("" ::"v"(b[(g + 1) & 1][i][c]));
            }
        }
    }
}
constexpr int CFR_PF = 24;            // stages per batch of loads of the running sums in the register-resident kernel (256 registers: one batch for N - c* <= 24)
template <typename T, int CPW>
__global__ void __launch_bounds__(CF_THREADS, 2) k_chain_sweep_reg(SweepArgs<T> a, ChainArgs<T> c) {
    typedef typename Mfma16<T>::acc_t acc_t;
    constexpr int CT = CPW == 1 ? 2 : 3;     // 16-column tiles: Lc <= 32 (CPW = 1), 2 Lc <= 48 (CPW = 2; the host checks)
    constexpr int NW = CF_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    const int Lc = c.Lc, SB = c.SB, SV = c.SV, SO = c.SO;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny;
    const int top = a.chainStage, K = a.K;
    const int chain0 = (int)blockIdx.x * CPW;
    const int nCh = K - chain0 < CPW ? K - chain0 : CPW;          // chains of this workgroup (the last one of an odd K: one)
    const int R = nCh * Lc;                                          // live columns
    T *sB = reinterpret_cast<T *>(cf_smem);                          // [CPW Lc][SB]; second product on: [CPW Lc][SO]
    T *sV = sB + (size_t)CPW * Lc * (SB > SO ? SB : SO);             // [CPW Lc][SV]
    T *sO = sB;
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (32-bit node and element indices: the host checks nodes * max(2 nv, ny) < 2^31)
    const int node00 = a.tr.stageCum[top] + chain0;                  // column r = (chain ci, stage k): node00 + ci + k K,  ci = r / Lc, k = r % Lc
    const int tilesV = (c.mV + 15) / 16, tilesL = (c.mL + 15) / 16;  // <= NW, <= 2 NW (the host checks)
    const int ksV = c.kpV / 4, ksL = c.kpL / 4;                      // <= CFR_KSV, <= CFR_KSL
    CF_KT(0);
    // ---- the A fragments of the v product's row tile and the product's initial values m1_i: requested first, they arrive during phase A
    T aV[CFR_KSV];
    cfr_load_a<T, CFR_KSV>(aV, c.MV + (size_t)(wave < tilesV ? wave : 0) * 16 + col + (size_t)kq * c.mpV, (size_t)4 * c.mpV, ksV);
    int colN[CT]; bool live[CT]; int nodeC[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        const int cn = ct * 16 + col;
        live[ct] = cn < R;
        colN[ct] = live[ct] ? cn : R - 1;
        const int ci = colN[ct] / Lc;
        nodeC[ct] = node00 + ci + (colN[ct] - ci * Lc) * K;
    }
    acc_t accV[CT];
    {
        const int t = wave < tilesV ? wave : 0;
#pragma unroll
        for (int ct = 0; ct < CT; ct++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int gr = t * 16 + Mfma16<T>::row(lane, reg), grc = gr < c.mV ? gr : c.mV - 1;
                accV[ct][reg] = a.my[nodeC[ct] * 2 * nv + grc];
            }
        if (a.splitFirst < a.nodes) {              // (uniform) the second partial m1 of the nodes of k_stream_gemv's split last round
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg), grc = gr < c.mV ? gr : c.mV - 1;
                    const bool has = nodeC[ct] >= a.splitFirst;
                    accV[ct][reg] += has ? a.my2[(nodeC[ct] - (has ? a.splitFirst : 0)) * 2 * nv + grc] : (T)0;
                }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    CF_KT(1);
    // ---- phase A: running sums leaf -> top, a thread per (chain, component)
    {
        const int per = nv + nx;
        const int ci = tid / per, t = tid - ci * per;
        if (ci < nCh) {
            const int nodeTop = node00 + ci;
            const T scale = (T)(-0.5) / a.tr.prob[nodeTop];          // p_i is the same along a chain: -1 / (2 p_i) goes onto the columns [s; kappa]
            T *sBc = sB + ci * Lc * SB;
            if (t < nv) {
                T rho = 0;
                T mx[STREAM_SPLIT_STAGES];           // the second partial m2 of the last stages (see k_chain_sweep)
#pragma unroll
                for (int j = 0; j < STREAM_SPLIT_STAGES; j++) {
                    const int kk = Lc - 1 - j >= 0 ? Lc - 1 - j : 0;
                    const int node = nodeTop + kk * K;
                    const bool has = Lc - 1 - j >= 0 && node >= a.splitFirst;
                    mx[j] = has ? a.my2[(node - (has ? a.splitFirst : 0)) * 2 * nv + nv + t] : (T)0;
                }
                for (int k0 = Lc - 1; k0 >= 0; k0 -= CFR_PF) {
                    T b[CFR_PF], m[CFR_PF];
#pragma unroll
                    for (int j = 0; j < CFR_PF; j++) {
                        const int kk = k0 - j >= 0 ? k0 - j : 0;
                        const int node = nodeTop + kk * K;
                        b[j] = a.beta[node * nv + t];
                        m[j] = a.my[node * 2 * nv + nv + t];
                    }
                    if (k0 == Lc - 1) {
#pragma unroll
                        for (int j = 0; j < STREAM_SPLIT_STAGES && j < CFR_PF; j++) m[j] += mx[j];
                    }
#pragma unroll
                    for (int j = 0; j < CFR_PF; j++) {
                        if (k0 - j >= 0) {
                            const T sv = b[j] + rho;
                            rho = sv + m[j];
                            sBc[(k0 - j) * SB + t] = scale * (a.structured ? rho : sv);
                        }
                    }
                }
                a.rkq[nodeTop * (nv + 2 * nx) + t] = rho;
            } else {
                const int j0 = t - nv;
                T kap = 0, q = 0;
                for (int k0 = Lc - 1; k0 >= 0; k0 -= CFR_PF) {
                    T av[CFR_PF];
#pragma unroll
                    for (int j = 0; j < CFR_PF; j++) {
                        const int kk = k0 - j >= 0 ? k0 - j : 0;
                        av[j] = a.qa[(nodeTop + kk * K) * nx + j0];
                    }
#pragma unroll
                    for (int j = 0; j < CFR_PF; j++) {
                        if (k0 - j >= 0) {
                            kap += q;
                            sBc[(k0 - j) * SB + nv + j0] = scale * kap;
                            q += av[j];
                        }
                    }
                }
                a.rkq[nodeTop * (nv + 2 * nx) + nv + j0] = kap;
                a.rkq[nodeTop * (nv + 2 * nx) + nv + nx + j0] = q;
            }
        }
        // zero where the products read what nobody writes: the K padding of the two B operands, k in [nv + nx, kpV) and [nv, kpL) (the operators'
        // columns there are zero, but 0 * garbage is not 0); the columns beyond are never read
        const int padB = c.kpV - per, padV = c.kpL - nv;
        for (int i = tid; i < R * padB; i += CF_THREADS) sB[(i / padB) * SB + per + i % padB] = (T)0;
        for (int i = tid; i < R * padV; i += CF_THREADS) sV[(i / padV) * SV + nv + i % padV] = (T)0;
    }
    CF_KT(2);
    __syncthreads();
    CF_KT(3);
    // ---- phase B: v = m1 + RT (scaled [s; kappa])
    T aL[2][CFR_KSL];
    {
        const int t = wave;
        const T *Bp[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) Bp[ct] = sB + colN[ct] * SB + kq;
        if (t < tilesV) cfr_mfma<T, CFR_KSV, CT>(accV, aV, Bp, ksV / 4);
        CF_KT(4);
        // the second product's A fragments (row tiles wave, wave + 8): their round trip overlaps the epilogue and the barrier
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int tl = wave + NW * j;
            cfr_load_a<T, CFR_KSL>(aL[j], c.ML + (size_t)(tl < tilesL ? tl : 0) * 16 + col + (size_t)kq * c.mpL, (size_t)4 * c.mpL, ksL);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t < tilesV) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                    if (live[ct] && gr < c.mV) {
                        sV[colN[ct] * SV + gr] = accV[ct][reg];
                        if (a.writePrimal) a.v[nodeC[ct] * nv + gr] = accV[ct][reg];
                    }
                }
        }
    }
    CF_KT(5);
    __syncthreads();
    CF_KT(6);
    // ---- phase C: [L v ; B L v] into LDS (over the [s; kappa] columns: last read in front of the barrier above)
    {
        const T *Bp[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) Bp[ct] = sV + colN[ct] * SV + kq;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int t = wave + NW * j;
            if (t < tilesL) {
                acc_t acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ct++) acc[ct] = acc_t{0, 0, 0, 0};
                cfr_mfma<T, CFR_KSL, CT>(acc, aL[j], Bp, ksL / 4);
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                        if (live[ct] && gr < c.mL) sO[colN[ct] * SO + gr] = acc[ct][reg];
                    }
            }
        }
    }
    CF_KT(7);
    // ---- phase D: prefix sums top -> leaf, a thread per (chain, component); the constants are requested in front of the barrier
    {
        const int w = nu + nx;
        const int ci = tid / w, tc0 = tid - ci * w;
        const bool on = ci < nCh;
        const int tc = on ? tc0 : 0, cic = on ? ci : 0;
        const bool isU = tc < nu;
        const int j0 = isU ? 0 : tc - nu;
        const int nodeTop = node00 + cic;
        const T *sOc = sO + cic * Lc * SO;
        T run = 0, xr = 0;
        for (int k0 = 0; k0 < Lc; k0 += CFR_PF) {
            T cst[CFR_PF];
#pragma unroll
            for (int j = 0; j < CFR_PF; j++) {
                const int kk = k0 + j < Lc ? k0 + j : Lc - 1;
                const int node = nodeTop + kk * K;
                cst[j] = isU ? a.uhat[node * nu + tc] : a.eb[node * nx + j0];
            }
            if (k0 == 0) { __syncthreads(); CF_KT(8); }
            if (on) {
#pragma unroll
                for (int j = 0; j < CFR_PF; j++) {
                    if (k0 + j < Lc) {
                        const int o = (nodeTop + (k0 + j) * K) * ny;
                        run += sOc[(k0 + j) * SO + tc];
                        if (isU) c.p[o + 2 * nx + tc] = cst[j] + run;
                        else {
                            xr += cst[j] + run;
                            c.p[o + j0] = xr;
                            c.p[o + nx + j0] = xr;
                        }
                    }
                }
            }
        }
    }
    CF_KT(9);
}
// the crown workgroup with register-resident A fragments (see above): k_crown_small's steps with every dependent round trip to global memory
// taken out of the workgroup's critical path that can be -- the first measurement of a single-workgroup crown was 35 us, of which the MFMAs
// are 6: the rest were ~20 strided loops whose trips each waited for their own loads.  Here the inputs of the leaf-to-root step of the stage
// above the chains are requested in batches of CRN_UB items per thread, and the constants of the root-to-leaf pass (uhat / eb, the scaling
// of Hx, the parent) of all of a thread's items in ONE batch behind the second product.
constexpr int CRN_UB = 6, CRN_DB = 7;
template <typename T, int CT>
__global__ void __launch_bounds__(CF_THREADS, 2) k_crown_small_reg(SweepArgs<T> a, CrownArgs<T> c) {
    typedef typename Mfma16<T>::acc_t acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char cf_smem[];
    const int SB = c.SB, SV = c.SV, SO = c.SO, nC = c.nCrown;
    T *sB = reinterpret_cast<T *>(cf_smem);        // [CT * 16][SB]
    T *sV = sB + (size_t)CT * 16 * SB;             // [CT * 16][SV]
    T *sO = sV + (size_t)CT * 16 * SV;             // [CT * 16][SO]
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = CF_THREADS / 64;
    const int nv = a.nv, nx = a.nx, nu = a.nu, ny = a.ny, w3 = nv + 2 * nx, per = nv + nx;
    const int cs = a.cutStage;
    const GemmArgs<T> &gV = c.gV, &gL = c.gL;
    const int tilesV = (gV.m + 15) / 16, tilesL = (gL.m + 15) / 16, ksV = gV.kp / 4, ksL = gL.kp / 4;
    CR_KT(0);
    T aV[CFR_KSV];
    cfr_load_a<T, CFR_KSV>(aV, gV.M + (size_t)(wave < tilesV ? wave : 0) * 16 + col + (size_t)kq * gV.mp, (size_t)4 * gV.mp, ksV);
    // the products' epilogue operands (m1_i, 1 / p_i)
    int nodeC[CT]; bool live[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) { const int n = ct * 16 + col; live[ct] = n < nC; nodeC[ct] = live[ct] ? n : nC - 1; }
    T auxv[CT][4], scale[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        scale[ct] = (T)(-0.5) / gV.prob[nodeC[ct]];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int gr = (wave < tilesV ? wave : 0) * 16 + Mfma16<T>::row(lane, reg);
            auxv[ct][reg] = gV.aux[(size_t)nodeC[ct] * gV.ldaux + (gr < gV.m ? gr : gV.m - 1)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    CR_KT(1);
    T *rk = sV;
    const int RS = SV + SO;
    // ---- leaf-to-root: stage c* - 1 from the children sums (batches of CRN_UB items per thread: loads first) ...
    {
        const int s0 = a.tr.stageCum[cs - 1], n1 = a.tr.stageCum[cs] - s0, total = n1 * per;
        for (int i0 = tid; i0 < total; i0 += CF_THREADS * CRN_UB) {
            T c0[CRN_UB], c1[CRN_UB], b0[CRN_UB], b1[CRN_UB];
#pragma unroll
            for (int u = 0; u < CRN_UB; u++) {
                const int i = i0 + u * CF_THREADS;
                const bool on = i < total;
                const int pos = on ? i / per : 0, t = on ? i - pos * per : 0, node = s0 + pos;
                if (t < nv) {
                    c0[u] = a.cutSums[(size_t)pos * w3 + t]; c1[u] = 0;
                    b0[u] = a.beta[(size_t)node * nv + t]; b1[u] = a.my[(size_t)node * 2 * nv + nv + t];
                } else {
                    const int j0 = t - nv;
                    c0[u] = a.cutSums[(size_t)pos * w3 + nv + j0]; c1[u] = a.cutSums[(size_t)pos * w3 + nv + nx + j0];
                    b0[u] = a.qa[(size_t)node * nx + j0]; b1[u] = 0;
                }
            }
#pragma unroll
            for (int u = 0; u < CRN_UB; u++) {
                const int i = i0 + u * CF_THREADS;
                if (i < total) {
                    const int pos = i / per, t = i - pos * per, node = s0 + pos;
                    if (t < nv) {
                        const T sv = b0[u] + c0[u];
                        const T rho = sv + b1[u];
                        sB[node * SB + t] = a.structured ? rho : sv;
                        rk[node * RS + t] = rho;
                    } else {
                        const int j0 = t - nv;
                        const T kap = c0[u] + c1[u];
                        sB[node * SB + nv + j0] = kap;
                        rk[node * RS + nv + j0] = kap;
                        rk[node * RS + nv + nx + j0] = c1[u] + b0[u];
                    }
                }
            }
        }
    }
    CR_KT(2);
    // ... the stages above it from their children (ascending order); a thread's first item has its own terms requested in front of the barrier
    for (int k = cs - 2; k >= 0; k--) {
        const int s0 = a.tr.stageCum[k], nk = a.tr.stageCum[k + 1] - s0, total = nk * per;
        T own0 = 0, own1 = 0;
        int ch0 = 0, nc = 0;
        if (tid < total) {
            const int node = s0 + tid / per, t = tid % per;
            ch0 = a.tr.childStart[node]; nc = a.tr.childCount[node];
            own0 = t < nv ? a.beta[(size_t)node * nv + t] : a.qa[(size_t)node * nx + (t - nv)];
            own1 = t < nv ? a.my[(size_t)node * 2 * nv + nv + t] : (T)0;
        }
        __syncthreads();
        for (int i = tid; i < total; i += CF_THREADS) {
            const int node = s0 + i / per, t = i % per;
            if (i != tid) {
                ch0 = a.tr.childStart[node]; nc = a.tr.childCount[node];
                own0 = t < nv ? a.beta[(size_t)node * nv + t] : a.qa[(size_t)node * nx + (t - nv)];
                own1 = t < nv ? a.my[(size_t)node * 2 * nv + nv + t] : (T)0;
            }
            if (t < nv) {
                T sum = 0;
                for (int ch = 0; ch < nc; ch++) sum += rk[(ch0 + ch) * RS + t];
                const T sv = own0 + sum;
                const T rho = sv + own1;
                sB[node * SB + t] = a.structured ? rho : sv;
                rk[node * RS + t] = rho;
            } else {
                const int j0 = t - nv;
                T kap = 0, q = 0;
                for (int ch = 0; ch < nc; ch++) { const T qc = rk[(ch0 + ch) * RS + nv + nx + j0]; kap += rk[(ch0 + ch) * RS + nv + j0] + qc; q += qc; }
                sB[node * SB + nv + j0] = kap;
                rk[node * RS + nv + j0] = kap;
                rk[node * RS + nv + nx + j0] = q + own0;
            }
        }
    }
    __syncthreads();
    CR_KT(3);
    {   // zero where the products read what nobody writes: the K padding of the B operands' live columns (rk, which lay over sV, is dead)
        const int padB = gV.kp - per, padV = gL.kp - nv;
        for (int i = tid; i < nC * padB; i += CF_THREADS) sB[(i / padB) * SB + per + i % padB] = (T)0;
        for (int i = tid; i < nC * padV; i += CF_THREADS) sV[(i / padV) * SV + nv + i % padV] = (T)0;
    }
    __syncthreads();
    CR_KT(4);
    // ---- the two products (wide_product's epilogues: v = m1 - acc / (2 p), then [L v; B L v])
    T aL[2][CFR_KSL];
    {
        acc_t acc[CT];
        const int t = wave;
#pragma unroll
        for (int ct = 0; ct < CT; ct++) acc[ct] = acc_t{0, 0, 0, 0};
        const T *Bp[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) Bp[ct] = sB + nodeC[ct] * SB + kq;
        if (t < tilesV) cfr_mfma<T, CFR_KSV, CT>(acc, aV, Bp, ksV / 4);
        CR_KT(5);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int tl = wave + NW * j;
            cfr_load_a<T, CFR_KSL>(aL[j], gL.M + (size_t)(tl < tilesL ? tl : 0) * 16 + col + (size_t)kq * gL.mp, (size_t)4 * gL.mp, ksL);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t < tilesV) {
#pragma unroll
            for (int ct = 0; ct < CT; ct++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                    const T r = auxv[ct][reg] + scale[ct] * acc[ct][reg];
                    if (gr < gV.m && live[ct]) {
                        if (gV.out) gV.out[(size_t)nodeC[ct] * gV.ldout + gr] = r;
                        sV[nodeC[ct] * SV + gr] = r;
                    }
                }
        }
    }
    __syncthreads();
    CR_KT(6);
    {
        const T *Bp[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ct++) Bp[ct] = sV + nodeC[ct] * SV + kq;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int t = wave + NW * j;
            if (t < tilesL) {
                acc_t acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ct++) acc[ct] = acc_t{0, 0, 0, 0};
                cfr_mfma<T, CFR_KSL, CT>(acc, aL[j], Bp, ksL / 4);
#pragma unroll
                for (int ct = 0; ct < CT; ct++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const int gr = t * 16 + Mfma16<T>::row(lane, reg);
                        if (gr < gL.m && live[ct]) sO[nodeC[ct] * SO + gr] = acc[ct][reg];
                    }
            }
        }
    }
    CR_KT(7);
    // the root-to-leaf pass: item i = (crown node, component of [u | x]); a thread's items tid, tid + 512, ...; the first CRN_DB of them have their
    // constants requested here, behind the second product's MFMAs (its registers are free) and in front of the barrier
    const int wd = nu + nx, nItems = nC * wd;
    T dCst[CRN_DB], dF0[CRN_DB], dF1[CRN_DB];
    int dNode[CRN_DB], dPar[CRN_DB], dStage[CRN_DB];
#pragma unroll
    for (int u = 0; u < CRN_DB; u++) {
        const int i = tid + u * CF_THREADS;
        const bool on = i < nItems;
        const int node = on ? i / wd : 0, t = on ? i - node * wd : 0;
        const int k = a.tr.stageOf[node];
        const T *dy = a.tr.dy + (size_t)k * ny;
        const T sp = a.tr.sqrtp[node];
        dNode[u] = on ? node : -1; dPar[u] = a.tr.parent[node]; dStage[u] = k;
        if (t < nu) { dCst[u] = a.uhat[(size_t)node * nu + t]; dF0[u] = sp * dy[2 * nx + t]; dF1[u] = (T)0; }
        else { const int j0 = t - nu; dCst[u] = a.eb[(size_t)node * nx + j0]; dF0[u] = sp * dy[j0]; dF1[u] = sp * dy[nx + j0]; }
    }
    CR_KT(8);
    __syncthreads();
    CR_KT(9);
    // ---- root-to-leaf over the crown, stage by stage; du, bw, x of the crown nodes in LDS ([node][nu + 2 nx] over sB and sV, both dead by now)
    T *dn = sB;
    const int DS = SB + SV;
    for (int k = 0; k < cs; k++) {
#pragma unroll
        for (int u = 0; u < CRN_DB; u++) {
            if (dNode[u] >= 0 && dStage[u] == k) {
                const int node = dNode[u], par = dPar[u], t = tid + u * CF_THREADS - node * wd;
                if (t < nu) {
                    const T wanc = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : dn[par * DS + t];
                    const T uv = dCst[u] + wanc + sO[node * SO + t];
                    dn[node * DS + t] = uv - dCst[u];
                    a.u[(size_t)node * nu + t] = uv;
                    a.hx[(size_t)node * ny + 2 * nx + t] = dF0[u] * uv;
                } else {
                    const int j0 = t - nu;
                    const T bw = (par < 0 ? a.bw0[j0] : dn[par * DS + nu + j0]) + sO[node * SO + nu + j0];
                    const T xv = (par < 0 ? a.curX[j0] : dn[par * DS + nu + nx + j0]) + dCst[u] + bw;
                    dn[node * DS + nu + j0] = bw;
                    dn[node * DS + nu + nx + j0] = xv;
                    a.bw[(size_t)node * nx + j0] = bw;
                    a.x[(size_t)node * nx + j0] = xv;
                    a.hx[(size_t)node * ny + j0] = dF0[u] * xv;
                    a.hx[(size_t)node * ny + nx + j0] = dF1[u] * xv;
                }
            }
        }
        // (crowns with more than CRN_DB * 512 items: the rest with their loads where they are used)
        for (int i = tid + CRN_DB * CF_THREADS; i < nItems; i += CF_THREADS) {
            const int node = i / wd, t = i - node * wd;
            if (a.tr.stageOf[node] != k) continue;
            const int par = a.tr.parent[node];
            const T sp = a.tr.sqrtp[node];
            const T *dy = a.tr.dy + (size_t)k * ny;
            if (t < nu) {
                const T wanc = par < 0 ? (a.prevU[t] - a.prevUhat[t]) : dn[par * DS + t];
                const T uh = a.uhat[(size_t)node * nu + t];
                const T uv = uh + wanc + sO[node * SO + t];
                dn[node * DS + t] = uv - uh;
                a.u[(size_t)node * nu + t] = uv;
                a.hx[(size_t)node * ny + 2 * nx + t] = sp * dy[2 * nx + t] * uv;
            } else {
                const int j0 = t - nu;
                const T bw = (par < 0 ? a.bw0[j0] : dn[par * DS + nu + j0]) + sO[node * SO + nu + j0];
                const T xv = (par < 0 ? a.curX[j0] : dn[par * DS + nu + nx + j0]) + a.eb[(size_t)node * nx + j0] + bw;
                dn[node * DS + nu + j0] = bw;
                dn[node * DS + nu + nx + j0] = xv;
                a.bw[(size_t)node * nx + j0] = bw;
                a.x[(size_t)node * nx + j0] = xv;
                a.hx[(size_t)node * ny + j0] = sp * dy[j0] * xv;
                a.hx[(size_t)node * ny + nx + j0] = sp * dy[nx + j0] * xv;
            }
        }
        __syncthreads();
    }
    CR_KT(10);
    {
        const int s0 = a.tr.stageCum[cs - 1], n1 = a.tr.stageCum[cs] - s0;
        for (int i = tid; i < n1 * ny; i += CF_THREADS) {
            const int pos = i / ny, r = i - pos * ny, node = s0 + pos;
            T o0, o1;
            if (r < 2 * nx) { const int j0 = r < nx ? r : r - nx; o0 = dn[node * DS + nu + nx + j0]; o1 = dn[node * DS + nu + j0]; }
            else { o0 = dn[node * DS + (r - 2 * nx)]; o1 = (T)0; }
            c.off0[(size_t)pos * ny + r] = o0;
            c.off1[(size_t)pos * ny + r] = o1;
        }
    }
    CR_KT(11);
}

// ------------------------------------------------------------------------------------------------------
// P -> Hx in place for the chain nodes (every consumer of Hx other than k_dual_stage OFFS: the exact path's dual update and its
// fix-up pass, the last iteration of a batch, step-wise calls); writePrimal: x and u of the chain nodes as well.
// One workgroup per (stage, group of nodes); the expression is cf_primal's, so the bits are k_dual_stage OFFS's.
template <typename T>
struct FinishArgs {
    T *hx;                    // in: P, out: Hx
    const T *off0, *off1;
    const int *chainPar;      // [K] position of the chain's parent within stage c* - 1
    const T *sqrtp, *dy;
    T *x, *u;
    int nx, nu, ny, cs, K, node0, N, writePrimal;
};
template <typename T>
__global__ void __launch_bounds__(ELT_THREADS) k_hx_finish(FinishArgs<T> f) {
    const long long total = (long long)(f.N - f.cs) * f.K * f.ny;
    for (long long i = (long long)blockIdx.x * ELT_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * ELT_THREADS) {
        const long long nl = i / f.ny;
        const int r = (int)(i - nl * f.ny);
        const int sIdx = (int)(nl / f.K), s = (int)(nl - (long long)sIdx * f.K);
        const long long node = f.node0 + nl;
        const int pos = f.chainPar[s];
        const T pr = cf_primal<T>(f.hx[node * f.ny + r], (T)(sIdx + 1), f.off1[(size_t)pos * f.ny + r], f.off0[(size_t)pos * f.ny + r]);
        f.hx[node * f.ny + r] = (f.sqrtp[node] * f.dy[(size_t)(f.cs + sIdx) * f.ny + r]) * pr;
        if (f.writePrimal) {
            if (r < f.nx) f.x[node * f.nx + r] = pr;
            else if (r >= 2 * f.nx) f.u[node * f.nu + (r - 2 * f.nx)] = pr;
        }
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
#endif
