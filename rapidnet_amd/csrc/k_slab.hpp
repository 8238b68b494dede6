// k_slab.hpp -- products with the SHARED operators on the matrix cores: k_gemm_shared, the slab kernels k_gemm_slab / k_gemm_prep_m2 / k_gemm_vlv / k_gemm_vlv_wide
// (part of the kernel sources of librapidnet_hip.so; kernels.hpp includes every family header, the translation units k_*.hip instantiate them)
#pragma once
#include "common.hpp"
#include "k_walks.hpp"

namespace rn {

// ------------------------------------------------------------------------------------------------------
// Batched product with a SHARED small matrix:  out_i = epi( M * in_i )  for all nodes i, as a tiled GEMM
// [m x k] * [k x nodes] on the matrix cores (MFMA 16x16x4, one wave = 64 rows x 16 nodes, operands straight
// from L2-resident M and the node-major vectors; these products are genuine dense contractions).
// The reference issues these as per-stage cublasSgemm / per-node SgemmBatched calls on K identical copies of the
// matrices (SmpcController.cu:604-611, :692-736).  Epilogues:
//   EPI_V : v_i  = m1_i - acc / (2 p_i)        M = [Rinv | Rinv Bbt], in = [s_i; kappa_i]     (:604-623)
//   EPI_LV: lv_i = acc                          M = L,  in = v_i                                (:692,:701,:727)
//   EPI_Z : z_i  = e_i + acc                    M = B,  in = u_i                                (:695,:715,:736)
enum { EPI_V = 0, EPI_LV = 1, EPI_Z = 2, EPI_VT = 3 };   // EPI_LV is also used for the structured m2_i = [Bbt | L'] [a_i; b_i]; EPI_VT: EPI_V's epilogue on a product
                                                          // that multiplies transposed (slab kernels: results leave as whole lines; k_gemm_comp)
#define RN_EPI_IS_V(EPI) ((EPI) == EPI_V || (EPI) == EPI_VT)
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
#define RN_SLAB_TR(EPI) (RN_SLAB_T == 1 || (EPI) == EPI_VT || (RN_SLAB_T == 2 && (EPI) == EPI_LV))
#ifndef RN_SLAB_KU
#define RN_SLAB_KU 4   // k-steps per group of operands; the operators' K is stored padded to whole groups (host: pad_k).  (Groups of 8 for the waves that own
                       // one tile -- the v product: 7 tiles on 8 waves -- were measured in round 6: 29.6 against 28.8 us dense, 41.7 against 41.1 us in the
                       // structured linear form: the lean loop is not waiting for its operand requests; profiles/r06_notes.md)
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
            scale[reg] = RN_EPI_IS_V(EPI) ? (T)(-0.5) / g.prob[nodeC] : (T)0;
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
    for (int reg = 0; reg < 4; reg++) scale[reg] = RN_EPI_IS_V(EPI) ? (T)(-0.5) / g.prob[nodeC] : (T)0;
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
            if (RN_EPI_IS_V(EPI)) r = auxv[j][reg] + scale[reg] * r;
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
                if (t < nx) { val = sp * (dy[t] * a.w[y + t] + dy[nx + t] * a.w[y + nx + t]); a.qa[(size_t)node * nx + t] = val; }
                else { const int j = t - nx; val = sp * dy[2 * nx + j] * a.w[y + 2 * nx + j]; }
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
        if (a.lin) up_crown_node_lin<T>(a, 0, 0, threadIdx.x, blockDim.x); else up_crown_node<T>(a, 0, 0, threadIdx.x, blockDim.x);
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

// Structured mode, linear form with the subtree sums of beta as a constant of the control step: v_i only ever feeds [L v_i ; B L v_i], so the two
// products are ONE with the composite operator [L ; B L] [T1 | T2] ((nu + nx) x (nx + nu), formed by the host in fp64):
//   [L v_i ; B L v_i] = [L ; B L] c_i - ([L ; B L] [T1 | T2]) [q_i + kappa_i ; Bu_i] / (2 p_i)
// -- the first term a constant of the control step (GemmArgs::aux), the second one slab product with the v product's epilogue: no v tile, no barrier
// between two products, fewer MFMAs (12 tiles x 45 k-steps against 7 x 45 + 12 x 25 on the Barcelona network).  v_i itself is an output only: the
// iterations that store the primal iterates run its product as a launch of its own.  foldRoot (0 / 1) as in k_gemm_vlv.
template <typename T, bool PIPE>
__global__ void __launch_bounds__(64 * SLAB_MAX_WAVES) k_gemm_comp(GemmArgs<T> g, int SB, SweepArgs<T> a, int foldRoot) {
    extern __shared__ unsigned char gemm_smem[];
    T *sB = reinterpret_cast<T *>(gemm_smem);   // [16][SB] slab of [q + kappa ; Bu]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    if (foldRoot == 1 && blockIdx.x == 0) {
        up_crown_node_lin<T>(a, 0, 0, threadIdx.x, blockDim.x);
        __threadfence_block();                   // same workgroup reads sk2 of node 0 back below (same CU, same L1)
        __syncthreads();
    }
    const int node0 = blockIdx.x * 16;
    const int cnt = g.nodes - node0 < 16 ? g.nodes - node0 : 16;
    slab_load<T>(sB, SB, g.in, g.ldin, g.k, g.kp, node0, cnt, wave, nw, lane);
    __syncthreads();
#ifndef RN_COMP_T
#define RN_COMP_T 1
#endif
    slab_product<T, RN_COMP_T ? EPI_VT : EPI_V, PIPE>(g, sB, SB, node0, wave, nw, lane, nullptr, 0);
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
        if (a.lin) up_crown_node_lin<T>(a, 0, 0, threadIdx.x, blockDim.x); else up_crown_node<T>(a, 0, 0, threadIdx.x, blockDim.x);
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


}  // namespace rn
