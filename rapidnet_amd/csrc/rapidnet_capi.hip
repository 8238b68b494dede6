// rapidnet_capi.hip -- implementation of include/rapidnet.h (librapidnet_hip.so).
//
// Host side of the boundary: owns every device allocation, drives the kernels of kernels.hpp on one HIP
// stream, never synchronises inside the APG loop.  No torch types, no cuBLAS/rocBLAS: the only external
// libraries are the HIP runtime and (lazily, for multi-GPU) RCCL.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/rapidnet.h"
#include "../../include/rapidnet_debug.h"
#include "kernels.hpp"
#include "fbe_kernels.hpp"
#include "partition.hpp"

// Every kernel this file launches is compiled in ONE of the units k_stream.hip ... k_fbe.hip; here the instantiations are only declared.
// (RN_NO_EXTERN_TEMPLATES: the one-unit build tools/gen_instantiations.py uses to find out what is launched.)
#ifndef RN_NO_EXTERN_TEMPLATES
#define RN_LINKAGE extern
#include "instantiations/stream.inc"
#include "instantiations/walks.inc"
#include "instantiations/slab.inc"
#include "instantiations/dual.inc"
#include "instantiations/misc.inc"
#include "instantiations/fbe.inc"
#undef RN_LINKAGE
#endif

#ifndef RN_FIXUP_BLOCKS
#define RN_FIXUP_BLOCKS 64
#endif
#ifndef RN_FOLD_CROWN_DOWN
#define RN_FOLD_CROWN_DOWN 1
#endif
#ifndef RN_OPT_LOCAL_MIN
#define RN_OPT_LOCAL_MIN 16   // shortest rn_apg_iterate batch that takes the optimistic single-GPU path
#endif
#ifndef RN_OPT_BACKOFF
#define RN_OPT_BACKOFF 8      // exact batches after an optimistic batch had to be replayed
#endif
#ifndef RN_DUAL_REGEN
#define RN_DUAL_REGEN 1
#endif
#ifndef RN_DOWN_PF_UNSC
#define RN_DOWN_PF_UNSC 12   // stages per batch of loads of the unscaled walk (24: one batch, 212 registers, two waves per SIMD: 16.6 -> 16.8 us)
#endif
#ifndef RN_FOLD_ROOT
#define RN_FOLD_ROOT 1
#endif
#ifndef RN_DUAL_STAGE
#define RN_DUAL_STAGE 1   // 1: stage-tiled main pass of the fused dual update (k_dual_stage) whenever the shape allows it
#endif
#ifndef RN_DUAL_STAGE_MAX_BLOCKS
#define RN_DUAL_STAGE_MAX_BLOCKS 16384
#endif
#ifndef RN_GEMM_SLAB
#define RN_GEMM_SLAB 1   // 1: slab kernels (k_gemm_slab / fused k_gemm_vlv); 0: always the tile kernel k_gemm_shared
#endif
#ifndef RN_GEMM_KS
#define RN_GEMM_KS 3   // k-steps whose operands a wave requests at once in k_gemm_shared (tuning knob)
#endif

namespace rn {

#define RN_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            char b_[512];                                                                              \
            snprintf(b_, sizeof b_, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #call); \
            err = b_;                                                                                  \
            return RN_E_HIP;                                                                           \
        }                                                                                              \
    } while (0)

#define RN_CHECK(cond, code, msg)  \
    do {                           \
        if (!(cond)) {             \
            err = (msg);           \
            return (code);         \
        }                          \
    } while (0)

// ---- minimal RCCL binding, resolved at run time (so single-GPU users never load it) ---------------------
struct UniqueId128 { char b[128]; };
struct NcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*InitRank)(void **, int, struct UniqueId128, int) = nullptr;
    int (*GetAsyncError)(void *, int *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    std::string path;   // file the bound image was loaded from (rn_comm_library)
    bool load() {
        if (h) return true;
        // an RCCL image that is already part of the process (PyTorch links one under the soname librccl.so.1) is reused:
        // RTLD_NOLOAD only returns a handle to a loaded object; a second image of the library is never mapped
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (h) break; }
        if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
        AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
        CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
        CommCount = (int (*)(void *, int *))dlsym(h, "ncclCommCount");
        GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
        InitRank = (int (*)(void **, int, UniqueId128, int))dlsym(h, "ncclCommInitRank");
        GetAsyncError = (int (*)(void *, int *))dlsym(h, "ncclCommGetAsyncError");
        CommAbort = (int (*)(void *))dlsym(h, "ncclCommAbort");
        Dl_info info;
        if (AllReduce && dladdr((void *)AllReduce, &info) && info.dli_fname) path = info.dli_fname;
        return GetUniqueId && AllReduce && InitRank;
    }
};
static NcclApi g_nccl;
constexpr int NCCL_IN_PROGRESS = 7;   // ncclInProgress
static std::atomic<long> g_guardContexts{0}, g_guardBadBytes{0};   // guard mode: contexts checked when they were destroyed / red-zone bytes found overwritten
static std::atomic<long> g_liveContexts{0};   // contexts of this process (rn_device_memory_info: a device-wide figure is only the caller's own while this is 1)

struct CtxBase {
    std::string err;
    virtual ~CtxBase() {}
    virtual int factor_step(const rn_system *) = 0;
    virtual int set_tree_errors(const double *, const double *) = 0;
    virtual int set_uncertainty(int, int, double) = 0;
    virtual int update_state_control(const double *, const double *, const double *) = 0;
    virtual int eliminate(const double *, const double *) = 0;
    virtual int set_parameters(double, double, double) = 0;
    virtual int apg_reset() = 0;
    virtual int apg_iterate(int, double *) = 0;
    virtual int control_action(const double *, const double *, const double *, const double *, const double *, int, int,
                               double *) = 0;
    virtual int extrapolate(double) = 0;
    virtual int solve_step() = 0;
    virtual int prox() = 0;
    virtual int residual() = 0;
    virtual int dual_update() = 0;
    virtual int primal_infeasibility(double *) = 0;
    virtual int prox_distances(double *, double *) = 0;
    virtual size_t buffer_size(int) const = 0;
    virtual int get(int, double *, size_t) = 0;
    virtual int set(int, const double *, size_t) = 0;
    virtual int get_operator(int, int, double *, size_t) = 0;
    virtual int device_pointer(int, void **, size_t *, int *) = 0;
    virtual int get_range(int, size_t, size_t, double *) = 0;
    virtual int set_range(int, size_t, size_t, const double *) = 0;
    virtual int profile_enable(int) = 0;
    virtual int profile_reset() = 0;
    virtual int profile_read(double *, long *) = 0;
    virtual int algorithmic_bytes(double *, double *) const = 0;
    virtual int synchronize() = 0;
    virtual void *stream_handle() = 0;
    virtual int comm_init(int, int, const void *, double timeoutSeconds = -1.0) = 0;
    virtual int comm_check() = 0;
    virtual int set_cut_stage(int) = 0;
    virtual int hist_parts(int, int, double *) = 0;
    virtual int counters(long *) = 0;
    virtual int kernel_info(int *) = 0;
    virtual int sweep_phase(int) = 0;
    virtual int set_operator_mode(int) = 0;
    virtual int get_operator_mode(int *, int *) = 0;
    virtual int set_operator(int, int, const double *, size_t) = 0;
    virtual int set_warm_start(int) = 0;
    virtual int set_exchange_mode(int) = 0;
    virtual int set_cut_moments(const double *, const double *, size_t) = 0;
    virtual int cut_buffer(int, double *, size_t) = 0;
    virtual int measure_hbm(size_t, int, double *, double *) = 0;
    // global FBE / NAMA
    virtual int set_algorithm(int, int) = 0;
    virtual int fbe_reset() = 0;
    virtual int hessian_oracle() = 0;
    virtual int gradient_fbe() = 0;
    virtual int nama_residual() = 0;
    virtual int lbfgs_direction() = 0;
    virtual int lbfgs_update_buffer() = 0;
    virtual int lbfgs_two_loop() = 0;
    virtual int value_fbe(double *) = 0;
    virtual int line_search_fbe(double, double *) = 0;
    virtual int line_search_ame(double, double *) = 0;
    virtual int algorithm_fbe_nama(int, double *, double *, double *) = 0;
    virtual int lbfgs_state(int, int *, int *, double *, double *) = 0;
    virtual int lbfgs_column(int, int, int, double *, size_t) = 0;
    // multi-GPU through the boundary
    virtual int set_allreduce(rn_allreduce_fn, void *) = 0;
    virtual int shard_info(int *) = 0;
    virtual int shard_global_nodes(int *, size_t) = 0;
    virtual void set_global_nodes(const int *, int, int) = 0;
    virtual int device_ordinal() const = 0;
    virtual int join_local_group(struct LocalGroup *, int) = 0;
    virtual int guard_check(long *) = 0;
    virtual int memory_info(size_t *) = 0;
    virtual int reserve_iterations(int) = 0;
    virtual int profile_read_collective(double *, long *) = 0;
    virtual int inject_allocation(size_t) = 0;
    virtual int guard_poke(int) = 0;
    virtual int peer_inbox_create(void *) = 0;
    virtual int peer_inbox_connect(const void *, int) = 0;
    virtual int peer_inbox_connect_local(CtxBase **, int) = 0;
    virtual unsigned long long *peer_inbox_ptr() = 0;
    virtual int set_exchange_transport(int) = 0;
    virtual int exchange_autotune_api(int, double *) = 0;
    virtual int exchange_prepare_api() = 0;
    virtual int set_fused_walk_dual(int) = 0;
    virtual int set_knob(int, int) = 0;
    virtual int debug_peer_seq(unsigned int) = 0;
    virtual int fbe_counters(long *) = 0;
};

// ---- in-process stand-in for the communicator (rn_debug_local_group_*): `n` contexts of one process, one host thread each ----
struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    bool broken = false;
    int timeoutSeconds = 120;   // $RAPIDNET_GROUP_TIMEOUT_S at creation (tests of the failure path use a short one)
    std::vector<void *> bufs;
    std::vector<size_t> counts;
    std::vector<int> f64;
    // false: a rank did not arrive in time (or the group is already broken) -- every waiter gives up with an error
    bool barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const long g = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(timeoutSeconds), [&] { return gen != g || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
};
struct LocalMember { LocalGroup *g = nullptr; int rank = 0; void *recv = nullptr; size_t recvBytes = 0; };
constexpr int LOCAL_GROUP_MAX = 16;
struct PeerPtrs { const void *p[LOCAL_GROUP_MAX]; };
template <typename T>
__global__ void k_sum_ranks(T *out, PeerPtrs peers, int n, size_t count, int op) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        T s = reinterpret_cast<const T *>(peers.p[0])[i];
        for (int r = 1; r < n; r++) {                                    // rank order: the same bits on every rank
            const T v = reinterpret_cast<const T *>(peers.p[r])[i];
            s = op == 2 ? (v > s ? v : s) : s + v;
        }
        out[i] = s;
    }
}
static int local_allreduce(void *user, void *devBuf, size_t count, int isF64, int op, void *streamHandle) {
    LocalMember *me = static_cast<LocalMember *>(user);
    LocalGroup *g = me->g;
    hipStream_t stream = (hipStream_t)streamHandle;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;          // this rank's payload is complete
    g->bufs[me->rank] = devBuf; g->counts[me->rank] = count; g->f64[me->rank] = isF64;
    if (!g->barrier()) return 2;                                        // ... and so is everybody else's
    for (int r = 0; r < g->n; r++)
        if (g->counts[r] != count || g->f64[r] != isF64) { std::lock_guard<std::mutex> lk(g->m); g->broken = true; g->cv.notify_all(); return 3; }
    const size_t bytes = count * (isF64 ? 8 : 4);
    if (bytes > me->recvBytes) {
        if (me->recv) (void)hipFree(me->recv);
        me->recv = nullptr; me->recvBytes = 0;
        if (hipMalloc(&me->recv, bytes * 2) != hipSuccess) return 4;
        me->recvBytes = bytes * 2;
    }
    PeerPtrs pp{};
    for (int r = 0; r < g->n; r++) pp.p[r] = g->bufs[r];
    const int blocks = (int)std::min<size_t>(256, (count + 255) / 256);
    if (isF64) hipLaunchKernelGGL(k_sum_ranks<double>, dim3(blocks), dim3(256), 0, stream, (double *)me->recv, pp, g->n, count, op);
    else hipLaunchKernelGGL(k_sum_ranks<float>, dim3(blocks), dim3(256), 0, stream, (float *)me->recv, pp, g->n, count, op);
    if (hipStreamSynchronize(stream) != hipSuccess) return 5;
    if (!g->barrier()) return 6;                                        // nobody still reads this rank's payload
    if (hipMemcpyAsync(devBuf, me->recv, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 7;
    return 0;
}

// dense host helpers (fp64, column-major) ---------------------------------------------------------------
static void h_gemm(bool ta, bool tb, int m, int n, int k, const double *A, int lda, const double *B, int ldb, double *C) {
    for (int j = 0; j < n; j++)
        for (int i = 0; i < m; i++) {
            double s = 0;
            for (int p = 0; p < k; p++)
                s += (ta ? A[p + (size_t)i * lda] : A[i + (size_t)p * lda]) * (tb ? B[j + (size_t)p * ldb] : B[p + (size_t)j * ldb]);
            C[i + (size_t)j * m] = s;
        }
}
static int h_inverse(int n, std::vector<double> A, double *Ainv) {  // LU with partial pivoting
    std::vector<int> piv(n);
    for (int k = 0; k < n; k++) {
        int p = k; double mx = std::fabs(A[k + (size_t)k * n]);
        for (int i = k + 1; i < n; i++) if (std::fabs(A[i + (size_t)k * n]) > mx) { mx = std::fabs(A[i + (size_t)k * n]); p = i; }
        piv[k] = p;
        if (mx == 0.0 || !std::isfinite(mx)) return k + 1;
        if (p != k) for (int j = 0; j < n; j++) std::swap(A[k + (size_t)j * n], A[p + (size_t)j * n]);
        const double d = A[k + (size_t)k * n];
        for (int i = k + 1; i < n; i++) A[i + (size_t)k * n] /= d;
        for (int j = k + 1; j < n; j++) { const double akj = A[k + (size_t)j * n]; for (int i = k + 1; i < n; i++) A[i + (size_t)j * n] -= A[i + (size_t)k * n] * akj; }
    }
    std::vector<double> b(n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) b[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(b[k], b[piv[k]]);
        for (int k = 0; k < n; k++) { const double bk = b[k]; for (int i = k + 1; i < n; i++) b[i] -= A[i + (size_t)k * n] * bk; }
        for (int k = n - 1; k >= 0; k--) { b[k] /= A[k + (size_t)k * n]; const double bk = b[k]; for (int i = 0; i < k; i++) b[i] -= A[i + (size_t)k * n] * bk; }
        for (int i = 0; i < n; i++) Ainv[i + (size_t)c * n] = b[i];
    }
    return 0;
}

template <typename T>
struct Ctx : CtxBase {
    rn_dims d{};
    int ny = 0, LD = 0, device = 0, numCUs = 256;
    size_t strideA = 0;      // values between the operator blocks of consecutive nodes (>= ny * LD, whole cache lines)
    hipStream_t stream = nullptr;
    bool factored = false, affine_ready = false;
    // host copies of the tree and of the shared factors (fp64)
    std::vector<int> h_stageCum, h_parent, h_childStart, h_childCount, h_stageOf;
    std::vector<double> h_prob, h_Rinv, h_Bbt, h_diag /* [N][nu|nx|nx] as given */;
    // device allocations
    std::vector<void *> allocs;
    int *d_stageCum = nullptr, *d_parent = nullptr, *d_childStart = nullptr, *d_childCount = nullptr, *d_stageOf = nullptr;
    T *d_sqrtp = nullptr, *d_prob = nullptr, *d_dy = nullptr;
    T *d_A = nullptr, *d_Rinv = nullptr, *d_Bbt = nullptr, *d_L = nullptr, *d_B = nullptr, *d_Lt = nullptr, *d_WLt = nullptr;
    T *d_T1 = nullptr, *d_T2 = nullptr, *d_Gd = nullptr, *d_Lhat = nullptr, *d_alpha1 = nullptr, *d_blo = nullptr, *d_bhi = nullptr;
    T *d_errD = nullptr, *d_errP = nullptr, *d_dhat = nullptr, *d_ahat = nullptr;
    T *d_curX = nullptr, *d_prevU = nullptr, *d_prevUhat = nullptr, *d_prevD = nullptr;
    T *d_beta = nullptr, *d_uhat = nullptr, *d_e = nullptr, *d_alpha = nullptr;
    T *d_x = nullptr, *d_u = nullptr, *d_v = nullptr, *d_hx = nullptr;
    T *d_my = nullptr, *d_qa = nullptr, *d_sk = nullptr, *d_rkq = nullptr, *d_lvb = nullptr, *d_eb = nullptr, *d_bw0 = nullptr, *d_bw = nullptr;
    T *d_LBLp = nullptr;
    bool aux_dirty = true;   // eb = e + B uhat and bw0 = B (prevU - prevUhat) must be refreshed before the next sweep
    T *d_RTp = nullptr, *d_Lp = nullptr, *d_Bp = nullptr, *d_BLp = nullptr, *d_ab = nullptr;
    // Operator storage (rn_set_operator_mode).  opsMode is what the caller asked for, `structured` what the context runs: RN_OPS_AUTO
    // (the default) is structured for as long as every block is the factor step's own -- block = shared matrix x stage diagonal x power
    // of p_i, Engine.cu:721-745 -- and becomes dense the moment a caller hands in a block of its own (rn_set_operator: materialise_dense)
    int opsMode = RN_OPS_AUTO, structured = 1, warmStart = 0;
    struct SavedSystem { std::vector<double> B, Gd, L, Lhat, W, diag, xmin, xmax, xsafe, umin, umax, alpha1; } h_sys;   // the factor step's inputs (AUTO: for the dense re-factor)
    int optimistic = 1;      // multi-GPU: 1 = one collective per iteration + verification, 0 = exact two-collective path
    bool pendingFin = false; // optimistic exchange: the previous iteration's bookkeeping has not been launched yet (it rides in the next k_cut_partial_sums)
    bool carryTail = false;  // the cut payload carries 2 extra reals (rank-local dist^2 of the previous iteration)
    T *d_ck[3] = {nullptr, nullptr, nullptr};   // checkpoint of (y, y+, w) for the exact fallback
    long fallbacks = 0;          // optimistic batches that had to be replayed through the exact path
    long optBatches = 0, exactBatches = 0;
    bool inReplay = false;
    int optHold = 0;             // batches still to run through the exact path after a replay (back-off: a state that violates its
                                 // soft bounds would otherwise pay checkpoint + replay on every batch of every control step)
    std::vector<double> h_T1, h_T2, h_Lt;   // zero-padded (rows % 16, cols % 4) copies for the MFMA GEMMs
    int chainStage = 0;
    int *h_verdict = nullptr;    // host-mapped word the last launch of a single-GPU optimistic batch writes the batch's verdict into (nullptr: not granted -- a copy is used)
    bool hxUnscaled = false;     // d_hx holds the primal values of every node (k_down_chain<T, true>): the next dual update applies sqrt(p_i) d_k (k_dual_stage SCALE)
    bool unscaled_on() const { return knob[RN_KNOB_UNSCALED_WALK] != 0; }   // inner iterations of optimistic batches take that pair of kernels (default on)
    // rn_debug_set_knob (include/rapidnet_debug.h): launch-shape choices the library otherwise makes by problem size, forced by tests and A/B tools
    // on trees that would not take them by themselves; -1 = the library's own choice.  Before the factor step only.
    int knob[RN_KNOB_COUNT];
    int set_knob(int k, int v) override {
        RN_CHECK(k >= 0 && k < RN_KNOB_COUNT, RN_E_ARG, "rn_debug_set_knob: unknown knob");
        RN_CHECK(!factored, RN_E_STATE, "rn_debug_set_knob must precede rn_factor_step");
        knob[k] = v;
        if (k == RN_KNOB_DUAL_TRIPS || k == RN_KNOB_DUAL_PIPE) dual_stage_setup();
        return RN_OK;
    }
    void hx_scale_now() {        // (safety net: a consumer of Hx other than k_dual_stage SCALE behind an unscaled walk)
        const long long total = ntot();
        const int blocks = (int)std::min<long long>((total + ELT_THREADS - 1) / ELT_THREADS, (long long)numCUs * 16);
        hipLaunchKernelGGL(k_hx_scale<T>, dim3(blocks), dim3(ELT_THREADS), 0, stream, d_hx, d_sqrtp, d_dy, d_stageOf, ny, total);
        hxUnscaled = false;
    }
    T *d_lo = nullptr, *d_hi = nullptr, *d_z = nullptr, *d_res = nullptr;
    T *d_ybuf[2] = {nullptr, nullptr}, *d_wbuf[2] = {nullptr, nullptr};
    T *d_tmp = nullptr;  // nodes*max(2nx,nu) staging for reference-layout get/set
    T *d_cut = nullptr;  // multi-GPU all-reduce payload
    T *d_momE = nullptr, *d_momP = nullptr;  // multi-GPU: children moments of the cut parents (static tree data)
    bool moments_set = false;
    // logical views (see DESIGN.md "iterate buffers")
    T *p_xi = nullptr, *p_upd = nullptr, *p_acc = nullptr, *p_acc_other = nullptr, *p_acc_view = nullptr;
    bool acc_ready = false;
    IterState *d_state = nullptr;
    Partial *d_partials = nullptr, *d_partials2 = nullptr;   // main pass / fix-up pass
    double *d_lam = nullptr, *d_hist = nullptr, *d_histParts = nullptr, *d_dist2 = nullptr;
    double *d_histGlob = nullptr;   // sharded: payload of the per-batch MAX all-reduce (verdict + the batch's history entries)
    int lamCap = 0, histCap = 0;
    int h_it = 0;
    double theta0 = 1, theta1 = 1;
    std::vector<double> h_lam;
    double stepSize = 1e-4, penX = 1e6, penXs = 1e4, wEco = 1.0;
    int useErrD = 1, useErrP = 1;
    int eltBlocks = 1;
    DualStageShape dshape{};   // k_dual_stage launch shape (dual_stage_setup)
    int dualU = 0;             // vectors a k_dual_stage thread keeps in flight; 0: shape not eligible, the flat k_dual_fused runs
    int dualBlocks = 1;        // workgroups (= partials) of the main pass of the fused dual update
    int mainPartials = 1;      // partials the most recent main pass left in d_partials (k_dual_stage or k_dual_fused)
    // profiling
    int prof = 0;
    struct EvPair { int cls; hipEvent_t a, b; };
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> freeEvents;
    double prof_ms[5] = {0, 0, 0, 0, 0};   // classes 0-3 as in rn_profile_read; 4 = the collectives (rn_profile_read_collective)
    long prof_n[5] = {0, 0, 0, 0, 0};
    // multi-GPU
    void *comm = nullptr;
    int rank = 0, nranks = 1, cutStage = -1;
    rn_allreduce_fn arHook = nullptr;   // rn_debug_set_allreduce: called in place of ncclAllReduce
    void *arUser = nullptr;
    LocalMember *localMember = nullptr; // rn_debug_local_group_join (owned)
    std::vector<int> globalNode;        // sharded through rn_create_sharded: index of every local node in the full tree
    int fullNodes = 0;
    bool has_comm() const { return comm != nullptr || arHook != nullptr; }
    // sum all-reduce, in place, on the solver's stream: the library's RCCL communicator, or the installed stand-in
    int all_reduce(void *buf, size_t count, bool f64, const char *what, int op = 0 /* ncclSum; 2 = ncclMax */) {
        if (arHook) {
            hipEvent_t ev = prof_begin(4);
            const int rc = arHook(arUser, buf, count, f64 ? 1 : 0, op, (void *)stream);
            prof_end(ev);
            RN_CHECK(rc == 0, RN_E_COMM, std::string(what) + ": the installed all-reduce callback failed (" + std::to_string(rc) + ")");
            return RN_OK;
        }
        RN_CHECK(comm != nullptr, RN_E_STATE, std::string(what) + ": no communicator");
        hipEvent_t ev = prof_begin(4);
        const int rc = g_nccl.AllReduce(buf, buf, count, f64 ? 8 /*ncclFloat64*/ : 7 /*ncclFloat32*/, op, comm, stream);
        prof_end(ev);
        RN_CHECK(rc == 0, RN_E_COMM, std::string(what) + " failed: " + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "?"));
        return RN_OK;
    }

    ~Ctx() override {
        g_liveContexts--;
        if (commJob) { std::lock_guard<std::mutex> lk(commJob->m); commJob->abandoned = true; }   // a helper still inside ncclCommInitRank destroys what it gets
        if (comm && g_nccl.CommDestroy) g_nccl.CommDestroy(comm);
        (void)hipSetDevice(device);
        if (guardMode && stream) {   // guard mode: a context never goes away with an overwritten red zone unnoticed
            long bad = 0;
            if (guard_check(&bad) == RN_OK) { g_guardContexts++; g_guardBadBytes += bad; if (bad) fprintf(stderr, "librapidnet_hip: %s\n", err.c_str()); }
        }
        if (localMember) { if (localMember->recv) (void)hipFree(localMember->recv); delete localMember; }
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto &p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        for (auto e : freeEvents) (void)hipEventDestroy(e);
        for (void *p : ipcOpened) (void)hipIpcCloseMemHandle(p);
        if (h_verdict) (void)hipHostFree(h_verdict);
        if (d_inbox) (void)hipFree(d_inbox);
        for (void *p : allocs) (void)hipFree(p);
        if (evFork) (void)hipEventDestroy(evFork);
        if (evJoin) (void)hipEventDestroy(evJoin);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }

    // ---- device allocations ---------------------------------------------------------------------------------
    // Every buffer of the context comes from here.  RAPIDNET_GUARD=1 (read when the context is created) is the library's
    // stand-in for a GPU address sanitizer, which this pool does not offer: every buffer gets a red zone of GUARD_BYTES on both
    // sides, and red zones AND payload of floating-point buffers start out as 0xFF bytes -- a NaN in fp64 and in fp32 -- so a
    // kernel that reads outside its buffer, or reads what nobody has written, carries a NaN into the iterates (the parity
    // tests then fail); integer tables get zero red zones (an index read from one stays in range).  A kernel that WRITES
    // outside its buffer changes a red zone: rn_guard_check / rn_destroy compare them with the pattern.
    static constexpr size_t GUARD_BYTES = 128 * 1024;
    struct GuardRec { void *base; size_t bytes; unsigned char pat; };
    std::vector<GuardRec> guards;
    bool guardMode = false;
    size_t allocatedBytes = 0;   // payload bytes of all live allocations of this context (rn_device_memory_info)
    template <typename U> int dalloc(U **p, size_t n) {
        void *q = nullptr;
        const size_t bytes = (n ? n : 1) * sizeof(U);
        if (!guardMode) {
            RN_HIP(hipMalloc(&q, bytes));
            allocs.push_back(q);
            allocatedBytes += bytes;
            *p = (U *)q;
            return RN_OK;
        }
        const size_t padded = (bytes + 255) / 256 * 256;
        RN_HIP(hipMalloc(&q, padded + 2 * GUARD_BYTES));
        allocs.push_back(q);
        allocatedBytes += bytes;
        const unsigned char pat = std::is_floating_point<U>::value || std::is_class<U>::value ? 0xFF : 0x00;
        // filled on the context's OWN stream and waited for: the stream is non-blocking (no implicit ordering with the null
        // stream), and everything that touches the buffer afterwards -- uploads on either stream, kernels -- must come after the fill
        RN_HIP(hipMemsetAsync(q, pat, padded + 2 * GUARD_BYTES, stream));
        RN_HIP(hipStreamSynchronize(stream));
        guards.push_back(GuardRec{q, bytes, pat});
        *p = (U *)((char *)q + GUARD_BYTES);
        return RN_OK;
    }
    // number of red-zone bytes that no longer hold their pattern (0 outside guard mode)
    int guard_check(long *bad) override {
        RN_CHECK(bad, RN_E_ARG, "rn_guard_check: null output");
        *bad = 0;
        if (!guardMode) return RN_OK;
        RN_HIP(hipSetDevice(device));
        RN_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned char> h(GUARD_BYTES + 256);
        for (const GuardRec &g : guards) {
            const size_t padded = (g.bytes + 255) / 256 * 256;
            RN_HIP(hipMemcpy(h.data(), g.base, GUARD_BYTES, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < GUARD_BYTES; i++) if (h[i] != g.pat) (*bad)++;
            // the back red zone starts right behind the payload (the round-up to 256 bytes belongs to it)
            const size_t back = GUARD_BYTES + (padded - g.bytes);
            RN_HIP(hipMemcpy(h.data(), (char *)g.base + GUARD_BYTES + g.bytes, back, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < back; i++) if (h[i] != g.pat) (*bad)++;
        }
        if (*bad) err = "rn_guard_check: " + std::to_string(*bad) + " red-zone bytes were overwritten (a kernel wrote outside its buffer)";
        return RN_OK;
    }
    // test of the detector: `n` bytes right behind the payload of the context's first buffer are overwritten
    int guard_poke(int n) override {
        RN_CHECK(guardMode && !guards.empty(), RN_E_STATE, "rn_debug_guard_poke: the context was not created under RAPIDNET_GUARD=1");
        RN_CHECK(n >= 1 && n <= 256, RN_E_ARG, "rn_debug_guard_poke: 1 .. 256 bytes");
        RN_HIP(hipSetDevice(device));
        RN_HIP(hipStreamSynchronize(stream));
        RN_HIP(hipMemset((char *)guards[0].base + GUARD_BYTES + guards[0].bytes, 0xA5, (size_t)n));
        return RN_OK;
    }
    // free / total bytes of the device and the bytes this context holds: what cudaMemGetInfo reports in the reference's leak check
    // (SmpcController.cu:1612, :1619) plus a figure that other contexts on the same device cannot disturb
    int memory_info(size_t *out) override {
        RN_CHECK(out, RN_E_ARG, "rn_device_memory_info: null output");
        RN_HIP(hipSetDevice(device));
        size_t fr = 0, tot = 0;
        RN_HIP(hipMemGetInfo(&fr, &tot));
        out[0] = fr; out[1] = tot; out[2] = allocatedBytes; out[3] = (size_t)g_liveContexts.load();
        return RN_OK;
    }
    int upload(T *dst, const double *src, size_t n) {
        std::vector<T> tmp(n);
        for (size_t i = 0; i < n; i++) tmp[i] = (T)src[i];
        RN_HIP(hipMemcpyAsync(dst, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice, stream));
        RN_HIP(hipStreamSynchronize(stream));  // tmp goes out of scope
        return RN_OK;
    }
    int download(double *dst, const T *src, size_t n) {
        std::vector<T> tmp(n);
        RN_HIP(hipMemcpyAsync(tmp.data(), src, n * sizeof(T), hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < n; i++) dst[i] = (double)tmp[i];
        return RN_OK;
    }
    int upload_int(int *dst, const std::vector<int> &v) {
        RN_HIP(hipMemcpy(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
        return RN_OK;
    }

    static int pad16(int v) { return (v + 63) / 64 * 64; }   // rows padded to one wave tile (GEMM_RT * 16)
    static int pad4(int v) { return (v + 4 * RN_SLAB_KU - 1) / (4 * RN_SLAB_KU) * (4 * RN_SLAB_KU); }   // pad_k: K of the shared operators in whole groups of the MFMA loop (zero columns)
    int upload_padded(T *dst, const double *src, int m, int k) {   // col-major m x k -> zero-padded pad16(m) x pad4(k)
        const int mp = pad16(m), kp = pad4(k);
        std::vector<double> tmp((size_t)mp * kp, 0.0);
        for (int j = 0; j < k; j++) for (int i = 0; i < m; i++) tmp[i + (size_t)j * mp] = src[i + (size_t)j * m];
        return upload(dst, tmp.data(), tmp.size());
    }
    // the shared operators of the slab products once more in MFMA fragment order (kernels.hpp, GemmArgs::Mf): [16-row tile][pair of k-steps][lane][2],
    // zero-padded like the column-major copies (pad16(m) rows, pad4(k) columns): a wave's A operands of two k-steps are one contiguous request
    T *d_RTf = nullptr, *d_LBLf = nullptr, *d_BLf = nullptr;
    // structured mode, linear form of the leaf-to-root recursion (k_walks.hpp, k_up_chain_lin): the operator [Rinv | T1 | T2] of the v product
    // (column-major padded, and in fragment order) and the running sums' buffers; allocated by the factor step of a structured context
    T *d_RT2p = nullptr, *d_RT2f = nullptr, *d_sk2 = nullptr, *d_rkq2 = nullptr;
    // Bs_i = beta_i + sum_c Bs_c only changes with the affine terms (once per control step): its term of v_i, c_i = -Rinv Bs_i / (2 p_i), is computed by
    // lin_const_refresh when they change and enters the v product as its epilogue operand -- the product itself is [T1 | T2] [q_i + kappa_i ; Bu_i],
    // K = nx + nu instead of nv + nx + nu, and the walks leave the Bs columns alone.  rn_debug_set_knob(RN_KNOB_STRUCT_LINEAR, 3): Bs in every iteration's product.
    T *d_T12p = nullptr, *d_T12f = nullptr, *d_vconst = nullptr;
    bool linConstValid = false;
    bool lin_const() const { return knob[RN_KNOB_STRUCT_LINEAR] != 3; }
    // ... and since v_i then only feeds [L v_i ; B L v_i], the two products are one with the composite operator MC = [L ; B L] [T1 | T2] (k_gemm_comp);
    // d_lvconst = [L ; B L] c_i.  RN_KNOB_STRUCT_LINEAR 4: the two products kept apart.
    T *d_MCp = nullptr, *d_MCf = nullptr, *d_lvconst = nullptr;
    bool lin_comp() const { return lin_const() && knob[RN_KNOB_STRUCT_LINEAR] != 4 && d_MCp != nullptr; }
    // ... and the forward walk's affine terms (uhat_i - uhat_anc, eb_i - eb_anc) join that constant (k_fold_affine), so the walk requests neither
    // uhat nor eb (SweepArgs::lin bit 2).  RN_KNOB_STRUCT_LINEAR 5: the composite operator without them.
    bool lin_fold() const { return lin_comp() && knob[RN_KNOB_STRUCT_LINEAR] != 5; }
    // v_i is an output only in this form (nothing of the sweep reads it): the sweeps that store the primal iterates leave it PENDING and the product
    // runs when somebody asks for RN_BUF_V (rn_get / rn_set / rn_get_range / rn_set_range) -- from the product's input of that very sweep, which is
    // still in place: a later sweep drops the pending product (its state is no longer observable) or, a Hessian sweep, runs it first.  A context
    // whose raw v pointer has been handed out (rn_device_pointer) computes v with every such sweep, as the pointer's contract says.
    bool vPending = false, vEager = false;
    int v_flush() {
        if (!vPending) return RN_OK;
        vPending = false;
        launch_gemm<EPI_V>(d_T12p, d.nv, d.nx + d.nu, d_sk2 + d.nv, d.nv + d.nx + d.nu, d_v, d.nv, d_vconst, d.nv);
        RN_HIP(hipGetLastError());
        return RN_OK;
    }
    // the form applies to unsharded structured sweeps whose v / Lv slab (16 nodes x (nv + nx + nu) and 16 x nv values) fits a workgroup's 64 KB;
    // otherwise (the wide fp32 network) the structured sweep keeps its first product k_gemm_prep_m2
    bool lin_fits() const { return (size_t)16 * (slab_stride(pad4(d.nv + d.nx + d.nu)) + slab_stride(pad4(d.nv))) * sizeof(T) <= 64 * 1024; }
    bool lin_on() const { return structured && d_sk2 != nullptr && knob[RN_KNOB_STRUCT_LINEAR] != 0; }
    bool frag_on() const { return knob[RN_KNOB_SLAB_FRAG] != 0; }   // the slab products take their A operands from the fragment-ordered copies (default on)
    int upload_fragments(T *dst, const double *src, int m, int k) {
        const int mp = pad16(m), kp = pad4(k), tiles = mp / 16, pairs = kp / 8;
        std::vector<double> tmp((size_t)mp * kp, 0.0);
        for (int t = 0; t < tiles; t++)
            for (int p2 = 0; p2 < pairs; p2++)
                for (int l = 0; l < 64; l++)
                    for (int e = 0; e < 2; e++) {
                        const int row = t * 16 + (l & 15), col = 4 * (2 * p2 + e) + (l >> 4);
                        if (row < m && col < k) tmp[(((size_t)t * pairs + p2) * 64 + l) * 2 + e] = src[row + (size_t)col * m];
                    }
        return upload(dst, tmp.data(), tmp.size());
    }
    TreeDev<T> tree_dev() const { return TreeDev<T>{d_stageCum, d_parent, d_childStart, d_childCount, d_stageOf, d_sqrtp, d_prob, d_dy}; }
    SweepArgs<T> sweep_args() const {
        SweepArgs<T> a{};
        a.tr = tree_dev();
        a.nx = d.nx; a.nu = d.nu; a.nv = d.nv; a.ny = ny; a.LD = LD; a.strideA = strideA; a.N = d.N; a.nodes = d.nodes;
        a.cutSums = (cutStage > 0) ? d_cut : nullptr;
        a.s1 = h_stageCum.size() > 1 ? h_stageCum[1] : d.nodes; a.e1 = h_stageCum.size() > 2 ? h_stageCum[2] : d.nodes;
        a.rootC0 = h_childStart.empty() ? 0 : h_childStart[0]; a.rootNc = h_childCount.empty() ? 0 : h_childCount[0];
        a.cutStage = cutStage;
        a.chainStage = a.cutSums ? std::max(chainStage, cutStage) : chainStage;
        a.K = h_stageCum[a.chainStage + 1] - h_stageCum[a.chainStage];
        a.A = d_A; a.RT = d_RTp; a.L = d_L; a.B = d_B; a.structured = structured; a.ab = d_ab;
        a.lin = (lin_on() && cutStage <= 0) ? 1 : 0; a.sk2 = d_sk2; a.rkq2 = d_rkq2;
        a.beta = d_beta; a.uhat = d_uhat; a.e = d_e; a.curX = d_curX; a.prevU = d_prevU; a.prevUhat = d_prevUhat;
        a.w = p_acc;
        a.my = d_my; a.my2 = d_my2; a.splitFirst = (splitFirst >= 0 && !structured) ? splitFirst : d.nodes; a.qa = d_qa; a.sk = d_sk; a.rkq = d_rkq; a.v = d_v; a.lvb = d_lvb; a.eb = d_eb; a.bw0 = d_bw0; a.bw = d_bw;
        a.x = d_x; a.u = d_u; a.hx = d_hx;
        a.distTail = (carryTail && a.cutSums) ? d_cut + cut_tail_offset() : nullptr;
        a.thrX = penX / stepSize; a.thrS = penXs / stepSize; a.iterState = d_state;
        a.writePrimal = 1;
        a.chain0 = h_stageCum[a.chainStage]; a.chainAnc = chainAncStage == a.chainStage ? d_chainAnc : nullptr;
        a.cut0 = cutStage > 0 ? h_stageCum[cutStage - 1] : 0;
        return a;
    }
    // The crown path of every chain as a table (SweepArgs::chainAnc; k_down_chain / k_down_chain_dual with foldCrown): built for the stage the chains
    // start at -- the tree's own, or the cut stage of a shard -- the first time a sweep folds the crown into the forward walk, and again if that stage changes.
    int *d_chainAnc = nullptr;
    int chainAncStage = -1;
    int ensure_chain_anc(int cs) {
        if (chainAncStage == cs && d_chainAnc) return RN_OK;
        RN_CHECK(cs >= 0 && cs <= CROWN_MAX_DEPTH && cs + 1 < (int)h_stageCum.size(), RN_E_STATE, "ensure_chain_anc: the crown is deeper than the folded walk supports");
        const int K = h_stageCum[cs + 1] - h_stageCum[cs];
        std::vector<int> tab((size_t)K * CROWN_MAX_DEPTH, 0);
        for (int c = 0; c < K; c++) {
            int n = h_stageCum[cs] + c;
            bool first = true;
            for (int dd = 0; dd < cs; dd++) {
                const int p = h_parent[n];
                first = first && h_childStart[p] == n;
                tab[(size_t)c * CROWN_MAX_DEPTH + dd] = p | (first ? (1 << 30) : 0);
                n = p;
            }
        }
        if (!d_chainAnc) {      // room for the widest stage, once (rn_create): a control step never allocates (the reference's leak check, SmpcController.cu:1593-1667)
            int widest = 1;
            for (size_t k = 0; k + 1 < h_stageCum.size(); k++) widest = std::max(widest, h_stageCum[k + 1] - h_stageCum[k]);
            if (int rc = dalloc(&d_chainAnc, (size_t)widest * CROWN_MAX_DEPTH)) return rc;
        }
        RN_HIP(hipStreamSynchronize(stream));   // (a sweep in flight may still read the previous table)
        RN_HIP(hipMemcpy(d_chainAnc, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        chainAncStage = cs;
        return RN_OK;
    }
    long long ntot() const { return (long long)d.nodes * ny; }

    // ---- construction --------------------------------------------------------------------------------
    int init(const rn_dims *dims, const rn_tree *tr, int dev) {
        d = *dims; device = dev;
        g_liveContexts++;
        for (int &k : knob) k = -1;
        if (const char *e = std::getenv("RAPIDNET_EXCHANGE")) {   // default transport of new contexts: collective | oneshot | auto
            const std::string v(e);
            if (v == "collective" || v == "rccl") { transportReq = RN_EXCHANGE_COLLECTIVE; tuned = true; }
            else if (v == "oneshot") { transportReq = RN_EXCHANGE_ONESHOT; tuned = true; }      // (takes effect once the inboxes are connected: exchange_prepare)
        }
        { const char *e = std::getenv("RAPIDNET_GUARD"); guardMode = e && std::atoi(e) != 0; }
        RN_CHECK(d.nx > 0 && d.nu > 0 && d.nv > 0 && d.nd > 0 && d.N > 0 && d.K > 0 && d.nodes > 0, RN_E_ARG, "rn_create: non-positive dimension");
        RN_CHECK(d.nv <= d.nu, RN_E_ARG, "rn_create: nv must not exceed nu");
        ny = 2 * d.nx + d.nu;
        {   // columns are whole 16-byte slots, node blocks whole 128-byte lines (k_stream_gemv walks a block in slots and
            // its spans are only line-aligned if the block is)
            const int vps = 16 / (int)sizeof(T), vpl = 128 / (int)sizeof(T);
            LD = (2 * d.nv + vps - 1) / vps * vps;
            strideA = ((size_t)ny * LD + vpl - 1) / vpl * vpl;
        }
        RN_HIP(hipSetDevice(device));
        RN_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) numCUs = cu; }
        if (hipHostMalloc((void **)&h_verdict, sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); h_verdict = nullptr; } else *h_verdict = 0;
        {   // The library's device code is six code objects (one per kernel unit) and the HIP runtime loads a code object when its first kernel is
            // looked up: asked for here, so that no later call -- a control step under the caller's leak check (SmpcController.cu:1612-1623) --
            // is the one during which the device's free memory shrinks by a code object
            hipFuncAttributes fa;
            for (const void *fn : {(const void *)k_stream_gemv<T, 1, false>, (const void *)k_dual_fused<T, false, false>, (const void *)k_up_chain<T, false>,
                                   (const void *)k_gemm_slab<T, EPI_Z, false>, (const void *)k_pack<T>, (const void *)k_dots<T>})
                if (hipFuncGetAttributes(&fa, fn) != hipSuccess) (void)hipGetLastError();
        }
        const int N = d.N, nodes = d.nodes;
        // validate and convert the tree (reference conventions -> 0-based parent/children ranges)
        h_stageCum.assign(tr->nodesPerStageCumul, tr->nodesPerStageCumul + N + 1);
        RN_CHECK(h_stageCum[0] == 0 && h_stageCum[N] == nodes, RN_E_ARG, "rn_create: nodesPerStageCumul inconsistent with nodes");
        h_parent.resize(nodes); h_childStart.assign(nodes, 0); h_childCount.assign(nodes, 0); h_stageOf.resize(nodes); h_prob.resize(nodes);
        for (int k = 0; k < N; k++) {
            RN_CHECK(tr->nodesPerStage[k] == h_stageCum[k + 1] - h_stageCum[k] && tr->nodesPerStage[k] > 0, RN_E_ARG, "rn_create: nodesPerStage inconsistent");
            for (int i = h_stageCum[k]; i < h_stageCum[k + 1]; i++) {
                RN_CHECK(tr->stages[i] == k, RN_E_ARG, "rn_create: nodes are not numbered stage by stage");
                h_stageOf[i] = k;
            }
        }
        RN_CHECK(h_stageCum[1] == 1 && tr->ancestor[0] == 0, RN_E_ARG, "rn_create: node 0 must be the only root");
        h_parent[0] = -1;
        for (int i = 0; i < nodes; i++) {
            h_prob[i] = tr->probNode[i];
            RN_CHECK(h_prob[i] > 0.0 && std::isfinite(h_prob[i]), RN_E_ARG, "rn_create: probNode must be positive");
            if (i == 0) continue;
            const int par = tr->ancestor[i] - 1;
            RN_CHECK(par >= 0 && par < i && h_stageOf[par] == h_stageOf[i] - 1, RN_E_ARG, "rn_create: ancestor must be a node of the previous stage");
            RN_CHECK(par >= h_parent[i - 1] || h_stageOf[i - 1] != h_stageOf[i], RN_E_ARG, "rn_create: children of a node must be contiguous");
            h_parent[i] = par;
            if (h_childCount[par] == 0) h_childStart[par] = i;
            RN_CHECK(h_childStart[par] + h_childCount[par] == i, RN_E_ARG, "rn_create: children of a node must be contiguous");
            h_childCount[par]++;
        }
        int leaves = 0, nonleaf = 0;
        for (int i = 0; i < nodes; i++) (h_childCount[i] == 0 ? leaves : nonleaf)++;
        RN_CHECK(nonleaf == d.nNonLeafNodes, RN_E_ARG, "rn_create: nNonLeafNodes inconsistent with ancestor[]");
        (void)leaves;
        // c*: first stage from which the tree is K parallel chains (every node has exactly one child, same position)
        chainStage = N - 1;
        while (chainStage > 0) {
            const int k = chainStage - 1;
            bool chain = (h_stageCum[k + 1] - h_stageCum[k]) == (h_stageCum[k + 2] - h_stageCum[k + 1]);
            for (int i = h_stageCum[k]; chain && i < h_stageCum[k + 1]; i++) chain = (h_childCount[i] == 1);
            if (!chain) break;
            chainStage--;
        }
        if (int rc = dalloc(&d_stageCum, N + 1)) return rc;
        if (int rc = dalloc(&d_parent, nodes)) return rc;
        if (int rc = dalloc(&d_childStart, nodes)) return rc;
        if (int rc = dalloc(&d_childCount, nodes)) return rc;
        if (int rc = dalloc(&d_stageOf, nodes)) return rc;
        if (int rc = upload_int(d_stageCum, h_stageCum)) return rc;
        if (int rc = upload_int(d_parent, h_parent)) return rc;
        if (int rc = upload_int(d_childStart, h_childStart)) return rc;
        if (int rc = upload_int(d_childCount, h_childCount)) return rc;
        if (int rc = upload_int(d_stageOf, h_stageOf)) return rc;
        if (chainStage <= CROWN_MAX_DEPTH) { if (int rc = ensure_chain_anc(chainStage)) return rc; }
        const size_t n = nodes;
        const int nx = d.nx, nu = d.nu, nv = d.nv, nd = d.nd;
#define DA(ptr, cnt) if (int rc = dalloc(&ptr, (size_t)(cnt))) return rc;
        DA(d_sqrtp, n) DA(d_prob, n) DA(d_dy, (size_t)N * ny)
        DA(d_Rinv, nv * nv) DA(d_Bbt, nv * nx) DA(d_L, nu * nv) DA(d_B, nx * nu) DA(d_Lt, nv * nu) DA(d_WLt, nv * nu) DA(d_W, nu * nu)
        DA(d_T1, nv * nx) DA(d_T2, nv * nu) DA(d_Gd, nx * nd) DA(d_Lhat, nu * nd) DA(d_alpha1, nu) DA(d_blo, ny) DA(d_bhi, ny)
        DA(d_errD, n * nd) DA(d_errP, n * nu) DA(d_dhat, (size_t)N * nd) DA(d_ahat, (size_t)N * nu)
        DA(d_curX, nx) DA(d_prevU, nu) DA(d_prevUhat, nu) DA(d_prevD, nd)
        DA(d_beta, n * nv) DA(d_uhat, n * nu) DA(d_e, n * nx) DA(d_alpha, n * nu)
        DA(d_x, n * nx) DA(d_u, n * nu) DA(d_v, n * nv) DA(d_hx, n * ny)
        DA(d_my, n * 2 * nv) DA(d_qa, n * nx) DA(d_sk, n * (nv + nx)) DA(d_rkq, n * (nv + 2 * nx)) DA(d_lvb, n * (nu + nx)) DA(d_eb, n * nx) DA(d_bw0, nx) DA(d_bw, n * nx)
        DA(d_LBLp, (size_t)pad16(nu + nx) * pad4(nv))
        DA(d_LBLf, (size_t)pad16(nu + nx) * pad4(nv)) DA(d_BLf, (size_t)pad16(nv) * pad4(nx + nu)) DA(d_RTf, (size_t)pad16(nv) * pad4(nv + nx))
        DA(d_BLp, (size_t)pad16(nv) * pad4(nx + nu)) DA(d_ab, n * (nx + nu))
        DA(d_RTp, (size_t)pad16(nv) * pad4(nv + nx)) DA(d_Lp, (size_t)pad16(nu) * pad4(nv)) DA(d_Bp, (size_t)pad16(nx) * pad4(nu))
        DA(d_lo, n * ny) DA(d_hi, n * ny) DA(d_z, n * ny) DA(d_res, n * ny)
        DA(d_ybuf[0], n * ny) DA(d_ybuf[1], n * ny) DA(d_wbuf[0], n * ny) DA(d_wbuf[1], n * ny)
        DA(d_tmp, n * (size_t)std::max(2 * nx, std::max(nu, nv)))
        DA(d_cut, (size_t)nodes * (nv + 2 * nx))  // upper bound on cut parents
        DA(d_tune, 8) DA(d_state, 1) DA(d_partials, std::max(ELT_MAX_BLOCKS, RN_DUAL_STAGE_MAX_BLOCKS)) DA(d_partials2, ELT_MAX_BLOCKS) DA(d_dist2, 2)
#undef DA
        std::vector<double> sq(nodes);
        for (int i = 0; i < nodes; i++) sq[i] = std::sqrt(h_prob[i]);
        if (int rc = upload(d_sqrtp, sq.data(), nodes)) return rc;
        if (int rc = upload(d_prob, h_prob.data(), nodes)) return rc;
        RN_HIP(hipMemsetAsync(d_errD, 0, n * nd * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_errP, 0, n * nu * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_z, 0, n * ny * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_res, 0, n * ny * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_x, 0, n * nx * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_u, 0, n * nu * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_v, 0, n * nv * sizeof(T), stream));
        RN_HIP(hipMemsetAsync(d_hx, 0, n * ny * sizeof(T), stream));
        const long long want = (ntot() + ELT_THREADS * 4 - 1) / (ELT_THREADS * 4);
        eltBlocks = (int)std::max<long long>(1, std::min<long long>(ELT_MAX_BLOCKS, want));
        dual_stage_setup();
        p_xi = d_ybuf[0]; p_upd = d_ybuf[1]; p_acc = d_wbuf[0]; p_acc_other = d_wbuf[1]; p_acc_view = p_acc;
        return apg_reset();
    }

    // ---- Engine ----------------------------------------------------------------------------------------
    int factor_step(const rn_system *s) override {
        RN_CHECK(s && s->matB && s->matGd && s->matL && s->matLhat && s->costW && s->matDiagPrecnd && s->vecXmin && s->vecXmax &&
                     s->vecXsafe && s->vecUmin && s->vecUmax && s->costAlpha1, RN_E_ARG, "rn_factor_step: null input");
        RN_HIP(hipSetDevice(device));
        if (int rc = v_flush()) return rc;       // (a pending v is the previous operators')
        const int nx = d.nx, nu = d.nu, nv = d.nv, nd = d.nd, N = d.N;
        if (opsMode == RN_OPS_AUTO && s->matB != h_sys.B.data()) {   // (not when materialise_dense re-runs the step on the saved copy)
            h_sys.B.assign(s->matB, s->matB + (size_t)nx * nu); h_sys.Gd.assign(s->matGd, s->matGd + (size_t)nx * nd);
            h_sys.L.assign(s->matL, s->matL + (size_t)nu * nv); h_sys.Lhat.assign(s->matLhat, s->matLhat + (size_t)nu * nd);
            h_sys.W.assign(s->costW, s->costW + (size_t)nu * nu); h_sys.diag.assign(s->matDiagPrecnd, s->matDiagPrecnd + (size_t)N * (2 * nx + nu));
            h_sys.xmin.assign(s->vecXmin, s->vecXmin + nx); h_sys.xmax.assign(s->vecXmax, s->vecXmax + nx); h_sys.xsafe.assign(s->vecXsafe, s->vecXsafe + nx);
            h_sys.umin.assign(s->vecUmin, s->vecUmin + nu); h_sys.umax.assign(s->vecUmax, s->vecUmax + nu); h_sys.alpha1.assign(s->costAlpha1, s->costAlpha1 + nu);
        }
        // Rbar = L' W L  (Engine.cu:412-416), inverse once: Omega_i = Rbar^-1 / p_i (Engine.cu:707-714)
        std::vector<double> WL((size_t)nu * nv), Rbar((size_t)nv * nv), Lt((size_t)nv * nu), WLt((size_t)nv * nu);
        h_gemm(false, false, nu, nv, nu, s->costW, nu, s->matL, nu, WL.data());
        h_gemm(true, false, nv, nv, nu, s->matL, nu, WL.data(), nu, Rbar.data());
        h_Rinv.assign((size_t)nv * nv, 0.0);
        RN_CHECK(h_inverse(nv, Rbar, h_Rinv.data()) == 0, RN_E_SINGULAR, "rn_factor_step: L'WL is singular");
        h_Bbt.assign((size_t)nv * nx, 0.0);
        h_gemm(true, true, nv, nx, nu, s->matL, nu, s->matB, nx, h_Bbt.data());  // Bbar' = L'B' (Engine.cu:702-705)
        std::vector<double> T1((size_t)nv * nx), T2((size_t)nv * nu);
        for (int i = 0; i < nu; i++) for (int j = 0; j < nv; j++) { Lt[j + (size_t)i * nv] = s->matL[i + (size_t)j * nu]; WLt[j + (size_t)i * nv] = WL[i + (size_t)j * nu]; }
        h_gemm(false, false, nv, nx, nv, h_Rinv.data(), nv, h_Bbt.data(), nv, T1.data());
        h_gemm(false, false, nv, nu, nv, h_Rinv.data(), nv, Lt.data(), nv, T2.data());
        // preconditioner diagonal re-ordered to y order: (d_x | d_xs | d_u) per stage
        h_diag.assign(s->matDiagPrecnd, s->matDiagPrecnd + (size_t)N * (2 * nx + nu));
        std::vector<double> dy((size_t)N * ny), blo(ny), bhi(ny);
        for (int k = 0; k < N; k++) {
            const double *dk = s->matDiagPrecnd + (size_t)k * (2 * nx + nu);
            for (int t = 0; t < 2 * nx; t++) dy[(size_t)k * ny + t] = dk[nu + t];
            for (int t = 0; t < nu; t++) dy[(size_t)k * ny + 2 * nx + t] = dk[t];
        }
        // "no upper bound" on the safety half: the reference memsets bytes 0x7F (Engine.cu:454-455)
        T big; std::memset(&big, 0x7F, sizeof(T));
        for (int t = 0; t < nx; t++) { blo[t] = s->vecXmin[t]; bhi[t] = s->vecXmax[t]; blo[nx + t] = s->vecXsafe[t]; bhi[nx + t] = (double)big; }
        for (int t = 0; t < nu; t++) { blo[2 * nx + t] = s->vecUmin[t]; bhi[2 * nx + t] = s->vecUmax[t]; }
#define UP(dst, src, cnt) if (int rc = upload(dst, src, (size_t)(cnt))) return rc;
        h_T1 = T1; h_T2 = T2; h_Lt = Lt;
        {   // [Bbt | L'] for the structured m2 GEMM
            std::vector<double> BL((size_t)nv * (nx + nu));
            std::copy(h_Bbt.begin(), h_Bbt.end(), BL.begin());
            std::copy(Lt.begin(), Lt.end(), BL.begin() + (size_t)nv * nx);
            if (int rc = upload_padded(d_BLp, BL.data(), nv, nx + nu)) return rc;
            if (int rc = upload_fragments(d_BLf, BL.data(), nv, nx + nu)) return rc;
        }
        std::vector<double> LBL((size_t)(nu + nx) * nv);
        {   // [L ; B L]  ((nu+nx) x nv) for the forward GEMM
            std::vector<double> BLm((size_t)nx * nv);
            h_gemm(false, false, nx, nv, nu, s->matB, nx, s->matL, nu, BLm.data());
            for (int j = 0; j < nv; j++) {
                for (int i = 0; i < nu; i++) LBL[i + (size_t)j * (nu + nx)] = s->matL[i + (size_t)j * nu];
                for (int i = 0; i < nx; i++) LBL[nu + i + (size_t)j * (nu + nx)] = BLm[i + (size_t)j * nx];
            }
            if (int rc = upload_padded(d_LBLp, LBL.data(), nu + nx, nv)) return rc;
            if (int rc = upload_fragments(d_LBLf, LBL.data(), nu + nx, nv)) return rc;
        }
        std::vector<double> RTm((size_t)nv * (nv + nx));
        std::copy(h_Rinv.begin(), h_Rinv.end(), RTm.begin());
        std::copy(T1.begin(), T1.end(), RTm.begin() + (size_t)nv * nv);
        if (int rc = upload_padded(d_RTp, RTm.data(), nv, nv + nx)) return rc;
        if (int rc = upload_fragments(d_RTf, RTm.data(), nv, nv + nx)) return rc;
        if (structured && lin_fits()) {   // [Rinv | T1 | T2]: nv x (nv + nx + nu)
            const size_t no = (size_t)pad16(nv) * pad4(nv + nx + nu);
            if (!d_RT2p) {
                if (int rc = dalloc(&d_RT2p, no)) return rc;
                if (int rc = dalloc(&d_RT2f, no)) return rc;
                if (int rc = dalloc(&d_sk2, (size_t)d.nodes * (nv + nx + nu))) return rc;
                if (int rc = dalloc(&d_rkq2, (size_t)d.nodes * (nv + 2 * nx + nu))) return rc;
                if (int rc = dalloc(&d_T12p, (size_t)pad16(nv) * pad4(nx + nu))) return rc;
                if (int rc = dalloc(&d_T12f, (size_t)pad16(nv) * pad4(nx + nu))) return rc;
                if (int rc = dalloc(&d_vconst, (size_t)d.nodes * nv)) return rc;
                if (int rc = dalloc(&d_MCp, (size_t)pad16(nu + nx) * pad4(nx + nu))) return rc;
                if (int rc = dalloc(&d_MCf, (size_t)pad16(nu + nx) * pad4(nx + nu))) return rc;
                if (int rc = dalloc(&d_lvconst, (size_t)d.nodes * (nu + nx))) return rc;
            }
            std::vector<double> RT2((size_t)nv * (nv + nx + nu));
            std::copy(RTm.begin(), RTm.end(), RT2.begin());
            std::copy(T2.begin(), T2.end(), RT2.begin() + (size_t)nv * (nv + nx));
            if (int rc = upload_padded(d_RT2p, RT2.data(), nv, nv + nx + nu)) return rc;
            if (int rc = upload_fragments(d_RT2f, RT2.data(), nv, nv + nx + nu)) return rc;
            if (int rc = upload_padded(d_T12p, RT2.data() + (size_t)nv * nv, nv, nx + nu)) return rc;       // [T1 | T2]: the columns behind Rinv
            if (int rc = upload_fragments(d_T12f, RT2.data() + (size_t)nv * nv, nv, nx + nu)) return rc;
            {   // MC = [L ; B L] [T1 | T2]: (nu + nx) x (nx + nu)
                std::vector<double> MC((size_t)(nu + nx) * (nx + nu));
                h_gemm(false, false, nu + nx, nx + nu, nv, LBL.data(), nu + nx, RT2.data() + (size_t)nv * nv, nv, MC.data());
                if (int rc = upload_padded(d_MCp, MC.data(), nu + nx, nx + nu)) return rc;
                if (int rc = upload_fragments(d_MCf, MC.data(), nu + nx, nx + nu)) return rc;
            }
            linConstValid = false;
        }
        if (int rc = upload_padded(d_Lp, s->matL, nu, nv)) return rc;
        if (int rc = upload_padded(d_Bp, s->matB, nx, nu)) return rc;
        UP(d_Rinv, h_Rinv.data(), nv * nv) UP(d_Bbt, h_Bbt.data(), nv * nx) UP(d_L, s->matL, nu * nv) UP(d_B, s->matB, nx * nu)
        UP(d_Lt, Lt.data(), nv * nu) UP(d_WLt, WLt.data(), nv * nu) UP(d_T1, T1.data(), nv * nx) UP(d_T2, T2.data(), nv * nu)
        UP(d_Gd, s->matGd, nx * nd) UP(d_Lhat, s->matLhat, nu * nd) UP(d_alpha1, s->costAlpha1, nu) UP(d_W, s->costW, nu * nu)
        UP(d_dy, dy.data(), (size_t)N * ny) UP(d_blo, blo.data(), ny) UP(d_bhi, bhi.data(), ny)
#undef UP
        if (d_Wp) { if (int rc = upload_padded(d_Wp, s->costW, nu, nu)) return rc; }   // k_value_mfma's copy of W (rn_set_algorithm may come before the factor step)
        if (int rc = stream_split_setup()) return rc;
        if (!structured && !d_A) {   // the dense per-node blocks are only allocated when they are used
            if (int rc = dalloc(&d_A, (size_t)d.nodes * strideA)) return rc;
        }
        RN_HIP(hipMemsetAsync(d_my, 0, (size_t)d.nodes * 2 * nv * sizeof(T), stream));   // structured mode never writes m1
        ExpandArgs<T> ea{};
        ea.tr = tree_dev(); ea.nx = nx; ea.nu = nu; ea.nv = nv; ea.ny = ny; ea.LD = LD; ea.strideA = strideA; ea.nodes = d.nodes;
        ea.T1 = d_T1; ea.T2 = d_T2; ea.Bbt = d_Bbt; ea.Lt = d_Lt; ea.A = d_A; ea.skipBlocks = structured; ea.blo = d_blo; ea.bhi = d_bhi; ea.lo = d_lo; ea.hi = d_hi;
        const int colChunks = std::min(ny, 8);
        hipLaunchKernelGGL(k_expand_operators<T>, dim3(d.nodes, colChunks), dim3(LD >= 192 ? 256 : (LD >= 96 ? 128 : 64)), 0, stream, ea);
        RN_HIP(hipGetLastError());
        factored = true;
        if (int rc = refresh_bounds_copies()) return rc;
        RN_HIP(hipStreamSynchronize(stream));
        return RN_OK;
    }
    int set_tree_errors(const double *ed, const double *ep) override {
        RN_CHECK(ed && ep, RN_E_ARG, "rn_set_tree_errors: null input");
        RN_HIP(hipSetDevice(device));
        if (int rc = upload(d_errD, ed, (size_t)d.nodes * d.nd)) return rc;
        return upload(d_errP, ep, (size_t)d.nodes * d.nu);
    }
    int set_uncertainty(int dflag, int pflag, double w) override { useErrD = dflag ? 1 : 0; useErrP = pflag ? 1 : 0; wEco = w; return RN_OK; }
    int update_state_control(const double *x0, const double *up, const double *dp) override {
        RN_CHECK(factored, RN_E_STATE, "rn_update_state_control before rn_factor_step");
        RN_CHECK(x0 && up && dp, RN_E_ARG, "rn_update_state_control: null input");
        RN_HIP(hipSetDevice(device));
        if (int rc = upload(d_curX, x0, d.nx)) return rc;
        if (int rc = upload(d_prevU, up, d.nu)) return rc;
        if (int rc = upload(d_prevD, dp, d.nd)) return rc;
        hipLaunchKernelGGL(k_gemv_small<T>, dim3(1), dim3(128), 0, stream, d_Lhat, d.nu, d.nd, d_prevD, d_prevUhat);  // Engine.cu:1314
        RN_HIP(hipGetLastError());
        aux_dirty = true;
        return RN_OK;
    }
    int eliminate(const double *dhat, const double *ahat) override {
        RN_CHECK(factored, RN_E_STATE, "rn_eliminate_input_disturbance_coupling before rn_factor_step");
        RN_CHECK(dhat && ahat, RN_E_ARG, "rn_eliminate_input_disturbance_coupling: null input");
        RN_HIP(hipSetDevice(device));
        if (int rc = upload(d_dhat, dhat, (size_t)d.N * d.nd)) return rc;
        if (int rc = upload(d_ahat, ahat, (size_t)d.N * d.nu)) return rc;
        AffineArgs<T> a{};
        a.tr = tree_dev(); a.nx = d.nx; a.nu = d.nu; a.nv = d.nv; a.nd = d.nd;
        a.Gd = d_Gd; a.Lhat = d_Lhat; a.WLt = d_WLt; a.Lt = d_Lt; a.errD = d_errD; a.errP = d_errP; a.dhat = d_dhat; a.ahat = d_ahat;
        a.alpha1 = d_alpha1; a.prevUhat = d_prevUhat; a.wEco = (T)wEco; a.useErrD = useErrD; a.useErrP = useErrP;
        a.e = d_e; a.uhat = d_uhat; a.alpha = d_alpha; a.beta = d_beta;
        RN_CHECK(cutStage <= 0 || nranks == 1 || moments_set, RN_E_STATE, "sharded context: rn_set_cut_children_moments must be called before the affine terms");
        a.momE = (cutStage > 0 && moments_set) ? d_momE : nullptr; a.momP = d_momP; a.cutStage = cutStage;
        const size_t sh1 = (size_t)(((d.nd + 3) & ~3) + ((std::max(d.nx, d.nu) + 3) & ~3) + AFF_THREADS) * sizeof(T);
        const size_t sh2 = (size_t)(2 * ((d.nu + 3) & ~3) + ((std::max(d.nv, d.nu) + 3) & ~3) + ((d.nv + 3) & ~3) + ((d.nd + 3) & ~3) + AFF_THREADS) * sizeof(T);
        hipEvent_t e0 = prof_begin(3);
        hipLaunchKernelGGL(k_affine_demand<T>, dim3(d.nodes), dim3(AFF_THREADS), sh1, stream, a);
        hipLaunchKernelGGL(k_affine_beta<T>, dim3(d.nodes), dim3(AFF_THREADS), sh2, stream, a);
        prof_end(e0);
        RN_HIP(hipGetLastError());
        affine_ready = true; aux_dirty = true;
        return RN_OK;
    }
    int set_parameters(double step, double px, double pxs) override {
        RN_CHECK(step > 0 && std::isfinite(step), RN_E_ARG, "rn_set_parameters: stepSize must be positive");
        stepSize = step; penX = px; penXs = pxs;
        return RN_OK;
    }

    // ---- profiling helpers -----------------------------------------------------------------------------
    hipEvent_t get_event() {
        if (!freeEvents.empty()) { hipEvent_t e = freeEvents.back(); freeEvents.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    // returns the closing event of the interval (a handle: intervals nest -- the collectives sit inside class 1 -- and `pending`
    // may grow in between), nullptr while profiling is off
    int profOpen = 0;
    hipEvent_t prof_begin(int cls) {
        if (!prof) return nullptr;
        if (pending.size() >= 60000 && profOpen == 0) (void)prof_flush();
        pending.push_back(EvPair{cls, get_event(), get_event()});
        (void)hipEventRecord(pending.back().a, stream);
        profOpen++;
        return pending.back().b;
    }
    void prof_end(hipEvent_t e) { if (e) { (void)hipEventRecord(e, stream); profOpen--; } }
    int prof_flush() {
        if (pending.empty()) return RN_OK;
        RN_HIP(hipStreamSynchronize(stream));
        for (auto &p : pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { prof_ms[p.cls] += ms; prof_n[p.cls]++; }
            freeEvents.push_back(p.a); freeEvents.push_back(p.b);
        }
        pending.clear();
        profOpen = 0;
        return RN_OK;
    }
    int profile_enable(int on) override { if (!on) { int rc = prof_flush(); prof = 0; return rc; } prof = on; return RN_OK; }
    int profile_reset() override { int rc = prof_flush(); for (int i = 0; i < 5; i++) { prof_ms[i] = 0; prof_n[i] = 0; } return rc; }
    int profile_read(double *ms, long *n) override {
        int rc = prof_flush();
        for (int i = 0; i < 4; i++) { ms[i] = prof_ms[i]; n[i] = prof_n[i]; }
        return rc;
    }
    // time between hipEvents recorded on the solver's stream around every all-reduce (RCCL or the installed stand-in) while
    // profiling is on: from the end of the kernel in front of the collective to the end of the collective, i.e. wire latency
    // AND the wait for the slowest peer.  The launches of class 1 (recursion + shared products) contain these intervals.
    int profile_read_collective(double *ms, long *n) override {
        RN_CHECK(ms && n, RN_E_ARG, "rn_profile_read_collective: null output");
        int rc = prof_flush();
        *ms = prof_ms[4]; *n = prof_n[4];
        return rc;
    }
    int algorithmic_bytes(double *bwd, double *dual) const override {
        // k_stream_gemv, one launch = the whole tree: A_i (2nv x ny, unpadded) read once + y_i read + m1,m2,a_i written
        const double s = sizeof(T), n = d.nodes;
        if (bwd) *bwd = structured ? 0.0 : n * ((double)2 * d.nv * ny + ny + 2.0 * d.nv + d.nx) * s;
        // fused dual update: Hx, w, y+prev read, y+, w_next written (+ the two scaled-bound streams unless they are regenerated)
        if (dual) *dual = (RN_DUAL_REGEN ? 5.0 : 7.0) * (double)ntot() * s;
        return RN_OK;
    }
    int synchronize() override { RN_HIP(hipSetDevice(device)); RN_HIP(hipStreamSynchronize(stream)); return RN_OK; }
    void *stream_handle() override { return (void *)stream; }

    // ---- the sweep ---------------------------------------------------------------------------------------
    // span of the streaming kernel: G columns = G*SPC 16-byte slots, NL slots per thread.  Preferred: spans that are whole
    // 128-byte lines (G*SPC*16 % 128 == 0) -- otherwise every span boundary splits a cache line between two wave-loads
    // that are issued at different times, and the non-temporal stream fetches that line twice (measured with
    // FETCH_SIZE: 4.35 GB per launch with G = 5, 4.14 GB with G = 8 on the 493-scenario tree, 6 % of the kernel's time)
    // -- with at most 2 slots per thread, then the best lane utilisation, then the larger span.  Without a line-aligned
    // candidate: the G that fills a multiple of STREAM_THREADS slots best.
    void stream_shape(int *G, int *NL) const {
        const int SPC = LD * (int)sizeof(T) / 16;
        int bestG = 0, bestNL = 0, bestClass = -1; double bestU = -1;
        for (int g = 1; g <= std::min(ny, 64); g++) {
            const int slots = g * SPC, nl = (slots + STREAM_THREADS - 1) / STREAM_THREADS;
            if (nl > STREAM_NLMAX) break;
            const bool aligned = ((long long)slots * 16) % 128 == 0;
            const int cls = aligned ? (nl <= 2 ? 2 : 1) : 0;
            const double u = (double)slots / ((double)nl * STREAM_THREADS);
            const bool tie = std::fabs(u - bestU) <= 1e-9;
            if (cls > bestClass || (cls == bestClass && (u > bestU + 1e-9 || (tie && cls > 0)))) { bestClass = cls; bestU = u; bestG = g; bestNL = nl; }
        }
        if (bestG == 0) { bestG = 1; bestNL = (SPC + STREAM_THREADS - 1) / STREAM_THREADS; }
#ifdef RN_STREAM_G   // tuning builds: force the span
        bestG = RN_STREAM_G; bestNL = (bestG * SPC + STREAM_THREADS - 1) / STREAM_THREADS;
#endif
        *G = bestG; *NL = bestNL;
    }
    size_t stream_lds(int G) const { return (size_t)(((ny + 3) & ~3) + (size_t)G * LD) * sizeof(T); }
    // k_stream_gemv's split last round (kernels.hpp, StreamSplit): applies when the launch is longer than one round, its last round
    // is at most half full (two workgroups per block still fit one round), every block of that round lies in the last
    // STREAM_SPLIT_STAGES stages of the chain region (the consumers add the second partial there) and a block has at least two
    // groups of spans.  splitFirst = nodes: no split.
    T *d_my2 = nullptr;
    int splitFirst = -1, splitSpanHalf = 0;
    bool streamTwoPerCU = false;
    int stream_split_setup() {
        if (splitFirst >= 0) return RN_OK;
        splitFirst = d.nodes;
        const int mode = knob[RN_KNOB_STREAM_SPLIT] != 0;
        int G, NL;
        stream_shape(&G, &NL);
        const int D = NL <= 2 ? RN_STREAM_D : RN_STREAM_D_WIDE, groups = (ny / G) / D;
        const int r = d.nodes % numCUs, cs = chainStage, K = h_stageCum[cs + 1] - h_stageCum[cs];
        // which instantiation the launches run: the one with the split's second code path allocates fewer registers (122 instead of
        // 150), so two workgroups share a CU -- start-up and drain of the launch overlap better, the steady state is slower in fp64 and
        // FASTER in fp32 (half-size blocks: the per-workgroup prologue and epilogue weigh double there).  Same-box A/B, one | two per
        // CU, ms per iteration: fp64 whole tree (42 rounds) 0.6733 | 0.6854, 1/2 shard 0.3712 | 0.3736, 1/4 shard (10.7 rounds)
        // 0.2101 | 0.2119, 1/8 shard (5.4 rounds) 0.1352 | 0.1318; fp32 whole tree 0.3838 | 0.3752 (the streaming kernel 313 -> 303 us
        // = 0.84 of the HBM peak).  So: fp32 always, fp64 for launches of fewer than 8 rounds.
        streamTwoPerCU = sizeof(T) == 4 || d.nodes < 8 * numCUs;
        if (knob[RN_KNOB_STREAM_TWO_PER_CU] >= 0) streamTwoPerCU = knob[RN_KNOB_STREAM_TWO_PER_CU] != 0;
        if (!mode || !streamTwoPerCU || structured || d.nodes <= numCUs || r == 0 || 2 * r > numCUs || groups < 2) return RN_OK;
        if (d.N - cs < STREAM_SPLIT_STAGES || r > STREAM_SPLIT_STAGES * K) return RN_OK;     // (the cut never moves the chain region's END)
        if (int rc = dalloc(&d_my2, (size_t)r * 2 * d.nv)) return rc;
        splitFirst = d.nodes - r;
        splitSpanHalf = (groups + 1) / 2 * D;       // whole groups; the first half gets the odd one (the second also walks the ragged last span)
        return RN_OK;
    }
    bool ensure_stream_lds() {   // > 64 KB of dynamic LDS needs the function attribute (once)
        static const bool ok = [] {
            bool all = true;
            for (const void *fn : {(const void *)k_stream_gemv<T, 1, false>, (const void *)k_stream_gemv<T, 2, false>, (const void *)k_stream_gemv<T, 3, false>, (const void *)k_stream_gemv<T, 4, false>,
                                   (const void *)k_stream_gemv<T, 1, true>, (const void *)k_stream_gemv<T, 2, true>, (const void *)k_stream_gemv<T, 3, true>, (const void *)k_stream_gemv<T, 4, true>})
                all = all && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            return all;
        }();
        return ok;
    }
    // second != nullptr: two right-hand sides in one pass (k_stream_gemv NR = 2; see stream_pair_ok)
    int launch_stream(const SweepArgs<T> &a, const StreamRhs2<T> *second = nullptr) {
        int G, NL;
        stream_shape(&G, &NL);
        RN_CHECK(NL <= STREAM_NLMAX, RN_E_ARG, "k_stream_gemv: more than 4096 values per operator column are not supported (2*nv too large)");
        const size_t lds = stream_lds(G);
        RN_CHECK(lds <= 64 * 1024 || (lds <= 160 * 1024 && ensure_stream_lds()), RN_E_ARG, "k_stream_gemv: a span of operator columns does not fit the CU's LDS");
        const int node0 = h_stageCum[a.chainStage];   // first node of the chain region as this sweep sees it
        const StreamSplit<T> sp{a.splitFirst, splitSpanHalf, d_my2};
        const int grid = d.nodes + (d.nodes - a.splitFirst);      // two workgroups for every block of the split round
        const StreamRhs2<T> none{nullptr, nullptr, nullptr};
        if (second) {
            RN_CHECK(a.splitFirst >= d.nodes && 2 * lds <= 64 * 1024, RN_E_STATE, "k_stream_gemv with two right-hand sides: unsplit launches only");
            switch (NL) {
                case 1: hipLaunchKernelGGL((k_stream_gemv<T, 1, false, 2>), dim3(grid), dim3(STREAM_THREADS), 2 * lds, stream, a, G, node0, sp, *second); break;
                case 2: hipLaunchKernelGGL((k_stream_gemv<T, 2, false, 2>), dim3(grid), dim3(STREAM_THREADS), 2 * lds, stream, a, G, node0, sp, *second); break;
                case 3: hipLaunchKernelGGL((k_stream_gemv<T, 3, false, 2>), dim3(grid), dim3(STREAM_THREADS), 2 * lds, stream, a, G, node0, sp, *second); break;
                default: hipLaunchKernelGGL((k_stream_gemv<T, 4, false, 2>), dim3(grid), dim3(STREAM_THREADS), 2 * lds, stream, a, G, node0, sp, *second); break;
            }
        } else if (streamTwoPerCU) {
            switch (NL) {
                case 1: hipLaunchKernelGGL((k_stream_gemv<T, 1, true>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                case 2: hipLaunchKernelGGL((k_stream_gemv<T, 2, true>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                case 3: hipLaunchKernelGGL((k_stream_gemv<T, 3, true>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                default: hipLaunchKernelGGL((k_stream_gemv<T, 4, true>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
            }
        } else {
            switch (NL) {
                case 1: hipLaunchKernelGGL((k_stream_gemv<T, 1, false>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                case 2: hipLaunchKernelGGL((k_stream_gemv<T, 2, false>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                case 3: hipLaunchKernelGGL((k_stream_gemv<T, 3, false>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
                default: hipLaunchKernelGGL((k_stream_gemv<T, 4, false>), dim3(grid), dim3(STREAM_THREADS), lds, stream, a, G, node0, sp, none); break;
            }
        }
        return RN_OK;
    }
    // two right-hand sides in one streaming pass are possible for: dense per-node blocks, no split last round, both LDS sets in 64 KB
    bool stream_pair_ok() const {
        int G, NL;
        stream_shape(&G, &NL);
        return !structured && !(splitFirst >= 0 && splitFirst < d.nodes) && 2 * stream_lds(G) <= 64 * 1024;
    }
    static int slab_stride(int kp) { return (kp + 59) / 64 * 64 + 4; }   // >= kp, = 4 (mod 64): conflict-free MFMA B reads
    // waves per slab workgroup.  Many slabs (more workgroups than CUs): the count in {4, 6, 8} that wastes the least SIMD
    // time on idle tile slots (throughput).  Few slabs (small or sharded trees: every workgroup has a CU to itself): the
    // count with the shortest critical path per workgroup (latency).
    // which MFMA loop the slab kernels run (slab_mfma): the software-pipelined one when a launch has at most one workgroup per CU
    // (small or sharded trees: nothing else hides the operand latency) and in fp32 (half the registers: the workgroups still share
    // a CU); the lean one otherwise.  Measured, k_gemm_vlv, lean | pipelined: 31-scenario tree 14.0 | 12.5 us, 1/8 shard 19.4 |
    // 17.7, 1/2 shard (340 slabs) 27.3 | 28.1, whole 493-scenario tree 30.0 | 35.8; fp32 wide network (k_gemm_prep_m2) 954 | 574.
    bool few_slabs() const {
        const int force = knob[RN_KNOB_SLAB_PIPE];
        return force >= 0 ? force != 0 : ((d.nodes + 15) / 16 <= numCUs || sizeof(T) == 4);
    }
    int slab_waves(int tilesA, int kstepsA, int tilesB, int kstepsB) const {
        const bool few = (d.nodes + 15) / 16 <= numCUs;
        int best = 4; long bestCost = -1;
        for (int nw : {4, 6, 8, 12, 16}) {
            if (nw > SLAB_MAX_WAVES || (nw > 8 && !few)) continue;
            const long path = (long)((tilesA + nw - 1) / nw) * kstepsA + (long)((tilesB + nw - 1) / nw) * kstepsB;
            const long cost = few ? path * 16 + nw : (long)nw * path;
            if (bestCost < 0 || cost < bestCost) { best = nw; bestCost = cost; }
        }
        return best;
    }
    template <int EPI>
    void launch_gemm(const T *Mp, int m, int k, const T *in, int ldin, T *out, int ldout, const T *aux, int ldaux) {
        GemmArgs<T> g{Mp, m, k, pad16(m), pad4(k), in, ldin, out, ldout, aux, ldaux, d_prob, d.nodes};
        if (EPI == EPI_V && !structured && splitFirst >= 0 && splitFirst < d.nodes) { g.aux2 = d_my2; g.auxSplit = splitFirst; }   // k_stream_gemv's split last round
#if RN_GEMM_SLAB
        const int SB = slab_stride(g.kp);
        const size_t lds = (size_t)16 * SB * sizeof(T);
        if (lds <= 64 * 1024) {   // slab kernel (default); falls back to the tile kernel when the slab does not fit
            const int nw = slab_waves((m + 15) / 16, g.kp / 4, 0, 0);
            if (few_slabs()) hipLaunchKernelGGL((k_gemm_slab<T, EPI, true>), dim3((d.nodes + 15) / 16), dim3(64 * nw), lds, stream, g, SB);
            else hipLaunchKernelGGL((k_gemm_slab<T, EPI, false>), dim3((d.nodes + 15) / 16), dim3(64 * nw), lds, stream, g, SB);
            return;
        }
#endif
        const int units = (g.mp / (16 * GEMM_RT)) * ((d.nodes + 15) / 16);   // one 64 x 16 output tile per workgroup
        hipLaunchKernelGGL((k_gemm_shared<T, EPI, RN_GEMM_KS>), dim3(units), dim3(GEMM_THREADS), 0, stream, g);
    }
    // structured mode, (1) of the sweep: a_i = F_i' xi_i, b_i = G_i' psi_i and m2_i = [Bbt | L'] [a_i; b_i]
    void launch_prep_m2(const SweepArgs<T> &a) {
        const int nx = d.nx, nu = d.nu, nv = d.nv;
#if RN_GEMM_SLAB
        GemmArgs<T> g{d_BLp, nv, nx + nu, pad16(nv), pad4(nx + nu), d_ab, nx + nu, d_my + nv, 2 * nv, nullptr, 0, d_prob, d.nodes};
        if (frag_on()) g.Mf = d_BLf;
        const int SB = slab_stride(g.kp);
        const size_t lds = (size_t)16 * SB * sizeof(T);
        if (lds <= 64 * 1024) {
            const int nw = slab_waves((nv + 15) / 16, g.kp / 4, 0, 0);
            if (few_slabs()) hipLaunchKernelGGL((k_gemm_prep_m2<T, true>), dim3((d.nodes + 15) / 16), dim3(64 * nw), lds, stream, g, a, SB);
            else hipLaunchKernelGGL((k_gemm_prep_m2<T, false>), dim3((d.nodes + 15) / 16), dim3(64 * nw), lds, stream, g, a, SB);
            return;
        }
#endif
        const long long tot = (long long)d.nodes * (nx + nu);
        const int blocks = (int)std::min<long long>(4096, (tot + 255) / 256);
        hipLaunchKernelGGL(k_struct_prep<T>, dim3(blocks), dim3(256), 0, stream, a);
        launch_gemm<EPI_LV>(d_BLp, nv, nx + nu, d_ab, nx + nu, d_my + nv, 2 * nv, nullptr, 0);
    }
    bool v_lv_is_slab() const {
#if RN_GEMM_SLAB
        return (size_t)16 * (slab_stride(pad4(d.nv + d.nx)) + slab_stride(pad4(d.nv))) * sizeof(T) <= 64 * 1024;
#else
        return false;
#endif
    }
    // slabs per workgroup of k_gemm_vlv_wide (0: use k_gemm_vlv): 3 when every CU gets many slabs (more than four per CU: the
    // shared operators' stream from L2, re-read per slab, is then what the launch waits for; fp32 wide network, 21 slabs per CU:
    // 1 047 -> 780 us), and only if the workgroup's LDS (ct slab buffers; > 64 KB needs the function attribute, set once) is
    // granted.  With two or three slabs per CU (493-scenario tree) the plain kernel is as fast or faster (30.0 vs 31.2 us).
    int wideCt = -1;
    int wide_ct(int nSlabs, size_t ldsPerSlab) {
        if (wideCt >= 0) return wideCt;
        wideCt = 0;
        int want = nSlabs > 4 * numCUs ? 3 : 0;
        if (knob[RN_KNOB_VLV_WIDE] >= 0) want = std::min(3, knob[RN_KNOB_VLV_WIDE]);   // 0 / 1: off
        for (; want >= 2; want--) {
            const size_t bytes = ldsPerSlab * want;
            if (bytes > 160 * 1024) continue;
            const void *fn = want == 2 ? (const void *)k_gemm_vlv_wide<T, 2> : (const void *)k_gemm_vlv_wide<T, 3>;
            // (the attribute belongs to the function, not to this context: always the device's whole LDS)
            if (bytes <= 64 * 1024 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess) { wideCt = want; break; }
            (void)hipGetLastError();
        }
        return wideCt;
    }
    // values of LDS scratch behind the slab buffers of k_gemm_vlv for the root's children (0: not granted / switched off)
    int crownScratchVals = -1;
    int crown_scratch(size_t ldsSlab) {
        if (crownScratchVals >= 0) return crownScratchVals;
        crownScratchVals = 0;
        const size_t want = (size_t)h_childCount[0] * (d.nv + 2 * d.nx), bytes = ldsSlab + want * sizeof(T);
        if (bytes > 160 * 1024) return 0;
        if (bytes > 64 * 1024) {
            // the attribute belongs to the function, not to this context: always the device's whole LDS, never a smaller value later
            if (hipFuncSetAttribute((const void *)k_gemm_vlv<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void *)k_gemm_vlv<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
        }
        crownScratchVals = (int)want;
        return crownScratchVals;
    }
    static int wide_waves() { return std::min(RN_WIDE_THREADS / 64, 8); }
    size_t crown_lin_lds(int lin) const {      // k_up_crown_lin's LDS: [parts][W2], parts as the kernel counts them
        const int w2 = d.nv + 2 * d.nx + d.nu, span = w2 - ((lin & 2) ? d.nv : 0), wp = (span + 63) / 64 * 64;
        return (size_t)std::max(1, CROWN_THREADS / wp) * w2 * sizeof(T);
    }
    // Structured mode, linear form: Bs of every node (one full leaf-to-root pass of the linear form: chain walks, crown stages, the root) and
    // c_i = -Rinv Bs_i / (2 p_i) (one product with the first nv columns of [Rinv | T1 | T2]) -- when the affine terms have changed, i.e. once per control step.
    int lin_const_refresh(const SweepArgs<T> &a0) {
        SweepArgs<T> c = a0;
        c.lin = 1;
        const int cs = c.chainStage;
        hipLaunchKernelGGL(k_up_chain_lin<T>, dim3(c.K), dim3(CHAIN_THREADS), 0, stream, c, FinArgs{});
        for (int k = cs - 1; k >= 0; k--)
            hipLaunchKernelGGL(k_up_crown_lin<T>, dim3(h_stageCum[k + 1] - h_stageCum[k]), dim3(CROWN_THREADS), crown_lin_lds(1), stream, c, k, h_stageCum[k + 1] - h_stageCum[k], FinArgs{});
        launch_gemm<EPI_V>(d_RT2p, d.nv, d.nv, d_sk2, d.nv + d.nx + d.nu, d_vconst, d.nv, nullptr, 0);
        if (d_MCp) launch_gemm<EPI_LV>(d_LBLp, d.nu + d.nx, d.nv, d_vconst, d.nv, d_lvconst, d.nu + d.nx, nullptr, 0);      // [L ; B L] c_i
        if (lin_fold()) {
            const long long tot = (long long)d.nodes * (d.nu + d.nx);
            hipLaunchKernelGGL(k_fold_affine<T>, dim3((unsigned)std::min<long long>((tot + 255) / 256, (long long)numCUs * 8)), dim3(256), 0, stream, d_lvconst, d_uhat, d_eb, d_prevUhat,
                               d_parent, d.nodes, d.nu, d.nx);
        }
        RN_HIP(hipGetLastError());
        linConstValid = true;
        return RN_OK;
    }
    // (3) of the sweep: v_i and [L v_i ; B L v_i] for all nodes
    void launch_v_lv(const SweepArgs<T> &a, int foldRoot) {
        const int nx = d.nx, nu = d.nu, nv = d.nv;
#if RN_GEMM_SLAB
        GemmArgs<T> gV{d_RTp, nv, nv + nx, pad16(nv), pad4(nv + nx), a.sk, nv + nx, a.v, nv, a.my, 2 * nv, d_prob, d.nodes, a.my2, a.splitFirst};
        if (a.lin) gV = GemmArgs<T>{d_RT2p, nv, nv + nx + nu, pad16(nv), pad4(nv + nx + nu), a.sk2, nv + nx + nu, a.v, nv, nullptr, 0, d_prob, d.nodes, nullptr, d.nodes};
        if (!a.writePrimal) gV.out = nullptr;   // slab kernel only: v stays in LDS for the second product
        if (structured) gV.aux = nullptr;       // m1_i is folded into the v product (d_my's m1 half stays zero): the epilogue has nothing to fetch
        if (a.lin & 2) {        // Bs's term is the constant c_i (lin_const_refresh): v_i = c_i - [T1 | T2] [q_i + kappa_i ; Bu_i] / (2 p_i); a Hessian sweep (beta = 0) has none
            gV = GemmArgs<T>{d_T12p, nv, nx + nu, pad16(nv), pad4(nx + nu), a.sk2 + nv, nv + nx + nu, a.writePrimal ? a.v : nullptr, nv,
                             a.beta == d_zero ? nullptr : d_vconst, nv, d_prob, d.nodes, nullptr, d.nodes};
        }
        GemmArgs<T> gL{d_LBLp, nu + nx, nv, pad16(nu + nx), pad4(nv), a.v, nv, a.lvb, nu + nx, nullptr, 0, d_prob, d.nodes};
        if ((a.lin & 2) && lin_comp()) {      // one product with the composite operator (k_gemm_comp); v_i, when it is stored, by a launch of its own
            const bool hess = a.beta == d_zero;
            GemmArgs<T> gC{d_MCp, nu + nx, nx + nu, pad16(nu + nx), pad4(nx + nu), a.sk2 + nv, nv + nx + nu, a.lvb, nu + nx,
                           hess ? nullptr : d_lvconst, nu + nx, d_prob, d.nodes, nullptr, d.nodes};
            if (frag_on()) gC.Mf = d_MCf;
            const int SBc = slab_stride(gC.kp), nSlabs = (d.nodes + 15) / 16;
            const int nwc = slab_waves((nu + nx + 15) / 16, gC.kp / 4, 0, 0);
            if (few_slabs()) hipLaunchKernelGGL((k_gemm_comp<T, true>), dim3(nSlabs), dim3(64 * nwc), (size_t)16 * SBc * sizeof(T), stream, gC, SBc, a, foldRoot);
            else hipLaunchKernelGGL((k_gemm_comp<T, false>), dim3(nSlabs), dim3(64 * nwc), (size_t)16 * SBc * sizeof(T), stream, gC, SBc, a, foldRoot);
            if (a.writePrimal) {
                if (hess || vEager || a.v != d_v) launch_gemm<EPI_V>(d_T12p, nv, nx + nu, a.sk2 + nv, nv + nx + nu, a.v, nv, hess ? nullptr : d_vconst, nv);
                else vPending = true;
            }
            return;
        }
        const int SB = slab_stride(gV.kp), SV = slab_stride(gL.kp);
        const size_t lds = (size_t)16 * (SB + SV) * sizeof(T);
        if (frag_on()) { gV.Mf = (a.lin & 2) ? d_T12f : (a.lin ? d_RT2f : d_RTf); gL.Mf = d_LBLf; }
        if (lds <= 64 * 1024) {
            const int nSlabs = (d.nodes + 15) / 16;
            // many slabs per CU: one workgroup per CU with CT slabs each, every A fragment (from L2) used CT times (k_gemm_vlv_wide)
            const int ct = wide_ct(nSlabs, lds);
            if (ct >= 2 && foldRoot != 2) {
                const int grid = (nSlabs + ct - 1) / ct, threads = 64 * wide_waves();
                if (ct == 2) hipLaunchKernelGGL((k_gemm_vlv_wide<T, 2>), dim3(grid), dim3(threads), lds * 2, stream, gV, gL, SB, SV, a, foldRoot);
                else hipLaunchKernelGGL((k_gemm_vlv_wide<T, 3>), dim3(grid), dim3(threads), lds * 3, stream, gV, gL, SB, SV, a, foldRoot);
                return;
            }
            const int nw = slab_waves((nv + 15) / 16, gV.kp / 4, (nu + nx + 15) / 16, gL.kp / 4);
            // sharded two-stage crown (foldRoot = 2): scratch for the root's children behind the slab buffers, so that workgroup 0 -- the
            // launch's critical path -- fills its own slab in LDS instead of draining its stores and reading them back
            const int scratch = foldRoot == 2 ? crown_scratch(lds) : 0;
            const size_t ldsAll = lds + (size_t)scratch * sizeof(T);
            if (few_slabs()) hipLaunchKernelGGL((k_gemm_vlv<T, true>), dim3(nSlabs), dim3(64 * nw), ldsAll, stream, gV, gL, SB, SV, a, foldRoot, scratch);
            else hipLaunchKernelGGL((k_gemm_vlv<T, false>), dim3(nSlabs), dim3(64 * nw), ldsAll, stream, gV, gL, SB, SV, a, foldRoot, scratch);
            return;
        }
#endif
        launch_gemm<EPI_V>(d_RTp, nv, nv + nx, a.sk, nv + nx, a.v, nv, a.my, 2 * nv);
        launch_gemm<EPI_LV>(d_LBLp, nu + nx, nv, a.v, nv, a.lvb, nu + nx, nullptr, 0);   // [L v_i ; B L v_i]
    }
    // crown handling of the forward sweep: 0 = crown launches of their own; 1 = every chain workgroup walks its crown path and
    // the first descendant chain of a crown node writes it; 2 (sharded) = crown nodes dealt round-robin to the workgroups
    int fold_crown_mode(int cs, bool sharded) const {
        return (RN_FOLD_CROWN_DOWN && cs >= 1 && cs <= CROWN_MAX_DEPTH) ? (sharded ? (h_stageCum[cs] <= 256 ? 2 : 0) : 1) : 0;
    }
    // lanes per chain of k_up_chain_cut, or 0 when the merged launch does not apply: the cut lies right above the chains and every
    // cut parent's local chains fit side by side in one workgroup
    int up_cut_lanes() const {
        if (cutStage <= 0 || cutStage < chainStage) return 0;
        const int lanesPer = (d.nv + d.nx + 63) / 64 * 64;
        if (lanesPer > UPCUT_THREADS) return 0;
        int most = 0;
        for (int i = h_stageCum[cutStage - 1]; i < h_stageCum[cutStage]; i++) most = std::max(most, h_childCount[i]);
        if (most * lanesPer > UPCUT_THREADS) return 0;
        return lanesPer;
    }
    // phase: 0 = whole sweep; 1 = up to (and including) the cut parents' partial children sums; 2 = the rest,
    // assuming the summed payload is in d_cut (tests emulate the all-reduce between two contexts on one GPU)
    // hessianInput != nullptr: SmpcController::computeHessianOracalGlobalFbe (SmpcController.cu:884-1055) -- the same
    // sweep evaluated at `hessianInput` with sigma = 0 and every affine term zero, writing xdir / udir / H * dir
    // primalOut = false (inner iterations of a batch): x, u and v are not stored, only Hx (what the dual update reads)
    // hessianInput2 (with hessianInput; stream_pair_ok()): TWO Hessian sweeps whose inputs do not depend on each other's results share
    // ONE pass over the operator blocks (k_stream_gemv NR = 2); the vector recursions and shared-operator products then run once per
    // right-hand side.  The first sweep's results go to the pair buffers (d_xdirB, d_udirB, d_hxDirB), the second's where a Hessian
    // sweep always leaves them (d_xdir, d_udir, d_hxDir).  Bitwise the results of two launch_sweep calls.
    int launch_sweep(int phase = 0, const T *hessianInput = nullptr, bool primalOut = true, const T *hessianInput2 = nullptr, bool allowPending = false) {
        SweepArgs<T> a = sweep_args();
        a.writePrimal = primalOut ? 1 : 0;
        if (vPending) { if (hessianInput) { if (int rc = v_flush()) return rc; } else vPending = false; }      // (see v_flush)
        if (hessianInput) {
            a.w = hessianInput;
            a.beta = d_zero; a.uhat = d_zero; a.e = d_zero; a.eb = d_zero; a.bw0 = d_zero;
            a.curX = d_zero; a.prevU = d_zero; a.prevUhat = d_zero;
            a.x = d_xdir; a.u = d_udir; a.hx = d_hxDir;
        }
        const bool pair = hessianInput2 != nullptr;
        if (pair) {
            RN_CHECK(hessianInput && phase == 0 && d_myB && stream_pair_ok(), RN_E_STATE, "launch_sweep: a pair of sweeps needs two Hessian inputs, the pair buffers and an unsplit dense launch");
            a.x = d_xdirB; a.u = d_udirB; a.hx = d_hxDirB;
        }
        RN_CHECK(phase == 0 || a.cutSums, RN_E_STATE, "rn_debug_sweep_phase needs rn_set_cut_stage and nranks > 1");
        // one-shot exchange (inside rn_apg_iterate batches only): the launch that produces the cut parents' local sums pushes them
        // to every peer, the crown launch gathers them -- no collective in between
        const bool oneShot = transport == 1 && peerReady && inBatch && phase == 0 && a.cutSums != nullptr && hessianInput == nullptr;
        if (oneShot) {
            // 0 is the "never written" tag; the wrap skips TWO values (... fffffffe, ffffffff, 2, 3 ...) so that the parity -- which of the
            // two inbox buffers an exchange uses -- keeps alternating (the tag 2 of four billion exchanges ago is long overwritten)
            if (++peerSeq == 0) peerSeq = 2;
            a.peer = h_peer; a.peerSeq = peerSeq; a.peerTail = (pendingFin && carryTail) ? 1 : 0;
        }
        const int nx = d.nx, nu = d.nu, nv = d.nv, cs = a.chainStage;
        auto nk = [&](int k) { return h_stageCum[k + 1] - h_stageCum[k]; };
        // (1) all per-node mat-vecs of the backward sweep in one streaming launch
        if (aux_dirty) {   // iteration-invariant pieces of the forward sweep (once per control step)
            launch_gemm<EPI_Z>(d_Bp, nx, nu, d_uhat, nu, d_eb, nx, d_e, nx);            // eb_i = e_i + B uhat_i
            hipLaunchKernelGGL(k_bw0<T>, dim3(1), dim3(128), 0, stream, d_B, nx, nu, d_prevU, d_prevUhat, d_bw0);
            aux_dirty = false;
            linConstValid = false;      // (the affine terms may have changed: beta)
        }
        if (a.lin && lin_const()) {
            if (!hessianInput && !linConstValid) { if (int rc = lin_const_refresh(a)) return rc; }
            a.lin = lin_fold() ? 7 : 3;      // the walks leave the Bs columns alone (bit 1); the forward walk's affine terms ride in the product's constant (bit 2)
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (phase != 2 && !a.lin) {      // (the structured mode's linear form has no product in front of the chain walks: class 0 stays empty)
            e0 = prof_begin(0);
            if (structured) {
                launch_prep_m2(a);
            } else {
                const StreamRhs2<T> r2{hessianInput2, d_myB, d_qaB};
                if (int rc = launch_stream(a, pair ? &r2 : nullptr)) return rc;
            }
            prof_end(e0);
        }
        // one-shot exchange: gathered by the launch that produces the cut parents' sums, every parent's workgroup its own
        auto helpers = [&](SweepArgs<T> &a) -> int {
        e1 = prof_begin(1);
        const bool rodeUp = upDone && a.lin && phase == 0 && !hessianInput;
        upDone = false;
        FinArgs linFin{};
        // (2) leaf-to-root vector recursion: chains in one launch, crown stage by stage
        // sharded, cut right above the chains, few local chains per cut parent: one launch does the chain walks AND the cut
        // parents' local children sums (k_up_chain_cut)
        const bool mergedCut = phase != 2 && a.cutSums && up_cut_lanes() > 0;
        if (mergedCut) {
            const int k = cutStage - 1, lanesPer = up_cut_lanes();
            FinArgs fin{};
            if (pendingFin) fin = FinArgs{d_partials, main_partials(), d_state, (void *)(d_cut + cut_tail_offset()), d_hist, d_histParts, histCap, -1.0, -1.0};
            const size_t ldsCut = (size_t)(UPCUT_THREADS / lanesPer) * (nv + 2 * nx) * sizeof(T);
            const int grid = nk(k) + (pendingFin ? 1 : 0);
            if (oneShot) {
                if (a.splitFirst < d.nodes) hipLaunchKernelGGL((k_up_chain_cut<T, true, true>), dim3(grid), dim3(UPCUT_THREADS), ldsCut, stream, a, d_cut, nk(k), lanesPer, fin);
                else hipLaunchKernelGGL((k_up_chain_cut<T, false, true>), dim3(grid), dim3(UPCUT_THREADS), ldsCut, stream, a, d_cut, nk(k), lanesPer, fin);
                a.peer.nranks = 0;      // d_cut holds the all-rank sums: every later launch of this sweep is the collective path's
            } else if (a.splitFirst < d.nodes) hipLaunchKernelGGL((k_up_chain_cut<T, true>), dim3(grid), dim3(UPCUT_THREADS), ldsCut, stream, a, d_cut, nk(k), lanesPer, fin);
            else hipLaunchKernelGGL((k_up_chain_cut<T, false>), dim3(grid), dim3(UPCUT_THREADS), ldsCut, stream, a, d_cut, nk(k), lanesPer, fin);
            pendingFin = false;
        } else if (phase != 2) {
            // single-GPU optimistic bookkeeping: the previous iteration's fold / history entry / distance check rides here
            FinArgs fin{};
            const bool ride = pendingFin && !a.cutSums;
            if (rodeUp) {
                // the chain walk of this sweep rode in the previous iteration's fused walk + dual update; the bookkeeping goes with the first crown launch
                if (ride) { linFin = FinArgs{d_partials, main_partials(), d_state, nullptr, d_hist, d_histParts, histCap, penX / stepSize, penXs / stepSize}; pendingFin = false; }
            } else {
            if (ride) { fin = FinArgs{d_partials, main_partials(), d_state, nullptr, d_hist, d_histParts, histCap, penX / stepSize, penXs / stepSize}; pendingFin = false; }
            if (a.lin) hipLaunchKernelGGL(k_up_chain_lin<T>, dim3(a.K + (ride ? 1 : 0)), dim3(CHAIN_THREADS), 0, stream, a, fin);
            else if (a.splitFirst < d.nodes) hipLaunchKernelGGL((k_up_chain<T, true>), dim3(a.K + (ride ? 1 : 0)), dim3(CHAIN_THREADS), 0, stream, a, fin);
            else hipLaunchKernelGGL((k_up_chain<T, false>), dim3(a.K + (ride ? 1 : 0)), dim3(CHAIN_THREADS), 0, stream, a, fin);
            }
        }
        // small crowns are walked by ONE workgroup per direction (stage after stage inside the kernel)
        const bool fusedCrown = cs > 0 && h_stageCum[cs] <= 64;
        auto all_reduce_cut = [&](int k) -> int {   // multi-GPU: all-reduce the children sums of the cut parents
            if (phase == 2) return RN_OK;              // payload already summed by the caller
            // optimistic exchange: the bookkeeping of the previous iteration's dual update rides in this launch
            if (!mergedCut) {   // (k_up_chain_cut has already left the payload in d_cut)
                FinArgs fin{};
                if (pendingFin) fin = FinArgs{d_partials, main_partials(), d_state, (void *)(d_cut + cut_tail_offset()), d_hist, d_histParts, histCap, -1.0, -1.0};
                if (oneShot) {
                    hipLaunchKernelGGL((k_cut_partial_sums<T, true>), dim3(nk(k) + (pendingFin ? 1 : 0)), dim3(CUT_THREADS), 0, stream, a, d_cut, nk(k), fin);
                    a.peer.nranks = 0;      // d_cut holds the all-rank sums
                } else hipLaunchKernelGGL((k_cut_partial_sums<T, false>), dim3(nk(k) + (pendingFin ? 1 : 0)), dim3(CUT_THREADS), 0, stream, a, d_cut, nk(k), fin);
                pendingFin = false;
            }
            if (phase == 1 || !has_comm()) return RN_OK;     // emulation, or a single-rank "sharded" run
            if (oneShot) return RN_OK;                       // the payload has gone to the peers' inboxes straight from the kernel
            const size_t cnt = (size_t)nk(k) * (nv + 2 * nx);
            return all_reduce(d_cut, cnt + (carryTail ? 2 : 0), sizeof(T) == 8, "ncclAllReduce(cut payload)");
        };
        // the root's own recursion step is folded into workgroup 0 of the v / Lv launch (one launch less) whenever that
        // launch is the slab kernel and stage 0 is not the multi-GPU exchange stage (single GPU: also when the root IS the whole crown --
        // the 31-scenario tree: one launch of 4.6 us less, helper path 40.2 -> 38.8 us)
        // 2: sharded with a two-stage crown whose stage 1 is the exchange stage -- its (presummed) step is folded as well
        int foldRoot = (RN_FOLD_ROOT && phase == 0 && cs >= (a.cutSums ? 2 : 1) && !(a.cutSums && cutStage == 1) && v_lv_is_slab()) ? 1 : 0;
        if (foldRoot && a.cutSums && cs == 2 && cutStage == 2) foldRoot = 2;
        {
            const int w = nv + 2 * nx, wp = (w + 63) / 64 * 64;
            const size_t ldsCrown = (size_t)std::max(1, CROWN_THREADS / wp) * w * sizeof(T);
            for (int k = cs - 1; k >= (foldRoot ? 1 : 0); k--) {
                if (phase == 2 && k > cutStage - 1) continue;          // done in phase 1
                if (a.cutSums && k == cutStage - 1) {
                    if (int rc = all_reduce_cut(k)) return rc;
                    if (phase == 1) { prof_end(e1); RN_HIP(hipGetLastError()); return RN_OK; }
                    if (foldRoot == 2) continue;                       // done by the v / Lv launch
                }
                if (a.lin) {
                    const bool host = linFin.partials != nullptr;       // (first crown launch of a sweep whose chain walk rode along)
                    hipLaunchKernelGGL(k_up_crown_lin<T>, dim3(nk(k) + (host ? 1 : 0)), dim3(CROWN_THREADS), crown_lin_lds(a.lin), stream, a, k, nk(k), linFin);
                    linFin = FinArgs{};
                }
                else hipLaunchKernelGGL(k_up_crown<T>, dim3(nk(k)), dim3(CROWN_THREADS), ldsCrown, stream, a, k);
            }
        }
        // (3) v_i = m1_i - (Rinv s_i + Rinv Bbt kappa_i) / (2 p_i) ; lv_i = L v_i    (batched over all nodes, MFMA)
        launch_v_lv(a, foldRoot);
        // (4) root-to-leaf: u, x and Hx in one pass (crown, then the chains)
        // shallow crowns: every chain workgroup walks its own crown path (k_down_chain, foldCrown) -- no crown launch.
        // single GPU: the first descendant chain of a crown node writes it (1); sharded: workgroup 0 writes them all (2),
        // because a replicated crown node may have no chain on this rank while its Hx still feeds the replicated duals
        const int foldCrown = fold_crown_mode(cs, a.cutSums != nullptr);
        if (foldCrown) { if (int rc = ensure_chain_anc(cs)) return rc; a.chainAnc = d_chainAnc; }
        if (!foldCrown) {
            if (fusedCrown) hipLaunchKernelGGL(k_down_crown_all<T>, dim3(1), dim3(CROWN_THREADS), 0, stream, a, cs);
            else for (int k = 0; k < cs; k++) hipLaunchKernelGGL(k_down_crown<T>, dim3(nk(k)), dim3(CHAIN_THREADS), 0, stream, a, k);
        }
        // sharded (foldCrown = 2): one more workgroup per replicated crown node, which writes that node while the chains are walked
        const int nCrownWg = foldCrown == 2 ? h_stageCum[cs] : 0;
        const int downGrid = a.K + nCrownWg;
        const size_t fuseLds = (size_t)d.N * ny * sizeof(T);      // Hx of the chain's N - cs nodes and of up to cs crown nodes
        const int split = fuse_split();                           // workgroups per chain of the fused launch
        const int fuseGrid = a.K * split + nCrownWg;
        // (every workgroup leaves one entry in d_partials: trees with more chains than it holds take the two launches)
        if (fuseReq && foldCrown && phase == 0 && !hessianInput && fuseLds <= 64 * 1024 && fuseGrid <= std::max(ELT_MAX_BLOCKS, RN_DUAL_STAGE_MAX_BLOCKS)) {
            // structured mode, linear form, inner iteration of a batch: the NEXT sweep's chain walk rides in this launch (phase C) when that sweep
            // has a crown launch to host the bookkeeping workgroup the chain walk otherwise carries (one workgroup per chain only: the walk needs the chain's rows in one tile)
            const bool upRide = a.lin && !fuseMat && foldCrown == 1 && split == 1 && (cs - 1 >= (foldRoot ? 1 : 0)) && knob[RN_KNOB_STRUCT_LINEAR] != 2;
            if (upRide) { hipLaunchKernelGGL((k_down_chain_dual<T, false, true>), dim3(fuseGrid), dim3(CHAIN_THREADS), fuseLds, stream, a, foldCrown, fuseArgs, fuseLn, 1); upDone = true; }
            else if (fuseMat) hipLaunchKernelGGL((k_down_chain_dual<T, true>), dim3(fuseGrid), dim3(CHAIN_THREADS), fuseLds, stream, a, foldCrown, fuseArgs, fuseLn, split);
            else hipLaunchKernelGGL((k_down_chain_dual<T, false>), dim3(fuseGrid), dim3(CHAIN_THREADS), fuseLds, stream, a, foldCrown, fuseArgs, fuseLn, split);
            fuseDone = true; mainPartials = fuseGrid;
        } else
        {
            // inner iterations of an optimistic batch whose dual update is the stage-tiled kernel reading w: the walk leaves the primal values and
            // the dual update scales them (k_down_chain UNSC / k_dual_stage SCALE: the walk requests no preconditioner entries; bitwise the same Hx)
            const bool unsc = allowPending && !a.writePrimal && phase == 0 && !hessianInput && foldCrown && dualU != 0 && a.hx == d_hx && unscaled_on();
            if (unsc) { hipLaunchKernelGGL((k_down_chain<T, true, RN_DOWN_PF_UNSC>), dim3(downGrid), dim3(CHAIN_THREADS), 0, stream, a, foldCrown); hxUnscaled = true; }
            else hipLaunchKernelGGL((k_down_chain<T, false>), dim3(downGrid), dim3(CHAIN_THREADS), 0, stream, a, foldCrown);
        }
        prof_end(e1);
        RN_HIP(hipGetLastError());
        return RN_OK;
        };
        if (pair && stream2 && !prof) {      // (while profiling every interval is bracketed on the one stream)
            // the two right-hand sides' helper chains (five dependent, latency-bound launches each) do not touch each other's buffers:
            // the second runs on a stream of its own with a scratch set of its own, forked behind the streaming pass and joined here
            SweepArgs<T> b = a;
            b.w = hessianInput2; b.my = d_myB; b.qa = d_qaB;
            b.x = d_xdir; b.u = d_udir; b.hx = d_hxDir;
            b.sk = d_skB; b.rkq = d_rkqB; b.v = d_vB; b.lvb = d_lvbB; b.bw = d_bwB;
            RN_HIP(hipEventRecord(evFork, stream));
            RN_HIP(hipStreamWaitEvent(stream2, evFork, 0));
            std::swap(stream, stream2);
            int rcB = helpers(b);
            if (rcB == RN_OK && hipEventRecord(evJoin, stream) != hipSuccess) rcB = RN_E_HIP;
            std::swap(stream, stream2);
            if (rcB) return rcB;
            if (int rc = helpers(a)) return rc;
            RN_HIP(hipStreamWaitEvent(stream, evJoin, 0));
            return RN_OK;
        }
        if (int rc = helpers(a)) return rc;
        if (pair) {
            SweepArgs<T> b = a;
            b.w = hessianInput2; b.my = d_myB; b.qa = d_qaB;
            b.x = d_xdir; b.u = d_udir; b.hx = d_hxDir;
            if (int rc = helpers(b)) return rc;
        }
        return RN_OK;
    }
    // launch shape of k_dual_stage: one tile of ELT_THREADS * trips 16-byte vectors per workgroup, tiles never cross a stage
    void dual_stage_setup() {
        dualU = 0; dualBlocks = eltBlocks;
        constexpr int VN = 16 / (int)sizeof(T);
        if (!RN_DUAL_REGEN || !RN_DUAL_STAGE || ny % VN) return;
        const int vpn = ny / VN, cs = chainStage, K = h_stageCum[cs + 1] - h_stageCum[cs], node0 = h_stageCum[cs];
        if (vpn < 2) return;   // one vector per node: floor(2^32 / 1) + 1 does not fit the 32-bit magic -- the flat kernel runs
        const unsigned long long lim = (1ull << 32) / (unsigned)vpn;          // range in which umulhi(j, magic) == j / vpn
        if ((unsigned long long)K * vpn >= lim || (unsigned long long)node0 * vpn >= lim || (unsigned long long)d.nodes * vpn >= (1ull << 31)) return;
        const int forced = knob[RN_KNOB_DUAL_TRIPS] > 0 ? knob[RN_KNOB_DUAL_TRIPS] : 0;
        // workgroups: at most one resident round (numCUs x 8 workgroups of 4 waves) so that every workgroup's loads start at
        // once; measured on the 493-scenario tree: 1 trip (5 113 workgroups) 21.9 us, 2-3 trips 20.8, 4-6 trips 21.6-24 us
        // vectors that do not fit the 256 MiB Infinity Cache stream from HBM: there 4x as many (smaller) workgroups and the
        // double-buffered variant measured best (wide4096 fp32, 1.3 GB per launch: 279 -> 257 us)
        const bool cacheResident = 5.0 * (double)ntot() * sizeof(T) <= 256.0 * 1024 * 1024;
        const long long resident = std::min<long long>((long long)numCUs * (cacheResident ? 8 : 32), RN_DUAL_STAGE_MAX_BLOCKS);   // d_partials holds that many
        for (int pass = 0; pass < 2 && dualU == 0; pass++)
            for (int trips : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64}) {
                if (forced > 0 && trips != forced) continue;
                const long long tile = (long long)ELT_THREADS * trips;
                const long long bps = ((long long)K * vpn + tile - 1) / tile, cb = ((long long)node0 * vpn + tile - 1) / tile;
                const long long blocks = cb + bps * (d.N - cs);
                if (blocks > (pass == 0 && forced <= 0 ? resident : (long long)RN_DUAL_STAGE_MAX_BLOCKS)) continue;
                dshape = DualStageShape{cs, K, node0, (int)bps, (int)cb, vpn, (unsigned int)((1ull << 32) / (unsigned)vpn) + 1u, trips, 0.0};
                dualU = (!cacheResident && trips >= 2) ? 2 : 1;
                if (knob[RN_KNOB_DUAL_PIPE] > 0) dualU = knob[RN_KNOB_DUAL_PIPE] >= 2 ? 2 : 1;
                dualBlocks = (int)blocks;
                break;
            }
    }
    // main pass of the fused dual update (prox as a pure projection, residual, dual update, arg-max partials, next
    // extrapolation); `flat` forces the grid-stride kernel (eltBlocks partials), which the exact multi-GPU path folds
    void launch_dual_main(const DualArgs<T> &a, bool materialize, bool flat = false) {
        if (flat || dualU == 0) {
            if (hxUnscaled) hx_scale_now();
            if (materialize) hipLaunchKernelGGL((k_dual_fused<T, true, false>), dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
            else hipLaunchKernelGGL((k_dual_fused<T, false, false>), dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
            mainPartials = eltBlocks;
            return;
        }
        mainPartials = dualBlocks;
        DualStageShape g = dshape;
        g.lnNext = h_lam[h_it + 1];   // ensure_tables(h_it + n) has run: the table covers every iteration of the batch
        if (hxUnscaled && !materialize) {   // unscaled walk: Hx = sqrt(p_i) d_k * (primal value) is formed here
            if (dualU == 1) hipLaunchKernelGGL((k_dual_stage<T, false, 1, true>), dim3(dualBlocks), dim3(ELT_THREADS), 0, stream, a, g);
            else hipLaunchKernelGGL((k_dual_stage<T, false, 2, true>), dim3(dualBlocks), dim3(ELT_THREADS), 0, stream, a, g);
            hxUnscaled = false;
            return;
        }
        if (hxUnscaled) hx_scale_now();
#define RN_LAUNCH_DSTAGE(MAT, PIPE) hipLaunchKernelGGL((k_dual_stage<T, MAT, PIPE>), dim3(dualBlocks), dim3(ELT_THREADS), 0, stream, a, g)
        if (dualU == 1) { if (materialize) RN_LAUNCH_DSTAGE(true, 1); else RN_LAUNCH_DSTAGE(false, 1); }
        else { if (materialize) RN_LAUNCH_DSTAGE(true, 2); else RN_LAUNCH_DSTAGE(false, 2); }
#undef RN_LAUNCH_DSTAGE
    }
    int main_partials() const { return mainPartials; }
    DualArgs<T> dual_args() const {
        DualArgs<T> a{};
        a.hx = d_hx; a.w = p_acc; a.yprev = p_upd; a.lo = d_lo; a.hi = d_hi;
        a.ynew = p_xi; a.wnext = p_acc_other; a.z = d_z; a.res = d_res;
        a.n = ntot(); a.nx = d.nx; a.ny = ny;
        a.lambda = (T)stepSize; a.invLambda = (T)(1.0 / stepSize);
        a.lamNext = d_lam; a.thrX = penX / stepSize; a.thrS = penXs / stepSize;
        a.st = d_state; a.partials = d_partials;
        a.crownElems = 0; a.countCrown = 1;
        a.finalizedEarly = 0; a.hist = d_hist; a.histParts = d_histParts; a.histCap = histCap;
        if (cutStage > 0) { a.crownElems = h_stageCum[cutStage] * ny; a.countCrown = (rank == 0); }
        a.regen = RN_DUAL_REGEN; a.stageOf = d_stageOf; a.sqrtp = d_sqrtp; a.dy = d_dy; a.blo = d_blo; a.bhi = d_bhi;
        return a;
    }
    int lamUploaded = 0;   // leading entries of h_lam that are on the device (kept across restarts: the theta recursion always starts at (1, 1))
    int ensure_tables(int upto) {  // lambda table and history capacity for iterations [0, upto]
        if (upto + 2 > lamCap) {
            const int cap = std::max(1024, 2 * (upto + 2));
            double *nl = nullptr, *nh = nullptr, *np = nullptr;
            if (int rc = dalloc(&nl, cap)) return rc;
            if (int rc = dalloc(&nh, cap)) return rc;
            if (int rc = dalloc(&np, (size_t)4 * cap)) return rc;
            if (int rc = dalloc(&d_histGlob, (size_t)4 * cap + 2)) return rc;
            RN_HIP(hipStreamSynchronize(stream));
            if (d_hist && histCap) {
                RN_HIP(hipMemcpy(nh, d_hist, histCap * sizeof(double), hipMemcpyDeviceToDevice));
                RN_HIP(hipMemcpy(np, d_histParts, (size_t)4 * histCap * sizeof(double), hipMemcpyDeviceToDevice));
            }
            d_lam = nl; d_hist = nh; d_histParts = np; lamCap = cap; histCap = cap;   // old arrays are freed with the context
            lamUploaded = 0;
        }
        // the theta recursion (SmpcController.cu:1513-1520) is a fixed sequence: the table is filled up to the device array's capacity at once
        // and uploaded ONCE per capacity, so that a batch in steady state starts without a copy and a host sync of its own
        while ((int)h_lam.size() < lamCap) {
            h_lam.push_back(theta1 * (1.0 / theta0 - 1.0));
            theta0 = theta1;
            theta1 = 0.5 * (std::sqrt(std::pow(theta1, 4) + 4 * std::pow(theta1, 2)) - std::pow(theta1, 2));
        }
        if (lamUploaded >= upto + 2) return RN_OK;
        RN_HIP(hipMemcpyAsync(d_lam, h_lam.data(), (size_t)lamCap * sizeof(double), hipMemcpyHostToDevice, stream));
        RN_HIP(hipStreamSynchronize(stream));   // (pageable source)
        lamUploaded = lamCap;
        return RN_OK;
    }
    int apg_reset() override {
        RN_HIP(hipSetDevice(device));
        // the optimistic paths' checkpoint buffers: allocated here, not inside a (possibly timed) rn_apg_iterate
        if (optimistic) for (int i = 0; i < 3; i++) if (!d_ck[i]) { if (int rc = dalloc(&d_ck[i], (size_t)ntot())) return rc; }
        const size_t bytes = (size_t)ntot() * sizeof(T);
        for (int i = 0; i < 2; i++) { RN_HIP(hipMemsetAsync(d_ybuf[i], 0, bytes, stream)); RN_HIP(hipMemsetAsync(d_wbuf[i], 0, bytes, stream)); }
        RN_HIP(hipMemsetAsync(d_hx, 0, bytes, stream));
        RN_HIP(hipMemsetAsync(d_z, 0, bytes, stream));
        RN_HIP(hipMemsetAsync(d_state, 0, sizeof(IterState), stream));
        p_xi = d_ybuf[0]; p_upd = d_ybuf[1]; p_acc = d_wbuf[0]; p_acc_other = d_wbuf[1]; p_acc_view = p_acc;
        acc_ready = true;  // w_0 = (1+l) 0 - l 0 = 0
        poisoned = false; carryTail = false; pendingFin = false; hxUnscaled = false; upDone = false;
        h_it = 0;      // (the lambda table is the same fixed sequence after every restart: kept, with its device copy)
        return ensure_tables(0);
    }
    // warm start: keep the duals of the previous control step, restart the momentum (theta = {1,1} => w_0 = y+)
    int apg_restart_keep_duals() {
        RN_HIP(hipSetDevice(device));
        RN_HIP(hipMemcpyAsync(p_xi, p_upd, (size_t)ntot() * sizeof(T), hipMemcpyDeviceToDevice, stream));   // y := y+
        RN_HIP(hipMemsetAsync(d_state, 0, sizeof(IterState), stream));
        h_it = 0;
        acc_ready = false;   // the first iteration re-derives w_0 = (1 + 0) y+ - 0 y
        return ensure_tables(0);
    }
    int set_warm_start(int on) override { warmStart = on ? 1 : 0; return RN_OK; }
    size_t cut_tail_offset() const { return (size_t)(h_stageCum[cutStage] - h_stageCum[cutStage - 1]) * (d.nv + 2 * d.nx); }
    // Multi-GPU, optimistic exchange: ONE collective per iteration.  The prox runs as a pure projection (what happens
    // unless a tree-global distance exceeds gamma/lambda, SmpcController.cu:793/811); every rank's dist^2 of iteration t
    // rides in the tail of iteration t+1's cut all-reduce and is checked on the device.  If a threshold was ever
    // exceeded, the batch is replayed from a checkpoint with the exact two-collective path.  Results are exact either way.
    int apg_iterate_optimistic(int n, double *primalInfs) {
        const int first = h_it;
        for (int i = 0; i < 3; i++) if (!d_ck[i]) { if (int rc = dalloc(&d_ck[i], (size_t)ntot())) return rc; }
        if (int rc = ensure_tables(h_it + n)) return rc;
        // checkpoint
        const size_t tail = cut_tail_offset();
        if (int rc = batch_open(d_cut + tail)) return fail_batch(rc);  // checkpoint of (y, y+, w), the payload's dist^2 tail and the verdict flag cleared: one launch
        const IterSave saved = save_iterates();
        carryTail = true; inBatch = true;
        for (int k = 0; k < n; k++) {
            if (!acc_ready) {
                hipLaunchKernelGGL(k_extrapolate<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, p_acc, p_xi, p_upd, (T)h_lam[h_it], ntot());
                acc_ready = true;
            }
            fuseReq = fuse_want(); fuseDone = false;
            if (fuseReq) { fuseArgs = dual_args(); fuseMat = k == n - 1; fuseLn = h_lam[h_it + 1]; }
            if (int rc = launch_sweep(0, nullptr, k == n - 1, nullptr, true)) { carryTail = false; pendingFin = false; fuseReq = false; return fail_batch(rc); }
            fuseReq = false;
            DualArgs<T> a = dual_args();
            hipEvent_t e2 = prof_begin(2);
            if (!fuseDone) launch_dual_main(a, k == n - 1, false);
            prof_end(e2);
            // bookkeeping of this iteration: folded into the next iteration's k_cut_partial_sums; the last one of the
            // batch gets a launch of its own
            if (k == n - 1) {
                hipEvent_t e3 = prof_begin(3);
                hipLaunchKernelGGL(k_finalize_optimistic<T>, dim3(1), dim3(ELT_THREADS), 0, stream, d_partials, main_partials(), d_state, d_cut + tail,
                                   d_hist, d_histParts, histCap, -1.0, -1.0);
                prof_end(e3);
            } else pendingFin = true;
            std::swap(p_xi, p_upd);
            p_acc_view = p_acc; std::swap(p_acc, p_acc_other);
            h_it++;
        }
        carryTail = false; inBatch = false;
        if (n > 0) {   // the last iteration's distances: one 2-element all-reduce per BATCH
            if (int rc = all_reduce(d_cut + tail, 2, sizeof(T) == 8, "ncclAllReduce(dist tail)")) return fail_batch(rc);
            // one more all-reduce per BATCH (MAX): the ranks agree on the verdict (every rank takes the same replay decision even if
            // an all-reduce algorithm ever delivered sums that differ in the last bit between ranks) and the batch's history
            // entries become tree-global (vecPrimalInfs, SmpcController.cu:1521)
            if (int rc = globalize_history(first, n, d_cut + tail)) return fail_batch(rc);
        }
        RN_HIP(hipGetLastError());
        int violated = 0;
        if (n > 0) {
            double votes = 0;
            RN_HIP(hipMemcpyAsync(&votes, d_histGlob, sizeof(double), hipMemcpyDeviceToHost, stream));
            RN_HIP(hipStreamSynchronize(stream));
            violated = votes > 0 ? 1 : 0;
        } else RN_HIP(hipStreamSynchronize(stream));
        if (transport == 1 && peerReady) { if (int rc = check_comm_fail()) return fail_batch(rc); }
        if (violated) return replay_exact(saved, n, primalInfs);
        if (primalInfs && n > 0) RN_HIP(hipMemcpy(primalInfs, d_hist + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        return RN_OK;
    }
    // The iterate state a batch starts from, as far as it is not in the checkpoint buffers (d_ck: y, y+, w -- written by batch_open):
    // which buffer plays which role, and the iteration count.  restore_iterates puts a context back there (the replay of a tripped batch;
    // the timing runs of the exchange auto-tuner).
    struct IterSave { T *xi, *upd, *acc, *other; bool ready; int it; };
    IterSave save_iterates() const { return IterSave{p_xi, p_upd, p_acc, p_acc_other, acc_ready, h_it}; }
    int restore_iterates(const IterSave &sv) {
        const size_t bytes = (size_t)ntot() * sizeof(T);
        p_xi = sv.xi; p_upd = sv.upd; p_acc = sv.acc; p_acc_other = sv.other; p_acc_view = p_acc; acc_ready = sv.ready;
        RN_HIP(hipMemcpyAsync(p_xi, d_ck[0], bytes, hipMemcpyDeviceToDevice, stream));
        RN_HIP(hipMemcpyAsync(p_upd, d_ck[1], bytes, hipMemcpyDeviceToDevice, stream));
        RN_HIP(hipMemcpyAsync(p_acc, d_ck[2], bytes, hipMemcpyDeviceToDevice, stream));
        h_it = sv.it;
        RN_HIP(hipMemcpyAsync(&d_state->it, &sv.it, sizeof(int), hipMemcpyHostToDevice, stream));
        RN_HIP(hipStreamSynchronize(stream));   // (sv.it is the caller's)
        return RN_OK;
    }
    // an optimistic batch whose verdict says "a threshold was exceeded": the same n iterations once more from the checkpoint, through the exact path
    int replay_exact(const IterSave &sv, int n, double *primalInfs) {
        fallbacks++;
        if (int rc = restore_iterates(sv)) return fail_batch(rc);
        const int keep = optimistic;
        optimistic = 0; inReplay = true; optHold = RN_OPT_BACKOFF;
        const int rc = apg_iterate(n, primalInfs);
        optimistic = keep; inReplay = false;
        return rc;
    }
    int batch_open(T *tail) {
        const long long n = ntot();
        const int blocks = (int)std::max<long long>(1, std::min<long long>((n / (16 / (long long)sizeof(T)) + ELT_THREADS - 1) / ELT_THREADS, (long long)numCUs * 8));
        hipLaunchKernelGGL(k_batch_open<T>, dim3(blocks), dim3(ELT_THREADS), 0, stream, (const T *)p_xi, (const T *)p_upd, (const T *)p_acc, d_ck[0], d_ck[1], d_ck[2], n, d_state, tail);
        RN_HIP(hipGetLastError());   // a checkpoint that was not taken must not be replayed from
        return RN_OK;
    }
    // Single GPU, optimistic bookkeeping: the same idea without a collective.  The fused dual update runs the prox as a
    // pure projection; instead of a decision launch after every iteration (the 64-workgroup fix-up launch, ~5 us that
    // almost always exits at once) the fold of its partials, the history entry and the distance check of iteration t ride
    // in k_up_chain of iteration t+1 as one more workgroup.  If a threshold was exceeded anywhere in the batch, the batch
    // is replayed from its checkpoint through the exact path -- the result is exact either way.
    int apg_iterate_optimistic_local(int n, double *primalInfs) {
        const int first = h_it;
        for (int i = 0; i < 3; i++) if (!d_ck[i]) { if (int rc = dalloc(&d_ck[i], (size_t)ntot())) return rc; }
        if (int rc = ensure_tables(h_it + n)) return rc;
        if (int rc = batch_open(nullptr)) return fail_batch(rc);       // checkpoint of (y, y+, w) + the verdict flag cleared: one launch
        const IterSave saved = save_iterates();
        for (int k = 0; k < n; k++) {
            if (!acc_ready) {
                hipLaunchKernelGGL(k_extrapolate<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, p_acc, p_xi, p_upd, (T)h_lam[h_it], ntot());
                acc_ready = true;
            }
            fuseReq = fuse_want(); fuseDone = false;
            if (fuseReq) { fuseArgs = dual_args(); fuseMat = k == n - 1; fuseLn = h_lam[h_it + 1]; }
            if (int rc = launch_sweep(0, nullptr, k == n - 1, nullptr, true)) { pendingFin = false; fuseReq = false; return fail_batch(rc); }
            fuseReq = false;
            DualArgs<T> a = dual_args();
            hipEvent_t e2 = prof_begin(2);
            if (!fuseDone) launch_dual_main(a, k == n - 1, false);
            prof_end(e2);
            if (k == n - 1) {   // the last iteration's bookkeeping gets a launch of its own
                hipEvent_t e3 = prof_begin(3);
                hipLaunchKernelGGL(k_finalize_optimistic<T>, dim3(1), dim3(ELT_THREADS), 0, stream, d_partials, main_partials(), d_state, (T *)nullptr,
                                   d_hist, d_histParts, histCap, penX / stepSize, penXs / stepSize, h_verdict);
                prof_end(e3);
            } else pendingFin = true;
            std::swap(p_xi, p_upd);
            p_acc_view = p_acc; std::swap(p_acc, p_acc_other);
            h_it++;
        }
        RN_HIP(hipGetLastError());
        int violated = 0;
        if (h_verdict && n > 0) {      // written by the batch's last launch (k_finalize_optimistic): no copy, one synchronisation
            RN_HIP(hipStreamSynchronize(stream));
            violated = *(volatile int *)h_verdict;
        } else {
            RN_HIP(hipMemcpyAsync(&violated, &d_state->violated, sizeof(int), hipMemcpyDeviceToHost, stream));
            RN_HIP(hipStreamSynchronize(stream));
        }
        if (violated) return replay_exact(saved, n, primalInfs);
        if (primalInfs && n > 0) RN_HIP(hipMemcpy(primalInfs, d_hist + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        return RN_OK;
    }
    // SmpcController::allocateApgAlgorithm sizes its per-iteration storage by maxIterations once (SmpcController.cu:124-151):
    // the iteration tables (lambda_k, vecPrimalInfs and its parts) and the optimistic paths' checkpoint buffers for `n` iterations
    // are allocated here, so that no control step allocates device memory (the reference's leak check, :1612-1623, would flag it)
    // ---- one-shot exchange at the cut (kernels.hpp, PeerTable) -----------------------------------------------------------------
    // The inbox of this rank: 2 buffers x nranks sources x slots elements of 8-byte packets, in UNCACHED device memory (no L2 of
    // either side may hold a packet back), exported as a hipIpcMemHandle_t; peers map it and write into it from their kernels.
    unsigned long long *d_inbox = nullptr;
    size_t inboxBytes = 0;
    PeerTable h_peer{};   // handed to the kernels by value (SweepArgs::peer)
    std::vector<void *> ipcOpened;
    bool peerReady = false, inBatch = false;
    // Transport of the per-iteration exchange at the cut.  transportReq is what the caller asked for (RN_EXCHANGE_AUTO unless told:
    // rn_set_exchange_transport, $RAPIDNET_EXCHANGE), `transport` what the batches run: 0 = the communicator's all-reduce on the solver's
    // stream, 1 = one-shot peer writes.  AUTO is resolved by exchange_autotune -- by the first device-resident batch, or when the caller
    // asks (rn_exchange_autotune): both candidates run the context's own iterations, the ranks agree on the faster one.
    int transportReq = RN_EXCHANGE_AUTO, transport = 0;
    bool tuned = false, inTune = false, oneShotBroken = false;
    double tuneInfo[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rn_exchange_autotune: {chosen, candidates, us/it collective (max over ranks), us/it one-shot (max), own collective, own one-shot, iterations, tunes run}
    double *d_tune = nullptr;    // 8 doubles: the ranks' agreement on the timings (allocated with the context: a control step never allocates)
    T *d_ckView = nullptr;       // the tuner's copy of the accelerated dual a getter would show (allocated when the one-shot transport becomes a candidate)
    // The forward walk and the dual update of the nodes it has walked in ONE launch (k_down_chain_dual): the optimistic batches ask for
    // it per iteration (fuseReq + the dual update's arguments), the sweep says whether it happened.
    // Default (round 6): on, with the number of workgroups per chain BY SHAPE (fuse_split) -- one where the chains (nearly) fill the chip: the whole
    // 493-scenario tree -2.2 % per iteration dense, -5.6 % structured, a 1/2 shard -0.7 %; several on small trees and smaller shards (with one, the 62
    // workgroups of a 1/8 shard cannot keep as many bytes in flight as the stage-tiled kernel's grid: +2.7 %; profiles/r06_ab_fuse_by_shape.txt).  rn_set_fused_walk_dual(ctx, 0 / 1) or
    // $RAPIDNET_FUSE_DOWN_DUAL = 0 / 1 (read when the context runs its first batch) force it either way; while the per-launch profiling
    // of rn_profile_enable is on, the dual update always runs as a launch of its own (the kernel north_star's roofline target names).
    bool fuseReq = false, fuseDone = false, fuseMat = false;
    bool upDone = false;   // the chain walk of the NEXT plain sweep has been done by the last fused launch (k_down_chain_dual UPLIN): consumed by that sweep
    DualArgs<T> fuseArgs{};
    double fuseLn = 0.0;
    int fuseMode = -2;     // -2: not decided yet ($RAPIDNET_FUSE_DOWN_DUAL, else by shape), -1: by shape, 0 / 1: forced
    int set_fused_walk_dual(int on) override { RN_CHECK(on >= -1 && on <= 1, RN_E_ARG, "rn_set_fused_walk_dual: 0, 1 or -1 (by shape)"); fuseMode = on; return RN_OK; }
    bool fuse_by_shape() const {      // (profiles/r06_ab_fuse_by_shape.txt: 493 chains -2.2 % dense / -5.6 % structured, 247 chains of a 1/2 shard -0.7 %, 124 chains: a tie,
        const int cs = chainStage, K = h_stageCum[cs + 1] - h_stageCum[cs];      //  62 and 31 chains: +3-4 % with ONE workgroup per chain)
        return 4 * K >= 3 * numCUs;
    }
    // workgroups per chain of the fused launch: one where the chains (nearly) fill the chip; otherwise up to four, as many as give every CU a
    // workgroup, each updating the dual of its own slice of the chain's rows (k_down_chain_dual, P) -- the dual update of a 1/8 shard's 62 chains
    // then runs on 248 workgroups instead of 62.  With that the one launch is never slower than the two (profiles/r06_ab_fuse_split.txt: 31-scenario
    // tree -1 ... -3 % dense, -4.4 % structured; 1/4 shard -1.1 %; 1/8 shard -0.5 %; more than four redo too much of the walk).
    // rn_debug_set_knob(RN_KNOB_FUSE_SPLIT): n > 0 forces n, 0 = one per chain and only by fuse_by_shape() (the rule before the split existed).
    int fuse_split() const {
        const int cs = chainStage, K = h_stageCum[cs + 1] - h_stageCum[cs], L = d.N - cs;
        if (knob[RN_KNOB_FUSE_SPLIT] > 0) return std::max(1, std::min(knob[RN_KNOB_FUSE_SPLIT], L));
        if (knob[RN_KNOB_FUSE_SPLIT] == 0 || fuse_by_shape()) return 1;
        return std::max(1, std::min(std::min(L, 4), numCUs / std::max(K, 1)));
    }
    bool fuse_want() {
        if (fuseMode == -2) { const char *e = std::getenv("RAPIDNET_FUSE_DOWN_DUAL"); fuseMode = e ? (std::atoi(e) != 0 ? 1 : 0) : -1; }
        const bool shapeSaysYes = knob[RN_KNOB_FUSE_SPLIT] == 0 ? fuse_by_shape() : true;
        return (fuseMode == 1 || (fuseMode == -1 && shapeSaysYes)) && dualU != 0 && !prof;
    }
    unsigned int peerSeq = 0;     // sequence number of the last one-shot exchange (the same on every rank: they issue the same exchanges)
    unsigned int peer_slots() const { return (unsigned int)((size_t)(h_stageCum[cutStage] - h_stageCum[cutStage - 1]) * (d.nv + 2 * d.nx) + 2); }
    int peer_inbox_create(void *handle64) override {
        RN_CHECK(handle64, RN_E_ARG, "rn_peer_inbox_create: null output");
        // (one rank: the context writes to and reads from its own inbox -- what a rank executes except the wire, for timing runs)
        RN_CHECK(cutStage > 0 && nranks >= 1 && nranks <= PEER_MAX, RN_E_STATE, "rn_peer_inbox_create: a sharded context (cut stage set) of at most 16 ranks is required");
        RN_HIP(hipSetDevice(device));
        if (!d_inbox) {
            inboxBytes = (size_t)2 * nranks * peer_slots() * PeerPk<T>::N * sizeof(unsigned long long);
            RN_HIP(hipExtMallocWithFlags((void **)&d_inbox, inboxBytes, hipDeviceMallocUncached));
            RN_HIP(hipMemset(d_inbox, 0, inboxBytes));      // tag 0 = "never written" (sequence numbers start at 1)
            RN_HIP(hipDeviceSynchronize());
        }
        hipIpcMemHandle_t h;
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "rn_peer_inbox_create hands out 64-byte handles");
        RN_HIP(hipIpcGetMemHandle(&h, d_inbox));
        std::memcpy(handle64, &h, sizeof h);
        return RN_OK;
    }
    unsigned long long *peer_inbox_ptr() override { return d_inbox; }
    int peer_table_install() {
        h_peer.nranks = nranks; h_peer.rank = rank; h_peer.slots = peer_slots();
        double ms = 2000.0;   // a reader waits at most this long for a peer's packets, then gives up with RN_E_COMM
        if (const char *e = std::getenv("RAPIDNET_ONESHOT_TIMEOUT_MS")) { const double v = std::atof(e); if (v > 0) ms = v; }
        h_peer.timeoutTicks = (unsigned long long)(ms * 1e5);   // 100 MHz wall clock
        h_peer.own = h_peer.inbox[rank];
        peerReady = true; peerSeq = 0;
        return RN_OK;
    }
    int peer_inbox_connect(const void *handles, int n) override {
        RN_CHECK(handles && n == nranks, RN_E_ARG, "rn_peer_inbox_connect: one 64-byte handle per rank expected");
        RN_CHECK(d_inbox != nullptr, RN_E_STATE, "rn_peer_inbox_connect before rn_peer_inbox_create");
        // once per context: the sequence numbers of the ranks advance together from the connect on, and a second connect of one rank
        // would restart its tags over packets its peers still hold (a new set of peers needs a new context)
        RN_CHECK(!peerReady, RN_E_STATE, "rn_peer_inbox_connect: the inboxes of this context are already connected");
        RN_HIP(hipSetDevice(device));
        for (int r = 0; r < nranks; r++) {
            if (r == rank) { h_peer.inbox[r] = d_inbox; continue; }
            hipIpcMemHandle_t h;
            std::memcpy(&h, (const char *)handles + (size_t)64 * r, sizeof h);
            void *ptr = nullptr;
            RN_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
            ipcOpened.push_back(ptr);
            h_peer.inbox[r] = (unsigned long long *)ptr;
        }
        return peer_table_install();
    }
    // the ranks are contexts of ONE process (tests): a peer's inbox is an ordinary device pointer of the same address space
    int peer_inbox_connect_local(CtxBase **peers, int n) override {
        RN_CHECK(peers && n == nranks, RN_E_ARG, "rn_debug_peer_inbox_connect_local: one context per rank expected");
        for (int r = 0; r < nranks; r++) {
            RN_CHECK(peers[r] && peers[r]->peer_inbox_ptr(), RN_E_STATE, "rn_debug_peer_inbox_connect_local: a rank has no inbox (rn_peer_inbox_create)");
            RN_CHECK(!peerReady, RN_E_STATE, "rn_debug_peer_inbox_connect_local: the inboxes of this context are already connected");
            h_peer.inbox[r] = peers[r]->peer_inbox_ptr();
        }
        RN_CHECK(h_peer.inbox[rank] == d_inbox, RN_E_ARG, "rn_debug_peer_inbox_connect_local: contexts must be given in rank order");
        RN_HIP(hipSetDevice(device));
        return peer_table_install();
    }
    int debug_peer_seq(unsigned int seq) override {      // test hook: the next exchange gets sequence number seq + 1 (all ranks alike)
        RN_CHECK(peerReady, RN_E_STATE, "rn_debug_peer_seq: connect the inboxes first");
        peerSeq = seq;
        return RN_OK;
    }
    int set_exchange_transport(int t) override {
        RN_CHECK(t == RN_EXCHANGE_COLLECTIVE || t == RN_EXCHANGE_ONESHOT || t == RN_EXCHANGE_AUTO, RN_E_ARG,
                 "rn_set_exchange_transport: RN_EXCHANGE_COLLECTIVE, RN_EXCHANGE_ONESHOT or RN_EXCHANGE_AUTO");
        if (t == RN_EXCHANGE_ONESHOT && !peerReady) { if (int rc = exchange_prepare()) return rc; }   // (a real communicator: the library wires the inboxes itself)
        RN_CHECK(t != RN_EXCHANGE_ONESHOT || peerReady, RN_E_STATE, "rn_set_exchange_transport: the peers' inboxes are not connected (rn_peer_inbox_connect, or a communicator over which the library can exchange the handles)");
        transportReq = t;
        transport = t == RN_EXCHANGE_ONESHOT ? 1 : 0;
        tuned = t != RN_EXCHANGE_AUTO;
        return RN_OK;
    }
    // ---- the exchange chooses itself (round 6) ----------------------------------------------------------------------------------
    // all-reduce of raw integers over the library's own communicator (the IPC handles travel this way: every rank fills its own slot of a
    // zeroed buffer, the sum is the gather)
    int all_reduce_bytes(void *buf, size_t count, const char *what) {
        RN_CHECK(comm != nullptr, RN_E_STATE, std::string(what) + ": no communicator");
        const int rc = g_nccl.AllReduce(buf, buf, count, 1 /* ncclUint8 */, 0 /* ncclSum */, comm, stream);
        RN_CHECK(rc == 0, RN_E_COMM, std::string(what) + " failed: " + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "?"));
        return RN_OK;
    }
    // One-shot transport made available WITHOUT the caller's help: the inbox of this rank, its IPC handle gathered over the RCCL
    // communicator, the peers' inboxes mapped.  Collective (every rank of the communicator calls it at the same point: rn_comm_init /
    // rn_create_sharded / rn_set_exchange_transport(ONESHOT) do).  A rank that cannot create or map an inbox makes EVERY rank give the
    // transport up (a MAX all-reduce of the failure flags): returns RN_OK with peerReady == false, and AUTO then has one candidate.
    bool prepared = false;
    int exchange_prepare() {
        if (prepared || peerReady || oneShotBroken) return RN_OK;
        // (stand-in communicators: the test wires the inboxes.  A ONE-rank communicator goes through all of it too -- the rank's own inbox, the
        //  gather, the agreement: what a one-GPU box can exercise of this path)
        if (comm == nullptr || nranks < 1 || nranks > PEER_MAX || cutStage <= 0) return RN_OK;
        prepared = true;
        RN_HIP(hipSetDevice(device));
        unsigned char handle[64] = {0};
        int bad = peer_inbox_create(handle) != RN_OK ? 1 : 0;
        unsigned char *d_h = nullptr;
        const size_t hb = (size_t)64 * nranks + 8;      // + the failure flags' slot
        if (hipMalloc((void **)&d_h, hb) != hipSuccess) { (void)hipGetLastError(); d_h = nullptr; bad = 1; }
        std::vector<unsigned char> all(hb, 0);
        if (d_h) {
            std::memcpy(all.data() + (size_t)64 * rank, handle, 64);
            all[(size_t)64 * nranks] = (unsigned char)(bad ? 1 : 0);      // (sum over <= 16 ranks: no overflow)
            if (hipMemcpyAsync(d_h, all.data(), hb, hipMemcpyHostToDevice, stream) != hipSuccess) bad = 1;
        }
        // every rank MUST issue the same collectives whatever happened to it: a rank without a buffer contributes through a fallback slot
        int rc = RN_OK;
        if (d_h) rc = all_reduce_bytes(d_h, hb, "ncclAllReduce(inbox handles)");
        else { err = "exchange_prepare: no device memory for the handle exchange"; rc = RN_E_HIP; }
        if (rc == RN_OK && (hipMemcpyAsync(all.data(), d_h, hb, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) { (void)hipGetLastError(); rc = RN_E_HIP; }
        if (d_h) (void)hipFree(d_h);
        if (rc != RN_OK) { oneShotBroken = true; return rc; }      // the communicator itself failed: the caller must know
        if (all[(size_t)64 * nranks] == 0) {
            if (peer_inbox_connect(all.data(), nranks) != RN_OK) { (void)hipGetLastError(); bad = 1; }
        } else bad = 1;
        // second agreement: did every rank map every peer?
        double flag = bad ? 1.0 : 0.0;
        RN_HIP(hipMemcpyAsync(d_tune, &flag, sizeof flag, hipMemcpyHostToDevice, stream));
        if (int rc2 = all_reduce(d_tune, 1, true, "ncclAllReduce(inbox set-up verdict)", 2 /* ncclMax */)) { oneShotBroken = true; peerReady = false; return rc2; }
        RN_HIP(hipMemcpyAsync(&flag, d_tune, sizeof flag, hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        if (flag != 0.0) { peerReady = false; oneShotBroken = true; }      // somebody could not: nobody uses the transport (err keeps this rank's reason, if it was this rank)
        else if (transportReq == RN_EXCHANGE_ONESHOT) transport = 1;
        if (peerReady && transportReq == RN_EXCHANGE_AUTO && !d_ckView) { if (int rc3 = dalloc(&d_ckView, (size_t)ntot())) return rc3; }   // the tuner will run: its buffer now, not inside a control step
        return RN_OK;
    }
    // Times the candidates on THIS context's own iterations and keeps the faster: `iters` device-resident iterations per candidate (after
    // a warm-up run of half as many) from the current iterate state, which is restored afterwards -- iterates, iteration count, batch
    // counters: a caller cannot tell that the runs happened, except by the clock.  The ranks' times are combined by a MAX all-reduce, so
    // every rank takes the same decision.  Collective: every rank calls it at the same point (rn_apg_iterate does, in its first batch).
    int exchange_autotune(int iters) {
        RN_CHECK(factored && affine_ready, RN_E_STATE, "rn_exchange_autotune before the factor step / affine terms");
        RN_CHECK(!poisoned, RN_E_STATE, "rn_exchange_autotune: an earlier batch failed half-way; call rn_apg_reset first");
        RN_CHECK(iters >= 1, RN_E_ARG, "rn_exchange_autotune: iterations >= 1");
        RN_HIP(hipSetDevice(device));
        tuned = true;
        const bool sharded = has_comm() && cutStage > 0 && nranks >= 1 && optimistic;
        if (sharded && !peerReady) { if (int rc = exchange_prepare()) return rc; }
        const bool canOne = sharded && peerReady && !oneShotBroken;
        tuneInfo[1] = 1.0 + (canOne ? 2.0 : 0.0); tuneInfo[6] = 0; tuneInfo[2] = tuneInfo[3] = tuneInfo[4] = tuneInfo[5] = 0.0;
        if (!canOne) { transport = 0; tuneInfo[0] = 0; return RN_OK; }     // one candidate: nothing to time
        for (int i = 0; i < 3; i++) if (!d_ck[i]) { if (int rc = dalloc(&d_ck[i], (size_t)ntot())) return rc; }
        // what the timing runs must leave as they found it (the iterates themselves are in the batch's own checkpoint)
        const IterSave sv = save_iterates();
        const long kOpt = optBatches, kExact = exactBatches, kFall = fallbacks; const int kHold = optHold;
        // ... and what RN_BUF_ACC_* shows: after a batch that is w_t, in the buffer the timing runs are about to reuse (the checkpoint holds w_{t+1}, the next input)
        T *const view = p_acc_view;
        const size_t vbytes = (size_t)ntot() * sizeof(T);
        if (view != sv.acc) {
            if (!d_ckView) { if (int rc = dalloc(&d_ckView, (size_t)ntot())) return rc; }
            RN_HIP(hipMemcpyAsync(d_ckView, view, vbytes, hipMemcpyDeviceToDevice, stream));
        }
        double own[2] = {0, 0};
        int failed[2] = {0, 0};
        inTune = true;
        for (int cand = 0; cand < 2; cand++) {
            transport = cand;
            for (int pass = 0; pass < 2 && !failed[cand]; pass++) {
                const int n = pass == 0 ? std::max(iters / 2, 8) : iters;
                optHold = 0;
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = apg_iterate_optimistic(n, nullptr);     // ends with a synchronisation
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rc != RN_OK) {      // (a one-shot reader's time-out is every rank's RN_E_COMM: all ranks come here together)
                    failed[cand] = 1; poisoned = false;
                    (void)hipMemsetAsync(&d_state->commFail, 0, sizeof(int), stream);
                }
                if (pass == 1 && rc == RN_OK) own[cand] = us / n;
                if (int rc2 = restore_iterates(sv)) { inTune = false; return fail_batch(rc2); }
            }
        }
        inTune = false;
        optBatches = kOpt; exactBatches = kExact; fallbacks = kFall; optHold = kHold;
        if (view != sv.acc) { RN_HIP(hipMemcpyAsync(view, d_ckView, vbytes, hipMemcpyDeviceToDevice, stream)); p_acc_view = view; }
        if (knob[RN_KNOB_TUNE_BIAS_US] != -1) own[1] += (double)knob[RN_KNOB_TUNE_BIAS_US];      // test of the selection: this rank's one-shot time, biased
        // every rank the same figures: MAX over the ranks of {time collective, time one-shot, failed collective, failed one-shot}
        double h[4] = {own[0], own[1], (double)failed[0], (double)failed[1]};
        RN_HIP(hipMemcpyAsync(d_tune, h, sizeof h, hipMemcpyHostToDevice, stream));
        transport = 0;
        if (int rc = all_reduce(d_tune, 4, true, "ncclAllReduce(exchange timings)", 2 /* ncclMax */)) return rc;
        RN_HIP(hipMemcpyAsync(h, d_tune, sizeof h, hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        RN_CHECK(h[2] == 0.0, RN_E_COMM, "rn_exchange_autotune: the collective exchange failed on a rank (" + err + ")");
        if (h[3] != 0.0) oneShotBroken = true;
        transport = (h[3] == 0.0 && h[1] < h[0]) ? 1 : 0;
        tuneInfo[0] = transport; tuneInfo[2] = h[0]; tuneInfo[3] = h[3] != 0.0 ? -1.0 : h[1]; tuneInfo[4] = own[0]; tuneInfo[5] = own[1]; tuneInfo[6] = iters; tuneInfo[7] += 1.0;
        return RN_OK;
    }
    int exchange_prepare_api() override { return transportReq == RN_EXCHANGE_COLLECTIVE ? RN_OK : exchange_prepare(); }
    int exchange_autotune_api(int iters, double *out) override {
        RN_CHECK(iters >= 0, RN_E_ARG, "rn_exchange_autotune: iterations >= 0");
        if (iters > 0) {
            RN_CHECK(transportReq == RN_EXCHANGE_AUTO, RN_E_STATE, "rn_exchange_autotune: the transport was fixed by rn_set_exchange_transport");
            if (int rc = exchange_autotune(iters)) return rc;
        }
        if (out) { for (int i = 0; i < 8; i++) out[i] = tuneInfo[i]; out[0] = transport; }
        return RN_OK;
    }
    // after a batch in one-shot mode: did a reader give up waiting?  (one 4-byte read-back; the stream has been synchronised)
    int check_comm_fail() {
        int f = 0;
        RN_HIP(hipMemcpyAsync(&f, &d_state->commFail, sizeof(int), hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        RN_CHECK(f == 0, RN_E_COMM, "one-shot exchange: a peer's packets did not arrive within the time-out (a rank is missing or far behind)");
        return RN_OK;
    }
    size_t injectBytes = 0;
    int inject_allocation(size_t bytes) override { injectBytes = bytes; return RN_OK; }
    int reserve_iterations(int n) override {
        RN_CHECK(n >= 0, RN_E_ARG, "rn_reserve_iterations: negative iteration count");
        RN_HIP(hipSetDevice(device));
        for (int i = 0; i < 3; i++) if (!d_ck[i]) { if (int rc = dalloc(&d_ck[i], (size_t)ntot())) return rc; }
        if (n + 2 > lamCap) return ensure_tables(n);
        return RN_OK;
    }
    int set_exchange_mode(int mode) override {
        RN_CHECK(mode == 0 || mode == 1, RN_E_ARG, "rn_set_exchange_mode: 0 exact, 1 optimistic");
        optimistic = mode; optHold = 0;
        return RN_OK;
    }
    int apg_iterate(int n, double *primalInfs) override {
        RN_CHECK(factored && affine_ready, RN_E_STATE, "rn_apg_iterate before the factor step / affine terms");
        RN_CHECK(n >= 0, RN_E_ARG, "rn_apg_iterate: negative iteration count");
        RN_CHECK(!poisoned, RN_E_STATE, "rn_apg_iterate: an earlier batch failed half-way; call rn_apg_reset first");
        RN_HIP(hipSetDevice(device));
        // optimistic batches (prox as a pure projection, verified afterwards); after a replay the next RN_OPT_BACKOFF batches
        // go straight through the exact path (every rank sees the same verdicts, so sharded ranks stay in step)
        const bool wantOptSharded = has_comm() && cutStage > 0 && optimistic && n > 0;
        // single GPU: worth a checkpoint (3 vector copies) and a read-back per batch once the batch is long enough
        const bool wantOptLocal = !has_comm() && cutStage <= 0 && optimistic && n >= RN_OPT_LOCAL_MIN;
        if (wantOptSharded && transportReq == RN_EXCHANGE_AUTO && !tuned && !inTune && !inReplay) {   // the exchange chooses itself, once
            if (int rc = exchange_autotune(std::min(std::max(n, 20), 100))) return rc;
        }
        if ((wantOptSharded || wantOptLocal) && optHold > 0 && !inReplay) optHold--;
        else if (wantOptSharded) { optBatches++; return apg_iterate_optimistic(n, primalInfs); }
        else if (wantOptLocal) { optBatches++; return apg_iterate_optimistic_local(n, primalInfs); }
        if (!inReplay && n > 0) exactBatches++;
        const int first = h_it;
        if (int rc = ensure_tables(h_it + n)) return rc;
        inBatch = true;
        for (int k = 0; k < n; k++) {
            if (!acc_ready) {   // re-derive w_t after manual buffer edits (SmpcController.cu:1514)
                hipLaunchKernelGGL(k_extrapolate<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, p_acc, p_xi, p_upd, (T)h_lam[h_it], ntot());
                acc_ready = true;
            }
            const bool last = (k == n - 1);
            if (int rc = launch_sweep(0, nullptr, last)) return fail_batch(rc);
            DualArgs<T> a = dual_args();
            hipEvent_t e2 = prof_begin(2);
            const bool exactSharded = has_comm() && cutStage > 0;   // its fix-up pass and k_finalize fold eltBlocks partials: flat kernel
            launch_dual_main(a, last, exactSharded);
            prof_end(e2);
            hipEvent_t e3 = prof_begin(3);
            if (exactSharded) {   // tree-global distances: sum the ranks' dist^2 (2 doubles) before deciding
                hipLaunchKernelGGL(k_reduce_dist<>, dim3(1), dim3(ELT_THREADS), 0, stream, d_partials, eltBlocks, d_dist2);
                if (int rc = all_reduce(d_dist2, 2, true, "ncclAllReduce(dist)")) return fail_batch(rc);
                hipLaunchKernelGGL(k_decide_from<>, dim3(1), dim3(1), 0, stream, d_dist2, d_state, a.thrX, a.thrS);
                if (last) hipLaunchKernelGGL((k_dual_fused<T, true, true>), dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
                else hipLaunchKernelGGL((k_dual_fused<T, false, true>), dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
                hipLaunchKernelGGL(k_finalize<>, dim3(1), dim3(ELT_THREADS), 0, stream, d_partials, eltBlocks, d_state, d_hist, d_histParts, histCap);
            } else {   // single GPU: the (small) fix-up launch also decides and does the bookkeeping (kernels.hpp, decideHere)
                a.finalizedEarly = 1;
                a.decideHere = 1; a.itHost = h_it; a.nMain = main_partials(); a.mainPartials = d_partials; a.partials = d_partials2;
                const int fixBlocks = std::min(eltBlocks, RN_FIXUP_BLOCKS);
                if (last) hipLaunchKernelGGL((k_dual_fused<T, true, true>), dim3(fixBlocks), dim3(ELT_THREADS), 0, stream, a);
                else hipLaunchKernelGGL((k_dual_fused<T, false, true>), dim3(fixBlocks), dim3(ELT_THREADS), 0, stream, a);
            }
            prof_end(e3);
            // rotate: y_t := y+_{t-1} (old upd), y+_t := buffer just written; w_{t+1} becomes the sweep input
            std::swap(p_xi, p_upd);
            p_acc_view = p_acc;
            std::swap(p_acc, p_acc_other);
            h_it++;
        }
        inBatch = false;
        RN_HIP(hipGetLastError());
        // sharded: the batch's history entries become tree-global (one MAX all-reduce per batch)
        if (has_comm() && cutStage > 0 && n > 0) { if (int rc = globalize_history(first, n, nullptr)) return fail_batch(rc); }
        if (transport == 1 && peerReady && has_comm() && cutStage > 0 && n > 0) { if (int rc = check_comm_fail()) return fail_batch(rc); }
        if (primalInfs && n > 0) {
            RN_HIP(hipStreamSynchronize(stream));
            RN_HIP(hipMemcpy(primalInfs, d_hist + first, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
        }
        return RN_OK;
    }
    // Sharded contexts, once per batch: vecPrimalInfs[first .. first + n) of this rank (arg-max over its own nodes) -> the
    // tree-global values, on every rank (SmpcController.cu:1480-1496, :1521); element 0 of the payload carries the ranks'
    // verdict on the optimistic batch (tail = the all-reduced dist^2 of its last iteration; nullptr: no verdict).
    int globalize_history(int first, int n, T *tail) {
        hipLaunchKernelGGL(k_batch_close_pack<T>, dim3(1), dim3(ELT_THREADS), 0, stream, (const T *)tail, d_state, penX / stepSize, penXs / stepSize,
                           (const double *)d_histParts, first, n, d_histGlob);
        // (the last element: one-shot exchange, "a reader of this rank gave up waiting" -- the MAX makes it every rank's verdict, so all
        //  ranks return RN_E_COMM for the batch together instead of the late one alone)
        if (int rc = all_reduce(d_histGlob, (size_t)4 * n + 2, true, "ncclAllReduce(verdict + primal infeasibilities)", 2 /* ncclMax */)) return rc;
        hipLaunchKernelGGL(k_batch_close_unpack<>, dim3(1), dim3(ELT_THREADS), 0, stream, (const double *)d_histGlob, d_hist, first, n, d_state);
        if (int rc = comm_check()) return rc;
        RN_HIP(hipGetLastError());
        return RN_OK;
    }
    // A batch that fails half-way (a launch or a collective returned an error after some of its iterations were enqueued) leaves
    // the rotating iterate buffers, the device-side iteration counter and the host's view of them out of step: the context
    // refuses further iterations until rn_apg_reset / rn_fbe_reset (or rn_control_action, which resets) instead of continuing
    // from an inconsistent accelerated dual.
    bool poisoned = false;
    int fail_batch(int rc) {
        poisoned = true; carryTail = false; pendingFin = false; inBatch = false; hxUnscaled = false; upDone = false;
        err += " -- the batch was abandoned half-way: call rn_apg_reset before iterating again";
        return rc;
    }
    // which kernels a batch of this context launches (bench.py names them in its JSON line instead of assuming)
    int kernel_info(int *out) override {
        RN_CHECK(out, RN_E_ARG, "rn_get_kernel_info: null output");
        int G = 0, NL = 0;
        stream_shape(&G, &NL);
        const bool exactSharded = has_comm() && cutStage > 0 && !optimistic;
        out[0] = (dualU > 0 && !exactSharded) ? 1 : 0;   // 1: k_dual_stage is the main pass of the fused dual update, 0: the flat k_dual_fused
        out[1] = (dualU > 0 && !exactSharded) ? dualBlocks : eltBlocks;
        out[2] = dualU > 0 ? dshape.trips : 0;
        out[3] = dualU;                                  // 2: double-buffered variant
        out[4] = G; out[5] = NL;                         // k_stream_gemv: columns per span, 16-byte slots per thread and span
        out[6] = chainStage;
        out[7] = v_lv_is_slab() ? 1 : 0;                 // 1: k_gemm_vlv (slab), 0: two k_gemm_shared launches
        return RN_OK;
    }
    int counters(long *out) override {
        RN_CHECK(out, RN_E_ARG, "rn_get_counters: null output");
        out[0] = optBatches; out[1] = exactBatches; out[2] = fallbacks; out[3] = optHold;
        return RN_OK;
    }
    int hist_parts(int first, int n, double *out) override {
        RN_CHECK(first >= 0 && n >= 0 && first + n <= h_it, RN_E_ARG, "rn_get_history_parts: range outside the iterations run");
        RN_HIP(hipStreamSynchronize(stream));
        RN_HIP(hipMemcpy(out, d_histParts + (size_t)4 * first, (size_t)4 * n * sizeof(double), hipMemcpyDeviceToHost));
        return RN_OK;
    }
    int control_action(const double *x0, const double *up, const double *dp, const double *dhat, const double *ahat, int maxIt,
                       int project, double *u0) override {
        RN_CHECK(u0, RN_E_ARG, "rn_control_action: null output");
        if (injectBytes) {   // rn_debug_inject_allocation: a deliberate leak inside this control step (test of the callers' leak check)
            char *leak = nullptr;
            if (int rc = dalloc(&leak, injectBytes)) return rc;
            injectBytes = 0;
        }
        if (int rc = update_state_control(x0, up, dp)) return rc;
        if (int rc = eliminate(dhat, ahat)) return rc;
        if (warmStart && h_it > 0 && !poisoned) { if (int rc = apg_restart_keep_duals()) return rc; }   // y, y+ kept; theta = {1,1}
        else if (int rc = apg_reset()) return rc;
        if (int rc = apg_iterate(maxIt, nullptr)) return rc;
        T *src = d_u;
        if (project) {  // SmpcController.cu:1647-1650: clamp with the (scaled) bounds of the root node
            RN_HIP(hipMemcpyAsync(d_tmp, d_u, d.nu * sizeof(T), hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_clamp_vec<T>, dim3(1), dim3(128), 0, stream, d_tmp, d_lo + 2 * d.nx, d_hi + 2 * d.nx, d.nu);
            src = d_tmp;
        }
        return download(u0, src, d.nu);
    }

    // ---- step-wise API -----------------------------------------------------------------------------------
    int extrapolate(double lambda) override {
        RN_HIP(hipSetDevice(device));
        hipLaunchKernelGGL(k_extrapolate<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, p_acc, p_xi, p_upd, (T)lambda, ntot());
        RN_HIP(hipGetLastError());
        p_acc_view = p_acc; acc_ready = true;
        return RN_OK;
    }
    int solve_step() override {
        RN_CHECK(factored && affine_ready, RN_E_STATE, "rn_solve_step before the factor step / affine terms");
        RN_HIP(hipSetDevice(device));
        p_acc_view = p_acc;
        return launch_sweep();
    }
    int prox() override {
        RN_CHECK(factored, RN_E_STATE, "rn_proximal_fun_g before the factor step");
        RN_HIP(hipSetDevice(device));
        DualArgs<T> a = dual_args();
        a.w = p_acc_view;
        hipLaunchKernelGGL(k_prox_clamp<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
        hipLaunchKernelGGL(k_decide<>, dim3(1), dim3(ELT_THREADS), 0, stream, d_partials, eltBlocks, d_state, a.thrX, a.thrS);
        hipLaunchKernelGGL(k_prox_soft<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, a);
        RN_HIP(hipGetLastError());
        return RN_OK;
    }
    int residual() override {
        RN_HIP(hipSetDevice(device));
        hipLaunchKernelGGL(k_axpby<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, d_res, d_hx, d_z, (T)1, (T)-1, ntot());
        RN_HIP(hipGetLastError());
        return RN_OK;
    }
    int dual_update() override {
        RN_HIP(hipSetDevice(device));
        if (algorithm != RN_ALG_APG) return fbe_dual_update();   // SmpcController.cu:866-880
        hipLaunchKernelGGL(k_axpby<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, p_upd, p_acc_view, d_res, (T)1, (T)stepSize, ntot());
        RN_HIP(hipGetLastError());
        acc_ready = false;
        return RN_OK;
    }
    int primal_infeasibility(double *value) override {
        RN_CHECK(value, RN_E_ARG, "rn_update_primal_infeasibility: null output");
        RN_HIP(hipSetDevice(device));
        hipLaunchKernelGGL(k_absmax<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, d_res, ntot(), d.nx, ny, d_partials);
        RN_HIP(hipGetLastError());
        std::vector<Partial> hp(eltBlocks);
        RN_HIP(hipMemcpyAsync(hp.data(), d_partials, eltBlocks * sizeof(Partial), hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        double aX = -1, vX = 0, aP = -1, vP = 0; long long iX = 0x7fffffffffffffffLL, iP = iX;
        for (auto &p : hp) {
            if (p.absXi > aX || (p.absXi == aX && p.idxXi < iX)) { aX = p.absXi; vX = p.valXi; iX = p.idxXi; }
            if (p.absPsi > aP || (p.absPsi == aP && p.idxPsi < iP)) { aP = p.absPsi; vP = p.valPsi; iP = p.idxPsi; }
        }
        if (cutStage > 0 && has_comm() && d_scal) {
            // sharded quasi-Newton loops: the tree-global arg-max from the ranks' local ones.  A max all-reduce of (v, -v) per
            // part gives the largest magnitude and its sign (equal magnitudes of opposite sign on two ranks: the positive one)
            double h[4] = {vX, -vX, vP, -vP};
            RN_HIP(hipMemcpyAsync(d_scal + 32, h, sizeof h, hipMemcpyHostToDevice, stream));
            if (int rc = all_reduce(d_scal + 32, 4, true, "ncclAllReduce(primal infeasibility)", 2 /* ncclMax */)) return rc;
            RN_HIP(hipMemcpyAsync(h, d_scal + 32, sizeof h, hipMemcpyDeviceToHost, stream));
            RN_HIP(hipStreamSynchronize(stream));
            vX = h[0] >= h[1] ? h[0] : -h[1];
            vP = h[2] >= h[3] ? h[2] : -h[3];
        }
        *value = vX > vP ? vX : vP;
        return RN_OK;
    }
    int prox_distances(double *dx, double *ds) override {
        IterState st;
        RN_HIP(hipSetDevice(device));
        RN_HIP(hipMemcpyAsync(&st, d_state, sizeof st, hipMemcpyDeviceToHost, stream));
        RN_HIP(hipStreamSynchronize(stream));
        if (dx) *dx = st.distX;
        if (ds) *ds = st.distS;
        return RN_OK;
    }

    // ---- raw access --------------------------------------------------------------------------------------
    // maps a buffer id to (y-layout base, offset, dim) or to a plain array
    bool ymap(int id, T **base, int *off, int *dim) {
        const int nx = d.nx, nu = d.nu;
        switch (id) {
            case RN_BUF_XI: *base = p_xi; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_PSI: *base = p_xi; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_ACC_XI: *base = p_acc_view; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_ACC_PSI: *base = p_acc_view; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_UPD_XI: *base = p_upd; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_UPD_PSI: *base = p_upd; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_PRIMAL_XI: *base = d_hx; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_PRIMAL_PSI: *base = d_hx; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_DUAL_XI: *base = d_z; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_DUAL_PSI: *base = d_z; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_RES_XI: *base = d_res; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_RES_PSI: *base = d_res; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_XMIN: *base = d_lo; *off = 0; *dim = nx; return true;
            case RN_BUF_XMAX: *base = d_hi; *off = 0; *dim = nx; return true;
            case RN_BUF_XS: *base = d_lo; *off = nx; *dim = nx; return true;
            case RN_BUF_UMIN: *base = d_lo; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_UMAX: *base = d_hi; *off = 2 * nx; *dim = nu; return true;
            default: break;
        }
        if (!d_matS) return false;   // the FBE / NAMA vectors exist once rn_set_algorithm selected one of them
        switch (id) {
            case RN_BUF_PREV_XI: *base = d_prevY; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_PREV_PSI: *base = d_prevY; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_LBFGS_CUR_YVEC_XI: *base = d_g; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_LBFGS_CUR_YVEC_PSI: *base = d_g; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_LBFGS_PREV_YVEC_XI: *base = d_gPrev; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_LBFGS_PREV_YVEC_PSI: *base = d_gPrev; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_LBFGS_DIR_XI: *base = d_dir; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_LBFGS_DIR_PSI: *base = d_dir; *off = 2 * nx; *dim = nu; return true;
            case RN_BUF_PRIMAL_XI_DIR: *base = d_hxDir; *off = 0; *dim = 2 * nx; return true;
            case RN_BUF_PRIMAL_PSI_DIR: *base = d_hxDir; *off = 2 * nx; *dim = nu; return true;
            default: return false;
        }
    }
    T *plain(int id, size_t *n) {
        const size_t N_ = d.nodes;
        switch (id) {
            case RN_BUF_X: *n = N_ * d.nx; return d_x;
            case RN_BUF_U: *n = N_ * d.nu; return d_u;
            case RN_BUF_V: *n = N_ * d.nv; return d_v;
            case RN_BUF_UHAT: *n = N_ * d.nu; return d_uhat;
            case RN_BUF_E: *n = N_ * d.nx; return d_e;
            case RN_BUF_BETA: *n = N_ * d.nv; return d_beta;
            case RN_BUF_ALPHA: *n = N_ * d.nu; return d_alpha;
            case RN_BUF_XDIR: *n = d_xdir ? N_ * d.nx : 0; return d_xdir;
            case RN_BUF_UDIR: *n = d_udir ? N_ * d.nu : 0; return d_udir;
            default: *n = 0; return nullptr;
        }
    }
    // The scaled bounds (Engine.cuh:294-314 getSysXmin ... getSysUmax; Engine.cu:433-463) live interleaved in the dual layout here ([node][ny]: the
    // fused dual update rebuilds them from tables).  A caller that asks for their device pointers gets node-major copies in the reference's layout,
    // made on the first request (5 arrays, (3 nx + 2 nu) reals per node) and refreshed by every later factor step.
    T *d_bnd[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int refresh_bounds_copies() {
        if (!d_bnd[0]) return RN_OK;
        const int ids[5] = {RN_BUF_XMIN, RN_BUF_XMAX, RN_BUF_XS, RN_BUF_UMIN, RN_BUF_UMAX};
        for (int i = 0; i < 5; i++) {
            T *b; int off, dim;
            if (!ymap(ids[i], &b, &off, &dim)) continue;
            hipLaunchKernelGGL(k_pack<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, b, d_bnd[i], ny, off, dim, (long long)d.nodes, 0);
        }
        RN_HIP(hipGetLastError());
        return RN_OK;
    }
    int device_pointer(int id, void **ptr, size_t *n, int *prec) override {
        RN_CHECK(ptr && n && prec, RN_E_ARG, "rn_device_pointer: null output");
        *ptr = nullptr; *n = 0; *prec = sizeof(T) == 8 ? RN_F64 : RN_F32;
        if (id >= RN_BUF_XMIN && id <= RN_BUF_UMAX) {
            RN_CHECK(factored, RN_E_STATE, "rn_device_pointer: the scaled bounds exist after rn_factor_step");
            RN_HIP(hipSetDevice(device));
            if (!d_bnd[0]) {
                const size_t dims[5] = {(size_t)d.nx, (size_t)d.nx, (size_t)d.nx, (size_t)d.nu, (size_t)d.nu};
                for (int i = 0; i < 5; i++) if (int rc = dalloc(&d_bnd[i], (size_t)d.nodes * dims[i])) return rc;
                if (int rc = refresh_bounds_copies()) return rc;
                RN_HIP(hipStreamSynchronize(stream));
            }
            const int i = id - RN_BUF_XMIN;
            *ptr = d_bnd[i]; *n = (size_t)d.nodes * (i < 3 ? d.nx : d.nu);
            return RN_OK;
        }
        size_t cnt = 0;
        if (id == RN_BUF_V) { vEager = true; if (int rc = v_flush()) return rc; }      // from now on v is computed with every sweep that stores the primal iterates
        T *p = plain(id, &cnt);
        RN_CHECK(p != nullptr && cnt > 0, RN_E_ARG, "rn_device_pointer: this buffer is not kept in the reference's layout on the device (or not allocated yet): use rn_get");
        *ptr = p; *n = cnt;
        return RN_OK;
    }
    size_t buffer_size(int id) const override {
        T *b; int off, dim; size_t n;
        Ctx *self = const_cast<Ctx *>(this);
        if (self->ymap(id, &b, &off, &dim)) return (size_t)d.nodes * dim;
        self->plain(id, &n);
        return n;
    }
    int get(int id, double *host, size_t n) override {
        RN_CHECK(host, RN_E_ARG, "rn_get: null host pointer");
        RN_HIP(hipSetDevice(device));
        T *b; int off, dim; size_t cnt;
        if (ymap(id, &b, &off, &dim)) {
            RN_CHECK(n == (size_t)d.nodes * dim, RN_E_ARG, "rn_get: size mismatch");
            hipLaunchKernelGGL(k_pack<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, b, d_tmp, ny, off, dim, (long long)d.nodes, 0);
            RN_HIP(hipGetLastError());
            return download(host, d_tmp, n);
        }
        if (id == RN_BUF_V) { if (int rc = v_flush()) return rc; }
        T *p = plain(id, &cnt);
        RN_CHECK(p != nullptr, RN_E_ARG, "rn_get: unknown buffer id");
        RN_CHECK(n == cnt, RN_E_ARG, "rn_get: size mismatch");
        return download(host, p, n);
    }
    int set(int id, const double *host, size_t n) override {
        RN_CHECK(host, RN_E_ARG, "rn_set: null host pointer");
        RN_HIP(hipSetDevice(device));
        T *b; int off, dim; size_t cnt;
        if (ymap(id, &b, &off, &dim)) {
            RN_CHECK(id < RN_BUF_XMIN || id > RN_BUF_UMAX, RN_E_ARG, "rn_set: the scaled bounds are read-only");
            RN_CHECK(n == (size_t)d.nodes * dim, RN_E_ARG, "rn_set: size mismatch");
            if (int rc = upload(d_tmp, host, n)) return rc;
            hipLaunchKernelGGL(k_pack<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, b, d_tmp, ny, off, dim, (long long)d.nodes, 1);
            RN_HIP(hipGetLastError());
            RN_HIP(hipStreamSynchronize(stream));
            if (id == RN_BUF_ACC_XI || id == RN_BUF_ACC_PSI) {   // the view becomes the sweep input
                if (p_acc_view != p_acc) std::swap(p_acc, p_acc_other);
                acc_ready = true;
            } else if (id == RN_BUF_XI || id == RN_BUF_PSI || id == RN_BUF_UPD_XI || id == RN_BUF_UPD_PSI) {
                acc_ready = false;
            }
            return RN_OK;
        }
        if (id == RN_BUF_V) vPending = false;      // the caller's values replace it
        T *p = plain(id, &cnt);
        RN_CHECK(p != nullptr, RN_E_ARG, "rn_set: unknown buffer id");
        RN_CHECK(n == cnt, RN_E_ARG, "rn_set: size mismatch");
        if (id == RN_BUF_UHAT || id == RN_BUF_E || id == RN_BUF_BETA) { affine_ready = true; aux_dirty = true; }
        return upload(p, host, n);
    }
    // elements [first, first + n) of a buffer in the reference's node-major layout; dual-shaped buffers: whole nodes only
    int get_range(int id, size_t first, size_t n, double *host) override {
        RN_CHECK(host, RN_E_ARG, "rn_get_range: null host pointer");
        RN_HIP(hipSetDevice(device));
        T *b; int off, dim; size_t cnt;
        if (ymap(id, &b, &off, &dim)) {
            RN_CHECK(first % dim == 0 && n % dim == 0 && first + n <= (size_t)d.nodes * dim, RN_E_ARG, "rn_get_range: dual-shaped buffers are addressed in whole nodes");
            hipLaunchKernelGGL(k_pack<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, b + (first / dim) * ny, d_tmp, ny, off, dim, (long long)(n / dim), 0);
            RN_HIP(hipGetLastError());
            return download(host, d_tmp, n);
        }
        if (id == RN_BUF_V) { if (int rc = v_flush()) return rc; }
        T *p = plain(id, &cnt);
        RN_CHECK(p != nullptr, RN_E_ARG, "rn_get_range: unknown buffer id");
        RN_CHECK(first + n <= cnt, RN_E_ARG, "rn_get_range: range outside the buffer");
        return download(host, p + first, n);
    }
    int set_range(int id, size_t first, size_t n, const double *host) override {
        RN_CHECK(host, RN_E_ARG, "rn_set_range: null host pointer");
        RN_HIP(hipSetDevice(device));
        T *b; int off, dim; size_t cnt;
        if (ymap(id, &b, &off, &dim)) {
            RN_CHECK(id < RN_BUF_XMIN || id > RN_BUF_UMAX, RN_E_ARG, "rn_set_range: the scaled bounds are read-only");
            RN_CHECK(first % dim == 0 && n % dim == 0 && first + n <= (size_t)d.nodes * dim, RN_E_ARG, "rn_set_range: dual-shaped buffers are addressed in whole nodes");
            if (int rc = upload(d_tmp, host, n)) return rc;
            hipLaunchKernelGGL(k_pack<T>, dim3(eltBlocks), dim3(ELT_THREADS), 0, stream, b + (first / dim) * ny, d_tmp, ny, off, dim, (long long)(n / dim), 1);
            RN_HIP(hipGetLastError());
            RN_HIP(hipStreamSynchronize(stream));
            if (id == RN_BUF_ACC_XI || id == RN_BUF_ACC_PSI) { if (p_acc_view != p_acc) std::swap(p_acc, p_acc_other); acc_ready = true; }
            else if (id == RN_BUF_XI || id == RN_BUF_PSI || id == RN_BUF_UPD_XI || id == RN_BUF_UPD_PSI) acc_ready = false;
            return RN_OK;
        }
        if (id == RN_BUF_V) { if (int rc = v_flush()) return rc; }      // (the rest of the buffer must be the sweep's)
        T *p = plain(id, &cnt);
        RN_CHECK(p != nullptr, RN_E_ARG, "rn_set_range: unknown buffer id");
        RN_CHECK(first + n <= cnt, RN_E_ARG, "rn_set_range: range outside the buffer");
        if (id == RN_BUF_UHAT || id == RN_BUF_E || id == RN_BUF_BETA) { affine_ready = true; aux_dirty = true; }
        return upload(p + first, host, n);
    }
    int get_operator(int op, int node, double *host, size_t n) override {
        RN_CHECK(factored, RN_E_STATE, "rn_get_operator before rn_factor_step");
        RN_CHECK(host && node >= 0 && node < d.nodes, RN_E_ARG, "rn_get_operator: bad node");
        const int nx = d.nx, nu = d.nu, nv = d.nv;
        const double p = h_prob[node];
        if (op == RN_OP_OMEGA) { RN_CHECK(n == (size_t)nv * nv, RN_E_ARG, "rn_get_operator: size"); for (size_t i = 0; i < n; i++) host[i] = h_Rinv[i] / p; return RN_OK; }
        if (op == RN_OP_G) { RN_CHECK(n == (size_t)nv * nx, RN_E_ARG, "rn_get_operator: size"); for (size_t i = 0; i < n; i++) host[i] = h_Bbt[i]; return RN_OK; }
        if (op == RN_OP_THETA) {
            RN_CHECK(n == (size_t)nv * nx, RN_E_ARG, "rn_get_operator: size");
            std::vector<double> t((size_t)nv * nx);
            h_gemm(false, false, nv, nx, nv, h_Rinv.data(), nv, h_Bbt.data(), nv, t.data());
            for (size_t i = 0; i < n; i++) host[i] = -0.5 * t[i] / p;
            return RN_OK;
        }
        RN_CHECK(op == RN_OP_PHI || op == RN_OP_PSI || op == RN_OP_D || op == RN_OP_F, RN_E_ARG, "rn_get_operator: unknown operator id");
        const bool xiCols = (op == RN_OP_PHI || op == RN_OP_D), top = (op == RN_OP_PHI || op == RN_OP_PSI);
        const int cols = xiCols ? 2 * nx : nu, c0 = xiCols ? 0 : 2 * nx, r0 = top ? 0 : nv;
        RN_CHECK(n == (size_t)nv * cols, RN_E_ARG, "rn_get_operator: size");
        RN_HIP(hipSetDevice(device));
        if (structured) {   // no stored blocks: evaluate the factor-step formula (kernels.hpp, k_expand_operators) on the host
            const int k = h_stageOf[node];
            const double sp = std::sqrt(p);
            const double *dk = h_diag.data() + (size_t)k * (2 * nx + nu);
            for (int c = 0; c < cols; c++) {
                const int j = xiCols ? c % nx : c;
                const double dc = xiCols ? dk[nu + c] : dk[c];
                const double *m = top ? (xiCols ? h_T1.data() : h_T2.data()) + (size_t)j * nv : (xiCols ? h_Bbt.data() : h_Lt.data()) + (size_t)j * nv;
                const double sc = top ? -0.5 * dc / sp : sp * dc;
                for (int r = 0; r < nv; r++) host[r + (size_t)c * nv] = sc * m[r];
            }
            return RN_OK;
        }
        std::vector<double> blk((size_t)ny * LD);
        if (int rc = download(blk.data(), d_A + (size_t)node * strideA, (size_t)ny * LD)) return rc;
        for (int c = 0; c < cols; c++) for (int r = 0; r < nv; r++) host[r + (size_t)c * nv] = blk[(size_t)(c0 + c) * LD + r0 + r];
        return RN_OK;
    }

    // ---- multi-GPU ---------------------------------------------------------------------------------------
    // The communicator of this context.  ncclCommInitRank blocks until every rank of the id has arrived -- for ever if one never does.
    // It therefore runs on a HELPER THREAD and the caller waits for it against the wall clock: after `timeoutSeconds` (< 0:
    // $RAPIDNET_COMM_TIMEOUT_S, default 120) the call returns RN_E_COMM and the context stays usable, without a communicator (the
    // reference would exit(), Configuration.h:38-81; this must neither exit nor hang).  The abandoned helper stays blocked inside RCCL
    // for as long as RCCL waits (it holds only its own copies of the arguments); should it ever come back with a communicator, it
    // destroys it.  (First form, round 5: ncclCommInitRankConfig with blocking = 0, polled with ncclCommGetAsyncError and aborted on
    // time-out -- on the one-GPU box a rank of a two-rank id never came back from that sequence, tests/test_gpu_comm_timeout.py; the
    // blocking call is also the one every earlier multi-process rehearsal used.)
    double commTimeoutS = 120.0;
    struct CommJob { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; int rc = -1; void *comm = nullptr; };
    std::shared_ptr<CommJob> commJob;     // a set-up whose helper thread has not come back
    int comm_init(int rk, int nr, const void *id, double timeoutSeconds = -1.0) override {
        RN_CHECK(nr >= 1 && rk >= 0 && rk < nr, RN_E_ARG, "rn_comm_init: bad rank");
        if (id == nullptr) {               // id == NULL: bookkeeping only (tests emulate the exchange); allowed at any time
            rank = rk; nranks = nr; optHold = 0;
            return RN_OK;
        }
        RN_CHECK(comm == nullptr, RN_E_STATE, "rn_comm_init: the context already has a communicator");
        RN_CHECK(g_nccl.load(), RN_E_COMM, "rn_comm_init: cannot load librccl.so");
        RN_HIP(hipSetDevice(device));
        if (timeoutSeconds < 0) { timeoutSeconds = 120.0; if (const char *e = std::getenv("RAPIDNET_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0) timeoutSeconds = v; } }
        commTimeoutS = timeoutSeconds;
        UniqueId128 u; std::memcpy(u.b, id, 128);
        auto job = std::make_shared<CommJob>();
        commJob = job;                     // kept: rn_destroy tells a helper that is still inside RCCL that nobody waits for it any more
        const int dev = device;
        std::thread([job, u, nr, rk, dev] {
            void *c = nullptr;
            int rc = hipSetDevice(dev) == hipSuccess ? g_nccl.InitRank(&c, nr, u, rk) : -2;
            std::unique_lock<std::mutex> lk(job->m);
            job->rc = rc; job->comm = rc == 0 ? c : nullptr; job->done = true;
            const bool orphan = job->abandoned;
            lk.unlock();
            job->cv.notify_all();
            if (orphan && rc == 0 && c) { if (g_nccl.CommAbort) (void)g_nccl.CommAbort(c); else if (g_nccl.CommDestroy) (void)g_nccl.CommDestroy(c); }
        }).detach();
        {
            std::unique_lock<std::mutex> lk(job->m);
            if (!job->cv.wait_for(lk, std::chrono::duration<double>(timeoutSeconds), [&] { return job->done; })) {
                job->abandoned = true;
                char b[256];
                snprintf(b, sizeof b, "ncclCommInitRank failed: rank %d of %d waited %.1f s for its peers (time-out; the context has no communicator)", rk, nr, timeoutSeconds);
                err = b;
                return RN_E_COMM;
            }
        }
        RN_CHECK(job->rc == 0 && job->comm, RN_E_COMM, std::string("ncclCommInitRank failed: ") + (job->rc == -2 ? "hipSetDevice on the helper thread" : (g_nccl.GetErrorString ? g_nccl.GetErrorString(job->rc) : "?")));
        commJob.reset();
        // The outcome must be the SAME on every rank.  A peer may have given up (its time-out) a moment before this rank's
        // ncclCommInitRank completed: this rank then holds a communicator whose peer is gone, and its first collective would wait for
        // ever.  So before the communicator is published every rank runs one tiny all-reduce with a BOUNDED wait (an event polled against
        // the same time-out, with ncclCommGetAsyncError): a rank whose handshake does not finish aborts its communicator and returns
        // RN_E_COMM like the peer that gave up.
        {
            void *c = job->comm;
            bool ok = true; std::string why;
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ok = false; why = "hipEventCreate"; }
            if (ok && hipMemsetAsync(d_tune, 0, sizeof(double), stream) != hipSuccess) { ok = false; why = "hipMemsetAsync"; }
            if (ok) {
                const int rc = g_nccl.AllReduce(d_tune, d_tune, 1, 8 /* ncclFloat64 */, 0 /* ncclSum */, c, stream);
                if (rc != 0) { ok = false; why = std::string("ncclAllReduce: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "?"); }
            }
            if (ok && hipEventRecord(ev, stream) != hipSuccess) { ok = false; why = "hipEventRecord"; }
            if (ok) {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t q = hipEventQuery(ev);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) { ok = false; why = std::string("hipEventQuery: ") + hipGetErrorString(q); break; }
                    int st = 0;
                    if (g_nccl.GetAsyncError && g_nccl.GetAsyncError(c, &st) == 0 && st != 0 && st != NCCL_IN_PROGRESS) { ok = false; why = std::string("RCCL: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(st) : "?"); break; }
                    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) { ok = false; why = "a peer left between ncclCommInitRank and the first collective (time-out)"; break; }
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
            }
            (void)hipGetLastError();
            if (ev) (void)hipEventDestroy(ev);
            if (!ok) {
                if (g_nccl.CommAbort) (void)g_nccl.CommAbort(c); else if (g_nccl.CommDestroy) (void)g_nccl.CommDestroy(c);
                err = "rn_comm_init: the communicator's handshake failed (" + why + "); the context has no communicator";
                return RN_E_COMM;
            }
        }
        comm = job->comm;
        rank = rk; nranks = nr;
        optHold = 0;                       // every rank starts its batches aligned (the back-off counter decides which path a batch takes)
        // the one-shot transport, set up by the library itself where the cut is already known (rn_create_sharded does it after the cut stage otherwise)
        if (cutStage > 0 && transportReq != RN_EXCHANGE_COLLECTIVE) return exchange_prepare();
        return RN_OK;
    }
    // asynchronous errors of the communicator (a peer that died, a link that went down): asked once per batch, never inside one
    int comm_check() override {
        if (!comm || !g_nccl.GetAsyncError) return RN_OK;
        int st = 0;
        const int q = g_nccl.GetAsyncError(comm, &st);
        RN_CHECK(q == 0 && (st == 0 || st == NCCL_IN_PROGRESS), RN_E_COMM,
                 std::string("RCCL reports an asynchronous error on the communicator: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(q != 0 ? q : st) : "?"));
        return RN_OK;
    }
    int set_allreduce(rn_allreduce_fn fn, void *user) override { arHook = fn; arUser = user; optHold = 0; return RN_OK; }
    int shard_info(int *info) override {
        RN_CHECK(info, RN_E_ARG, "rn_shard_info: null output");
        info[0] = rank; info[1] = nranks; info[2] = cutStage;
        info[3] = cutStage > 0 ? h_stageCum[cutStage] - h_stageCum[cutStage - 1] : 0;
        int cnt = 0;
        if (comm && g_nccl.CommCount) { if (g_nccl.CommCount(comm, &cnt) != 0) cnt = -1; }
        info[4] = cnt; info[5] = d.nodes; info[6] = fullNodes > 0 ? fullNodes : d.nodes;
        return RN_OK;
    }
    int shard_global_nodes(int *out, size_t n) override {
        RN_CHECK(out && n == (size_t)d.nodes, RN_E_ARG, "rn_shard_global_nodes: one entry per local node expected");
        for (int i = 0; i < d.nodes; i++) out[i] = globalNode.empty() ? i : globalNode[i];
        return RN_OK;
    }
    void set_global_nodes(const int *g, int n, int full) override { globalNode.assign(g, g + n); fullNodes = full; }
    int device_ordinal() const override { return device; }
    int join_local_group(LocalGroup *g, int rk) override {
        RN_CHECK(g && rk >= 0 && rk < g->n, RN_E_ARG, "rn_debug_local_group_join: bad rank");
        RN_CHECK(comm == nullptr, RN_E_STATE, "rn_debug_local_group_join: the context already has an RCCL communicator");
        if (!localMember) localMember = new LocalMember();
        localMember->g = g; localMember->rank = rk;
        rank = rk; nranks = g->n;
        return set_allreduce(local_allreduce, localMember);
    }
    int sweep_phase(int phase) override {
        RN_CHECK(factored && affine_ready, RN_E_STATE, "rn_debug_sweep_phase before the factor step / affine terms");
        RN_CHECK(phase == 1 || phase == 2, RN_E_ARG, "rn_debug_sweep_phase: phase must be 1 or 2");
        RN_HIP(hipSetDevice(device));
        p_acc_view = p_acc;
        return launch_sweep(phase);
    }
    int cut_buffer(int write, double *host, size_t n) override {
        RN_CHECK(cutStage > 0, RN_E_STATE, "rn_debug_cut_buffer: no cut stage set");
        const size_t cnt = (size_t)(h_stageCum[cutStage] - h_stageCum[cutStage - 1]) * (d.nv + 2 * d.nx);
        RN_CHECK(host && n == cnt, RN_E_ARG, "rn_debug_cut_buffer: size mismatch");
        RN_HIP(hipSetDevice(device));
        return write ? upload(d_cut, host, n) : download(host, d_cut, n);
    }
    // streaming ceilings of this box: flat read-only and copy, `bytes` per pass (temporary buffers), best of `reps`
    int measure_hbm(size_t bytes, int reps, double *readGBs, double *copyGBs) override {
        RN_CHECK(bytes >= (1u << 20) && reps >= 1 && readGBs && copyGBs, RN_E_ARG, "rn_measure_hbm: bytes >= 1 MiB, reps >= 1");
        RN_HIP(hipSetDevice(device));
        void *a = nullptr, *b = nullptr; double *sink = nullptr;
        RN_HIP(hipMalloc(&a, bytes));
        if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); err = "rn_measure_hbm: out of memory"; return RN_E_HIP; }
        if (hipMalloc((void **)&sink, 65536 * sizeof(double)) != hipSuccess) { (void)hipFree(a); (void)hipFree(b); err = "rn_measure_hbm: out of memory"; return RN_E_HIP; }
        (void)hipMemsetAsync(a, 0, bytes, stream); (void)hipMemsetAsync(b, 0, bytes, stream);
        const long long n = (long long)(bytes / 16);
        const int blocks = 256 * 16;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        double bestR = 0, bestC = 0;
        for (int r = 0; r < reps + 1; r++) {   // first pass = warm-up
            float ms = 0;
            (void)hipEventRecord(e0, stream);
            hipLaunchKernelGGL(k_bw_read<>, dim3(blocks), dim3(256), 0, stream, (const nat_d2 *)a, n, sink);
            (void)hipEventRecord(e1, stream); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms > 0) bestR = std::max(bestR, (double)bytes / (ms * 1e-3) / 1e9);
            (void)hipEventRecord(e0, stream);
            // copy: 4 workgroups per CU measured best for a read + write stream (tools/probes/probe_stream.hip: 6.2 TB/s with
            // 1 024 workgroups, 5.2-5.4 with 4 096 or 8 192 on a 1 GiB vector)
            hipLaunchKernelGGL(k_bw_copy<>, dim3(numCUs * 4), dim3(256), 0, stream, (const nat_d2 *)a, (nat_d2 *)b, n);
            (void)hipEventRecord(e1, stream); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && ms > 0) bestC = std::max(bestC, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
        RN_HIP(hipGetLastError());
        *readGBs = bestR; *copyGBs = bestC;
        return RN_OK;
    }
    int set_operator_mode(int mode) override {
        RN_CHECK(mode == RN_OPS_DENSE || mode == RN_OPS_STRUCTURED || mode == RN_OPS_AUTO, RN_E_ARG, "rn_set_operator_mode: RN_OPS_DENSE, RN_OPS_STRUCTURED or RN_OPS_AUTO");
        const int want = mode == RN_OPS_DENSE ? 0 : 1;
        RN_CHECK(!factored || want == structured, RN_E_STATE, "rn_set_operator_mode must precede rn_factor_step");
        opsMode = mode; structured = want;
        return RN_OK;
    }
    int get_operator_mode(int *requested, int *active) override {
        if (requested) *requested = opsMode;
        if (active) *active = structured ? RN_OPS_STRUCTURED : RN_OPS_DENSE;
        return RN_OK;
    }
    // RN_OPS_AUTO, a caller hands in a block of its own: from here on the context runs on dense per-node blocks -- the factor step once
    // more on the saved inputs, this time expanding every block (Engine.cu:721-745) into d_A
    int materialise_dense() {
        RN_CHECK(factored && structured && opsMode == RN_OPS_AUTO && !h_sys.B.empty(), RN_E_STATE, "materialise_dense: not an RN_OPS_AUTO context after its factor step");
        if (int rc = v_flush()) return rc;       // (a pending v belongs to the structured form's last sweep)
        RN_HIP(hipStreamSynchronize(stream));
        structured = 0; splitFirst = -1;      // (the streaming kernel's launch shape is decided by the factor step)
        rn_system sy{h_sys.B.data(), h_sys.Gd.data(), h_sys.L.data(), h_sys.Lhat.data(), h_sys.W.data(), h_sys.diag.data(), h_sys.xmin.data(), h_sys.xmax.data(),
                     h_sys.xsafe.data(), h_sys.umin.data(), h_sys.umax.data(), h_sys.alpha1.data()};
        if (int rc = factor_step(&sy)) { structured = 1; return rc; }
        if (algorithm == RN_ALG_NAMA) return set_algorithm(algorithm, lbfgsSize);   // (its paired sweep needs buffers of its own in dense mode)
        return RN_OK;
    }
    // One node's block as the caller wants it (the counterpart of rn_get_operator; the reference's Engine hands out the device pointers of
    // these arrays, Engine.cuh:170-230, so a caller may overwrite any block): Phi_i, Psi_i, D_i, Ftil_i -- the blocks solveStep multiplies
    // with at SmpcController.cu:617-638.  Omega / Theta / G are K identical copies of shared matrices in the reference: not per-node here.
    int set_operator(int op, int node, const double *host, size_t n) override {
        RN_CHECK(factored, RN_E_STATE, "rn_set_operator before rn_factor_step");
        RN_CHECK(host && node >= 0 && node < d.nodes, RN_E_ARG, "rn_set_operator: bad node");
        RN_CHECK(op == RN_OP_PHI || op == RN_OP_PSI || op == RN_OP_D || op == RN_OP_F, RN_E_ARG,
                 "rn_set_operator: only the per-node blocks Phi, Psi, D, F can be handed in (Omega, Theta, G are shared matrices scaled by p_i)");
        const int nx = d.nx, nu = d.nu, nv = d.nv;
        const bool xiCols = (op == RN_OP_PHI || op == RN_OP_D), top = (op == RN_OP_PHI || op == RN_OP_PSI);
        const int cols = xiCols ? 2 * nx : nu, c0 = xiCols ? 0 : 2 * nx, r0 = top ? 0 : nv;
        RN_CHECK(n == (size_t)nv * cols, RN_E_ARG, "rn_set_operator: size");
        RN_CHECK(opsMode != RN_OPS_STRUCTURED, RN_E_STATE, "rn_set_operator: the context was created with RN_OPS_STRUCTURED (no per-node blocks); use RN_OPS_AUTO or RN_OPS_DENSE");
        RN_HIP(hipSetDevice(device));
        if (structured) { if (int rc = materialise_dense()) return rc; }
        std::vector<double> blk((size_t)ny * LD);
        if (int rc = download(blk.data(), d_A + (size_t)node * strideA, (size_t)ny * LD)) return rc;
        for (int c = 0; c < cols; c++) for (int r = 0; r < nv; r++) blk[(size_t)(c0 + c) * LD + r0 + r] = host[r + (size_t)c * nv];
        return upload(d_A + (size_t)node * strideA, blk.data(), (size_t)ny * LD);
    }
    int set_cut_moments(const double *E, const double *P, size_t nParents) override {
        RN_CHECK(cutStage > 0, RN_E_STATE, "rn_set_cut_children_moments: set the cut stage first");
        RN_CHECK(E && P && nParents == (size_t)(h_stageCum[cutStage] - h_stageCum[cutStage - 1]), RN_E_ARG, "rn_set_cut_children_moments: one row per cut parent expected");
        RN_HIP(hipSetDevice(device));
        if (!d_momE) { if (int rc = dalloc(&d_momE, nParents * d.nd)) return rc; if (int rc = dalloc(&d_momP, nParents)) return rc; }
        if (int rc = upload(d_momE, E, nParents * d.nd)) return rc;
        if (int rc = upload(d_momP, P, nParents)) return rc;
        moments_set = true;
        return RN_OK;
    }
    int set_cut_stage(int c) override {
        RN_CHECK(c == -1 || (c >= 1 && c < d.N), RN_E_ARG, "rn_set_cut_stage: stage out of range");
        cutStage = c; moments_set = false;
        return RN_OK;
    }

#include "fbe_methods.inc"
};

}  // namespace rn

struct rn_ctx { rn::CtxBase *impl; };
static thread_local std::string g_create_error;   // rn_create failure message of this thread (rn_last_error(NULL))

#define RN_GUARD(ctx) if (!(ctx) || !(ctx)->impl) return RN_E_ARG

extern "C" {

int rn_create(const rn_dims *dims, const rn_tree *tree, int precision, int device, rn_ctx **out) {
    if (!dims || !tree || !out) return RN_E_ARG;
    if (!tree->stages || !tree->nodesPerStage || !tree->nodesPerStageCumul || !tree->ancestor || !tree->probNode) return RN_E_ARG;
    *out = nullptr;
    int rc;
    rn::CtxBase *impl;
    if (precision == RN_F64) { auto *c = new rn::Ctx<double>(); impl = c; rc = c->init(dims, tree, device); }
    else if (precision == RN_F32) { auto *c = new rn::Ctx<float>(); impl = c; rc = c->init(dims, tree, device); }
    else return RN_E_ARG;
    if (rc != RN_OK) { g_create_error = impl->err; delete impl; return rc; }
    *out = new rn_ctx{impl};
    return RN_OK;
}
int rn_destroy(rn_ctx *ctx) { RN_GUARD(ctx); delete ctx->impl; delete ctx; return RN_OK; }
const char *rn_last_error(const rn_ctx *ctx) { return (ctx && ctx->impl) ? ctx->impl->err.c_str() : g_create_error.c_str(); }
int rn_synchronize(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->synchronize(); }
int rn_factor_step(rn_ctx *ctx, const rn_system *sys) { RN_GUARD(ctx); return ctx->impl->factor_step(sys); }
int rn_set_tree_errors(rn_ctx *ctx, const double *ed, const double *ep) { RN_GUARD(ctx); return ctx->impl->set_tree_errors(ed, ep); }
int rn_set_uncertainty(rn_ctx *ctx, int dflag, int pflag, double w) { RN_GUARD(ctx); return ctx->impl->set_uncertainty(dflag, pflag, w); }
int rn_update_state_control(rn_ctx *ctx, const double *x, const double *u, const double *dm) { RN_GUARD(ctx); return ctx->impl->update_state_control(x, u, dm); }
int rn_eliminate_input_disturbance_coupling(rn_ctx *ctx, const double *dh, const double *ah) { RN_GUARD(ctx); return ctx->impl->eliminate(dh, ah); }
int rn_set_parameters(rn_ctx *ctx, double s, double px, double pxs) { RN_GUARD(ctx); return ctx->impl->set_parameters(s, px, pxs); }
int rn_apg_reset(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->apg_reset(); }
int rn_apg_iterate(rn_ctx *ctx, int n, double *h) { RN_GUARD(ctx); return ctx->impl->apg_iterate(n, h); }
int rn_algorithm_apg(rn_ctx *ctx, int n, double *h) { RN_GUARD(ctx); if (int rc = ctx->impl->apg_reset()) return rc; return ctx->impl->apg_iterate(n, h); }
int rn_control_action(rn_ctx *ctx, const double *x, const double *u, const double *dm, const double *dh, const double *ah, int maxIt,
                      int project, double *u0) { RN_GUARD(ctx); return ctx->impl->control_action(x, u, dm, dh, ah, maxIt, project, u0); }
int rn_dual_extrapolation_step(rn_ctx *ctx, double l) { RN_GUARD(ctx); return ctx->impl->extrapolate(l); }
int rn_solve_step(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->solve_step(); }
int rn_proximal_fun_g(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->prox(); }
int rn_compute_fixed_point_residual(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->residual(); }
int rn_dual_update(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->dual_update(); }
int rn_update_primal_infeasibility(rn_ctx *ctx, double *v) { RN_GUARD(ctx); return ctx->impl->primal_infeasibility(v); }
int rn_get_prox_distances(rn_ctx *ctx, double *a, double *b) { RN_GUARD(ctx); return ctx->impl->prox_distances(a, b); }
size_t rn_buffer_size(const rn_ctx *ctx, int id) { return (ctx && ctx->impl) ? ctx->impl->buffer_size(id) : 0; }
int rn_get(rn_ctx *ctx, int id, double *h, size_t n) { RN_GUARD(ctx); return ctx->impl->get(id, h, n); }
int rn_set(rn_ctx *ctx, int id, const double *h, size_t n) { RN_GUARD(ctx); return ctx->impl->set(id, h, n); }
int rn_get_range(rn_ctx *ctx, int id, size_t first, size_t n, double *h) { RN_GUARD(ctx); return ctx->impl->get_range(id, first, n, h); }
int rn_set_range(rn_ctx *ctx, int id, size_t first, size_t n, const double *h) { RN_GUARD(ctx); return ctx->impl->set_range(id, first, n, h); }
int rn_get_operator(rn_ctx *ctx, int op, int node, double *h, size_t n) { RN_GUARD(ctx); return ctx->impl->get_operator(op, node, h, n); }
int rn_device_pointer(rn_ctx *ctx, int id, void **p, size_t *n, int *prec) { RN_GUARD(ctx); return ctx->impl->device_pointer(id, p, n, prec); }
int rn_profile_enable(rn_ctx *ctx, int on) { RN_GUARD(ctx); return ctx->impl->profile_enable(on); }
int rn_profile_reset(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->profile_reset(); }
int rn_profile_read(rn_ctx *ctx, double ms[4], long n[4]) { RN_GUARD(ctx); return ctx->impl->profile_read(ms, n); }
int rn_algorithmic_bytes(const rn_ctx *ctx, double *b, double *dl) { RN_GUARD(ctx); return ctx->impl->algorithmic_bytes(b, dl); }
void *rn_stream(rn_ctx *ctx) { return (ctx && ctx->impl) ? ctx->impl->stream_handle() : nullptr; }
int rn_comm_unique_id(void *id128) {
    if (!id128) return RN_E_ARG;
    if (!rn::g_nccl.load()) return RN_E_COMM;
    return rn::g_nccl.GetUniqueId(id128) == 0 ? RN_OK : RN_E_COMM;
}
int rn_comm_init(rn_ctx *ctx, int rank, int nranks, const void *id128) { RN_GUARD(ctx); return ctx->impl->comm_init(rank, nranks, id128); }
int rn_comm_init_timeout(rn_ctx *ctx, int rank, int nranks, const void *id128, double timeoutSeconds) { RN_GUARD(ctx); return ctx->impl->comm_init(rank, nranks, id128, timeoutSeconds); }
int rn_comm_check(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->comm_check(); }
int rn_comm_library(char *buf, size_t n) {
    if (!buf || n == 0) return RN_E_ARG;
    if (!rn::g_nccl.load()) { buf[0] = 0; return RN_E_COMM; }
    snprintf(buf, n, "%s", rn::g_nccl.path.c_str());
    return RN_OK;
}
int rn_set_cut_stage(rn_ctx *ctx, int stage) { RN_GUARD(ctx); return ctx->impl->set_cut_stage(stage); }
int rn_get_history_parts(rn_ctx *ctx, int first, int n, double *out) { RN_GUARD(ctx); return ctx->impl->hist_parts(first, n, out); }
int rn_get_kernel_info(rn_ctx *ctx, int info[8]) { RN_GUARD(ctx); return ctx->impl->kernel_info(info); }
int rn_get_counters(rn_ctx *ctx, long out[4]) { RN_GUARD(ctx); return ctx->impl->counters(out); }
int rn_set_cut_children_moments(rn_ctx *ctx, const double *E, const double *P, size_t n) { RN_GUARD(ctx); return ctx->impl->set_cut_moments(E, P, n); }
int rn_set_operator_mode(rn_ctx *ctx, int mode) { RN_GUARD(ctx); return ctx->impl->set_operator_mode(mode); }
int rn_get_operator_mode(rn_ctx *ctx, int *requested, int *active) { RN_GUARD(ctx); return ctx->impl->get_operator_mode(requested, active); }
int rn_set_operator(rn_ctx *ctx, int op, int node, const double *h, size_t n) { RN_GUARD(ctx); return ctx->impl->set_operator(op, node, h, n); }
int rn_set_warm_start(rn_ctx *ctx, int on) { RN_GUARD(ctx); return ctx->impl->set_warm_start(on); }
int rn_set_exchange_mode(rn_ctx *ctx, int mode) { RN_GUARD(ctx); return ctx->impl->set_exchange_mode(mode); }
int rn_measure_hbm(rn_ctx *ctx, size_t bytes, int reps, double *r, double *c) { RN_GUARD(ctx); return ctx->impl->measure_hbm(bytes, reps, r, c); }
int rn_set_algorithm(rn_ctx *ctx, int alg, int m) { RN_GUARD(ctx); return ctx->impl->set_algorithm(alg, m); }
int rn_fbe_reset(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->fbe_reset(); }
int rn_compute_hessian_oracle(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->hessian_oracle(); }
int rn_compute_gradient_fbe(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->gradient_fbe(); }
int rn_update_fixed_point_residual_nama(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->nama_residual(); }
int rn_compute_lbfgs_direction(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->lbfgs_direction(); }
int rn_update_lbfgs_buffer(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->lbfgs_update_buffer(); }
int rn_two_loop_recursion_lbfgs(rn_ctx *ctx) { RN_GUARD(ctx); return ctx->impl->lbfgs_two_loop(); }
int rn_compute_value_fbe(rn_ctx *ctx, double *v) { RN_GUARD(ctx); return ctx->impl->value_fbe(v); }
int rn_line_search_lbfgs_update(rn_ctx *ctx, double vy, double *tau) { RN_GUARD(ctx); return ctx->impl->line_search_fbe(vy, tau); }
int rn_line_search_ame_lbfgs_update(rn_ctx *ctx, double vy, double *tau) { RN_GUARD(ctx); return ctx->impl->line_search_ame(vy, tau); }
int rn_algorithm_fbe_nama(rn_ctx *ctx, int n, double *h, double *v, double *t) { RN_GUARD(ctx); return ctx->impl->algorithm_fbe_nama(n, h, v, t); }
int rn_lbfgs_state(rn_ctx *ctx, int set, int *col, int *mem, double *H, double *rho) { RN_GUARD(ctx); return ctx->impl->lbfgs_state(set, col, mem, H, rho); }
int rn_lbfgs_column(rn_ctx *ctx, int set, int which, int col, double *h, size_t n) { RN_GUARD(ctx); return ctx->impl->lbfgs_column(set, which, col, h, n); }
int rn_debug_sweep_phase(rn_ctx *ctx, int phase) { RN_GUARD(ctx); return ctx->impl->sweep_phase(phase); }
int rn_debug_cut_buffer(rn_ctx *ctx, int write, double *host, size_t n) { RN_GUARD(ctx); return ctx->impl->cut_buffer(write, host, n); }

// ---- multi-GPU through the boundary ------------------------------------------------------------------------------------
int rn_default_cut_stage(const rn_dims *dims, const rn_tree *tree) { return rn::default_cut_stage(dims, tree); }
int rn_partition_create(const rn_dims *dims, const rn_tree *tree, const double *errD, const double *errP, int rank, int nranks, int cutStage,
                        rn_partition *out) {
    return rn::build_partition(dims, tree, errD, errP, rank, nranks, cutStage, out, g_create_error);
}
void rn_partition_destroy(rn_partition *p) {
    if (!p || !p->owner) return;
    delete static_cast<rn::PartitionData *>(p->owner);
    std::memset(p, 0, sizeof *p);
}
int rn_create_sharded(const rn_dims *dims, const rn_tree *tree, const double *errD, const double *errP, int precision, int device, int rank,
                      int nranks, int cutStage, const void *id128, rn_ctx **out) {
    if (!dims || !tree || !out) return RN_E_ARG;
    *out = nullptr;
    if (nranks == 1) {   // a plain unsharded context
        if (rank != 0) { g_create_error = "rn_create_sharded: bad rank"; return RN_E_ARG; }
        if (int rc = rn_create(dims, tree, precision, device, out)) return rc;
        if (errD && errP) {
            const int rc = (*out)->impl->set_tree_errors(errD, errP);
            if (rc != RN_OK) { g_create_error = (*out)->impl->err; rn_destroy(*out); *out = nullptr; return rc; }
        }
        return RN_OK;
    }
    rn_partition part;
    if (int rc = rn::build_partition(dims, tree, errD, errP, rank, nranks, cutStage, &part, g_create_error)) return rc;
    rn_ctx *ctx = nullptr;
    int rc = rn_create(&part.dims, &part.tree, precision, device, &ctx);
    if (rc == RN_OK) {
        rn::CtxBase *c = ctx->impl;
        c->set_global_nodes(part.globalNode, part.dims.nodes, dims->nodes);
        if (part.errorDemandNode && part.errorPriceNode) rc = c->set_tree_errors(part.errorDemandNode, part.errorPriceNode);
        if (rc == RN_OK) rc = c->comm_init(rank, nranks, id128);
        if (rc == RN_OK) rc = c->set_cut_stage(part.cutStage);
        if (rc == RN_OK) {
            std::vector<double> zeros;
            const double *E = part.momE;
            if (!E) { zeros.assign((size_t)part.nCutParents * dims->nd, 0.0); E = zeros.data(); }
            rc = c->set_cut_moments(E, part.momP, (size_t)part.nCutParents);
        }
        if (rc == RN_OK && id128) rc = c->exchange_prepare_api();    // AUTO / one-shot: this rank's inbox, the peers' mapped (collective; falls back by agreement)
        if (rc != RN_OK) { g_create_error = c->err; rn_destroy(ctx); ctx = nullptr; }
    }
    rn_partition_destroy(&part);
    *out = ctx;
    return rc;
}
int rn_shard_info(rn_ctx *ctx, int info[7]) { RN_GUARD(ctx); return ctx->impl->shard_info(info); }
int rn_shard_global_nodes(rn_ctx *ctx, int *g, size_t n) { RN_GUARD(ctx); return ctx->impl->shard_global_nodes(g, n); }
int rn_debug_set_allreduce(rn_ctx *ctx, rn_allreduce_fn fn, void *user) { RN_GUARD(ctx); return ctx->impl->set_allreduce(fn, user); }
int rn_debug_local_group_create(int nranks, void **group) {
    if (!group || nranks < 1 || nranks > rn::LOCAL_GROUP_MAX) return RN_E_ARG;
    rn::LocalGroup *g = new rn::LocalGroup();
    g->n = nranks; g->bufs.assign(nranks, nullptr); g->counts.assign(nranks, 0); g->f64.assign(nranks, 0);
    if (const char *e = std::getenv("RAPIDNET_GROUP_TIMEOUT_S")) { const int t = std::atoi(e); if (t > 0) g->timeoutSeconds = t; }
    *group = g;
    return RN_OK;
}
int rn_debug_local_group_join(rn_ctx *ctx, void *group, int rank) { RN_GUARD(ctx); return ctx->impl->join_local_group(static_cast<rn::LocalGroup *>(group), rank); }
int rn_debug_local_group_destroy(void *group) { if (!group) return RN_E_ARG; delete static_cast<rn::LocalGroup *>(group); return RN_OK; }
int rn_debug_inject_allocation(rn_ctx *ctx, size_t bytes) { RN_GUARD(ctx); return ctx->impl->inject_allocation(bytes); }
int rn_debug_guard_poke(rn_ctx *ctx, int nbytes) { RN_GUARD(ctx); return ctx->impl->guard_poke(nbytes); }
int rn_debug_set_knob(rn_ctx *ctx, int knob, int value) { RN_GUARD(ctx); return ctx->impl->set_knob(knob, value); }
int rn_debug_peer_seq(rn_ctx *ctx, unsigned int seq) { RN_GUARD(ctx); return ctx->impl->debug_peer_seq(seq); }
int rn_guard_report(long out[2]) { if (!out) return RN_E_ARG; out[0] = rn::g_guardContexts.load(); out[1] = rn::g_guardBadBytes.load(); return RN_OK; }
int rn_peer_inbox_create(rn_ctx *ctx, void *ipcHandle64) { RN_GUARD(ctx); return ctx->impl->peer_inbox_create(ipcHandle64); }
int rn_peer_inbox_connect(rn_ctx *ctx, const void *ipcHandles, int nranks) { RN_GUARD(ctx); return ctx->impl->peer_inbox_connect(ipcHandles, nranks); }
int rn_debug_peer_inbox_connect_local(rn_ctx **ctxs, int nranks) {
    if (!ctxs || nranks < 2 || nranks > rn::PEER_MAX) return RN_E_ARG;
    std::vector<rn::CtxBase *> impls(nranks);
    for (int r = 0; r < nranks; r++) { if (!ctxs[r] || !ctxs[r]->impl) return RN_E_ARG; impls[r] = ctxs[r]->impl; }
    for (int r = 0; r < nranks; r++) if (int rc = impls[r]->peer_inbox_connect_local(impls.data(), nranks)) return rc;
    return RN_OK;
}
int rn_set_exchange_transport(rn_ctx *ctx, int transport) { RN_GUARD(ctx); return ctx->impl->set_exchange_transport(transport); }
int rn_exchange_autotune(rn_ctx *ctx, int iterations, double info[8]) { RN_GUARD(ctx); return ctx->impl->exchange_autotune_api(iterations, info); }
int rn_set_fused_walk_dual(rn_ctx *ctx, int on) { RN_GUARD(ctx); return ctx->impl->set_fused_walk_dual(on); }
int rn_fbe_counters(rn_ctx *ctx, long out[4]) { RN_GUARD(ctx); return ctx->impl->fbe_counters(out); }
int rn_guard_check(rn_ctx *ctx, long *badBytes) { RN_GUARD(ctx); return ctx->impl->guard_check(badBytes); }
int rn_device_memory_info(rn_ctx *ctx, size_t info[4]) { RN_GUARD(ctx); return ctx->impl->memory_info(info); }
int rn_reserve_iterations(rn_ctx *ctx, int maxIterations) { RN_GUARD(ctx); return ctx->impl->reserve_iterations(maxIterations); }
int rn_profile_read_collective(rn_ctx *ctx, double *ms, long *launches) { RN_GUARD(ctx); return ctx->impl->profile_read_collective(ms, launches); }

}  // extern "C"

#ifdef RN_KTIMING
// debug builds only (not part of include/rapidnet.h): phase stamps of the instrumented kernels, 8 workgroups x 16 slots
extern "C" int rn_debug_ktiming(unsigned long long *out128) {
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(rn::g_ktiming), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : 1;
}
#endif
