/*
 * rapidnet.h -- C ABI of the MI355X-native APG solve path (librapidnet_hip.so).
 *
 * The reference (GPUEngineering/RapidNet) has no FFI layer: its seam is the C++ class surface
 * Engine / SmpcController and raw device pointers.  This header is the boundary a
 * maintainer binds instead: every entry point names the reference member function it replaces
 * (file:line relative to /root/reference/src).  The C++ classes in rapidnet_amd/csrc/host/ (same names and
 * signatures as the reference's) are a thin layer over exactly these functions.
 *
 * Conventions
 *   - plain pointers and sizes only; every HOST array is `double` whatever the device precision;
 *   - matrices are column-major, per-node vectors node-major `[node][dim]`, tree indices as in the
 *     reference's JSON (1-based `ancestor`, root = 0; `nodesPerStage` N+1 entries, `...Cumul` N+2);
 *   - every function returns 0 on success or a negative RN_E_* code and never calls exit();
 *     rn_last_error() returns a message for the last failure on that context;
 *   - one host thread per context, one HIP stream per context; contexts are independent.
 */
#ifndef RAPIDNET_H_
#define RAPIDNET_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rn_ctx rn_ctx;

enum { RN_F32 = 0, RN_F64 = 1 };

enum {
    RN_OK = 0,
    RN_E_ARG = -1,      /* bad argument / dimension mismatch */
    RN_E_HIP = -2,      /* HIP runtime error (message in rn_last_error) */
    RN_E_STATE = -3,    /* call order violated (e.g. iterate before factor step) */
    RN_E_SINGULAR = -4, /* L'WL singular: Engine::inverseBatchMat, Engine.cu:1334-1353 */
    RN_E_COMM = -5      /* RCCL error */
};

/* problem dimensions: DwnNetwork.cuh:67-87, SmpcConfiguration.cuh:60-75, ScenarioTree.cuh:64-84 */
typedef struct {
    int nx, nu, nv, nd; /* tanks, controls, reduced inputs (nu - ne), demands */
    int N, K;           /* prediction horizon, scenarios */
    int nodes, nNonLeafNodes;
} rn_dims;

/* scenario tree exactly as ScenarioTree's getters return it (ScenarioTree.cuh:92-154) */
typedef struct {
    const int *stages;             /* [nodes]   0-based stage of each node              */
    const int *nodesPerStage;      /* [N+1]                                             */
    const int *nodesPerStageCumul; /* [N+2]                                             */
    const int *ancestor;           /* [nodes]   1-based parent, root = 0                */
    const int *nChildren;          /* [nNonLeafNodes]                                   */
    const int *nChildrenCumul;     /* [nodes]                                           */
    const double *probNode;        /* [nodes]                                           */
} rn_tree;

/* inputs of the factor step: what Engine::initialiseSystemDevice (Engine.cu:382-463) reads from DwnNetwork
 * and SmpcConfiguration */
typedef struct {
    const double *matB;          /* nx x nu */
    const double *matGd;         /* nx x nd */
    const double *matL;          /* nu x nv  null-space basis of E (config "matL")      */
    const double *matLhat;       /* nu x nd  -pinv(E) Ed          (config "matLhat")   */
    const double *costW;         /* nu x nu */
    const double *matDiagPrecnd; /* [N][nu | nx | nx] dual preconditioner diagonal      */
    const double *vecXmin, *vecXmax, *vecXsafe; /* nx */
    const double *vecUmin, *vecUmax;            /* nu */
    const double *costAlpha1;                   /* nu */
} rn_system;

/* buffer ids for rn_get / rn_set: the protected device vectors of SmpcController (SmpcController.cuh:336-462)
 * and the affine-term arrays of Engine (Engine.cuh:426-648), in the reference's node-major layout */
enum {
    RN_BUF_X = 0,      /* devVecX            nodes*nx   */
    RN_BUF_U,          /* devVecU            nodes*nu   */
    RN_BUF_V,          /* devVecV            nodes*nv   */
    RN_BUF_XI,         /* devVecXi           nodes*2nx  (y)        */
    RN_BUF_PSI,        /* devVecPsi          nodes*nu              */
    RN_BUF_ACC_XI,     /* devVecAcceleratedXi   (w)                */
    RN_BUF_ACC_PSI,    /* devVecAcceleratedPsi                     */
    RN_BUF_UPD_XI,     /* devVecUpdateXi        (y+)               */
    RN_BUF_UPD_PSI,    /* devVecUpdatePsi                          */
    RN_BUF_PRIMAL_XI,  /* devVecPrimalXi        (Hx)               */
    RN_BUF_PRIMAL_PSI, /* devVecPrimalPsi                          */
    RN_BUF_DUAL_XI,    /* devVecDualXi          (z)                */
    RN_BUF_DUAL_PSI,   /* devVecDualPsi                            */
    RN_BUF_RES_XI,     /* devVecFixedPointResidualXi               */
    RN_BUF_RES_PSI,    /* devVecFixedPointResidualPsi              */
    RN_BUF_UHAT,       /* Engine devVecUhat  nodes*nu   */
    RN_BUF_E,          /* Engine devVecE     nodes*nx   */
    RN_BUF_BETA,       /* Engine devVecBeta  nodes*nv   */
    RN_BUF_ALPHA,      /* Engine devVecAlpha nodes*nu (getPriceAlpha) */
    RN_BUF_XMIN, RN_BUF_XMAX, RN_BUF_XS, /* scaled bounds, nodes*nx */
    RN_BUF_UMIN, RN_BUF_UMAX,            /* nodes*nu */
    /* global FBE / NAMA (valid after rn_set_algorithm selected one of them) */
    RN_BUF_PREV_XI, RN_BUF_PREV_PSI,                       /* devVecPrevXi / devVecPrevPsi                              */
    RN_BUF_LBFGS_CUR_YVEC_XI, RN_BUF_LBFGS_CUR_YVEC_PSI,   /* ptrLbfgsCurrentYvec*: devVecGradientFbe* (FBE) or
                                                              devVecCurrentFixedPointResidual* (NAMA), SmpcController.cu:513-524 */
    RN_BUF_LBFGS_PREV_YVEC_XI, RN_BUF_LBFGS_PREV_YVEC_PSI, /* ptrLbfgsPreviousYvec*                                     */
    RN_BUF_LBFGS_DIR_XI, RN_BUF_LBFGS_DIR_PSI,             /* devVecLbfgsDirXi / Psi                                    */
    RN_BUF_PRIMAL_XI_DIR, RN_BUF_PRIMAL_PSI_DIR,           /* devVecPrimalXiDir / PsiDir                                */
    RN_BUF_XDIR, RN_BUF_UDIR,                              /* devVecXdir nodes*nx, devVecUdir nodes*nu                  */
    RN_BUF_COUNT
};

/* per-node operator blocks of Engine::factorStep for rn_get_operator (Engine.cuh getMatPhi() ...) */
enum { RN_OP_PHI = 0 /* nv x 2nx */, RN_OP_PSI /* nv x nu */, RN_OP_D /* nv x 2nx */, RN_OP_F /* nv x nu */,
       RN_OP_OMEGA /* nv x nv */, RN_OP_THETA /* nv x nx */, RN_OP_G /* nv x nx */ };

/* ---- lifetime ------------------------------------------------------------------------------------ */
/* Engine::Engine + allocate*Device (Engine.cu:126-380) and SmpcController::allocateSmpcController /
 * allocateApgAlgorithm (SmpcController.cu:117-232).  `device` is the HIP device ordinal. */
int rn_create(const rn_dims *dims, const rn_tree *tree, int precision, int device, rn_ctx **out);
int rn_destroy(rn_ctx *ctx);
const char *rn_last_error(const rn_ctx *ctx);
int rn_synchronize(rn_ctx *ctx);

/* Operator storage, to be chosen BEFORE rn_factor_step.
 *   RN_OPS_DENSE: the reference's storage model -- dense per-node blocks Phi_i, Psi_i, D_i, Ftil_i
 *     (Engine.cu:166-189), streamed from HBM once per iteration (4.09 GB for the 493-scenario Barcelona tree).
 *   RN_OPS_STRUCTURED: the blocks are never materialised; the factor step's own formulas (block = shared matrix x
 *     stage diagonal x power of p_i, Engine.cu:721-745) are applied as shared-operator GEMMs.  Same iterates (the parity
 *     suite runs both), 6-7 x the iteration rate on the 493-scenario tree.
 *   RN_OPS_AUTO (the default of a new context): structured for as long as every block is the factor step's own -- which is
 *     always, unless the caller hands in a block through rn_set_operator; the first such call materialises the dense blocks
 *     (one more factor step on the inputs of the last one) and the context runs dense from then on.
 * rn_get_operator_mode: *requested = what was asked for, *active = RN_OPS_DENSE or RN_OPS_STRUCTURED, what the context runs. */
enum { RN_OPS_DENSE = 0, RN_OPS_STRUCTURED = 1, RN_OPS_AUTO = 2 };
int rn_set_operator_mode(rn_ctx *ctx, int mode);
int rn_get_operator_mode(rn_ctx *ctx, int *requested, int *active);

/* ---- Engine -------------------------------------------------------------------------------------- */
/* Engine::factorStep (Engine.cu:671-774) incl. initialiseSystemDevice / preconditioning kernels. */
int rn_factor_step(rn_ctx *ctx, const rn_system *sys);
/* tree errors uploaded once (the reference re-uploads them every control step, Engine.cu:1205,1228) */
int rn_set_tree_errors(rn_ctx *ctx, const double *errorDemandNode /* nodes*nd */, const double *errorPriceNode /* nodes*nu */);
/* Engine::setDemandUncertaintyFlag / setPriceUncertaintyFlag / SmpcConfiguration::getWeightEconomical */
int rn_set_uncertainty(rn_ctx *ctx, int demandUncertainty, int priceUncertainty, double weightEconomical);
/* Engine::updateStateControl (Engine.cu:1300-1316) */
int rn_update_state_control(rn_ctx *ctx, const double *currentX, const double *prevU, const double *prevDemand);
/* Engine::eliminateInputDistubanceCoupling (Engine.cu:1147-1298): nominalDemand [N][nd], nominalPrices [N][nu] */
int rn_eliminate_input_disturbance_coupling(rn_ctx *ctx, const double *nominalDemand, const double *nominalPrices);

/* ---- SmpcController: algorithm parameters -------------------------------------------------------- */
/* stepSize, penaltyStateX, penaltySafetyX (SmpcConfiguration.cuh:120-135) */
int rn_set_parameters(rn_ctx *ctx, double stepSize, double penaltyStateX, double penaltySafetyX);

/* ---- SmpcController: the APG loop ---------------------------------------------------------------- */
/* SmpcController::initialiseAlgorithm (SmpcController.cu:420-450): zero the duals, theta = {1,1} */
int rn_apg_reset(rn_ctx *ctx);
/* `n` iterations of the loop body of SmpcController::algorithmApg (SmpcController.cu:1512-1522), device
 * resident, no host synchronisation.  primalInfs (may be NULL) receives vecPrimalInfs[] for these n
 * iterations (this copy synchronises). */
int rn_apg_iterate(rn_ctx *ctx, int n, double *primalInfs);
/* SmpcController::algorithmApg (SmpcController.cu:1500-1525) = rn_apg_reset + rn_apg_iterate(maxIterations) */
int rn_algorithm_apg(rn_ctx *ctx, int maxIterations, double *primalInfs);
/* SmpcController::controlAction(real_t*) (SmpcController.cu:1607-1625): update state, eliminate, APG, copy u of the root node
 * (nu reals) to the host.  The reference's leak check around these calls (cudaMemGetInfo before and after, :1612-1623) is the
 * caller's, with rn_device_memory_info below -- the host class SmpcController::controlAction does exactly that. */
int rn_control_action(rn_ctx *ctx, const double *currentX, const double *prevU, const double *prevDemand,
                      const double *nominalDemand, const double *nominalPrices, int maxIterations,
                      int projectOnBounds, double *u0);

/* cudaMemGetInfo of the reference's leak check (SmpcController.cu:1612, :1619, :1641, :1657): info = {free bytes of the context's
 * device, total bytes, bytes THIS context holds, live contexts of this process}.  The device-wide figure moves with every other
 * context and process on the device; the context's own does not. */
int rn_device_memory_info(rn_ctx *ctx, size_t info[4]);
/* SmpcController::allocateApgAlgorithm sizes the per-iteration storage by maxIterations once (SmpcController.cu:124-151): reserves
 * the iteration tables and the checkpoint buffers for batches of up to maxIterations iterations, so that no later
 * rn_apg_iterate / rn_control_action allocates device memory.  (1 022 iterations are reserved by rn_create.) */
int rn_reserve_iterations(rn_ctx *ctx, int maxIterations);

/* Extension (the reference always cold-starts, SmpcController.cu:1509): when on,
 * rn_control_action keeps the duals of the previous control step and only restarts the momentum. */
int rn_set_warm_start(rn_ctx *ctx, int on);

/* step-wise entry points mirroring the protected methods the reference's known-answer tests call */
int rn_dual_extrapolation_step(rn_ctx *ctx, double lambda); /* SmpcController.cu:535-557  */
int rn_solve_step(rn_ctx *ctx);                             /* SmpcController.cu:563-755  */
int rn_proximal_fun_g(rn_ctx *ctx);                         /* SmpcController.cu:759-835  */
int rn_compute_fixed_point_residual(rn_ctx *ctx);           /* SmpcController.cu:839-850  */
int rn_dual_update(rn_ctx *ctx);                            /* SmpcController.cu:854-864  */
int rn_update_primal_infeasibility(rn_ctx *ctx, double *value); /* SmpcController.cu:1480-1496 */
/* tree-global distances computed by the last prox (SmpcController.cu:792,810) */
int rn_get_prox_distances(rn_ctx *ctx, double *distanceXcst, double *distanceXs);

/* ---- SmpcController: the global-FBE and NAMA loops --------------------- */
/* Engine's algorithm flags (Engine.cu:151-163, "algorithmName" of the controller configuration).  Selecting FBE or
 * NAMA allocates their vectors and the L-BFGS buffers (SmpcController::allocateGlobalFbeAlgorithm /
 * allocateNamaAlgorithm / allocateLbfgsBuffer, SmpcController.cu:234-330) and runs rn_fbe_reset.  On a sharded context
 * (rn_create_sharded) every dot product, value term and prox distance of the loops is all-reduced over the ranks (the
 * replicated crown counted once), so all ranks take the same skip / line-search decisions. */
enum { RN_ALG_APG = 0, RN_ALG_GLOBAL_FBE = 1, RN_ALG_NAMA = 2 };
int rn_set_algorithm(rn_ctx *ctx, int algorithm, int lbfgsBufferSize);
/* SmpcController::initialiseAlgorithm (FBE / NAMA part, :436-449) + initaliseLbfgBuffer (:453-468) */
int rn_fbe_reset(rn_ctx *ctx);
/* SmpcController::algorithmGlobalFbe (:1529-1555) or algorithmNama (:1559-1586), whichever is selected.
 * primalInfs[maxIterations] = vecPrimalInfs, valueFbe / tau [maxIterations-1] = vecValueFbe / vecTau (may be NULL). */
int rn_algorithm_fbe_nama(rn_ctx *ctx, int maxIterations, double *primalInfs, double *valueFbe, double *tau);
/* step-wise entry points (the reference's known-answer tests call these protected methods);
 * rn_dual_update takes the FBE / NAMA branch (SmpcController.cu:866-880) once one of them is selected */
int rn_compute_hessian_oracle(rn_ctx *ctx);                 /* computeHessianOracalGlobalFbe          :884-1055  */
int rn_compute_gradient_fbe(rn_ctx *ctx);                   /* computeGradientFbe                     :1077-1097 */
int rn_update_fixed_point_residual_nama(rn_ctx *ctx);       /* updateFixedPointResidualNamaAlgorithm  :1060-1072 */
int rn_compute_lbfgs_direction(rn_ctx *ctx);                /* computeLbfgsDirection                  :1103-1237 */
int rn_update_lbfgs_buffer(rn_ctx *ctx);                    /*   its first half:  updateLbfgsBuffer   :1103-1169 */
int rn_two_loop_recursion_lbfgs(rn_ctx *ctx);               /*   its second half: twoLoopRecursionLbfgs :1175-1229 */
int rn_compute_value_fbe(rn_ctx *ctx, double *value);       /* computeValueFbe                        :1416-1476 */
int rn_line_search_lbfgs_update(rn_ctx *ctx, double valueFbeY, double *tau);     /* computeLineSearchLbfgsUpdate    :1242-1305 */
int rn_line_search_ame_lbfgs_update(rn_ctx *ctx, double valueAmeY, double *tau); /* computeLineSearchAmeLbfgsUpdate :1311-1414 */
/* How the line searches ran (both loops evaluate the trials tau = 1, 1/2, 1/4 ... of SmpcController.cu:1272-1300 in batches of six
 * candidates -- two passes and one read-back per batch instead of eight launches and two read-backs per trial; identical tau):
 * out = {line searches, candidate batches evaluated, searches that ran trial by trial (a candidate's prox tripped the
 * soft-constraint branch, or the batch kernel's LDS tile does not fit), NAMA iterations whose two Hessian oracles (:1331, :1341-1345)
 * shared one pass over the operator blocks} */
int rn_fbe_counters(rn_ctx *ctx, long out[4]);
/* lbfgsBufferCol / lbfgsBufferMemory / lbfgsBufferHessian and lbfgsBufferRho (lbfgsBufferSize + 1 entries, the
 * reference addresses entries 1..lbfgsBufferSize; rho may be NULL): get (set = 0) or set (set = 1) */
int rn_lbfgs_state(rn_ctx *ctx, int set, int *col, int *mem, double *H, double *rho);
/* one column (0..lbfgsBufferSize) of devLbfgsBufferMatS (which = 0) / MatY (which = 1), nodes*(2nx+nu) reals in the
 * reference's order (all xi, then all psi) */
int rn_lbfgs_column(rn_ctx *ctx, int set, int which, int col, double *host, size_t n);

/* ---- raw access (tests, closed loop) ------------------------------------------------------------- */
size_t rn_buffer_size(const rn_ctx *ctx, int buffer_id); /* element count, 0 for a bad id */
int rn_get(rn_ctx *ctx, int buffer_id, double *host, size_t n);
int rn_set(rn_ctx *ctx, int buffer_id, const double *host, size_t n);
/* elements [first, first + n) only (e.g. the nx reals of node 0: first = 0, n = nx); dual-shaped buffers (RN_BUF_XI ...
 * RN_BUF_UMAX and the FBE vectors) are addressed in whole nodes (first and n multiples of the per-node dimension) */
int rn_get_range(rn_ctx *ctx, int buffer_id, size_t first, size_t n, double *host);
int rn_set_range(rn_ctx *ctx, int buffer_id, size_t first, size_t n, const double *host);
/* one node's operator block, reference layout (col-major, ld = nv) */
int rn_get_operator(rn_ctx *ctx, int op_id, int node, double *host, size_t n);
/* ... and its counterpart: a block handed in by the caller (the reference's Engine returns the device pointers of these arrays,
 * Engine.cuh:170-230 getMatPhi() ... getPtrMatF(), so its callers may overwrite any block).  RN_OP_PHI, _PSI, _D, _F of one node -- the
 * blocks solveStep multiplies with (SmpcController.cu:617-638); the next sweep uses them.  After rn_factor_step; a later rn_factor_step
 * recomputes every block.  RN_OPS_AUTO contexts switch to dense storage on the first call; RN_OPS_STRUCTURED contexts (no per-node
 * blocks by request): RN_E_STATE.  Omega_i, Theta_i, G_i are shared matrices scaled by p_i here (K identical copies in the reference,
 * Engine.cu:306-308): RN_E_ARG. */
int rn_set_operator(rn_ctx *ctx, int op_id, int node, const double *host, size_t n);
/* The raw device pointer of a buffer that the library keeps in the reference's own layout -- the counterpart of the reference's raw
 * getters and protected device vectors for those arrays (Engine.cuh:108-318 getVecUhat / getVecBeta / getVecE / getPriceAlpha ...,
 * SmpcController.cuh:336-462 devVecX / devVecU / devVecV): RN_BUF_X, _U, _V, _UHAT, _E, _BETA, _ALPHA, _XDIR, _UDIR, node-major
 * [node][dim].  *precision = RN_F64 / RN_F32 (the element type is the context's), *n = element count.  The pointer stays valid until
 * rn_destroy; its contents are those of the last call that has COMPLETED on the context's stream (rn_synchronize, or order your own
 * work behind rn_stream).  x, u, v are stored by the last iteration of a batch and by every step-wise call.  (In the structured operator
 * mode v feeds nothing inside the solver and is normally computed when rn_get asks for it; asking for ITS raw pointer switches the context
 * to computing it with every such iteration, so the sentence before holds for the pointer's whole life.)
 * The scaled bounds RN_BUF_XMIN, _XMAX, _XS, _UMIN, _UMAX (Engine.cuh:294-314 getSysXmin ... getSysUmax) are kept interleaved in the dual
 * layout; the first request for one of them (after rn_factor_step) makes node-major copies of all five in the reference's layout
 * ((3 nx + 2 nu) reals per node: this call allocates), which every later rn_factor_step refreshes.  The dual iterates (RN_BUF_XI ...
 * RN_BUF_RES_PSI) and the operator blocks are kept in other layouts: RN_E_ARG, use rn_get / rn_get_operator. */
int rn_device_pointer(rn_ctx *ctx, int buffer_id, void **devPtr, size_t *n, int *precision);

/* ---- measurement --------------------------------------------------------------------------------- */
/* per-kernel-class device time accumulated by hipEvents when profiling is on (costs a little; off by default).
 * classes: 0 backward sweep, 1 forward sweep, 2 fused dual update, 3 bookkeeping. ms[] receives the totals,
 * launches[] the launch counts since the last rn_profile_reset. */
int rn_profile_enable(rn_ctx *ctx, int on);
int rn_profile_reset(rn_ctx *ctx);
int rn_profile_read(rn_ctx *ctx, double ms[4], long launches[4]);
/* sharded contexts: device time between hipEvents recorded on the solver's stream around every all-reduce (RCCL or an installed
 * stand-in) since the last rn_profile_reset -- wire latency plus the wait for the slowest peer; these intervals lie INSIDE
 * class 1 of rn_profile_read. */
int rn_profile_read_collective(rn_ctx *ctx, double *ms, long *launches);
/* algorithmic HBM bytes of ONE launch of the dominant kernels, as defined in DESIGN.md */
int rn_algorithmic_bytes(const rn_ctx *ctx, double *backwardStageBytesTotal, double *dualUpdateBytes);
/* which kernels an iteration of this context launches: info = {1 if k_dual_stage is the main pass of the fused dual update
 * (0: the flat k_dual_fused), its workgroups, vectors per thread, pipeline depth, k_stream_gemv columns per span, slots per
 * thread and span, first chain stage c*, 1 if the v / Lv products run as the fused slab kernel k_gemm_vlv} */
int rn_get_kernel_info(rn_ctx *ctx, int info[8]);
/* streaming ceilings of THIS device measured with do-nothing kernels (16 B per lane per load, non-temporal): flat
 * grid-stride read-only GB/s and copy GB/s (read + written bytes), best of `reps` passes over `bytes`.
 * bench.py reports them beside the 8 TB/s spec peak as the practical denominator. */
int rn_measure_hbm(rn_ctx *ctx, size_t bytes, int reps, double *readGBs, double *copyGBs);
/* the context's stream as a hipStream_t (void* to keep HIP types out of this header) */
void *rn_stream(rn_ctx *ctx);

/* ---- multi-GPU: subtree sharding with one RCCL all-reduce at the cut per iteration ---------------- */
/* see DESIGN.md "multi-GPU"; ncclUniqueId bytes are produced by rn_comm_unique_id on rank 0 and
 * distributed by the caller (bench.py uses torch.distributed for that). */
int rn_comm_unique_id(void *id128 /* 128 bytes out */);
/* id128 == NULL records rank / nranks without creating a communicator (the exchange is then a test's job, rapidnet_debug.h).
 * ncclCommInitRank waits for every rank of the id; it runs on a helper thread and the caller waits against the wall clock: a rank
 * whose peers do not arrive within the time-out ($RAPIDNET_COMM_TIMEOUT_S, default 120 s; rn_comm_init_timeout: `timeoutSeconds`)
 * gets RN_E_COMM back -- never a hang (the reference exits on any failure, Configuration.h:38-81) -- and the context stays usable
 * without a communicator. */
int rn_comm_init(rn_ctx *ctx, int rank, int nranks, const void *id128);
int rn_comm_init_timeout(rn_ctx *ctx, int rank, int nranks, const void *id128, double timeoutSeconds);
/* asynchronous errors of the communicator (ncclCommGetAsyncError: a peer died, a link went down): RN_E_COMM if RCCL reports one.
 * The library asks once per batch of rn_apg_iterate by itself. */
int rn_comm_check(rn_ctx *ctx);
/* Path of the RCCL image the library bound (an image the process has already loaded -- e.g. the one PyTorch links --
 * is reused, never a second copy), written as a C string into buf; RN_E_COMM when RCCL cannot be loaded. */
int rn_comm_library(char *buf, size_t n);
/* stage c of the (rank-local) tree whose nodes are the roots of the sharded subtrees: the children sums of the
 * stage c-1 nodes (replicated on every rank) are all-reduced once per iteration; -1 switches sharding off. */
int rn_set_cut_stage(rn_ctx *ctx, int stage);
/* How the tree-global prox distances (SmpcController.cu:792,810) are obtained across ranks.
 *   1 (default) optimistic: the prox runs as a pure projection, every rank's dist^2 of iteration t rides in the tail of
 *     iteration t+1's cut all-reduce and is checked on the device -- ONE collective per iteration; if a threshold is
 *     ever exceeded the batch is replayed from a checkpoint with mode 0, so the result is exact either way;
 *   0 exact: a second 2-element all-reduce per iteration before the trip decision.
 * Single-GPU contexts use the same switch: with 1 (default) a batch of >= 16 iterations runs the prox as a pure projection,
 * the bookkeeping of iteration t (history entry, distance check) rides in a launch of iteration t+1 instead of a decision
 * launch of its own, and the batch is replayed from a checkpoint through the exact path if a distance ever exceeded its
 * threshold; with 0 every iteration decides before the next one starts.  Results are identical. */
int rn_set_exchange_mode(rn_ctx *ctx, int mode);
/* static tree data a shard cannot derive from its local children: for every cut parent i (stage-1 nodes of the cut, in
 * stage order) E_i = sum over ALL children c of p_c * errorDemand_c (nd reals) and P_i = sum_c p_c.  They replace the
 * children loop of calculateZeta (Utilities.cu:100-131) for those nodes: sum_c p_c uhat_c = Lhat (E_i + P_i dhat). */
int rn_set_cut_children_moments(rn_ctx *ctx, const double *E /* parents*nd */, const double *P /* parents */, size_t nParents);
/* per-iteration {max|res_xi|, signed entry, max|res_psi|, signed entry} over THIS RANK's nodes for iterations [first, first+n).
 * (The history rn_apg_iterate returns is tree-global on a sharded context with a communicator: one MAX all-reduce per batch
 * combines the ranks' parts into vecPrimalInfs, SmpcController.cu:1480-1496, :1521.  Without a communicator -- id128 == NULL,
 * the exchange emulated by a test -- it is rank-local and these parts are what the caller combines.) */
int rn_get_history_parts(rn_ctx *ctx, int first, int n, double *out /* 4*n */);
/* Batch bookkeeping of rn_apg_iterate: out = {optimistic batches, exact batches, optimistic batches that were replayed
 * through the exact path because a tree-global prox distance exceeded its threshold, exact batches still to run before the
 * optimistic path is tried again (back-off after a replay)}. */
int rn_get_counters(rn_ctx *ctx, long out[4]);

/* ---- multi-GPU through the boundary: partition + communicator + cut stage in one call --------------------------------
 * The reference is single-GPU: its seam for a whole solver is `Engine(SmpcConfiguration*)` / `SmpcController(string)`
 * (Engine.cuh:68, SmpcController.cuh:57-87).  A sharded solver is created the same way, one per process / GPU, from the
 * FULL scenario tree plus (rank, nranks): the library cuts the tree, keeps the rank's subtrees and
 * the replicated crown, creates the RCCL communicator and installs the static children moments -- the host classes pass
 * rank / nranks straight through (Engine(config, precision, device, rank, nranks, id128)). */

/* First stage at which the tree has all its K scenario chains (the most balanced cut: K subtrees dealt round-robin);
 * max(1, N-1) for a tree that branches until the end.  Negative RN_E_* on bad input. */
int rn_default_cut_stage(const rn_dims *dims, const rn_tree *tree);

/* The rank-local scenario tree of a subtree partition.  Nodes of stages < cutStage (the crown) are replicated on every
 * rank, the subtrees rooted at stage cutStage are dealt round-robin by position; local nodes keep their stage-by-stage
 * order, so a local tree is an ordinary scenario tree (ScenarioTree.cuh:92-154 conventions: 1-based ancestor, nChildren
 * lists the non-leaf nodes -- a cut parent without local children is a local leaf).  All arrays are owned by the
 * partition object and stay valid until rn_partition_destroy. */
typedef struct {
    rn_dims dims;                 /* local: nodes, K (local leaves), nNonLeafNodes; nx..N as given                    */
    rn_tree tree;                 /* local tree arrays                                                                 */
    const int *globalNode;        /* [dims.nodes] index of every local node in the full tree                          */
    const double *errorDemandNode, *errorPriceNode; /* local rows of the full tree's errors (NULL if none were given)  */
    int rank, nranks, cutStage, nCutParents;
    /* static children moments of the cut parents over ALL their children (rn_set_cut_children_moments): the replacement
     * of the children loop of calculateZeta (Utilities.cu:100-131) for nodes whose children live on other ranks */
    const double *momE;           /* [nCutParents][nd]  sum_c p_c errorDemand_c   (NULL if no errors were given)        */
    const double *momP;           /* [nCutParents]      sum_c p_c                                                      */
    void *owner;                  /* internal */
} rn_partition;
int rn_partition_create(const rn_dims *dims, const rn_tree *tree, const double *errorDemandNode /* nodes*nd or NULL */,
                        const double *errorPriceNode /* nodes*nu or NULL */, int rank, int nranks,
                        int cutStage /* <= 0: rn_default_cut_stage */, rn_partition *out);
void rn_partition_destroy(rn_partition *p);

/* rn_create for rank `rank` of `nranks`: partition (above) + rn_create on the local tree + rn_set_tree_errors (local rows)
 * + rn_comm_init + rn_set_cut_stage + rn_set_cut_children_moments.  id128 = the ncclUniqueId of rn_comm_unique_id (rank 0's,
 * distributed by the caller); NULL creates no RCCL communicator (rn_comm_init may follow -- a caller that wants to keep the context when
 * the communicator cannot be created within its time-out does it that way; here a failed set-up destroys the context).
 * nranks == 1 gives a plain unsharded context. */
int rn_create_sharded(const rn_dims *dims, const rn_tree *tree, const double *errorDemandNode, const double *errorPriceNode,
                      int precision, int device, int rank, int nranks, int cutStage, const void *id128, rn_ctx **out);
/* what a sharded context is: info = {rank, nranks, cut stage (-1: none), cut parents, ranks the RCCL communicator itself
 * reports (ncclCommCount; 0 without a communicator), local nodes, nodes of the full tree} */
int rn_shard_info(rn_ctx *ctx, int info[7]);
/* index in the FULL tree of every local node (rn_get returns local node-major arrays); identity for unsharded contexts */
int rn_shard_global_nodes(rn_ctx *ctx, int *globalNode, size_t n);

/* ---- transport of the per-iteration exchange at the cut ----------------------------------------------
 * The reference has no counterpart (single GPU).  What is exchanged is exactly what solveSumChildren computes at the cut
 * (Utilities.cu:168-201): the cut parents' children sums, once per iteration.  Two transports carry it:
 *   RN_EXCHANGE_COLLECTIVE  ncclAllReduce on the solver's stream (RCCL), one launch between the chain walks and the crown;
 *   RN_EXCHANGE_ONESHOT     the kernel that produces a rank's partial sums writes them straight into an inbox on every peer (xGMI peer
 *                           mappings) as self-validating {32 payload bits, 32-bit sequence tag} packets and the same workgroup adds the
 *                           n contributions for its cut parent in rank order (the same bits on every rank): no collective launch, no
 *                           launch boundary.  The per-BATCH collectives (dist^2 tail, verdict + history) stay with the communicator.
 *   RN_EXCHANGE_AUTO        (the default) the context times both on its OWN iterations and keeps the faster: rn_exchange_autotune.
 * Inboxes: with a communicator made by rn_comm_init / rn_create_sharded the library allocates this rank's inbox (uncached device
 * memory), gathers the hipIpcMemHandle_t of all ranks over the communicator and maps the peers' inboxes by itself, unless the
 * transport was fixed to RN_EXCHANGE_COLLECTIVE before; if any rank cannot, all ranks agree to do without (AUTO then has one
 * candidate).  rn_peer_inbox_create / rn_peer_inbox_connect do the same with handles the caller distributes.  A reader that waits
 * longer than 2 s ($RAPIDNET_ONESHOT_TIMEOUT_MS) for a peer's packets gives up: the batch returns RN_E_COMM (never a hang) on every
 * rank (the flag rides in the per-batch MAX all-reduce).  rn_peer_inbox_connect is once per context.  $RAPIDNET_EXCHANGE =
 * collective | oneshot | auto sets the default of contexts that were not told.  (Tests wire the inboxes of contexts of one process:
 * rapidnet_debug.h.) */
enum { RN_EXCHANGE_COLLECTIVE = 0, RN_EXCHANGE_ONESHOT = 1, RN_EXCHANGE_AUTO = 2 };
int rn_peer_inbox_create(rn_ctx *ctx, void *ipcHandle64 /* 64 bytes out */);
int rn_peer_inbox_connect(rn_ctx *ctx, const void *ipcHandles /* nranks x 64 bytes, rank order */, int nranks);
int rn_set_exchange_transport(rn_ctx *ctx, int transport);
/* The exchange chooses itself.  iterations > 0: every rank of the communicator calls this at the same point (after the factor step and
 * the affine terms); each candidate transport runs `iterations` device-resident APG iterations of the context (after a warm-up run of
 * half as many) from the current iterates, which are restored afterwards -- the dual iterates (y, y+, w), the iteration count and the
 * batch counters are what they were (the output buffers x, u, v, z, res hold the tuner's last iteration until the next batch writes
 * them); the ranks' times are combined by a MAX all-reduce, so every rank keeps the same transport.  A context whose transport is AUTO
 * does this by itself in its first device-resident batch (min(max(n, 20), 100) iterations).  iterations == 0: report only.
 * info = {transport in use (0 / 1), candidates timed (bit 0 collective, bit 1 one-shot), microseconds per iteration with the
 * collective (max over ranks), with the one-shot exchange (max over ranks; -1: it failed on a rank; 0: not a candidate), this rank's
 * own two figures, iterations per candidate, tunes run so far}. */
int rn_exchange_autotune(rn_ctx *ctx, int iterations, double info[8]);
/* The forward walk and the fused dual update of the nodes it has just walked in ONE launch (k_down_chain_dual: Hx stays in LDS between
 * the two; SmpcController.cu:676-747 + :759-864 per node) inside batches of >= 16 iterations; identical iterates (bitwise).
 * on = -1 (default): on, shaped by the context -- one workgroup per scenario chain where the (rank-local) tree has at least 3/4 as many chains as
 * the device has compute units (the 493-scenario tree: 2 % faster per iteration, 5.6 % in structured mode), up to four per chain -- each updating
 * the dual of its own slice of the chain's nodes -- on small trees and small shards (1-4 % faster; with one per chain those measured 3-4 % slower);
 * 0 / 1: off / on whenever the shape allows it.  $RAPIDNET_FUSE_DOWN_DUAL = 0 / 1 sets the default of contexts that were not told.  While
 * rn_profile_enable is on the dual update always runs as a launch of its own. */
int rn_set_fused_walk_dual(rn_ctx *ctx, int on);
/* Device-buffer guard mode ("race detection / sanitizers": no GPU address sanitizer exists on this pool).
 * With RAPIDNET_GUARD=1 in the environment when a context is created, every device buffer of the context gets a 128 KiB red
 * zone on both sides; red zones and payloads of floating-point buffers start as NaN (0xFF bytes), so an out-of-bounds or
 * uninitialised READ carries a NaN into the iterates, and an out-of-bounds WRITE changes a red zone: rn_guard_check counts the
 * red-zone bytes that no longer hold their pattern (0 outside guard mode); rn_destroy reports them on stderr. */
int rn_guard_check(rn_ctx *ctx, long *badBytes);
/* process-wide tally of the checks rn_destroy makes in guard mode: out = {contexts checked so far, red-zone bytes found overwritten} */
int rn_guard_report(long out[2]);

#ifdef __cplusplus
}
#endif
#endif /* RAPIDNET_H_ */
