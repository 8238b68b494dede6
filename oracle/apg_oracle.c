/*
 * apg_oracle.c -- CPU restatement of RapidNet's APG solve path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle and the "port" CPU baseline of bench.py.  It is NOT part of the
 * product: nothing under rapidnet_amd/ may include, link or call it.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * It restates, serially and with the reference's own storage model (dense per-node operator blocks,
 * Omega/Theta aliased by scenario position, q/r scratch indexed by position within the stage), the
 * following reference code (paths relative to /root/reference/src):
 *
 *   Engine::initialiseSystemDevice      Engine.cu:382-463   (Rbar = L'WL, F, G, scaled bounds)
 *   preconditionSystem / ConstraintU/X  Utilities.cu:33-58, 360-405
 *   Engine::factorStep, inverseBatchMat Engine.cu:671-774, 1318-1359
 *   Engine::updateStateControl          Engine.cu:1300-1316
 *   Engine::eliminateInputDistubanceCoupling  Engine.cu:1147-1298
 *   calculateDiffUhat / calculateZeta   Utilities.cu:69-131
 *   SmpcController::dualExtrapolationStep  SmpcController.cu:535-557
 *   SmpcController::solveStep           SmpcController.cu:563-755
 *   solveSumChildren / solveChildNodesUpdate  Utilities.cu:142-201
 *   SmpcController::proximalFunG        SmpcController.cu:759-835 (+ projectionBox Utilities.cu:237)
 *   SmpcController::computeFixedPointResidual  SmpcController.cu:839-850
 *   SmpcController::dualUpdate          SmpcController.cu:854-864
 *   SmpcController::updatePrimalInfeasibity    SmpcController.cu:1480-1496
 *   SmpcController::algorithmApg        SmpcController.cu:1500-1525
 * and, for the global-FBE / NAMA outer loops (second half of this file):
 *   SmpcController::dualUpdate (FBE / NAMA branch)          SmpcController.cu:866-880
 *   SmpcController::computeHessianOracalGlobalFbe           SmpcController.cu:884-1055
 *   SmpcController::updateFixedPointResidualNamaAlgorithm   SmpcController.cu:1060-1072
 *   SmpcController::computeGradientFbe                      SmpcController.cu:1077-1097
 *   SmpcController::updateLbfgsBuffer / twoLoopRecursionLbfgs / computeLbfgsDirection  SmpcController.cu:1103-1237
 *   SmpcController::computeLineSearchLbfgsUpdate / computeLineSearchAmeLbfgsUpdate     SmpcController.cu:1242-1414
 *   SmpcController::computeValueFbe                         SmpcController.cu:1416-1476
 *   SmpcController::algorithmGlobalFbe / algorithmNama      SmpcController.cu:1529-1586
 *   pinned by src/test/testDataFiles/{smpcFbeTest,smpcNamaTest}.json, see tests/test_oracle_fbe_nama.py
 * and, for the closed loop around the solve (third part of this file):
 *   SmpcController::controlAction (both overloads)          SmpcController.cu:1607-1667 (+ projectionBox<<<1,nu>>> :1649)
 *   SmpcController::moveForewardInTime (in-built simulator) SmpcController.cu:1679-1716
 *   SmpcController::updateKpi, get{Economic,Smooth,Network,Safety}Kpi   SmpcController.cu:1778-1859
 *   (Forecaster::predictDemand / predictPrices, Forecaster.cu:93-119, are file lookups: the caller passes the horizons)
 *   The reference holds no golden vectors for these; they are short scalar loops restated line by line, checked in
 *   tests/test_oracle_closed_loop.py against a hand computation in numpy.
 *
 * Third-party arithmetic the reference calls and that is absent from /root/reference: cuBLAS
 * (gemm/gemv/axpy/scal/nrm2/isamax, getrfBatched/getriBatched) from the CUDA toolkit 7.0/8.0.  These
 * are textbook BLAS/LAPACK semantics; they are restated here as plain loops (LU with partial pivoting
 * for getrf/getri).  Parity is pinned by the reference's own golden vectors
 * (src/test/testDataFiles/{engineTest,smpcTest}.json), see tests/test_oracle_reference_fixtures.py.
 *
 * Deliberate, documented divergences from the reference (none is reachable from the pinned fixtures):
 *   - Engine.cu:1208 scales nu*nodes elements of an nd*nodes array when demand uncertainty is off;
 *     here the whole demand-error array is zeroed.
 *   - SmpcController.cu:800/818 clobber devVecDiffXi on the soft-constraint branch; here the
 *     mathematically correct prox of gamma*dist(.,C) is computed for both halves.
 *   - XsUpper is "no upper bound" (the reference memsets bytes 0x7F, Engine.cu:454-455).
 *   - FBE / NAMA: the L-BFGS buffers get one spare column (the reference indexes columns 1..m of m-column buffers,
 *     SmpcController.cu:1146) and matY is zeroed completely at reset (the reference zeroes nu*nodes entries, :466);
 *     the value of g at the prox point when the soft branch trips is gamma*dist(prox point, C) (the reference's
 *     :798-808 reads a partially filled scratch vector).
 *
 * Build: gcc -O3 -march=native -fPIC -shared [-DORACLE_REAL=float] -o liboracle_f64.so apg_oracle.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL real;

typedef struct {
    int nx, nu, nv, nd, N, K, nodes, nNonLeaf;
    /* tree, reference JSON conventions (ScenarioTree.cu:66-105): ancestor is 1-based, root = 0 */
    int *stages, *nodesPerStage, *nodesPerStageCumul, *ancestor, *nChildren, *nChildrenCumul;
    real *prob;
    int finalBranchNode; /* ScenarioTree.cu:147-156 */
    /* system (Engine.cuh:426-648) */
    real *B, *L, *Lhat, *Gd, *alpha1;
    real *Wv;            /* W*L, nu x nv (devMatWv) */
    real *sysF;          /* nodes x (2nx x nx), col-major ld 2nx */
    real *sysG;          /* nodes x (nu x nu) */
    real *xmin, *xmax, *xs, *umin, *umax; /* scaled, node-major */
    real *Omega, *Theta; /* finalBranchNode x (nv x nv) / (nv x nx) */
    int *opIdx;          /* node -> index into Omega/Theta (aliasing, Engine.cu:210-221) */
    real *Phi, *D;       /* nodes x (nv x 2nx) */
    real *Psi, *Ftil;    /* nodes x (nv x nu)  (devMatPsi, devMatF) */
    real *Gtil;          /* nv x nx  (devMatG: K identical copies of Bbar') */
    /* affine terms */
    real *curX, *prevU, *prevUhat, *prevD;
    real *uhat, *e, *beta, *alpha, *sigma;
    /* iterates (SmpcController.cuh:336-462) */
    real *x, *u, *v;
    real *xi, *psi, *accXi, *accPsi, *updXi, *updPsi;
    real *primalXi, *primalPsi, *dualXi, *dualPsi, *resXi, *resPsi;
    real *q, *r;         /* K x nx, K x nv, indexed by position within the stage */
    real stepSize, penaltyX, penaltyXs;
    real distXcst, distXs;
    /* ---- global-FBE / NAMA outer loops (SmpcController.cuh:380-458) ---- */
    int algorithm;             /* 0 APG, 1 globalFbeAlgorithm, 2 namaAlgorithm (Engine.cu:151-163) */
    int lbfgsSize, lbfgsCol, lbfgsMem, lbfgsSkip;
    real lbfgsH;
    real *prevXi, *prevPsi, *gradXi, *gradPsi, *prevGradXi, *prevGradPsi;
    real *curResXi, *curResPsi, *prevResXi, *prevResPsi;
    real *dirXi, *dirPsi, *xdir, *udir, *primalXiDir, *primalPsiDir;
    real *matS, *matY, *rho;   /* (lbfgsSize + 1) columns: the reference indexes columns 1..lbfgsSize (see oracle_lbfgs_update) */
    real valueGxBox, valueGxSafe, valueGuBox;
    real *W;                   /* costW, nu x nu (computeValueFbe) */
    /* closed loop */
    real *controlAction;       /* devControlAction, nu (SmpcController.cuh) */
    real *stateUpdate;         /* devStateUpdate, nx */
    real economicKpi, smoothKpi, safeKpi, networkKpi;   /* SmpcController.cu:103-106 (zeroed by the constructor) */
} oracle_t;

static real *ralloc(size_t n) { real *p = (real *)calloc(n ? n : 1, sizeof(real)); return p; }
static int *ialloc_copy(const int *src, size_t n) {
    int *p = (int *)malloc((n ? n : 1) * sizeof(int));
    memcpy(p, src, n * sizeof(int));
    return p;
}

/* y[m] = alpha*A[m x n]*x[n] + beta*y, A column-major with leading dimension lda (cublas gemv N) */
static void gemv_n(int m, int n, real alpha, const real *A, int lda, const real *x, real beta, real *y) {
    for (int i = 0; i < m; i++) y[i] = (beta == (real)0) ? (real)0 : beta * y[i];
    for (int j = 0; j < n; j++) {
        real xj = alpha * x[j];
        const real *col = A + (size_t)j * lda;
        for (int i = 0; i < m; i++) y[i] += col[i] * xj;
    }
}
/* y[n] = alpha*A'[n x m]*x[m] + beta*y (cublas gemv T on an m x n col-major matrix) */
static void gemv_t(int m, int n, real alpha, const real *A, int lda, const real *x, real beta, real *y) {
    for (int j = 0; j < n; j++) {
        const real *col = A + (size_t)j * lda;
        real s = 0;
        for (int i = 0; i < m; i++) s += col[i] * x[i];
        y[j] = alpha * s + ((beta == (real)0) ? (real)0 : beta * y[j]);
    }
}
/* C[m x n] = alpha*op(A)*op(B) + beta*C, all col-major. ta/tb: 0 = N, 1 = T.
 * Column-saxpy order so that gcc vectorises it; terms whose op(B) entry is an exact zero are skipped (the
 * reference multiplies by dense-stored diagonal F_i/G_i; adding 0*a changes nothing). */
static void gemm(int ta, int tb, int m, int n, int k, real alpha, const real *A, int lda, const real *Bm, int ldb,
                 real beta, real *C, int ldc) {
    real *At = NULL;
    if (ta) { /* materialise op(A) = A' as an m x k col-major matrix */
        At = (real *)malloc((size_t)m * k * sizeof(real));
        for (int p = 0; p < k; p++)
            for (int i = 0; i < m; i++) At[i + (size_t)p * m] = A[p + (size_t)i * lda];
        A = At; lda = m;
    }
    real *acc = (real *)malloc((size_t)m * sizeof(real));
    for (int j = 0; j < n; j++) {
        for (int i = 0; i < m; i++) acc[i] = 0;
        for (int p = 0; p < k; p++) {
            real b = tb ? Bm[j + (size_t)p * ldb] : Bm[p + (size_t)j * ldb];
            if (b == 0) continue;
            const real *col = A + (size_t)p * lda;
            for (int i = 0; i < m; i++) acc[i] += col[i] * b;
        }
        real *c = C + (size_t)j * ldc;
        if (beta == (real)0) for (int i = 0; i < m; i++) c[i] = alpha * acc[i];
        else for (int i = 0; i < m; i++) c[i] = alpha * acc[i] + beta * c[i];
    }
    free(acc); free(At);
}

/* inverse of an n x n col-major matrix via LU with partial pivoting (getrfBatched + getriBatched,
 * Engine.cu:1318-1359). Returns nonzero if singular. src is overwritten by its LU factors, like cuBLAS. */
static int lu_inverse(int n, real *A, real *Ainv) {
    int *piv = (int *)malloc(n * sizeof(int));
    for (int k = 0; k < n; k++) {
        int p = k;
        real mx = fabs((double)A[k + (size_t)k * n]);
        for (int i = k + 1; i < n; i++) {
            real a = fabs((double)A[i + (size_t)k * n]);
            if (a > mx) { mx = a; p = i; }
        }
        piv[k] = p;
        if (mx == 0) { free(piv); return k + 1; }
        if (p != k)
            for (int j = 0; j < n; j++) {
                real t = A[k + (size_t)j * n]; A[k + (size_t)j * n] = A[p + (size_t)j * n]; A[p + (size_t)j * n] = t;
            }
        real d = A[k + (size_t)k * n];
        for (int i = k + 1; i < n; i++) A[i + (size_t)k * n] /= d;
        for (int j = k + 1; j < n; j++) {
            real akj = A[k + (size_t)j * n];
            for (int i = k + 1; i < n; i++) A[i + (size_t)j * n] -= A[i + (size_t)k * n] * akj;
        }
    }
    /* solve A X = I column by column: P A = L U */
    real *b = (real *)malloc(n * sizeof(real));
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) b[i] = (i == c) ? 1 : 0;
        for (int k = 0; k < n; k++) if (piv[k] != k) { real t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
        for (int k = 0; k < n; k++) { real bk = b[k]; for (int i = k + 1; i < n; i++) b[i] -= A[i + (size_t)k * n] * bk; }
        for (int k = n - 1; k >= 0; k--) {
            b[k] /= A[k + (size_t)k * n];
            real bk = b[k];
            for (int i = 0; i < k; i++) b[i] -= A[i + (size_t)k * n] * bk;
        }
        for (int i = 0; i < n; i++) Ainv[i + (size_t)c * n] = b[i];
    }
    free(b); free(piv);
    return 0;
}

/* Tests of rank-local (sharded) trees switch the reference's Omega/Theta aliasing off: a local tree may keep branching
 * probabilities that its own shape no longer shows, so every node gets its own (p_i Rbar)^-1. */
static int g_alias_operators = 1;
void oracle_config_aliasing(int on) { g_alias_operators = on; }

oracle_t *oracle_create(int nx, int nu, int nv, int nd, int N, int K, int nodes, int nNonLeaf,
                        const int *stages, const int *nodesPerStage, const int *nodesPerStageCumul,
                        const int *ancestor, const int *nChildren, const int *nChildrenCumul, const double *prob) {
    oracle_t *o = (oracle_t *)calloc(1, sizeof(oracle_t));
    o->nx = nx; o->nu = nu; o->nv = nv; o->nd = nd; o->N = N; o->K = K; o->nodes = nodes; o->nNonLeaf = nNonLeaf;
    o->stages = ialloc_copy(stages, nodes);
    o->nodesPerStage = ialloc_copy(nodesPerStage, N + 1);
    o->nodesPerStageCumul = ialloc_copy(nodesPerStageCumul, N + 2);
    o->ancestor = ialloc_copy(ancestor, nodes);
    o->nChildren = ialloc_copy(nChildren, nNonLeaf);
    o->nChildrenCumul = ialloc_copy(nChildrenCumul, nodes);
    o->prob = ralloc(nodes);
    for (int i = 0; i < nodes; i++) o->prob[i] = (real)prob[i];
    /* ScenarioTree::getFinalBranchNode, ScenarioTree.cu:147-156 */
    o->finalBranchNode = 0;
    for (int i = 0; i < N - 1; i++)
        if (nodesPerStage[i] == nodesPerStage[i + 1]) { o->finalBranchNode = nodesPerStageCumul[i + 1]; break; }
    if (!g_alias_operators) o->finalBranchNode = nodes;
    int fb = o->finalBranchNode;
    /* operator aliasing, Engine.cu:210-221 */
    o->opIdx = (int *)malloc(nodes * sizeof(int));
    for (int k = 0; k < N; k++) {
        int cum = nodesPerStageCumul[k];
        for (int j = 0; j < nodesPerStage[k]; j++) o->opIdx[cum + j] = (fb <= cum) ? (fb - K + j) : (cum + j);
    }
    size_t n = nodes;
    o->B = ralloc((size_t)nx * nu); o->L = ralloc((size_t)nu * nv); o->Lhat = ralloc((size_t)nu * nd);
    o->Gd = ralloc((size_t)nx * nd); o->alpha1 = ralloc(nu); o->Wv = ralloc((size_t)nu * nv);
    o->sysF = ralloc(n * 2 * nx * nx); o->sysG = ralloc(n * nu * nu);
    o->xmin = ralloc(n * nx); o->xmax = ralloc(n * nx); o->xs = ralloc(n * nx);
    o->umin = ralloc(n * nu); o->umax = ralloc(n * nu);
    o->Omega = ralloc((size_t)(fb > 0 ? fb : 1) * nv * nv); o->Theta = ralloc((size_t)(fb > 0 ? fb : 1) * nv * nx);
    o->Phi = ralloc(n * nv * 2 * nx); o->D = ralloc(n * nv * 2 * nx);
    o->Psi = ralloc(n * nv * nu); o->Ftil = ralloc(n * nv * nu); o->Gtil = ralloc((size_t)nv * nx);
    o->curX = ralloc(nx); o->prevU = ralloc(nu); o->prevUhat = ralloc(nu); o->prevD = ralloc(nd);
    o->uhat = ralloc(n * nu); o->e = ralloc(n * nx); o->beta = ralloc(n * nv); o->alpha = ralloc(n * nu);
    o->sigma = ralloc(n * nv);
    o->x = ralloc(n * nx); o->u = ralloc(n * nu); o->v = ralloc(n * nv);
    o->xi = ralloc(n * 2 * nx); o->psi = ralloc(n * nu);
    o->accXi = ralloc(n * 2 * nx); o->accPsi = ralloc(n * nu);
    o->updXi = ralloc(n * 2 * nx); o->updPsi = ralloc(n * nu);
    o->primalXi = ralloc(n * 2 * nx); o->primalPsi = ralloc(n * nu);
    o->dualXi = ralloc(n * 2 * nx); o->dualPsi = ralloc(n * nu);
    o->resXi = ralloc(n * 2 * nx); o->resPsi = ralloc(n * nu);
    o->q = ralloc((size_t)K * nx); o->r = ralloc((size_t)K * nv);
    o->stepSize = (real)1e-4; o->penaltyX = (real)1e6; o->penaltyXs = (real)1e4;
    o->W = ralloc((size_t)nu * nu);
    o->controlAction = ralloc(nu); o->stateUpdate = ralloc(nx);
    return o;
}

void oracle_destroy(oracle_t *o) {
    if (!o) return;
    void *ptrs[] = {o->stages, o->nodesPerStage, o->nodesPerStageCumul, o->ancestor, o->nChildren, o->nChildrenCumul,
                    o->prob, o->opIdx, o->B, o->L, o->Lhat, o->Gd, o->alpha1, o->Wv, o->sysF, o->sysG, o->xmin, o->xmax,
                    o->xs, o->umin, o->umax, o->Omega, o->Theta, o->Phi, o->D, o->Psi, o->Ftil, o->Gtil, o->curX,
                    o->prevU, o->prevUhat, o->prevD, o->uhat, o->e, o->beta, o->alpha, o->sigma, o->x, o->u, o->v,
                    o->xi, o->psi, o->accXi, o->accPsi, o->updXi, o->updPsi, o->primalXi, o->primalPsi, o->dualXi,
                    o->dualPsi, o->resXi, o->resPsi, o->q, o->r, o->controlAction, o->stateUpdate};
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(ptrs[i]);
    free(o);
}

void oracle_set_params(oracle_t *o, double stepSize, double penaltyX, double penaltyXs) {
    o->stepSize = (real)stepSize; o->penaltyX = (real)penaltyX; o->penaltyXs = (real)penaltyXs;
}

/* Engine::initialiseSystemDevice (Engine.cu:382-463) followed by Engine::factorStep (Engine.cu:671-774).
 * matL / matLhat are taken as given (the reference recomputes them by SVD, Engine.cu:466-669; the basis of
 * a null space is not unique, so the caller supplies the one to use).  Returns nonzero if p_i*Rbar is singular. */
int oracle_factor_step(oracle_t *o, const double *matB, const double *matL, const double *matLhat, const double *matGd,
                       const double *costW, const double *diagPrecnd, const double *xmin, const double *xmax,
                       const double *xsafe, const double *umin, const double *umax, const double *alpha1) {
    int nx = o->nx, nu = o->nu, nv = o->nv, nd = o->nd, N = o->N, nodes = o->nodes, fb = o->finalBranchNode;
    for (int i = 0; i < nx * nu; i++) o->B[i] = (real)matB[i];
    for (int i = 0; i < nu * nv; i++) o->L[i] = (real)matL[i];
    for (int i = 0; i < nu * nd; i++) o->Lhat[i] = (real)matLhat[i];
    for (int i = 0; i < nx * nd; i++) o->Gd[i] = (real)matGd[i];
    for (int i = 0; i < nu; i++) o->alpha1[i] = (real)alpha1[i];
    real *W = ralloc((size_t)nu * nu);
    for (int i = 0; i < nu * nu; i++) { W[i] = (real)costW[i]; o->W[i] = W[i]; }
    /* Wv = W*L ; Rbar = L' * Wv   (Engine.cu:412-416) */
    gemm(0, 0, nu, nv, nu, 1, W, nu, o->L, nu, 0, o->Wv, nu);
    real *Rbar = ralloc((size_t)nv * nv);
    gemm(1, 0, nv, nv, nu, 1, o->L, nu, o->Wv, nu, 0, Rbar, nv);
    memset(o->sysF, 0, (size_t)nodes * 2 * nx * nx * sizeof(real));
    memset(o->sysG, 0, (size_t)nodes * nu * nu * sizeof(real));
    /* per-node copies of the bounds and p_i*Rbar for the nodes that own an Omega (Engine.cu:421-437) */
    real *costWnode = ralloc((size_t)(fb > 0 ? fb : 1) * nv * nv);
    for (int i = 0; i < nodes; i++) {
        for (int j = 0; j < nx; j++) {
            o->xmin[(size_t)i * nx + j] = (real)xmin[j]; o->xmax[(size_t)i * nx + j] = (real)xmax[j];
            o->xs[(size_t)i * nx + j] = (real)xsafe[j];
        }
        for (int j = 0; j < nu; j++) { o->umin[(size_t)i * nu + j] = (real)umin[j]; o->umax[(size_t)i * nu + j] = (real)umax[j]; }
        if (i < fb)
            for (int j = 0; j < nv * nv; j++) costWnode[(size_t)i * nv * nv + j] = o->prob[i] * Rbar[j];
    }
    /* preconditionSystem / preconditionConstraintU / preconditionConstraintX (Utilities.cu:33-58, 360-405);
     * stage-k slice of matDiagPrecnd is (d_u[nu] | d_x[nx] | d_xs[nx]) */
    for (int k = 0; k < N; k++) {
        int cum = o->nodesPerStageCumul[k];
        const double *dk = diagPrecnd + (size_t)k * (2 * nx + nu);
        for (int j = 0; j < o->nodesPerStage[k]; j++) {
            int i = cum + j;
            real sp = (real)sqrt((double)o->prob[i]);
            real *G = o->sysG + (size_t)i * nu * nu;
            real *F = o->sysF + (size_t)i * 2 * nx * nx;
            for (int t = 0; t < nu; t++) G[(size_t)nu * t + t] = sp * (real)dk[t];
            for (int t = 0; t < nx; t++) {
                F[(size_t)2 * nx * t + t] = sp * (real)dk[nu + t];            /* rows 0..nx-1 : diag(d_x)  */
                F[(size_t)2 * nx * t + nx + t] = sp * (real)dk[nu + nx + t];  /* rows nx..2nx-1: diag(d_xs) */
            }
            for (int t = 0; t < nu; t++) {
                real s = sp * (real)dk[t];
                o->umax[(size_t)i * nu + t] *= s; o->umin[(size_t)i * nu + t] *= s;
            }
            for (int t = 0; t < nx; t++) {
                real sx = sp * (real)dk[nu + t], sxs = sp * (real)dk[nu + nx + t];
                o->xmax[(size_t)i * nx + t] *= sx; o->xmin[(size_t)i * nx + t] *= sx; o->xs[(size_t)i * nx + t] *= sxs;
            }
        }
    }
    /* factorStep: Bbar' = L'B' (nv x nx)  (Engine.cu:702-705) */
    gemm(1, 1, nv, nx, nu, 1, o->L, nu, o->B, nx, 0, o->Gtil, nv);
    /* Omega = (p Rbar)^-1 for the owning nodes (Engine.cu:707-714) */
    for (int i = 0; i < fb; i++)
        if (lu_inverse(nv, costWnode + (size_t)i * nv * nv, o->Omega + (size_t)i * nv * nv)) {
            free(W); free(Rbar); free(costWnode);
            return i + 1;
        }
    for (int k = N - 1; k > -1; k--) {
        int cum = o->nodesPerStageCumul[k];
        for (int j = 0; j < o->nodesPerStage[k]; j++) {
            int i = cum + j;
            const real *Om = o->Omega + (size_t)o->opIdx[i] * nv * nv;
            /* Ftil = L' G_i'  (nv x nu)   Engine.cu:721-722 */
            gemm(1, 1, nv, nu, nu, 1, o->L, nu, o->sysG + (size_t)i * nu * nu, nu, 0, o->Ftil + (size_t)i * nv * nu, nv);
            /* D = Bbar' F_i'  (nv x 2nx)  Engine.cu:726-727 */
            gemm(0, 1, nv, 2 * nx, nx, 1, o->Gtil, nv, o->sysF + (size_t)i * 2 * nx * nx, 2 * nx, 0,
                 o->D + (size_t)i * nv * 2 * nx, nv);
            /* Phi = -1/2 Omega D          Engine.cu:729-731 */
            gemm(0, 0, nv, 2 * nx, nv, (real)-0.5, Om, nv, o->D + (size_t)i * nv * 2 * nx, nv, 0,
                 o->Phi + (size_t)i * nv * 2 * nx, nv);
            /* Theta = -1/2 Omega Bbar'    Engine.cu:732-737 (owning nodes only) */
            if (fb > cum)
                gemm(0, 0, nv, nx, nv, (real)-0.5, Om, nv, o->Gtil, nv, 0, o->Theta + (size_t)i * nv * nx, nv);
            /* Psi = -1/2 Omega Ftil       Engine.cu:743-745 */
            gemm(0, 0, nv, nu, nv, (real)-0.5, Om, nv, o->Ftil + (size_t)i * nv * nu, nv, 0,
                 o->Psi + (size_t)i * nv * nu, nv);
        }
    }
    free(W); free(Rbar); free(costWnode);
    return 0;
}

/* Engine::updateStateControl, Engine.cu:1300-1316 */
void oracle_update_state_control(oracle_t *o, const double *currentX, const double *prevU, const double *prevDemand) {
    for (int i = 0; i < o->nx; i++) o->curX[i] = (real)currentX[i];
    for (int i = 0; i < o->nu; i++) o->prevU[i] = (real)prevU[i];
    for (int i = 0; i < o->nd; i++) o->prevD[i] = (real)prevDemand[i];
    gemv_n(o->nu, o->nd, 1, o->Lhat, o->nu, o->prevD, 0, o->prevUhat);
}

/* Engine::eliminateInputDistubanceCoupling, Engine.cu:1147-1298 */
void oracle_eliminate(oracle_t *o, const double *nominalDemand, const double *nominalPrices, const double *errDemand,
                      const double *errPrice, double weightEconomical, int demandUncertainty, int priceUncertainty) {
    int nx = o->nx, nu = o->nu, nv = o->nv, nd = o->nd, N = o->N, nodes = o->nodes;
    real *d = ralloc((size_t)nodes * nd);
    real *alphaHat = ralloc((size_t)N * nu);
    real *alphaBar = ralloc((size_t)nodes * nv);
    real *dU = ralloc((size_t)nodes * nu), *zeta = ralloc((size_t)nodes * nu);
    for (size_t i = 0; i < (size_t)nodes * nd; i++) d[i] = demandUncertainty ? (real)errDemand[i] : (real)0;
    /* d(node) = dhat(stage) + err(node);  e = Gd d   (Engine.cu:1211-1222) */
    for (int k = 0; k < N; k++) {
        int cum = o->nodesPerStageCumul[k];
        for (int j = 0; j < o->nodesPerStage[k]; j++) {
            int i = cum + j;
            for (int t = 0; t < nd; t++) d[(size_t)i * nd + t] += (real)nominalDemand[(size_t)k * nd + t];
            gemv_n(nx, nd, 1, o->Gd, nx, d + (size_t)i * nd, 0, o->e + (size_t)i * nx);
        }
    }
    /* uhat = Lhat d  (Engine.cu:1224-1225) */
    for (int i = 0; i < nodes; i++) gemv_n(nu, nd, 1, o->Lhat, nu, d + (size_t)i * nd, 0, o->uhat + (size_t)i * nu);
    /* alpha = w_e (errPrice + alphaHat(stage) + alpha1)  (Engine.cu:1227-1243) */
    for (int k = 0; k < N; k++)
        for (int t = 0; t < nu; t++) alphaHat[(size_t)k * nu + t] = (real)nominalPrices[(size_t)k * nu + t] + o->alpha1[t];
    for (int i = 0; i < nodes; i++) {
        int k = o->stages[i];
        for (int t = 0; t < nu; t++) {
            real a = priceUncertainty ? (real)errPrice[(size_t)i * nu + t] : (real)0;
            o->alpha[(size_t)i * nu + t] = (real)weightEconomical * (a + alphaHat[(size_t)k * nu + t]);
        }
    }
    /* alphaBar = L' alpha  (Engine.cu:1245-1246) */
    for (int i = 0; i < nodes; i++) gemv_t(nu, nv, 1, o->L, nu, o->alpha + (size_t)i * nu, 0, alphaBar + (size_t)i * nv);
    /* calculateDiffUhat, Utilities.cu:69-88 (ancestor is 1-based) */
    for (int i = 0; i < nodes; i++)
        for (int t = 0; t < nu; t++) {
            if (i == 0) dU[t] = o->uhat[t] - o->prevUhat[t];
            else dU[(size_t)i * nu + t] = o->uhat[(size_t)i * nu + t] - o->uhat[(size_t)(o->ancestor[i] - 1) * nu + t];
        }
    /* calculateZeta, Utilities.cu:100-131: children of node i are nodes cum[i-1]+1 .. cum[i] (0-based) */
    for (int i = 0; i < nodes; i++)
        for (int t = 0; t < nu; t++) {
            real z = o->prob[i] * dU[(size_t)i * nu + t];
            if (i < o->nNonLeaf) {
                int c0 = (i == 0) ? 0 : o->nChildrenCumul[i - 1];
                int nc = o->nChildrenCumul[i] - c0;
                for (int c = 0; c < nc; c++) {
                    int ch = c0 + c + 1;
                    z -= o->prob[ch] * dU[(size_t)ch * nu + t];
                }
            }
            zeta[(size_t)i * nu + t] = z;
        }
    /* beta = 2 (W L)' zeta + p alphaBar  (Engine.cu:1253-1261) */
    for (int i = 0; i < nodes; i++) {
        gemv_t(nu, nv, 2, o->Wv, nu, zeta + (size_t)i * nu, 0, o->beta + (size_t)i * nv);
        for (int t = 0; t < nv; t++) o->beta[(size_t)i * nv + t] += o->prob[i] * alphaBar[(size_t)i * nv + t];
    }
    free(d); free(alphaHat); free(alphaBar); free(dU); free(zeta);
}

/* SmpcController::initialiseAlgorithm, SmpcController.cu:420-450 (APG branch) */
void oracle_apg_reset(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    memset(o->xi, 0, nxi * sizeof(real)); memset(o->psi, 0, nps * sizeof(real));
    memset(o->accXi, 0, nxi * sizeof(real)); memset(o->accPsi, 0, nps * sizeof(real));
    memset(o->primalXi, 0, nxi * sizeof(real)); memset(o->primalPsi, 0, nps * sizeof(real));
    memset(o->dualXi, 0, nxi * sizeof(real)); memset(o->dualPsi, 0, nps * sizeof(real));
    memset(o->updXi, 0, nxi * sizeof(real)); memset(o->updPsi, 0, nps * sizeof(real));
}

/* SmpcController::dualExtrapolationStep, SmpcController.cu:535-557 */
void oracle_extrapolate(oracle_t *o, double lambda_) {
    real lambda = (real)lambda_;
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    for (size_t i = 0; i < nxi; i++) { o->accXi[i] = (1 + lambda) * o->updXi[i] + (-lambda) * o->xi[i]; o->xi[i] = o->updXi[i]; }
    for (size_t i = 0; i < nps; i++) { o->accPsi[i] = (1 + lambda) * o->updPsi[i] + (-lambda) * o->psi[i]; o->psi[i] = o->updPsi[i]; }
}

/* solveSumChildren, Utilities.cu:168-201: dst[parent position] = sum over its (contiguous) children of src */
static void sum_children(const oracle_t *o, const real *src, real *dst, int stageCumul, int stageNodes, int iStage, int dim) {
    for (int rel = 0; rel < stageNodes; rel++) {
        int offset = 0, numChild;
        if (iStage > 0) {
            offset = (o->nChildrenCumul[stageCumul + rel - 1] - o->nChildrenCumul[stageCumul - 1]) * dim;
            numChild = o->nChildren[stageCumul + rel];
        } else numChild = o->nChildren[rel];
        for (int t = 0; t < dim; t++) {
            real s = src[offset + t];
            for (int c = 1; c < numChild; c++) s += src[offset + t + c * dim];
            dst[rel * dim + t] = s;
        }
    }
}

/* backward substitution of SmpcController::solveStep (SmpcController.cu:593-673) for stages kHi .. kLo */
void oracle_backward_range(oracle_t *o, int kHi, int kLo) {
    int nx = o->nx, nu = o->nu, nv = o->nv, N = o->N, K = o->K;
    real *tmpQ = ralloc((size_t)K * nx), *tmpR = ralloc((size_t)K * nv);
    for (int k = kHi; k >= kLo; k--) {
        int cum = o->nodesPerStageCumul[k], nk = o->nodesPerStage[k];
        for (int j = 0; j < nk; j++) {
            int i = cum + j;
            real *sig = o->sigma + (size_t)i * nv, *v = o->v + (size_t)i * nv;
            real *qj = o->q + (size_t)j * nx, *rj = o->r + (size_t)j * nv;
            const real *xi = o->accXi + (size_t)i * 2 * nx, *psi = o->accPsi + (size_t)i * nu;
            if (k < N - 1) for (int t = 0; t < nv; t++) sig[t] += rj[t];                         /* :599 */
            gemv_n(nv, nv, (real)-0.5, o->Omega + (size_t)o->opIdx[i] * nv * nv, nv, sig, 0, v); /* :604 */
            if (k < N - 1) gemv_n(nv, nx, 1, o->Theta + (size_t)o->opIdx[i] * nv * nx, nv, qj, 1, v); /* :611 */
            gemv_n(nv, nu, 1, o->Psi + (size_t)i * nv * nu, nv, psi, 1, v);                      /* :617 */
            gemv_n(nv, 2 * nx, 1, o->Phi + (size_t)i * nv * 2 * nx, nv, xi, 1, v);               /* :623 */
            memcpy(rj, sig, nv * sizeof(real));                                                  /* :629 */
            gemv_n(nv, 2 * nx, 1, o->D + (size_t)i * nv * 2 * nx, nv, xi, 1, rj);                /* :633 */
            gemv_n(nv, nu, 1, o->Ftil + (size_t)i * nv * nu, nv, psi, 1, rj);                    /* :638 */
            if (k < N - 1) gemv_n(nv, nx, 1, o->Gtil, nv, qj, 1, rj);                            /* :644 */
            { /* q = F'xi (+ q)  :651/656 ; F_i = [diag; diag] stored dense, only its non-zeros are touched */
                const real *F = o->sysF + (size_t)i * 2 * nx * nx;
                for (int t = 0; t < nx; t++) {
                    real s = F[(size_t)2 * nx * t + t] * xi[t] + F[(size_t)2 * nx * t + nx + t] * xi[nx + t];
                    qj[t] = (k < N - 1) ? s + qj[t] : s;
                }
            }
        }
        if (k > 0) {
            int pn = o->nodesPerStage[k - 1], pc = o->nodesPerStageCumul[k - 1];
            if (nk - pn > 0) {                                                                   /* :661-672 */
                sum_children(o, o->q, tmpQ, pc, pn, k - 1, nx);
                sum_children(o, o->r, tmpR, pc, pn, k - 1, nv);
                memcpy(o->r, tmpR, (size_t)pn * nv * sizeof(real));
                memcpy(o->q, tmpQ, (size_t)pn * nx * sizeof(real));
            }
        }
    }
    free(tmpQ); free(tmpR);
}

void oracle_forward(oracle_t *o);

/* SmpcController::solveStep, SmpcController.cu:563-755 */
void oracle_solve_step(oracle_t *o) {
    memcpy(o->sigma, o->beta, (size_t)o->nodes * o->nv * sizeof(real));                          /* :587 */
    oracle_backward_range(o, o->N - 1, 0);
    oracle_forward(o);
}
/* multi-GPU emulation (tests only): phase 0 runs the backward sweep down to the cut stage and leaves, in q/r, the
 * sums over the LOCAL children of every cut parent; the caller all-reduces them over the ranks; phase 1 finishes. */
void oracle_solve_step_phase(oracle_t *o, int phase, int cutStage) {
    if (phase == 0) {
        memcpy(o->sigma, o->beta, (size_t)o->nodes * o->nv * sizeof(real));
        oracle_backward_range(o, o->N - 1, cutStage);
    } else {
        oracle_backward_range(o, cutStage - 1, 0);
        oracle_forward(o);
    }
}

/* forward substitution and Hx of SmpcController::solveStep (SmpcController.cu:676-747) */
void oracle_forward(oracle_t *o) {
    int nx = o->nx, nu = o->nu, nv = o->nv, N = o->N, nodes = o->nodes, K = o->K;
    real *Lv = ralloc((size_t)K * nu);
    /* forward substitution */
    memcpy(o->u, o->uhat, (size_t)nodes * nu * sizeof(real));
    for (int k = 0; k < N; k++) {
        int cum = o->nodesPerStageCumul[k], nk = o->nodesPerStage[k];
        if (k == 0) {
            for (int t = 0; t < nu; t++) o->u[t] += o->prevU[t];                                 /* :683 */
            for (int t = 0; t < nu; t++) o->u[t] += -o->prevUhat[t];                             /* :685 */
            for (int t = 0; t < nx; t++) o->x[t] = o->curX[t] + o->e[t];                         /* :688-690 */
            gemv_n(nu, nv, 1, o->L, nu, o->v, 1, o->u);                                          /* :692 */
            gemv_n(nx, nu, 1, o->B, nx, o->u, 1, o->x);                                          /* :695 */
        } else {
            int pc = o->nodesPerStageCumul[k - 1], pn = o->nodesPerStage[k - 1];
            if (nk - pn > 0) {
                for (int j = 0; j < nk; j++)
                    gemv_n(nu, nv, 1, o->L, nu, o->v + (size_t)(cum + j) * nv, 1, o->u + (size_t)(cum + j) * nu); /* :701 */
                for (size_t t = 0; t < (size_t)pn * nu; t++) Lv[t] = o->u[(size_t)pc * nu + t] - o->uhat[(size_t)pc * nu + t]; /* :705-707 */
                /* solveChildNodesUpdate, Utilities.cu:142-155 */
                int prevAnc = o->ancestor[cum];
                for (int j = 0; j < nk; j++) {
                    int a = o->ancestor[cum + j] - prevAnc;
                    for (int t = 0; t < nu; t++) o->u[(size_t)(cum + j) * nu + t] += Lv[(size_t)a * nu + t];    /* :709 */
                }
                memcpy(o->x + (size_t)cum * nx, o->e + (size_t)cum * nx, (size_t)nk * nx * sizeof(real));       /* :712 */
                for (int j = 0; j < nk; j++)
                    gemv_n(nx, nu, 1, o->B, nx, o->u + (size_t)(cum + j) * nu, 1, o->x + (size_t)(cum + j) * nx); /* :715 */
                for (int j = 0; j < nk; j++) {
                    int a = o->ancestor[cum + j] - prevAnc;
                    for (int t = 0; t < nx; t++) o->x[(size_t)(cum + j) * nx + t] += o->x[(size_t)(pc + a) * nx + t]; /* :718 */
                }
            } else {
                for (size_t t = 0; t < (size_t)nk * nu; t++) o->u[(size_t)cum * nu + t] += o->u[(size_t)pc * nu + t];     /* :722 */
                for (size_t t = 0; t < (size_t)nk * nu; t++) o->u[(size_t)cum * nu + t] += -o->uhat[(size_t)pc * nu + t]; /* :724 */
                for (int j = 0; j < nk; j++)
                    gemv_n(nu, nv, 1, o->L, nu, o->v + (size_t)(cum + j) * nv, 1, o->u + (size_t)(cum + j) * nu); /* :727 */
                for (size_t t = 0; t < (size_t)nk * nx; t++)
                    o->x[(size_t)cum * nx + t] = o->x[(size_t)pc * nx + t] + o->e[(size_t)cum * nx + t];        /* :730-733 */
                for (int j = 0; j < nk; j++)
                    gemv_n(nx, nu, 1, o->B, nx, o->u + (size_t)(cum + j) * nu, 1, o->x + (size_t)(cum + j) * nx); /* :736 */
            }
        }
    }
    /* Hx  (:744-747) */
    for (int i = 0; i < nodes; i++) {
        const real *F = o->sysF + (size_t)i * 2 * nx * nx, *G = o->sysG + (size_t)i * nu * nu;
        for (int t = 0; t < nx; t++) {
            o->primalXi[(size_t)i * 2 * nx + t] = F[(size_t)2 * nx * t + t] * o->x[(size_t)i * nx + t];
            o->primalXi[(size_t)i * 2 * nx + nx + t] = F[(size_t)2 * nx * t + nx + t] * o->x[(size_t)i * nx + t];
        }
        for (int t = 0; t < nu; t++) o->primalPsi[(size_t)i * nu + t] = G[(size_t)nu * t + t] * o->u[(size_t)i * nu + t];
    }
    free(Lv);
}

/* SmpcController::proximalFunG, SmpcController.cu:759-835.  proxW* is the vector ptrProximalXi/Psi points
 * at: the accelerated dual for APG (SmpcController.cu:510-511). */
void oracle_prox(oracle_t *o) {
    int nx = o->nx, nu = o->nu, nodes = o->nodes;
    real invLambda = 1 / o->stepSize;
    size_t nxi = (size_t)nodes * 2 * nx, nps = (size_t)nodes * nu;
    real *diff = ralloc(nxi);
    for (size_t i = 0; i < nxi; i++) o->dualXi[i] = o->primalXi[i] + invLambda * o->accXi[i];   /* :778-780 */
    for (size_t i = 0; i < nps; i++) o->dualPsi[i] = o->primalPsi[i] + invLambda * o->accPsi[i];
    memcpy(diff, o->dualXi, nxi * sizeof(real));
    double d2x = 0, d2s = 0;
    for (int i = 0; i < nodes; i++)
        for (int t = 0; t < nx; t++) {
            real *zb = &o->dualXi[(size_t)i * 2 * nx + t], *zs = &o->dualXi[(size_t)i * 2 * nx + nx + t];
            real lo = o->xmin[(size_t)i * nx + t], hi = o->xmax[(size_t)i * nx + t], ls = o->xs[(size_t)i * nx + t];
            if (*zb < lo) *zb = lo; else if (*zb > hi) *zb = hi;                                 /* :785 */
            if (*zs < ls) *zs = ls;                                                              /* :786 (upper = +BIG) */
            real db = diff[(size_t)i * 2 * nx + t] - *zb, ds = diff[(size_t)i * 2 * nx + nx + t] - *zs;
            diff[(size_t)i * 2 * nx + t] = db; diff[(size_t)i * 2 * nx + nx + t] = ds;           /* :789 */
            d2x += (double)db * db; d2s += (double)ds * ds;
        }
    o->distXcst = (real)sqrt(d2x); o->distXs = (real)sqrt(d2s);                                  /* :792, :810 */
    o->valueGxBox = 0; o->valueGxSafe = 0; o->valueGuBox = 0;   /* :808, :826, :828: zero unless the soft branch trips */
    if (o->distXcst > invLambda * o->penaltyX) {                                                 /* :793-797 */
        real sc = 1 - invLambda * o->penaltyX / o->distXcst;
        o->valueGxBox = o->penaltyX * (o->distXcst - invLambda * o->penaltyX);   /* gamma * dist(prox point, C) */
        for (int i = 0; i < nodes; i++)
            for (int t = 0; t < nx; t++) o->dualXi[(size_t)i * 2 * nx + t] += sc * diff[(size_t)i * 2 * nx + t];
    }
    if (o->distXs > invLambda * o->penaltyXs) {                                                  /* :811-815 */
        real sc = 1 - invLambda * o->penaltyXs / o->distXs;
        o->valueGxSafe = o->penaltyXs * (o->distXs - invLambda * o->penaltyXs);
        for (int i = 0; i < nodes; i++)
            for (int t = 0; t < nx; t++) o->dualXi[(size_t)i * 2 * nx + nx + t] += sc * diff[(size_t)i * 2 * nx + nx + t];
    }
    for (size_t i = 0; i < nps; i++) {                                                           /* :827 */
        if (o->dualPsi[i] < o->umin[i]) o->dualPsi[i] = o->umin[i];
        else if (o->dualPsi[i] > o->umax[i]) o->dualPsi[i] = o->umax[i];
    }
    free(diff);
}

/* SmpcController::computeFixedPointResidual, SmpcController.cu:839-850 */
void oracle_residual(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    for (size_t i = 0; i < nxi; i++) o->resXi[i] = o->primalXi[i] - o->dualXi[i];
    for (size_t i = 0; i < nps; i++) o->resPsi[i] = o->primalPsi[i] - o->dualPsi[i];
}

/* SmpcController::dualUpdate, SmpcController.cu:854-881 (APG branch :859-864, FBE / NAMA branch :866-880) */
void oracle_dual_update(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    if (o->algorithm == 0) {
        for (size_t i = 0; i < nxi; i++) o->updXi[i] = o->accXi[i] + o->stepSize * o->resXi[i];
        for (size_t i = 0; i < nps; i++) o->updPsi[i] = o->accPsi[i] + o->stepSize * o->resPsi[i];
        return;
    }
    real *curYXi = o->algorithm == 1 ? o->gradXi : o->curResXi, *curYPsi = o->algorithm == 1 ? o->gradPsi : o->curResPsi;
    real *prvYXi = o->algorithm == 1 ? o->prevGradXi : o->prevResXi, *prvYPsi = o->algorithm == 1 ? o->prevGradPsi : o->prevResPsi;
    memcpy(prvYXi, curYXi, nxi * sizeof(real)); memcpy(prvYPsi, curYPsi, nps * sizeof(real));      /* :868-869 */
    memcpy(o->prevXi, o->xi, nxi * sizeof(real)); memcpy(o->prevPsi, o->psi, nps * sizeof(real));  /* :871-872 */
    for (size_t i = 0; i < nxi; i++) o->xi[i] = o->accXi[i] + o->stepSize * o->resXi[i];           /* :874-877 */
    for (size_t i = 0; i < nps; i++) o->psi[i] = o->accPsi[i] + o->stepSize * o->resPsi[i];
    memcpy(o->accXi, o->xi, nxi * sizeof(real)); memcpy(o->accPsi, o->psi, nps * sizeof(real));    /* :879-880 */
}

/* SmpcController::updatePrimalInfeasibity, SmpcController.cu:1480-1496: isamax returns the FIRST index of
 * the maximum |.|; the value read back is the signed entry (reference quirk, kept). */
double oracle_primal_infeasibility(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    size_t ix = 0, ip = 0;
    for (size_t i = 1; i < nxi; i++) if (fabs((double)o->resXi[i]) > fabs((double)o->resXi[ix])) ix = i;
    for (size_t i = 1; i < nps; i++) if (fabs((double)o->resPsi[i]) > fabs((double)o->resPsi[ip])) ip = i;
    real a = o->resXi[ix], b = o->resPsi[ip];
    return (double)(a > b ? a : b);
}

/* SmpcController::algorithmApg, SmpcController.cu:1500-1525. hist (may be NULL) receives vecPrimalInfs. */
void oracle_apg(oracle_t *o, int maxIterations, double *hist) {
    real theta[2] = {1, 1};
    oracle_apg_reset(o);
    for (int it = 0; it < maxIterations; it++) {
        real lambda = theta[1] * (1 / theta[0] - 1);
        oracle_extrapolate(o, lambda);
        oracle_solve_step(o);
        oracle_prox(o);
        oracle_residual(o);
        oracle_dual_update(o);
        theta[0] = theta[1];
        theta[1] = (real)(0.5 * (sqrt(pow(theta[1], 4) + 4 * pow(theta[1], 2)) - pow(theta[1], 2)));
        double inf = oracle_primal_infeasibility(o);
        if (hist) hist[it] = inf;
    }
}

/* one APG iteration continuing from the current state (bench cpu_baseline leg) */
void oracle_apg_continue(oracle_t *o, int iters, double *theta01) {
    real theta[2] = {(real)theta01[0], (real)theta01[1]};
    for (int it = 0; it < iters; it++) {
        real lambda = theta[1] * (1 / theta[0] - 1);
        oracle_extrapolate(o, lambda);
        oracle_solve_step(o);
        oracle_prox(o);
        oracle_residual(o);
        oracle_dual_update(o);
        theta[0] = theta[1];
        theta[1] = (real)(0.5 * (sqrt(pow(theta[1], 4) + 4 * pow(theta[1], 2)) - pow(theta[1], 2)));
    }
    theta01[0] = theta[0]; theta01[1] = theta[1];
}


/* ======================================================================================================
 * global-FBE and NAMA outer loops.  Restated from SmpcController.cu with the
 * reference's exact control flow; pinned by src/test/testDataFiles/{smpcFbeTest,smpcNamaTest}.json through
 * tests/test_oracle_fbe_nama.py (the Python twin of TestSmpcController.cu:403-1040).
 * ====================================================================================================== */
void oracle_set_algorithm(oracle_t *o, int algorithm, int lbfgsBufferSize) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, n = nxi + nps;
    o->algorithm = algorithm;
    if (algorithm == 0 || o->matS) return;
    o->lbfgsSize = lbfgsBufferSize;
    o->prevXi = ralloc(nxi); o->prevPsi = ralloc(nps); o->gradXi = ralloc(nxi); o->gradPsi = ralloc(nps);
    o->prevGradXi = ralloc(nxi); o->prevGradPsi = ralloc(nps);
    o->curResXi = ralloc(nxi); o->curResPsi = ralloc(nps); o->prevResXi = ralloc(nxi); o->prevResPsi = ralloc(nps);
    o->dirXi = ralloc(nxi); o->dirPsi = ralloc(nps);
    o->xdir = ralloc((size_t)o->nodes * o->nx); o->udir = ralloc(nps); o->primalXiDir = ralloc(nxi); o->primalPsiDir = ralloc(nps);
    /* the reference allocates lbfgsSize columns but addresses columns 1..lbfgsSize (SmpcController.cu:1146, 256-257:
     * a device heap overrun); one spare column makes the same indexing safe here */
    o->matS = ralloc((size_t)(lbfgsBufferSize + 1) * n); o->matY = ralloc((size_t)(lbfgsBufferSize + 1) * n);
    o->rho = ralloc((size_t)lbfgsBufferSize + 1);
    o->lbfgsCol = 0; o->lbfgsMem = 0; o->lbfgsSkip = 0; o->lbfgsH = 1;
}

/* SmpcController::initialiseAlgorithm (FBE / NAMA parts, :436-449) + initaliseLbfgBuffer (:453-468) */
void oracle_fbe_reset(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, n = nxi + nps;
    oracle_apg_reset(o);
    memset(o->prevXi, 0, nxi * sizeof(real)); memset(o->prevPsi, 0, nps * sizeof(real));
    if (o->algorithm == 1) { memset(o->gradXi, 0, nxi * sizeof(real)); memset(o->gradPsi, 0, nps * sizeof(real)); }
    else { memset(o->resXi, 0, nxi * sizeof(real)); memset(o->resPsi, 0, nps * sizeof(real)); }
    o->lbfgsCol = 0; o->lbfgsMem = 0; o->lbfgsSkip = 0; o->lbfgsH = 1;
    memset(o->matS, 0, (size_t)(o->lbfgsSize + 1) * n * sizeof(real));
    memset(o->matY, 0, (size_t)(o->lbfgsSize + 1) * n * sizeof(real));
    memset(o->rho, 0, ((size_t)o->lbfgsSize + 1) * sizeof(real));
}

/* SmpcController::computeHessianOracalGlobalFbe, SmpcController.cu:884-1058: the sweep of solveStep with sigma = 0 and
 * no affine terms, applied to the direction (dXi, dPsi) -> xdir, udir, primalXiDir, primalPsiDir */
static void hessian_oracle(oracle_t *o, const real *dXi, const real *dPsi) {
    int nx = o->nx, nu = o->nu, nv = o->nv, N = o->N, nodes = o->nodes, K = o->K;
    real *tmpQ = ralloc((size_t)K * nx), *tmpR = ralloc((size_t)K * nv), *Lv = ralloc((size_t)K * nu);
    memset(o->sigma, 0, (size_t)nodes * nv * sizeof(real));                                        /* :907 */
    for (int k = N - 1; k > -1; k--) {
        int cum = o->nodesPerStageCumul[k], nk = o->nodesPerStage[k];
        for (int j = 0; j < nk; j++) {
            int i = cum + j;
            real *sig = o->sigma + (size_t)i * nv, *v = o->v + (size_t)i * nv;
            real *qj = o->q + (size_t)j * nx, *rj = o->r + (size_t)j * nv;
            const real *xi = dXi + (size_t)i * 2 * nx, *psi = dPsi + (size_t)i * nu;
            if (k < N - 1) for (int t = 0; t < nv; t++) sig[t] += rj[t];
            gemv_n(nv, nv, (real)-0.5, o->Omega + (size_t)o->opIdx[i] * nv * nv, nv, sig, 0, v);
            if (k < N - 1) gemv_n(nv, nx, 1, o->Theta + (size_t)o->opIdx[i] * nv * nx, nv, qj, 1, v);
            gemv_n(nv, nu, 1, o->Psi + (size_t)i * nv * nu, nv, psi, 1, v);
            gemv_n(nv, 2 * nx, 1, o->Phi + (size_t)i * nv * 2 * nx, nv, xi, 1, v);
            memcpy(rj, sig, nv * sizeof(real));
            gemv_n(nv, 2 * nx, 1, o->D + (size_t)i * nv * 2 * nx, nv, xi, 1, rj);
            gemv_n(nv, nu, 1, o->Ftil + (size_t)i * nv * nu, nv, psi, 1, rj);
            if (k < N - 1) gemv_n(nv, nx, 1, o->Gtil, nv, qj, 1, rj);
            const real *F = o->sysF + (size_t)i * 2 * nx * nx;
            for (int t = 0; t < nx; t++) {
                real s = F[(size_t)2 * nx * t + t] * xi[t] + F[(size_t)2 * nx * t + nx + t] * xi[nx + t];
                qj[t] = (k < N - 1) ? s + qj[t] : s;
            }
        }
        if (k > 0) {
            int pn = o->nodesPerStage[k - 1], pc = o->nodesPerStageCumul[k - 1];
            if (nk - pn > 0) {
                sum_children(o, o->q, tmpQ, pc, pn, k - 1, nx);
                sum_children(o, o->r, tmpR, pc, pn, k - 1, nv);
                memcpy(o->r, tmpR, (size_t)pn * nv * sizeof(real));
                memcpy(o->q, tmpQ, (size_t)pn * nx * sizeof(real));
            }
        }
    }
    for (int k = 0; k < N; k++) {                                                                  /* :998-1046 */
        int cum = o->nodesPerStageCumul[k], nk = o->nodesPerStage[k];
        if (k == 0) {
            gemv_n(nu, nv, 1, o->L, nu, o->v, 0, o->udir);
            gemv_n(nx, nu, 1, o->B, nx, o->udir, 0, o->xdir);
        } else {
            int pc = o->nodesPerStageCumul[k - 1], pn = o->nodesPerStage[k - 1];
            for (int j = 0; j < nk; j++) gemv_n(nu, nv, 1, o->L, nu, o->v + (size_t)(cum + j) * nv, 0, o->udir + (size_t)(cum + j) * nu);
            if (nk - pn > 0) {
                int prevAnc = o->ancestor[cum];
                for (int j = 0; j < nk; j++) {
                    int a = o->ancestor[cum + j] - prevAnc;
                    for (int t = 0; t < nu; t++) o->udir[(size_t)(cum + j) * nu + t] += o->udir[(size_t)(pc + a) * nu + t];
                }
                for (int j = 0; j < nk; j++) gemv_n(nx, nu, 1, o->B, nx, o->udir + (size_t)(cum + j) * nu, 0, o->xdir + (size_t)(cum + j) * nx);
                for (int j = 0; j < nk; j++) {
                    int a = o->ancestor[cum + j] - prevAnc;
                    for (int t = 0; t < nx; t++) o->xdir[(size_t)(cum + j) * nx + t] += o->xdir[(size_t)(pc + a) * nx + t];
                }
            } else {
                for (size_t t = 0; t < (size_t)nk * nu; t++) o->udir[(size_t)cum * nu + t] += o->udir[(size_t)pc * nu + t];
                memcpy(o->xdir + (size_t)cum * nx, o->xdir + (size_t)pc * nx, (size_t)nk * nx * sizeof(real));
                for (int j = 0; j < nk; j++) gemv_n(nx, nu, 1, o->B, nx, o->udir + (size_t)(cum + j) * nu, 1, o->xdir + (size_t)(cum + j) * nx);
            }
        }
    }
    for (int i = 0; i < nodes; i++) {                                                              /* :1049-1052 */
        const real *F = o->sysF + (size_t)i * 2 * nx * nx, *G = o->sysG + (size_t)i * nu * nu;
        for (int t = 0; t < nx; t++) {
            o->primalXiDir[(size_t)i * 2 * nx + t] = F[(size_t)2 * nx * t + t] * o->xdir[(size_t)i * nx + t];
            o->primalXiDir[(size_t)i * 2 * nx + nx + t] = F[(size_t)2 * nx * t + nx + t] * o->xdir[(size_t)i * nx + t];
        }
        for (int t = 0; t < nu; t++) o->primalPsiDir[(size_t)i * nu + t] = G[(size_t)nu * t + t] * o->udir[(size_t)i * nu + t];
    }
    free(tmpQ); free(tmpR); free(Lv);
}
/* the oracle's input is whatever devPtrVecHessianOracleXi/Psi point at: the FBE gradient (:356-357) or, for NAMA,
 * the fixed-point residual (:408-409) */
void oracle_hessian_oracle(oracle_t *o) {
    if (o->algorithm == 2) hessian_oracle(o, o->resXi, o->resPsi);
    else hessian_oracle(o, o->gradXi, o->gradPsi);
}

/* SmpcController::computeGradientFbe, SmpcController.cu:1077-1097 */
void oracle_gradient_fbe(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    for (size_t i = 0; i < nxi; i++) o->gradXi[i] = -o->resXi[i];
    for (size_t i = 0; i < nps; i++) o->gradPsi[i] = -o->resPsi[i];
    hessian_oracle(o, o->gradXi, o->gradPsi);
    for (size_t i = 0; i < nxi; i++) o->gradXi[i] += o->stepSize * o->primalXiDir[i];
    for (size_t i = 0; i < nps; i++) o->gradPsi[i] += o->stepSize * o->primalPsiDir[i];
}
/* SmpcController::updateFixedPointResidualNamaAlgorithm, SmpcController.cu:1060-1072 */
void oracle_nama_residual(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    for (size_t i = 0; i < nxi; i++) o->curResXi[i] = -o->resXi[i];
    for (size_t i = 0; i < nps; i++) o->curResPsi[i] = -o->resPsi[i];
}

static double dot2(const real *a, const real *b, size_t n) { double s = 0; for (size_t i = 0; i < n; i++) s += (double)a[i] * (double)b[i]; return s; }

/* SmpcController::updateLbfgsBuffer, SmpcController.cu:1103-1169 */
static void lbfgs_update(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, n = nxi + nps;
    real *curYXi = o->algorithm == 1 ? o->gradXi : o->curResXi, *curYPsi = o->algorithm == 1 ? o->gradPsi : o->curResPsi;
    real *prvYXi = o->algorithm == 1 ? o->prevGradXi : o->prevResXi, *prvYPsi = o->algorithm == 1 ? o->prevGradPsi : o->prevResPsi;
    real *S = ralloc(n), *Y = ralloc(n);
    for (size_t i = 0; i < nxi; i++) { S[i] = o->xi[i] - o->prevXi[i]; Y[i] = curYXi[i] - prvYXi[i]; }
    for (size_t i = 0; i < nps; i++) { S[nxi + i] = o->psi[i] - o->prevPsi[i]; Y[nxi + i] = curYPsi[i] - prvYPsi[i]; }
    real normGrad = (real)sqrt(dot2(curYXi, curYXi, nxi) + dot2(curYPsi, curYPsi, nps));
    real invRho = (real)dot2(S, Y, n), normY = (real)sqrt(dot2(Y, Y, n)), normS = (real)sqrt(dot2(S, S, n));
    if (normGrad < 1) normGrad = normGrad * normGrad * normGrad;                                   /* :1133-1135 */
    if (invRho / (normS * normS) > (real)1e-6 * normGrad) {                                        /* :1137 */
        o->lbfgsCol = 1 + (o->lbfgsCol % o->lbfgsSize);
        o->lbfgsMem = o->lbfgsMem + 1 < o->lbfgsSize ? o->lbfgsMem + 1 : o->lbfgsSize;
        memcpy(o->matY + (size_t)o->lbfgsCol * n, Y, n * sizeof(real));
        memcpy(o->matS + (size_t)o->lbfgsCol * n, S, n * sizeof(real));
        o->rho[o->lbfgsCol] = 1 / invRho;
    } else o->lbfgsSkip++;
    real gammaH = invRho / (normY * normY);                                                        /* :1151-1156 */
    if (gammaH < 0 || fabs((double)(gammaH - o->lbfgsH)) == 0) o->lbfgsH = 1; else o->lbfgsH = gammaH;
    for (size_t i = 0; i < nxi; i++) o->dirXi[i] = -curYXi[i];                                     /* :1158-1161 */
    for (size_t i = 0; i < nps; i++) o->dirPsi[i] = -curYPsi[i];
    free(S); free(Y);
}
/* SmpcController::twoLoopRecursionLbfgs, SmpcController.cu:1175-1229 */
static void lbfgs_two_loop(oracle_t *o) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, n = nxi + nps;
    real *alpha = ralloc((size_t)o->lbfgsSize + 1);
    for (int is = 0; is < o->lbfgsMem; is++) {
        int c = o->lbfgsCol - is;
        if (c < 0) c = o->lbfgsMem + c;
        const real *Sc = o->matS + (size_t)c * n, *Yc = o->matY + (size_t)c * n;
        alpha[c] = o->rho[c] * (real)(dot2(Sc, o->dirXi, nxi) + dot2(Sc + nxi, o->dirPsi, nps));
        for (size_t i = 0; i < nxi; i++) o->dirXi[i] += -alpha[c] * Yc[i];
        for (size_t i = 0; i < nps; i++) o->dirPsi[i] += -alpha[c] * Yc[nxi + i];
    }
    for (size_t i = 0; i < nxi; i++) o->dirXi[i] *= o->lbfgsH;
    for (size_t i = 0; i < nps; i++) o->dirPsi[i] *= o->lbfgsH;
    for (int is = o->lbfgsMem; is > 0; is--) {
        int c = o->lbfgsCol - is + 1;
        if (c < 0) c = o->lbfgsMem + c;
        const real *Sc = o->matS + (size_t)c * n, *Yc = o->matY + (size_t)c * n;
        real beta = o->rho[c] * (real)(dot2(Yc, o->dirXi, nxi) + dot2(Yc + nxi, o->dirPsi, nps));
        real sc = alpha[c] - beta;
        for (size_t i = 0; i < nxi; i++) o->dirXi[i] += sc * Sc[i];
        for (size_t i = 0; i < nps; i++) o->dirPsi[i] += sc * Sc[nxi + i];
    }
    free(alpha);
}
/* SmpcController::computeLbfgsDirection, SmpcController.cu:1234-1237 */
void oracle_lbfgs_direction(oracle_t *o) { lbfgs_update(o); lbfgs_two_loop(o); }

/* SmpcController::computeValueFbe, SmpcController.cu:1416-1476 */
double oracle_value_fbe(oracle_t *o) {
    int nu = o->nu, nodes = o->nodes;
    size_t nxi = (size_t)nodes * 2 * o->nx, nps = (size_t)nodes * nu;
    real cost = (real)(dot2(o->accXi, o->resXi, nxi) + dot2(o->accPsi, o->resPsi, nps));
    real nX = (real)sqrt(dot2(o->resXi, o->resXi, nxi)), nP = (real)sqrt(dot2(o->resPsi, o->resPsi, nps));
    cost = cost + (real)0.5 * o->stepSize * (nX * nX + nP * nP);
    cost = cost + o->valueGuBox + o->valueGxBox + o->valueGxSafe;
    real *dU = ralloc(nps), *WdU = ralloc(nps);
    for (int i = 0; i < nodes; i++)                                    /* calculateDiffUhat on u (Utilities.cu:69-88) */
        for (int t = 0; t < nu; t++)
            dU[(size_t)i * nu + t] = (i == 0) ? o->u[t] - o->prevU[t] : o->u[(size_t)i * nu + t] - o->u[(size_t)(o->ancestor[i] - 1) * nu + t];
    for (int i = 0; i < nodes; i++) gemv_n(nu, nu, 1, o->W, nu, dU + (size_t)i * nu, 0, WdU + (size_t)i * nu);
    double quad = 0, lin = 0;
    for (int i = 0; i < nodes; i++)
        for (int t = 0; t < nu; t++) {
            quad += (double)(o->prob[i] * dU[(size_t)i * nu + t]) * (double)WdU[(size_t)i * nu + t];
            lin += (double)(o->prob[i] * o->u[(size_t)i * nu + t]) * (double)o->alpha[(size_t)i * nu + t];
        }
    cost = cost + (real)quad;
    cost = cost + (real)lin;
    free(dU); free(WdU);
    return (double)cost;
}

/* common tail of the two line searches (SmpcController.cu:1272-1300 and 1381-1409) */
static real line_search_loop(oracle_t *o, real valueY) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, nxx = (size_t)o->nodes * o->nx;
    real tau = 1;
    int maxStep = 10, iStep = 0;
    while (iStep < maxStep + 1) {
        for (size_t i = 0; i < nxx; i++) o->x[i] += tau * o->xdir[i];
        for (size_t i = 0; i < nps; i++) o->u[i] += tau * o->udir[i];
        for (size_t i = 0; i < nxi; i++) o->accXi[i] += tau * o->dirXi[i];
        for (size_t i = 0; i < nps; i++) o->accPsi[i] += tau * o->dirPsi[i];
        for (size_t i = 0; i < nxi; i++) o->primalXi[i] += tau * o->primalXiDir[i];
        for (size_t i = 0; i < nps; i++) o->primalPsi[i] += tau * o->primalPsiDir[i];
        oracle_prox(o);
        oracle_residual(o);
        real val = (real)oracle_value_fbe(o);
        if (val <= valueY) {
            iStep = iStep + 1;
            if (iStep < maxStep) {
                if (iStep == 1) tau = -1;
                tau = tau + (real)(1 / pow(2, iStep));
            }
        } else iStep = maxStep + 1;
    }
    return tau;
}
/* SmpcController::computeLineSearchLbfgsUpdate, SmpcController.cu:1242-1305 */
double oracle_line_search_fbe(oracle_t *o, double valueFbeY) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu;
    real tau = 1;
    hessian_oracle(o, o->dirXi, o->dirPsi);   /* the swap / oracle / swap of :1251-1255 */
    real valueDirection = (real)(dot2(o->gradXi, o->dirXi, nxi) + dot2(o->gradPsi, o->dirPsi, nps));
    if (valueDirection > 0) { /* "LBFGS direction is positive": nothing is applied, tau stays 1 (:1262-1263) */ }
    else if (fabs((double)valueDirection) < 1e-4) tau = 0;
    else tau = line_search_loop(o, (real)valueFbeY);
    return fabs((double)tau);
}
/* SmpcController::computeLineSearchAmeLbfgsUpdate, SmpcController.cu:1311-1414 */
double oracle_line_search_ame(oracle_t *o, double valueAmeY) {
    size_t nxi = (size_t)o->nodes * 2 * o->nx, nps = (size_t)o->nodes * o->nu, nxx = (size_t)o->nodes * o->nx;
    real tau = 1, alpha = o->stepSize;
    real valueDirection = -(real)(dot2(o->resXi, o->dirXi, nxi) + dot2(o->resPsi, o->dirPsi, nps));
    hessian_oracle(o, o->resXi, o->resPsi);                                                        /* :1331 */
    for (size_t i = 0; i < nxi; i++) o->accXi[i] += alpha * o->resXi[i];                           /* :1332-1337 */
    for (size_t i = 0; i < nps; i++) o->accPsi[i] += alpha * o->resPsi[i];
    for (size_t i = 0; i < nxx; i++) o->x[i] += alpha * o->xdir[i];
    for (size_t i = 0; i < nps; i++) o->u[i] += alpha * o->udir[i];
    for (size_t i = 0; i < nxi; i++) o->primalXi[i] += alpha * o->primalXiDir[i];
    for (size_t i = 0; i < nps; i++) o->primalPsi[i] += alpha * o->primalPsiDir[i];
    for (size_t i = 0; i < nxi; i++) o->dirXi[i] += -o->stepSize * o->resXi[i];                    /* :1338-1340 */
    for (size_t i = 0; i < nps; i++) o->dirPsi[i] += -o->stepSize * o->resPsi[i];
    hessian_oracle(o, o->dirXi, o->dirPsi);                                                        /* swap / oracle / swap :1341-1345 */
    if (valueDirection > 0) { }
    else if (fabs((double)valueDirection) < 1e-4) tau = 0;
    else tau = line_search_loop(o, (real)valueAmeY);
    return fabs((double)tau);
}

/* SmpcController::algorithmGlobalFbe (:1529-1555) / algorithmNama (:1559-1586).  hist: vecPrimalInfs[iter];
 * valueFbe / tau (may be NULL): vecValueFbe[iter-1], vecTau[iter-1] */
void oracle_fbe_nama(oracle_t *o, int maxIterations, double *hist, double *valueFbe, double *tauHist) {
    oracle_fbe_reset(o);
    for (int it = 0; it < maxIterations; it++) {
        oracle_solve_step(o);
        oracle_prox(o);
        oracle_residual(o);
        if (o->algorithm == 1) oracle_gradient_fbe(o); else oracle_nama_residual(o);
        if (it > 0) {
            double val = oracle_value_fbe(o);
            oracle_lbfgs_direction(o);
            double tau = o->algorithm == 1 ? oracle_line_search_fbe(o, val) : oracle_line_search_ame(o, val);
            if (valueFbe) valueFbe[it - 1] = val;
            if (tauHist) tauHist[it - 1] = tau;
        }
        oracle_dual_update(o);
        double inf = oracle_primal_infeasibility(o);
        if (hist) hist[it] = inf;
    }
}
void oracle_lbfgs_state(oracle_t *o, int set, int *col, int *mem, double *H) {
    if (set) { o->lbfgsCol = *col; o->lbfgsMem = *mem; o->lbfgsH = (real)*H; }
    else { *col = o->lbfgsCol; *mem = o->lbfgsMem; *H = (double)o->lbfgsH; }
}

/* ====================================================================================================
 * Closed loop around the solve
 * ==================================================================================================== */

/* SmpcController::controlAction(fstream&) SmpcController.cu:1633-1667 (project = 1): updateStateControl and
 * eliminateInputDistubanceCoupling are the caller's (oracle_update_state_control / oracle_eliminate), then
 * algorithmApg (:1646), devControlAction = devVecU[0..nu) (:1647), projectionBox<<<1,nu>>> with the SCALED bounds
 * of node 0 (:1649; Utilities.cu:237-254), copy to the host (:1650).
 * controlAction(real_t*) :1607-1626 (project = 0) returns devVecU[0..nu) as it is and leaves devControlAction alone. */
void oracle_control_action(oracle_t *o, int maxIterations, int project, double *uOut) {
    oracle_apg(o, maxIterations, NULL);
    if (!project) {
        for (int i = 0; i < o->nu; i++) uOut[i] = (double)o->u[i];
        return;
    }
    for (int i = 0; i < o->nu; i++) {
        real v = o->u[i];
        if (v < o->umin[i]) v = o->umin[i]; else if (v > o->umax[i]) v = o->umax[i];
        o->controlAction[i] = v;
        uOut[i] = (double)v;
    }
}

/* SmpcController::updateKpi SmpcController.cu:1778-1813.  previousControl is the configuration's prevU BEFORE it is
 * shifted (moveForewardInTime calls updateKpi first, :1706-1708); variablePrice = the forecaster's current nominal
 * prices (the first nu entries are used: stage 0); constantPrice = DwnNetwork::getAlpha() = costAlpha1. */
static void update_kpi(oracle_t *o, const real *state, const real *control, const real *safeX, const real *variablePrice,
                       real weightEconomic) {
    real ecoKpi = 0, smKpi = 0, saKpi = 0, netKpi = 0;
    for (int i = 0; i < o->nu; i++) {
        ecoKpi = ecoKpi + weightEconomic * (o->alpha1[i] + variablePrice[i]) * (real)fabs((double)control[i]);   /* :1795 */
        real deltaU = o->prevU[i] - control[i];                                                                    /* :1796 */
        smKpi = smKpi + deltaU * deltaU;                                                                           /* :1797 */
    }
    for (int i = 0; i < o->nx; i++) {
        real waterLevel = state[i] - safeX[i];                 /* :1800 */
        if (waterLevel > 0) waterLevel = 0;                    /* :1801-1803 */
        saKpi = saKpi + (real)fabs((double)waterLevel);        /* :1805 */
        netKpi = netKpi + (real)fabs((double)state[i]);        /* :1806 */
    }
    o->economicKpi += ecoKpi; o->smoothKpi += smKpi; o->safeKpi += saKpi; o->networkKpi += netKpi;   /* :1809-1812 */
}

/* SmpcController::moveForewardInTime, in-built simulator branch (simulatorFlag = 1), SmpcController.cu:1680-1711.
 * plantMode 0 = the reference as written: devStateUpdate = currentX (:1692); the disturbance is added to devVecX -- the
 *   x of node 0 -- instead of the state update (:1695, a slip that only changes that output buffer); devStateUpdate +=
 *   B devControlAction (:1697-1698): the simulated plant is x+ = x + B u.
 * plantMode 1 = the plant of DwnNetwork.cuh:41-57 with the disturbance of node 0, x+ = x + e_0 + B u (what :1694's
 *   comment says it computes); devVecX is left alone.
 * Then updateKpi (:1706), setCurrentState / setPreviousControl / setpreviousdemand (:1707-1709) with
 * previousDemand = the forecaster's nominal demand (first nd entries).  The shifted triple is returned through xOut,
 * uOut, dOut and installed as the oracle's current state / previous control / previous demand. */
void oracle_move_forward(oracle_t *o, const double *nominalDemand, const double *nominalPrices, const double *xsafe,
                         double weightEconomic, int plantMode, double *xOut, double *uOut, double *dOut) {
    const int nx = o->nx, nu = o->nu, nd = o->nd;
    for (int i = 0; i < nx; i++) o->stateUpdate[i] = o->curX[i];
    if (plantMode == 0) { for (int i = 0; i < nx; i++) o->x[i] += o->e[i]; }
    else { for (int i = 0; i < nx; i++) o->stateUpdate[i] += o->e[i]; }
    gemv_n(nx, nu, 1, o->B, nx, o->controlAction, 1, o->stateUpdate);
    real *safeX = ralloc(nx), *price = ralloc(nu);
    for (int i = 0; i < nx; i++) safeX[i] = (real)xsafe[i];
    for (int i = 0; i < nu; i++) price[i] = (real)nominalPrices[i];
    update_kpi(o, o->stateUpdate, o->controlAction, safeX, price, (real)weightEconomic);
    free(safeX); free(price);
    for (int i = 0; i < nx; i++) { o->curX[i] = o->stateUpdate[i]; xOut[i] = (double)o->curX[i]; }
    for (int i = 0; i < nu; i++) { o->prevU[i] = o->controlAction[i]; uOut[i] = (double)o->prevU[i]; }
    for (int i = 0; i < nd; i++) { o->prevD[i] = (real)nominalDemand[i]; dOut[i] = (double)o->prevD[i]; }
}

/* get{Economic,Smooth,Network,Safety}Kpi SmpcController.cu:1819-1859; which = 0 economic, 1 smooth, 2 network, 3 safety */
double oracle_kpi(const oracle_t *o, int which, int simulationTime, const double *xsafe) {
    if (which == 0) { real v = o->economicKpi / 3600; return (double)(v / simulationTime); }      /* :1819-1822 */
    if (which == 1) { real v = o->smoothKpi / 3600; return (double)(v / simulationTime); }        /* :1828-1831 */
    if (which == 2) {                                                                              /* :1837-1846 */
        real safeLevelNorm = 0;
        for (int i = 0; i < o->nx; i++) safeLevelNorm = safeLevelNorm + (real)xsafe[i];
        return (double)(100 * simulationTime * safeLevelNorm / o->networkKpi);
    }
    return (double)o->safeKpi;                                                                     /* :1852-1854 */
}

int oracle_sizeof_real(void) { return (int)sizeof(real); }
int oracle_final_branch_node(const oracle_t *o) { return o->finalBranchNode; }
double oracle_dist(const oracle_t *o, int which) { return which ? (double)o->distXs : (double)o->distXcst; }

/* raw buffer access for the numpy wrapper: returns pointer and element count */
real *oracle_buffer(oracle_t *o, const char *name, long *count) {
    size_t n = o->nodes; int nx = o->nx, nu = o->nu, nv = o->nv;
#define BUF(nm, ptr, cnt) if (!strcmp(name, nm)) { *count = (long)(cnt); return ptr; }
    BUF("x", o->x, n * nx) BUF("u", o->u, n * nu) BUF("v", o->v, n * nv)
    BUF("xi", o->xi, n * 2 * nx) BUF("psi", o->psi, n * nu)
    BUF("accXi", o->accXi, n * 2 * nx) BUF("accPsi", o->accPsi, n * nu)
    BUF("updXi", o->updXi, n * 2 * nx) BUF("updPsi", o->updPsi, n * nu)
    BUF("primalXi", o->primalXi, n * 2 * nx) BUF("primalPsi", o->primalPsi, n * nu)
    BUF("dualXi", o->dualXi, n * 2 * nx) BUF("dualPsi", o->dualPsi, n * nu)
    BUF("resXi", o->resXi, n * 2 * nx) BUF("resPsi", o->resPsi, n * nu)
    BUF("uhat", o->uhat, n * nu) BUF("e", o->e, n * nx) BUF("beta", o->beta, n * nv) BUF("alpha", o->alpha, n * nu)
    BUF("sigma", o->sigma, n * nv) BUF("q", o->q, (size_t)o->K * nx) BUF("r", o->r, (size_t)o->K * nv)
    BUF("sysF", o->sysF, n * 2 * nx * nx) BUF("sysG", o->sysG, n * nu * nu)
    BUF("xmin", o->xmin, n * nx) BUF("xmax", o->xmax, n * nx) BUF("xs", o->xs, n * nx)
    BUF("umin", o->umin, n * nu) BUF("umax", o->umax, n * nu)
    BUF("Omega", o->Omega, (size_t)o->finalBranchNode * nv * nv) BUF("Theta", o->Theta, (size_t)o->finalBranchNode * nv * nx)
    BUF("Phi", o->Phi, n * nv * 2 * nx) BUF("D", o->D, n * nv * 2 * nx)
    BUF("Psi", o->Psi, n * nv * nu) BUF("Ftil", o->Ftil, n * nv * nu) BUF("Gtil", o->Gtil, (size_t)nv * nx)
    BUF("prevUhat", o->prevUhat, nu) BUF("curX", o->curX, nx) BUF("prevU", o->prevU, nu) BUF("prevD", o->prevD, o->nd)
    BUF("controlAction", o->controlAction, nu) BUF("stateUpdate", o->stateUpdate, nx)
    if (o->matS) {
        size_t nall = n * (2 * nx + nu);
        BUF("prevXi", o->prevXi, n * 2 * nx) BUF("prevPsi", o->prevPsi, n * nu)
        BUF("gradXi", o->gradXi, n * 2 * nx) BUF("gradPsi", o->gradPsi, n * nu)
        BUF("prevGradXi", o->prevGradXi, n * 2 * nx) BUF("prevGradPsi", o->prevGradPsi, n * nu)
        BUF("curResXi", o->curResXi, n * 2 * nx) BUF("curResPsi", o->curResPsi, n * nu)
        BUF("prevResXi", o->prevResXi, n * 2 * nx) BUF("prevResPsi", o->prevResPsi, n * nu)
        BUF("dirXi", o->dirXi, n * 2 * nx) BUF("dirPsi", o->dirPsi, n * nu)
        BUF("xdir", o->xdir, n * nx) BUF("udir", o->udir, n * nu)
        BUF("primalXiDir", o->primalXiDir, n * 2 * nx) BUF("primalPsiDir", o->primalPsiDir, n * nu)
        BUF("matS", o->matS, (size_t)(o->lbfgsSize + 1) * nall) BUF("matY", o->matY, (size_t)(o->lbfgsSize + 1) * nall)
        BUF("rho", o->rho, (size_t)o->lbfgsSize + 1)
    }
    BUF("L", o->L, (size_t)nu * nv) BUF("B", o->B, (size_t)nx * nu) BUF("Wv", o->Wv, (size_t)nu * nv)
#undef BUF
    *count = 0;
    return NULL;
}
