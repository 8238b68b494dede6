// test_host.cpp -- C++ tests of the host classes, written the way the reference tests itself:
//   loaders     : Testing::testNetwork / testScenarioTree / testForecaster / testControllerConfig  (src/test/Testing.cu:78-335)
//   engine      : Testing::testEngineTesting (Testing.cu:340-477) against engineTest.json
//   controller  : TestSmpcController::testExtrapolation / testSoveStep / testProximalStep / testFixedPointResidual /
//                 testDualUpdate (src/test/TestSmpcController.cu:114-420) against smpcTest.json, same tolerance rule
//   fbe / nama  : Testing::testSmpcFbeController / testSmpcNamaController (Testing.cu:536-590) against smpcFbeTest.json /
//                 smpcNamaTest.json
//   sharded     : (new capability, no reference counterpart) N rank-local controllers of ONE problem, created through the sharded
//                 constructors (rank, nranks) of SmpcController / Engine, one host thread each, on one GPU, against the unsharded controller
// usage: test_host <loaders|engine|controller|fbe|nama|closedloop|nullspace|warmstart|sharded> <directory with the fixture JSON files> [ranks]
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <iostream>
#include <thread>

#include "../../rapidnet_amd/csrc/host/SmpcController.hpp"
#include "../../include/rapidnet_debug.h"   // test hooks (leak injection, in-process communicator stand-in)

static int g_failures = 0;
static int g_ops = -1;   // operator mode of the controllers under test (-1: the default of the class surface)
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) { std::cerr << "FAILED " << #cond << " (" << __FILE__ << ":" << __LINE__ << ")\n"; g_failures++; } \
    } while (0)

// TestSmpcController::compareDeviceArray (TestSmpcController.cu:26-44): abs 1e-1, or 0.1 % relative above 100
static bool closeRef(const real_t *a, const jsonlite::Value &ref, size_t n, const char *what) {
    if (ref.Size() != n) { std::cerr << what << ": size " << ref.Size() << " vs " << n << "\n"; return false; }
    for (size_t i = 0; i < n; i++) {
        double v = a[i] - ref[i];
        if (std::fabs(a[i]) > 1e2) v = v / a[i] * 100;
        if (!(std::fabs(v) < 1e-1)) { std::cerr << what << "[" << i << "] = " << a[i] << " vs " << ref[i] << "\n"; return false; }
    }
    return true;
}
static bool closeAbs(const real_t *a, const double *ref, size_t n, double tol, const char *what) {
    for (size_t i = 0; i < n; i++)
        if (!(std::fabs(a[i] - ref[i]) < tol)) { std::cerr << what << "[" << i << "] = " << a[i] << " vs " << ref[i] << "\n"; return false; }
    return true;
}
// max-norm relative agreement of two vectors
static bool closeRel(const real_t *a, const real_t *b, size_t n, double tol, const char *what) {
    double d = 0, m = 1e-300;
    for (size_t i = 0; i < n; i++) { d = std::max(d, std::fabs((double)a[i] - (double)b[i])); m = std::max(m, std::fabs((double)b[i])); }
    if (!(d <= tol * m)) { std::cerr << what << ": max |a - b| = " << d << " against max |b| = " << m << "\n"; return false; }
    return true;
}
template <typename T> static bool sameAsJson(T *arr, const jsonlite::Value &ref, const char *what) {
    for (size_t i = 0; i < ref.Size(); i++)
        if (!(std::fabs((double)arr[i] - ref[i]) < 1e-2)) { std::cerr << what << "[" << i << "]\n"; return false; }  // Testing.cu:36
    return true;
}

static void testLoaders(const string &dir) {
    {   // Testing::testNetwork
        DwnNetwork net(dir + "/network.json");
        jsonlite::Document j(dir + "/network.json");
        CHECK(net.getNumTanks() == (uint_t)j["nx"][0] && net.getNumControls() == (uint_t)j["nu"][0]);
        CHECK(net.getNumDemands() == (uint_t)j["nd"][0] && net.getNumMixNodes() == (uint_t)j["ne"][0]);
        CHECK(sameAsJson(net.getMatA(), j["matA"], "matA")); CHECK(sameAsJson(net.getMatB(), j["matB"], "matB"));
        CHECK(sameAsJson(net.getMatGd(), j["matGd"], "matGd")); CHECK(sameAsJson(net.getMatE(), j["matE"], "matE"));
        CHECK(sameAsJson(net.getMatEd(), j["matEd"], "matEd")); CHECK(sameAsJson(net.getXmin(), j["vecXmin"], "vecXmin"));
        CHECK(sameAsJson(net.getXmax(), j["vecXmax"], "vecXmax")); CHECK(sameAsJson(net.getXsafe(), j["vecXsafe"], "vecXsafe"));
        CHECK(sameAsJson(net.getUmin(), j["vecUmin"], "vecUmin")); CHECK(sameAsJson(net.getUmax(), j["vecUmax"], "vecUmax"));
        CHECK(sameAsJson(net.getAlpha(), j["costAlpha1"], "costAlpha1"));
    }
    {   // Testing::testScenarioTree
        ScenarioTree tree(dir + "/scenarioTree.json");
        jsonlite::Document j(dir + "/scenarioTree.json");
        CHECK(tree.getPredHorizon() == (uint_t)j["N"][0] && tree.getNumScenarios() == (uint_t)j["K"][0]);
        CHECK(tree.getNumNodes() == (uint_t)j["nodes"][0] && tree.getNumNonleafNodes() == (uint_t)j["nNonLeafNodes"][0]);
        CHECK(tree.getNumChildrenTot() == (uint_t)j["nChildrenTot"][0]);
        CHECK(sameAsJson(tree.getStageNodes(), j["stages"], "stages"));
        CHECK(sameAsJson(tree.getNodesPerStage(), j["nodesPerStage"], "nodesPerStage"));
        CHECK(sameAsJson(tree.getNodesPerStageCumul(), j["nodesPerStageCumul"], "nodesPerStageCumul"));
        CHECK(sameAsJson(tree.getLeaveArray(), j["leaves"], "leaves")); CHECK(sameAsJson(tree.getChildArray(), j["children"], "children"));
        CHECK(sameAsJson(tree.getAncestorArray(), j["ancestor"], "ancestor"));
        CHECK(sameAsJson(tree.getNumChildren(), j["nChildren"], "nChildren"));
        CHECK(sameAsJson(tree.getNumChildrenCumul(), j["nChildrenCumul"], "nChildrenCumul"));
        CHECK(sameAsJson(tree.getProbArray(), j["probNode"], "probNode"));
        CHECK(sameAsJson(tree.getErrorDemandArray(), j["errorDemandNode"], "errorDemandNode"));
        CHECK(sameAsJson(tree.getErrorPriceArray(), j["errorPriceNode"], "errorPriceNode"));
        std::cout << "tree: N " << tree.getPredHorizon() << " K " << tree.getNumScenarios() << " nodes " << tree.getNumNodes()
                  << " finalBranchNode " << tree.getFinalBranchNode() << " finalBranchStage " << tree.getFinalBranchStage() << "\n";
    }
    {   // Testing::testForecaster: member order 4+2t / 5+2t
        Forecaster fc(dir + "/forecastor.json");
        jsonlite::Document j(dir + "/forecastor.json");
        CHECK(fc.getPredHorizon() == (uint_t)j["N"][0] && fc.getSimHorizon() == (uint_t)j["simHorizon"][0]);
        CHECK(fc.getDimDemand() == (uint_t)j["dimDemand"][0] && fc.getDimPrice() == (uint_t)j["dimPrices"][0]);
        for (uint_t t = 0; t < 2; t++) {
            CHECK(fc.predictDemand(t) == 1 && fc.predictPrices(t) == 1);
            CHECK(sameAsJson(fc.getNominalDemand(), j.MemberValue(4 + 2 * t), "nominalDemand"));
            CHECK(sameAsJson(fc.getNominalPrices(), j.MemberValue(5 + 2 * t), "nominalPrices"));
        }
        CHECK(fc.predictDemand(100000) == 0);
    }
    {   // Testing::testControllerConfig
        SmpcConfiguration cfg(dir + "/controllerConfig.json");
        jsonlite::Document j(dir + "/controllerConfig.json");
        CHECK(cfg.getNX() == (uint_t)j["nx"][0] && cfg.getNU() == (uint_t)j["nu"][0] && cfg.getND() == (uint_t)j["nd"][0] && cfg.getNV() == (uint_t)j["nv"][0]);
        CHECK(sameAsJson(cfg.getMatL(), j["matL"], "matL")); CHECK(sameAsJson(cfg.getMatLhat(), j["matLhat"], "matLhat"));
        CHECK(sameAsJson(cfg.getCostW(), j["costW"], "costW")); CHECK(sameAsJson(cfg.getMatPrcndDiag(), j["matDiagPrecnd"], "matDiagPrecnd"));
        CHECK(sameAsJson(cfg.getCurrentX(), j["currentX"], "currentX")); CHECK(sameAsJson(cfg.getPrevU(), j["prevU"], "prevU"));
        CHECK(sameAsJson(cfg.getPrevDemand(), j["prevDemand"], "prevDemand"));
        CHECK(std::fabs(cfg.getStepSize() - j["stepSize"][0]) < 1e-12 && cfg.getMaxIterations() == (uint_t)j["maxIterations"][0]);
        CHECK(cfg.getPenaltyState() == j["penaltyStateX"][0] && cfg.getPenaltySafety() == j["penaltySafetyX"][0]);
        CHECK(cfg.getOptimisationAlgorithm() == j["algorithmName"].str);
        std::vector<real_t> x(cfg.getNX(), 7.0);
        cfg.setCurrentState(x.data()); CHECK(cfg.getCurrentX()[0] == 7.0);
        cfg.setCurrentState(); CHECK(std::fabs(cfg.getCurrentX()[0] - j["currentX"][0]) < 1e-9);
        cfg.setPreviousDemand(); CHECK(std::fabs(cfg.getPrevDemand()[0] - j["prevDemand"][0]) < 1e-9);   // reference bug fixed
        CHECK(std::fabs(cfg.getPrevU()[0] - j["prevU"][0]) < 1e-9);
    }
    bool threw = false;
    try { DwnNetwork missing(dir + "/does_not_exist.json"); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
}

// reaches the protected step methods exactly as the reference's TestSmpcController does (TestSmpcController.cuh:80)
class TestSmpcController : public SmpcController {
public:
    explicit TestSmpcController(const string &cfg) : SmpcController(cfg, g_ops) {}
    void run(const string &dir) {
        jsonlite::Document j(dir + "/smpcTest.json");
        const uint_t nx = getDwnNetwork()->getNumTanks(), nu = getDwnNetwork()->getNumControls(), nodes = getScenarioTree()->getNumNodes();
        const size_t nxi = (size_t)2 * nx * nodes, nps = (size_t)nu * nodes;
        std::vector<real_t> a(nxi), b(nps), c(nodes * (size_t)nx);
        // testExtrapolation (:114)
        setVector(RN_BUF_XI, j["xi"].arr.data()); setVector(RN_BUF_PSI, j["psi"].arr.data());
        setVector(RN_BUF_UPD_XI, j["updateXi"].arr.data()); setVector(RN_BUF_UPD_PSI, j["updatePsi"].arr.data());
        dualExtrapolationStep(j["theta"][1] * (1 / j["theta"][0] - 1));
        getVector(RN_BUF_ACC_XI, a.data()); CHECK(closeRef(a.data(), j["acceleXi"], nxi, "acceleXi"));
        getVector(RN_BUF_ACC_PSI, b.data()); CHECK(closeRef(b.data(), j["accelePsi"], nps, "accelePsi"));
        getVector(RN_BUF_XI, a.data()); CHECK(closeRef(a.data(), j["finalXi"], nxi, "finalXi"));
        getVector(RN_BUF_PSI, b.data()); CHECK(closeRef(b.data(), j["finalPsi"], nps, "finalPsi"));
        // testSoveStep (:173)
        setVector(RN_BUF_ACC_XI, j["acceleXi"].arr.data()); setVector(RN_BUF_ACC_PSI, j["accelePsi"].arr.data());
        solveStep();
        getVector(RN_BUF_X, c.data()); CHECK(closeRef(c.data(), j["X"], c.size(), "X"));
        getVector(RN_BUF_U, b.data()); CHECK(closeRef(b.data(), j["U"], nps, "U"));
        // testProximalStep (:221)
        proximalFunG();
        getVector(RN_BUF_PRIMAL_XI, a.data()); CHECK(closeRef(a.data(), j["primalX"], nxi, "primalX"));
        getVector(RN_BUF_PRIMAL_PSI, b.data()); CHECK(closeRef(b.data(), j["primalU"], nps, "primalU"));
        getVector(RN_BUF_DUAL_XI, a.data()); CHECK(closeRef(a.data(), j["dualX"], nxi, "dualX"));
        getVector(RN_BUF_DUAL_PSI, b.data()); CHECK(closeRef(b.data(), j["dualU"], nps, "dualU"));
        // testFixedPointResidual (:345)
        setVector(RN_BUF_PRIMAL_XI, j["primalX"].arr.data()); setVector(RN_BUF_PRIMAL_PSI, j["primalU"].arr.data());
        setVector(RN_BUF_DUAL_XI, j["dualX"].arr.data()); setVector(RN_BUF_DUAL_PSI, j["dualU"].arr.data());
        computeFixedPointResidual();
        getVector(RN_BUF_RES_XI, a.data()); CHECK(closeRef(a.data(), j["fixedPointResidualXi"], nxi, "fixedPointResidualXi"));
        getVector(RN_BUF_RES_PSI, b.data()); CHECK(closeRef(b.data(), j["fixedPointResidualPsi"], nps, "fixedPointResidualPsi"));
        // testDualUpdate (:291)
        setVector(RN_BUF_ACC_XI, j["acceleXi"].arr.data()); setVector(RN_BUF_ACC_PSI, j["accelePsi"].arr.data());
        setVector(RN_BUF_RES_XI, j["fixedPointResidualXi"].arr.data()); setVector(RN_BUF_RES_PSI, j["fixedPointResidualPsi"].arr.data());
        dualUpdate();
        getVector(RN_BUF_UPD_XI, a.data()); CHECK(closeRef(a.data(), j["finalUpdateXi"], nxi, "finalUpdateXi"));
        getVector(RN_BUF_UPD_PSI, b.data()); CHECK(closeRef(b.data(), j["finalUpdatePsi"], nps, "finalUpdatePsi"));
        // whole algorithm runs and produces a finite, decreasing infeasibility history
        CHECK(algorithmApg() == 1);
        const real_t *h = getPrimalInfeasibility();
        CHECK(std::isfinite(h[0]) && std::isfinite(h[getSmpcConfiguration()->getMaxIterations() - 1]));
        CHECK(std::fabs(h[getSmpcConfiguration()->getMaxIterations() - 1]) < std::fabs(h[0]));
        std::cout << "primal infeasibility: first " << h[0] << " last " << h[getSmpcConfiguration()->getMaxIterations() - 1] << "\n";
    }
};

// Testing::testSmpcFbeController / testSmpcNamaController (Testing.cu:536-590): the FBE / NAMA known-answer tests of
// TestSmpcController.cu:403-1040 against smpcFbeTest.json / smpcNamaTest.json, same tolerance rule, same order
class TestFbeNamaController : public SmpcController {
public:
    explicit TestFbeNamaController(const string &cfg) : SmpcController(cfg, g_ops) {}
    void run(const string &dir) {
        const bool fbe = getEngine()->getGlobalFbeFlag();
        CHECK(fbe != getEngine()->getNamaFlag() && !getEngine()->getApgFlag());
        jsonlite::Document j(dir + (fbe ? "/smpcFbeTest.json" : "/smpcNamaTest.json"));
        const uint_t nx = getDwnNetwork()->getNumTanks(), nu = getDwnNetwork()->getNumControls(), nodes = getScenarioTree()->getNumNodes();
        const size_t nxi = (size_t)2 * nx * nodes, nps = (size_t)nu * nodes, nall = nxi + nps;
        const int m = (int)getSmpcConfiguration()->getLbfgsBufferSize();
        std::vector<real_t> a(nxi), b(nps), c(nodes * (size_t)nx), col(nall);
        auto put = [&](int id, const char *key) { setVector(id, j[key].arr.data()); };
        // testHessianOracalGlobalFbe (:403)
        put(fbe ? RN_BUF_LBFGS_CUR_YVEC_XI : RN_BUF_RES_XI, "fixedPointResidualXi");
        put(fbe ? RN_BUF_LBFGS_CUR_YVEC_PSI : RN_BUF_RES_PSI, "fixedPointResidualPsi");
        computeHessianOracalGlobalFbe();
        getVector(RN_BUF_UDIR, b.data()); CHECK(closeRef(b.data(), j[fbe ? "fbeHessianDirUdir" : "ameFixedPointDirUdir"], nps, "Udir"));
        getVector(RN_BUF_XDIR, c.data()); CHECK(closeRef(c.data(), j[fbe ? "fbeHessianDirXdir" : "ameFixedPointDirXdir"], c.size(), "Xdir"));
        if (fbe) {   // testFbeGradient (:458)
            put(RN_BUF_RES_XI, "fixedPointResidualXi"); put(RN_BUF_RES_PSI, "fixedPointResidualPsi");
            computeGradientFbe();
            getVector(RN_BUF_LBFGS_CUR_YVEC_XI, a.data()); CHECK(closeRef(a.data(), j["fbeGradXi"], nxi, "fbeGradXi"));
            getVector(RN_BUF_LBFGS_CUR_YVEC_PSI, b.data()); CHECK(closeRef(b.data(), j["fbeGradPsi"], nps, "fbeGradPsi"));
        }
        {   // testValueFbe (:683)
            put(RN_BUF_RES_XI, "fixedPointResidualXi"); put(RN_BUF_RES_PSI, "fixedPointResidualPsi");
            put(RN_BUF_ACC_XI, "acceleXi"); put(RN_BUF_ACC_PSI, "accelePsi"); put(RN_BUF_U, "U");
            const real_t v = computeValueFbe();
            CHECK(std::fabs((v - j["fbeObjDual"][0]) / j["fbeObjDual"][0] * 100) < 1e-1);
        }
        if (!fbe) {  // testUpdateFixedPointResidualNamaAlgorithm (:633)
            put(RN_BUF_RES_XI, "fixedPointResidualXi"); put(RN_BUF_RES_PSI, "fixedPointResidualPsi");
            updateFixedPointResidualNamaAlgorithm();
            getVector(RN_BUF_LBFGS_CUR_YVEC_XI, a.data()); CHECK(closeRef(a.data(), j["lbfgsCurrentYvecXi"], nxi, "lbfgsCurrentYvecXi"));
            getVector(RN_BUF_LBFGS_CUR_YVEC_PSI, b.data()); CHECK(closeRef(b.data(), j["lbfgsCurrentYvecPsi"], nps, "lbfgsCurrentYvecPsi"));
        }
        {   // testLbfgsDirection (:503)
            put(RN_BUF_PREV_XI, "xi"); put(RN_BUF_PREV_PSI, "psi"); put(RN_BUF_XI, "acceleXi"); put(RN_BUF_PSI, "accelePsi");
            put(RN_BUF_LBFGS_CUR_YVEC_XI, "lbfgsCurrentYvecXi"); put(RN_BUF_LBFGS_CUR_YVEC_PSI, "lbfgsCurrentYvecPsi");
            put(RN_BUF_LBFGS_PREV_YVEC_XI, "lbfgsPreviousYvecXi"); put(RN_BUF_LBFGS_PREV_YVEC_PSI, "lbfgsPreviousYvecPsi");
            CHECK(j["matS"].Size() == nall * m && j["matY"].Size() == nall * m);
            for (int k = 0; k < m; k++) {
                setLbfgsColumn(0, k, j["matS"].arr.data() + (size_t)k * nall, nall);
                setLbfgsColumn(1, k, j["matY"].arr.data() + (size_t)k * nall, nall);
            }
            std::vector<real_t> rho(m + 1, 0.0);
            for (int k = 0; k < m; k++) rho[k] = j["vecInvRho"][k] != 0 ? 1 / j["vecInvRho"][k] : 0;
            setLbfgsState((int)j["colLbfgs"][0], (int)j["memLbfgs"][0], j["H"][0], rho.data());
            computeLbfgsDirection();
            int colNow, memNow; real_t H;
            getLbfgsState(colNow, memNow, H, rho.data());
            CHECK(std::fabs(H - j["updateH"][0]) < 1e-1 && colNow == (int)j["updateColLbfgs"][0]);
            for (int k = 0; k < m; k++) CHECK(std::fabs(rho[k] - (j["updateVecInvRho"][k] != 0 ? 1 / j["updateVecInvRho"][k] : 0)) < 1e-1);
            std::vector<real_t> S((size_t)m * nall), Y((size_t)m * nall);
            for (int k = 0; k < m; k++) { getLbfgsColumn(0, k, S.data() + (size_t)k * nall, nall); getLbfgsColumn(1, k, Y.data() + (size_t)k * nall, nall); }
            CHECK(closeRef(S.data(), j["updateMatS"], S.size(), "updateMatS")); CHECK(closeRef(Y.data(), j["updateMatY"], Y.size(), "updateMatY"));
            getVector(RN_BUF_LBFGS_DIR_XI, a.data()); CHECK(closeRef(a.data(), j["lbfgsDirXi"], nxi, "lbfgsDirXi"));
            getVector(RN_BUF_LBFGS_DIR_PSI, b.data()); CHECK(closeRef(b.data(), j["lbfgsDirPsi"], nps, "lbfgsDirPsi"));
        }
        {   // testFbeLineSearch (:841) / testAmeLineSearch (:748)
            put(RN_BUF_RES_XI, "fixedPointResidualXi"); put(RN_BUF_RES_PSI, "fixedPointResidualPsi");
            put(RN_BUF_ACC_XI, "acceleXi"); put(RN_BUF_ACC_PSI, "accelePsi"); put(RN_BUF_X, "X"); put(RN_BUF_U, "U");
            put(RN_BUF_LBFGS_DIR_XI, "lbfgsDirXi"); put(RN_BUF_LBFGS_DIR_PSI, "lbfgsDirPsi");
            if (fbe) { put(RN_BUF_LBFGS_CUR_YVEC_XI, "fbeGradXi"); put(RN_BUF_LBFGS_CUR_YVEC_PSI, "fbeGradPsi"); }
            put(RN_BUF_PRIMAL_XI, "primalX"); put(RN_BUF_PRIMAL_PSI, "primalU");
            const real_t v = computeValueFbe();
            const real_t tau = fbe ? computeLineSearchLbfgsUpdate(v) : computeLineSearchAmeLbfgsUpdate(v);
            CHECK(std::fabs((v - j["fbeObjDual"][0]) / j["fbeObjDual"][0] * 100) < 1e-1);
            getVector(RN_BUF_ACC_XI, a.data()); CHECK(closeRef(a.data(), j["updateXi"], nxi, "updateXi"));
            getVector(RN_BUF_ACC_PSI, b.data()); CHECK(closeRef(b.data(), j["updatePsi"], nps, "updatePsi"));
            CHECK(std::fabs(tau - j["tau"][0]) < 1e-1);
            getVector(RN_BUF_RES_XI, a.data()); CHECK(closeRef(a.data(), j["updateResidualXi"], nxi, "updateResidualXi"));
            getVector(RN_BUF_RES_PSI, b.data()); CHECK(closeRef(b.data(), j["updateResidualPsi"], nps, "updateResidualPsi"));
            std::cout << (fbe ? "fbe" : "nama") << " line search: value " << v << " tau " << tau << "\n";
        }
        {   // testFbeDualUpdate (:938)
            put(RN_BUF_XI, "acceleXi"); put(RN_BUF_PSI, "accelePsi"); put(RN_BUF_ACC_XI, "updateXi"); put(RN_BUF_ACC_PSI, "updatePsi");
            put(RN_BUF_RES_XI, "updateResidualXi"); put(RN_BUF_RES_PSI, "updateResidualPsi");
            put(RN_BUF_LBFGS_CUR_YVEC_XI, "lbfgsCurrentYvecXi"); put(RN_BUF_LBFGS_CUR_YVEC_PSI, "lbfgsCurrentYvecPsi");
            dualUpdate();
            getVector(RN_BUF_XI, a.data()); CHECK(closeRef(a.data(), j["finalUpdateXi"], nxi, "finalUpdateXi"));
            getVector(RN_BUF_PSI, b.data()); CHECK(closeRef(b.data(), j["finalUpdatePsi"], nps, "finalUpdatePsi"));
            getVector(RN_BUF_LBFGS_PREV_YVEC_XI, a.data()); CHECK(closeAbs(a.data(), j["lbfgsCurrentYvecXi"].arr.data(), nxi, 1e-12, "previousYvecXi"));
            getVector(RN_BUF_PREV_XI, a.data()); CHECK(closeAbs(a.data(), j["acceleXi"].arr.data(), nxi, 1e-12, "prevXi"));
            getVector(RN_BUF_PREV_PSI, b.data()); CHECK(closeAbs(b.data(), j["accelePsi"].arr.data(), nps, 1e-12, "prevPsi"));
            std::vector<real_t> a2(nxi);
            getVector(RN_BUF_ACC_XI, a.data()); getVector(RN_BUF_XI, a2.data()); CHECK(closeAbs(a.data(), a2.data(), nxi, 1e-300 + 0.0 + 1e-12, "accelerated == xi"));
        }
        // the whole loop runs from a cold start and stays finite
        CHECK((fbe ? algorithmGlobalFbe() : algorithmNama()) == 1);
        const uint_t last = getSmpcConfiguration()->getMaxIterations() - 1;
        CHECK(std::isfinite(getPrimalInfeasibility()[last]) && std::isfinite(getValueFbe()[last - 1]) && getVecTau()[0] == 1.0);
        std::cout << (fbe ? "fbe" : "nama") << ": primal infeasibility first " << getPrimalInfeasibility()[0] << " last " << getPrimalInfeasibility()[last]
                  << ", dual value " << getValueFbe()[0] << " -> " << getValueFbe()[last - 1] << "\n";
    }
};

static void testEngine(const string &dir) {   // Testing::testEngineTesting
    SmpcConfiguration cfg(dir + "/controllerConfig.json");
    Forecaster fc(dir + "/forecastor.json");
    Engine eng(&cfg, RN_F64, 0, g_ops);
    fc.predictDemand(1); fc.predictPrices(1);
    eng.factorStep();
    eng.updateStateControl(cfg.getCurrentX(), cfg.getPrevU(), cfg.getPrevDemand());
    eng.eliminateInputDistubanceCoupling(fc.getNominalDemand(), fc.getNominalPrices());
    jsonlite::Document j(dir + "/engineTest.json");
    const uint_t nx = cfg.getNX(), nu = cfg.getNU(), nv = cfg.getNV(), N = eng.getScenarioTree()->getPredHorizon();
    struct { int id; const char *key; } all[] = {{RN_BUF_UHAT, "uHat"}, {RN_BUF_E, "vecE"}, {RN_BUF_BETA, "beta"}, {RN_BUF_ALPHA, "costAlpha"}};
    for (auto &q : all) {
        std::vector<real_t> v(eng.getBufferSize(q.id));
        eng.getBuffer(q.id, v.data());
        CHECK(v.size() == j[q.key].Size() && closeAbs(v.data(), j[q.key].arr.data(), v.size(), 1e-2, q.key));   // Testing.cu:62
    }
    {   // the reference's raw getters (Engine.cuh: getVecUhat, getVecBeta, getVecE, getPriceAlpha): device pointers of arrays kept in the
        // reference's node-major layout; an RN_F64 engine stores doubles
        CHECK(eng.getDevicePrecision() == RN_F64);
        size_t n = 0;
        void *dev = eng.getDevicePointer(RN_BUF_UHAT, &n);
        CHECK(dev != nullptr && dev == eng.getVecUhat() && n == eng.getBufferSize(RN_BUF_UHAT));
        CHECK(eng.getVecBeta() != nullptr && eng.getVecE() != nullptr && eng.getPriceAlpha() != nullptr);
        // (what a device-to-host copy from these pointers returns is checked through the C-ABI in tests/test_gpu_device_pointer.py:
        //  this driver does not link the HIP runtime itself)
        bool threw = false;
        try { eng.getDevicePointer(RN_BUF_XI); } catch (const std::exception &) { threw = true; }    // dual-shaped: kept interleaved, no raw pointer
        CHECK(threw);
    }
    struct { int id; const char *key; uint_t dim; } path[] = {{RN_BUF_XMIN, "xmin", nx}, {RN_BUF_XMAX, "xmax", nx}, {RN_BUF_XS, "xs", nx},
                                                              {RN_BUF_UMIN, "umin", nu}, {RN_BUF_UMAX, "umax", nu}};
    for (auto &q : path) {   // compareDeviceScenarioArray: along one scenario (1-based node ids in "scenarioNodes")
        std::vector<real_t> v(eng.getBufferSize(q.id));
        eng.getBuffer(q.id, v.data());
        for (uint_t i = 0; i < N; i++) {
            const uint_t node = (uint_t)j["scenarioNodes"][i] - 1;
            CHECK(closeAbs(v.data() + (size_t)node * q.dim, j[q.key].arr.data() + (size_t)i * q.dim, q.dim, 1e-2, q.key));
        }
    }
    struct { int op; const char *key; size_t dim; uint_t count; } ops[] = {
        {RN_OP_D, "d", (size_t)2 * nx * nv, N}, {RN_OP_F, "f", (size_t)nu * nv, N}, {RN_OP_PHI, "Phi", (size_t)2 * nx * nv, N},
        {RN_OP_PSI, "Psi", (size_t)nu * nv, N}, {RN_OP_OMEGA, "omega", (size_t)nv * nv, eng.getScenarioTree()->getFinalBranchStage()},
        {RN_OP_THETA, "Theta", (size_t)nx * nv, eng.getScenarioTree()->getFinalBranchStage()},
        {RN_OP_G, "g", (size_t)nx * nv, eng.getScenarioTree()->getFinalBranchStage()}};
    for (auto &q : ops) {
        std::vector<real_t> v(q.dim);
        for (uint_t i = 0; i < q.count; i++) {
            eng.getOperator(q.op, (uint_t)j["scenarioNodes"][i] - 1, v.data(), q.dim);
            CHECK(closeAbs(v.data(), j[q.key].arr.data() + (size_t)i * q.dim, q.dim, 1e-2, q.key));
        }
    }
}

// main.cu:27-63: two closed-loop control steps with the in-built simulator, control written to a stream
static void testClosedLoop(const string &dir) {
    SmpcController ctl(dir + "/controllerConfig.json", g_ops);
    std::fstream out((dir + "/controlOutput.tmp").c_str(), std::fstream::out);
    std::vector<real_t> u(ctl.getSmpcConfiguration()->getNU());
    for (uint_t t = 0; t < 2; t++) {
        ctl.getForecaster()->predictDemand(t);
        ctl.getForecaster()->predictPrices(t);
        if (t == 0) ctl.initialiseSmpcController();
        CHECK(ctl.controlAction(out) == 1);
        ctl.moveForewardInTime();
    }
    CHECK(ctl.controlAction(u.data()) == 1);
    for (real_t v : u) CHECK(std::isfinite(v));
    // the reference's leak check (SmpcController.cu:1612-1623): a control step that leaves device memory behind returns 0 (and
    // prints the reference's message); the clean steps before and after it return 1
    {
        size_t m0[4], m1[4];
        CHECK(rn_device_memory_info(ctl.getEngine()->getContext(), m0) == RN_OK && m0[1] > 0 && m0[2] > 0 && m0[3] == 1);
        CHECK(rn_debug_inject_allocation(ctl.getEngine()->getContext(), (size_t)8 << 20) == RN_OK);
        CHECK(ctl.controlAction(u.data()) == 0);
        CHECK(rn_device_memory_info(ctl.getEngine()->getContext(), m1) == RN_OK && m1[2] == m0[2] + ((size_t)8 << 20));
        CHECK(ctl.controlAction(u.data()) == 1);
        CHECK(rn_debug_inject_allocation(ctl.getEngine()->getContext(), (size_t)8 << 20) == RN_OK);
        std::fstream out2((dir + "/controlOutput2.tmp").c_str(), std::fstream::out);
        CHECK(ctl.controlAction(out2) == 0);
        CHECK(ctl.controlAction(out2) == 1);
        out2.close();
        std::remove((dir + "/controlOutput2.tmp").c_str());
    }
    // KPIs of the two simulated steps (main.cu:66-69)
    const real_t eco = ctl.getEconomicKpi(2), smooth = ctl.getSmoothKpi(2), safe = ctl.getSafetyKpi(2), net = ctl.getNetworkKpi(2);
    CHECK(std::isfinite(eco) && eco > 0);
    CHECK(std::isfinite(smooth) && smooth >= 0);
    CHECK(std::isfinite(safe) && safe >= 0);
    CHECK(std::isfinite(net) && net > 0);
    std::cout << "KPIs: economic " << eco << " smooth " << smooth << " safety " << safe << " network " << net << "\n";
    // updateKpi against a hand computation
    {
        SmpcConfiguration *cfg = ctl.getSmpcConfiguration();
        const uint_t nx = cfg->getNX(), nu = cfg->getNU();
        std::vector<real_t> st(nx, 1.0), cu(nu, 2.0);
        const real_t e0 = ctl.getEconomicKpi(1) * 3600, s0 = ctl.getSmoothKpi(1) * 3600;
        ctl.updateKpi(st.data(), cu.data());
        real_t de = 0, ds = 0;
        for (uint_t i = 0; i < nu; i++) {
            de += (ctl.getDwnNetwork()->getAlpha()[i] + ctl.getForecaster()->getNominalPrices()[i]) * 2.0;
            ds += (cfg->getPrevU()[i] - 2.0) * (cfg->getPrevU()[i] - 2.0);
        }
        CHECK(std::fabs(ctl.getEconomicKpi(1) * 3600 - e0 - de) < 1e-9 * (1 + std::fabs(de)));
        CHECK(std::fabs(ctl.getSmoothKpi(1) * 3600 - s0 - ds) < 1e-9 * (1 + std::fabs(ds)));
    }
    // external simulator (simulatorFlag = 0, SmpcController.cu:1712-1716): the state, the previous control and the previous
    // demand are re-read from the configuration file -- whatever the in-built simulator had left in memory is replaced
    {
        SmpcConfiguration fresh(dir + "/controllerConfig.json");
        SmpcConfiguration *cfg = ctl.getSmpcConfiguration();
        const uint_t nx = cfg->getNX(), nu = cfg->getNU(), nd = cfg->getND();
        bool moved = false;                                      // two simulated steps have changed the in-memory state
        for (uint_t i = 0; i < nx; i++) moved = moved || cfg->getCurrentX()[i] != fresh.getCurrentX()[i];
        CHECK(moved);
        ctl.setSimulatorFlag(false);
        ctl.moveForewardInTime();
        CHECK(closeAbs(cfg->getCurrentX(), fresh.getCurrentX(), nx, 1e-12, "currentX re-read"));
        CHECK(closeAbs(cfg->getPrevU(), fresh.getPrevU(), nu, 1e-12, "prevU re-read"));
        CHECK(closeAbs(cfg->getPrevDemand(), fresh.getPrevDemand(), nd, 1e-12, "prevDemand re-read"));
        ctl.setSimulatorFlag(true);
    }
    out.close();
    std::remove((dir + "/controlOutput.tmp").c_str());
}

// closed loop of main.cu:45-63 for `steps` control steps, every quantity the loop carries printed with 17 digits (one JSON
// object per step) for tests/test_gpu_closed_loop.py, which runs the CPU oracle through the same steps
static void dumpClosedLoop(const string &dir, uint_t steps, bool disturbance) {
    SmpcController ctl(dir + "/controllerConfig.json", g_ops);
    ctl.setSimulatorDisturbance(disturbance);
    std::fstream out((dir + "/controlOutput.tmp").c_str(), std::fstream::out);
    SmpcConfiguration *cfg = ctl.getSmpcConfiguration();
    const uint_t nx = cfg->getNX(), nu = cfg->getNU(), nd = cfg->getND();
    std::cout.precision(17);
    auto arr = [](const char *key, const real_t *v, size_t n) {
        std::cout << "\"" << key << "\": [";
        for (size_t i = 0; i < n; i++) std::cout << (i ? ", " : "") << v[i];
        std::cout << "]";
    };
    for (uint_t t = 0; t < steps; t++) {
        ctl.getForecaster()->predictDemand(t);
        ctl.getForecaster()->predictPrices(t);
        if (t == 0) ctl.initialiseSmpcController();
        CHECK(ctl.controlAction(out) == 1);
        std::vector<real_t> u0(ctl.getEngine()->getBufferSize(RN_BUF_U));
        ctl.getEngine()->getBuffer(RN_BUF_U, u0.data());
        ctl.moveForewardInTime();
        std::vector<real_t> xAll(ctl.getEngine()->getBufferSize(RN_BUF_X));
        ctl.getEngine()->getBuffer(RN_BUF_X, xAll.data());
        const real_t kpi[4] = {ctl.getEconomicKpi(t + 1), ctl.getSmoothKpi(t + 1), ctl.getNetworkKpi(t + 1), ctl.getSafetyKpi(t + 1)};
        std::cout << "CLSTEP {";
        arr("u_root_unprojected", u0.data(), nu); std::cout << ", ";
        arr("x", cfg->getCurrentX(), nx); std::cout << ", ";
        arr("prevU", cfg->getPrevU(), nu); std::cout << ", ";
        arr("prevD", cfg->getPrevDemand(), nd); std::cout << ", ";
        arr("x_node0", xAll.data(), nx); std::cout << ", ";
        arr("kpi", kpi, 4);
        std::cout << "}" << std::endl;
    }
    out.close();
    std::remove((dir + "/controlOutput.tmp").c_str());
}

// Engine::calculateMatLandMatLhat: E L = 0, L'L = I, E Lhat = -Ed, and the solve does not depend on the basis
static void testNullSpace(const string &dir) {
    SmpcConfiguration cfg(dir + "/controllerConfig.json");
    Forecaster fc(dir + "/forecastor.json");
    fc.predictDemand(1); fc.predictPrices(1);
    std::vector<real_t> xRef, uRef;
    for (int pass = 0; pass < 2; pass++) {
        Engine eng(&cfg, RN_F64, 0, g_ops);
        DwnNetwork *net = eng.getDwnNetwork();
        const uint_t ne = net->getNumMixNodes(), nu = net->getNumControls(), nd = net->getNumDemands(), nv = cfg.getNV();
        if (pass == 1) {
            eng.calculateMatLandMatLhat();
            const real_t *L = eng.getMatL(), *Lh = eng.getMatLhat(), *E = net->getMatE(), *Ed = net->getMatEd();
            for (uint_t j = 0; j < nv; j++) {
                for (uint_t i = 0; i < ne; i++) { real_t s = 0; for (uint_t k = 0; k < nu; k++) s += E[i + (size_t)k * ne] * L[k + (size_t)j * nu]; CHECK(std::fabs(s) < 1e-12); }
                for (uint_t j2 = 0; j2 < nv; j2++) { real_t s = 0; for (uint_t k = 0; k < nu; k++) s += L[k + (size_t)j * nu] * L[k + (size_t)j2 * nu]; CHECK(std::fabs(s - (j == j2)) < 1e-12); }
            }
            for (uint_t j = 0; j < nd; j++)
                for (uint_t i = 0; i < ne; i++) { real_t s = 0; for (uint_t k = 0; k < nu; k++) s += E[i + (size_t)k * ne] * Lh[k + (size_t)j * nu]; CHECK(std::fabs(s + Ed[i + (size_t)j * ne]) < 1e-12); }
            // Lhat is unique (minimum-norm particular solution): it must equal the configuration's
            CHECK(closeAbs(Lh, cfg.getMatLhat(), (size_t)nu * nd, 1e-6, "matLhat"));
        }
        eng.factorStep();
        eng.updateStateControl(cfg.getCurrentX(), cfg.getPrevU(), cfg.getPrevDemand());
        eng.eliminateInputDistubanceCoupling(fc.getNominalDemand(), fc.getNominalPrices());
        std::vector<real_t> hist(40), x(eng.getBufferSize(RN_BUF_X)), u(eng.getBufferSize(RN_BUF_U));
        if (rn_algorithm_apg(eng.getContext(), 40, hist.data()) != RN_OK) { CHECK(false); return; }
        eng.getBuffer(RN_BUF_X, x.data()); eng.getBuffer(RN_BUF_U, u.data());
        if (pass == 0) { xRef = x; uRef = u; }
        else {
            real_t ex = 0, eu = 0, nx_ = 0, nu_ = 0;
            for (size_t i = 0; i < x.size(); i++) { ex = std::max(ex, std::fabs(x[i] - xRef[i])); nx_ = std::max(nx_, std::fabs(xRef[i])); }
            for (size_t i = 0; i < u.size(); i++) { eu = std::max(eu, std::fabs(u[i] - uRef[i])); nu_ = std::max(nu_, std::fabs(uRef[i])); }
            std::cout << "null-space basis invariance: max rel diff x " << ex / nx_ << " u " << eu / nu_ << "\n";
            // the configuration's matL is a 7-digit print (orthonormal / in null(E) to ~1e-7 only), the computed basis is exact
            CHECK(ex < 1e-6 * nx_ && eu < 1e-6 * nu_);
        }
    }
}

// warm start (extension; the reference always cold-starts, SmpcController.cu:1509): the next control step starts
// from the duals the previous one ended with (momentum restarted), a cold start from zero
static void testWarmStart(const string &dir) {
    for (int warm = 0; warm < 2; warm++) {
        SmpcConfiguration cfg(dir + "/controllerConfig.json");
        Forecaster fc(dir + "/forecastor.json");
        Engine eng(&cfg, RN_F64, 0, g_ops);
        eng.setWarmStart(warm == 1);
        eng.factorStep();
        std::vector<real_t> u(cfg.getNU());
        fc.predictDemand(0); fc.predictPrices(0);
        CHECK(rn_control_action(eng.getContext(), cfg.getCurrentX(), cfg.getPrevU(), cfg.getPrevDemand(), fc.getNominalDemand(),
                                fc.getNominalPrices(), 60, 0, u.data()) == RN_OK);
        std::vector<real_t> y1(eng.getBufferSize(RN_BUF_UPD_XI)), y2(y1.size()), w2(y1.size());
        eng.getBuffer(RN_BUF_UPD_XI, y1.data());
        fc.predictDemand(1); fc.predictPrices(1);
        // zero iterations: only the (re)start of the next control step happens
        CHECK(rn_control_action(eng.getContext(), cfg.getCurrentX(), u.data(), cfg.getPrevDemand(), fc.getNominalDemand(),
                                fc.getNominalPrices(), 0, 0, u.data()) == RN_OK);
        eng.getBuffer(RN_BUF_UPD_XI, y2.data());
        real_t diff = 0, norm1 = 0, norm2 = 0;
        for (size_t i = 0; i < y1.size(); i++) { diff = std::max(diff, std::fabs(y1[i] - y2[i])); norm1 = std::max(norm1, std::fabs(y1[i])); norm2 = std::max(norm2, std::fabs(y2[i])); }
        CHECK(norm1 > 0);
        if (warm) CHECK(diff == 0);          // duals carried over
        else CHECK(norm2 == 0);              // cold start: zeroed (SmpcController.cu:420-450)
        // and the solve continues from there
        CHECK(rn_control_action(eng.getContext(), cfg.getCurrentX(), u.data(), cfg.getPrevDemand(), fc.getNominalDemand(),
                                fc.getNominalPrices(), 25, 0, u.data()) == RN_OK);
        for (real_t v : u) CHECK(std::isfinite(v));
    }
}

// Multi-GPU through the C++ class surface.  `world` controllers are constructed exactly as a launcher would construct them,
// one per rank -- SmpcController(path, rank, world, ncclUniqueId) -- from the SAME configuration file; the library partitions
// the tree, replicates the crown and all-reduces the cut parents' children sums once per APG iteration.  A one-GPU box cannot
// host several RCCL ranks, so the ranks are threads of this process, the unique id is NULL and the in-process stand-in
// (rn_debug_local_group_join) is called wherever ncclAllReduce would be.  Every rank runs the reference's own entry points
// (initialiseSmpcController, controlAction) -- a device-resident batch of maxIterations iterations with checkpoint, dist^2
// tail and verdict vote -- and the reassembled x, u, duals must equal the unsharded controller's at 1e-9.
class ShardedController : public SmpcController {
public:
    ShardedController(const string &cfg, int rank, int world) : SmpcController(cfg, rank, world, nullptr, 0, RN_F64, 0, g_ops) {}
    explicit ShardedController(const string &cfg) : SmpcController(cfg, g_ops) {}
    using SmpcController::getVector;
    using SmpcController::algorithmApg;
};
static void testSharded(const string &dir, int world) {
    const string cfgPath = dir + "/controllerConfig.json";
    ShardedController ref(cfgPath);
    ref.getForecaster()->predictDemand(0); ref.getForecaster()->predictPrices(0);
    ref.initialiseSmpcController();
    const uint_t nx = ref.getSmpcConfiguration()->getNX(), nu = ref.getSmpcConfiguration()->getNU();
    const uint_t nodes = ref.getScenarioTree()->getNumNodes();
    std::vector<real_t> uRef(nu);
    // SmpcController::algorithmApg records the tree-global vecPrimalInfs (SmpcController.cu:1521): a sharded controller's must be
    // the unsharded controller's on every rank
    const uint_t maxIt = ref.getSmpcConfiguration()->getMaxIterations();
    CHECK(ref.algorithmApg() == 1);
    const std::vector<real_t> infRef(ref.getPrimalInfeasibility(), ref.getPrimalInfeasibility() + maxIt);
    CHECK(ref.controlAction(uRef.data()) == 1);
    void *group = nullptr;
    CHECK(rn_debug_local_group_create(world, &group) == RN_OK);
    std::vector<ShardedController *> rk(world, nullptr);
    for (int r = 0; r < world; r++) {
        rk[r] = new ShardedController(cfgPath, r, world);
        CHECK(rn_debug_local_group_join(rk[r]->getEngine()->getContext(), group, r) == RN_OK);
        CHECK(rk[r]->getEngine()->getRank() == r && rk[r]->getEngine()->getNumRanks() == world);
    }
    std::vector<std::vector<real_t>> u0(world, std::vector<real_t>(nu));
    std::vector<int> ok(world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r]() {
            try {
                rk[r]->getForecaster()->predictDemand(0); rk[r]->getForecaster()->predictPrices(0);
                rk[r]->initialiseSmpcController();
                ok[r] = (int)rk[r]->algorithmApg();
                ok[r] = ok[r] && (int)rk[r]->controlAction(u0[r].data());
            } catch (const std::exception &e) { std::cerr << "rank " << r << ": " << e.what() << "\n"; ok[r] = 0; }
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < world; r++) CHECK(ok[r] == 1);
    {
        real_t scale = 0;
        for (real_t v : infRef) scale = std::max(scale, std::fabs(v));
        for (int r = 0; r < world; r++) {
            CHECK(closeAbs(rk[r]->getPrimalInfeasibility(), infRef.data(), maxIt, 1e-9 * scale, "vecPrimalInfs (tree-global on every rank)"));
            CHECK(std::memcmp(rk[r]->getPrimalInfeasibility(), rk[0]->getPrimalInfeasibility(), maxIt * sizeof(real_t)) == 0);
        }
    }
    // the root's control is replicated: identical bits on every rank, and the unsharded controller's value
    real_t un = 0;
    for (uint_t i = 0; i < nu; i++) un = std::max(un, std::fabs(uRef[i]));
    for (int r = 0; r < world; r++) {
        CHECK(std::memcmp(u0[r].data(), u0[0].data(), nu * sizeof(real_t)) == 0);
        CHECK(closeAbs(u0[r].data(), uRef.data(), nu, 1e-9 * un, "u0"));
    }
    // reassemble the node-major iterates through the ranks' global node maps
    struct Buf { int id; uint_t dim; const char *name; };
    const Buf bufs[] = {{RN_BUF_X, nx, "X"}, {RN_BUF_U, nu, "U"}, {RN_BUF_UPD_XI, 2 * nx, "updateXi"}, {RN_BUF_UPD_PSI, nu, "updatePsi"},
                        {RN_BUF_DUAL_XI, 2 * nx, "dualXi"}};
    std::vector<int> owners(nodes, 0);
    for (const Buf &b : bufs) {
        std::vector<real_t> full((size_t)nodes * b.dim, 0.0), want((size_t)nodes * b.dim);
        ref.getVector(b.id, want.data());
        for (int r = 0; r < world; r++) {
            const std::vector<int> g = rk[r]->getEngine()->getGlobalNodes();
            std::vector<real_t> loc((size_t)g.size() * b.dim);
            CHECK(rk[r]->getEngine()->getBufferSize(b.id) == loc.size());
            rk[r]->getVector(b.id, loc.data());
            for (size_t l = 0; l < g.size(); l++) {
                std::memcpy(&full[(size_t)g[l] * b.dim], &loc[l * b.dim], b.dim * sizeof(real_t));
                if (b.id == RN_BUF_X) owners[g[l]]++;
            }
        }
        real_t scale = 0;
        for (real_t v : want) scale = std::max(scale, std::fabs(v));
        CHECK(closeAbs(full.data(), want.data(), full.size(), 1e-9 * scale, b.name));
    }
    // the closed loop of main.cu:45-63 on the shards: two more control steps with the in-built simulator in between.  The
    // root's control is replicated, so every rank simulates the same plant and carries the same state, previous control,
    // previous demand and KPIs as the unsharded controller.
    auto closedLoop = [&](ShardedController *c, int tag, std::vector<real_t> &state, real_t kpi[4]) {
        const string tmp = dir + "/controlOutput_" + std::to_string(tag) + ".tmp";
        std::fstream out(tmp.c_str(), std::fstream::out);
        bool good = true;
        for (uint_t t = 0; t < 2; t++) {
            c->getForecaster()->predictDemand(t); c->getForecaster()->predictPrices(t);
            good = good && c->controlAction(out) == 1;
            c->moveForewardInTime();
        }
        out.close(); std::remove(tmp.c_str());
        SmpcConfiguration *cfg = c->getSmpcConfiguration();
        state.assign(cfg->getCurrentX(), cfg->getCurrentX() + nx);
        state.insert(state.end(), cfg->getPrevU(), cfg->getPrevU() + nu);
        kpi[0] = c->getEconomicKpi(2); kpi[1] = c->getSmoothKpi(2); kpi[2] = c->getNetworkKpi(2); kpi[3] = c->getSafetyKpi(2);
        return good;
    };
    std::vector<real_t> refState; real_t refKpi[4];
    CHECK(closedLoop(&ref, 999, refState, refKpi));
    std::vector<std::vector<real_t>> st(world); std::vector<std::array<real_t, 4>> kp(world);
    std::vector<std::thread> th2;
    for (int r = 0; r < world; r++)
        th2.emplace_back([&, r]() {
            try { ok[r] = closedLoop(rk[r], r, st[r], kp[r].data()) ? 1 : 0; }
            catch (const std::exception &e) { std::cerr << "rank " << r << ": " << e.what() << "\n"; ok[r] = 0; }
        });
    for (auto &t : th2) t.join();
    real_t sn = 0;
    for (real_t v : refState) sn = std::max(sn, std::fabs(v));
    for (int r = 0; r < world; r++) {
        CHECK(ok[r] == 1);
        CHECK(st[r].size() == refState.size() && closeAbs(st[r].data(), refState.data(), refState.size(), 1e-9 * sn, "closed-loop state / previous control"));
        for (int k = 0; k < 4; k++) CHECK(std::fabs(kp[r][k] - refKpi[k]) <= 1e-9 * (1 + std::fabs(refKpi[k])));
        CHECK(std::memcmp(st[r].data(), st[0].data(), st[0].size() * sizeof(real_t)) == 0);   // identical bits on every rank
    }
    int info[7];
    CHECK(rn_shard_info(rk[0]->getEngine()->getContext(), info) == RN_OK);
    const uint_t crown = ref.getScenarioTree()->getNodesPerStageCumul()[info[2]];
    for (uint_t i = 0; i < nodes; i++) CHECK(owners[i] == (i < crown ? world : 1));   // crown replicated, every other node owned once
    long cnt[4];
    CHECK(rn_get_counters(rk[0]->getEngine()->getContext(), cnt) == RN_OK);
    // (one algorithmApg and three control steps were run: four optimistic batches, or -- once a threshold has tripped -- one
    //  replayed batch and the back-off's exact batches after it)
    std::cout << "sharded: " << world << " ranks, cut stage " << info[2] << ", " << info[3] << " cut parents, crown " << crown << " nodes, batches optimistic/exact/replayed "
              << cnt[0] << "/" << cnt[1] << "/" << cnt[2] << "\n";
    for (int r = 0; r < world; r++) {
        long c2[4];
        CHECK(rn_get_counters(rk[r]->getEngine()->getContext(), c2) == RN_OK);
        CHECK(c2[0] == cnt[0] && c2[1] == cnt[1] && c2[2] == cnt[2] && c2[3] == cnt[3]);   // every rank took the same path through the batches
        delete rk[r]->getEngine();
        delete rk[r];
    }
    CHECK(rn_debug_local_group_destroy(group) == RN_OK);
    delete ref.getEngine();
}

// What a drop-in caller gets: SmpcController(path) on the reference's own configuration file (no "operatorMode" key) runs the
// structured form (RN_OPS_AUTO), the iterates are the dense form's, and handing in a block (Engine::setOperator, the counterpart of writing
// through the reference's getMatPhi() pointers, Engine.cuh:170-230) switches the engine to dense storage with that block in use.
static void testOperatorMode(const string &dir) {
    SmpcController a(dir + "/controllerConfig.json"), dn(dir + "/controllerConfig.json", RN_OPS_DENSE);
    CHECK(a.getSmpcConfiguration()->getOperatorMode() == "auto");
    for (SmpcController *c : {&a, &dn}) {
        c->getForecaster()->predictDemand(1);
        c->getForecaster()->predictPrices(1);
        c->initialiseSmpcController();
    }
    int req = -1, act = -1;
    CHECK(rn_get_operator_mode(a.getEngine()->getContext(), &req, &act) == RN_OK && req == RN_OPS_AUTO && act == RN_OPS_STRUCTURED);
    CHECK(dn.getEngine()->getOperatorMode() == RN_OPS_DENSE);
    std::cout << "default: auto -> " << (act == RN_OPS_STRUCTURED ? "structured" : "dense") << "\n";
    const uint_t nu = a.getSmpcConfiguration()->getNU(), nx = a.getSmpcConfiguration()->getNX(), nv = a.getSmpcConfiguration()->getNV();
    std::vector<real_t> ua(nu), ud(nu);
    CHECK(a.controlAction(ua.data()) == 1 && dn.controlAction(ud.data()) == 1);
    CHECK(closeRel(ua.data(), ud.data(), nu, 1e-9, "u0: auto vs dense"));
    // hand the factor step's own block of one node back in: the engine materialises the dense blocks and nothing changes ...
    const uint_t node = a.getScenarioTree()->getNumNodes() / 2;
    std::vector<real_t> phi((size_t)2 * nx * nv);
    a.getEngine()->getOperator(RN_OP_PHI, node, phi.data(), phi.size());
    a.getEngine()->setOperator(RN_OP_PHI, node, phi.data(), phi.size());
    CHECK(a.getEngine()->getOperatorMode() == RN_OPS_DENSE);
    std::cout << "after setOperator: " << (a.getEngine()->getOperatorMode() == RN_OPS_DENSE ? "dense" : "structured") << "\n";
    CHECK(a.controlAction(ua.data()) == 1);
    CHECK(closeRel(ua.data(), ud.data(), nu, 1e-9, "u0: materialised vs dense"));
    // ... and a block of the caller's own IS used by the next sweep, exactly as in a dense engine given the same block
    for (real_t &v : phi) v *= 1.5;
    a.getEngine()->setOperator(RN_OP_PHI, node, phi.data(), phi.size());
    dn.getEngine()->setOperator(RN_OP_PHI, node, phi.data(), phi.size());
    std::vector<real_t> back(phi.size());
    a.getEngine()->getOperator(RN_OP_PHI, node, back.data(), back.size());
    CHECK(closeRel(back.data(), phi.data(), phi.size(), 1e-15, "block read back"));
    std::vector<real_t> u2(nu), u3(nu);
    CHECK(a.controlAction(u2.data()) == 1 && dn.controlAction(u3.data()) == 1);
    CHECK(closeRel(u2.data(), u3.data(), nu, 1e-9, "u0 with the caller's block: materialised vs dense"));
    real_t diff = 0;
    for (uint_t i = 0; i < nu; i++) diff = std::max(diff, std::fabs(u2[i] - ud[i]));
    CHECK(diff > 0);      // the block matters
    // an engine that was told never to keep blocks refuses one
    SmpcController st(dir + "/controllerConfig.json", RN_OPS_STRUCTURED);
    st.getForecaster()->predictDemand(1); st.getForecaster()->predictPrices(1);
    st.initialiseSmpcController();
    CHECK(rn_set_operator(st.getEngine()->getContext(), RN_OP_PHI, (int)node, phi.data(), phi.size()) == RN_E_STATE);
    CHECK(rn_set_operator(a.getEngine()->getContext(), RN_OP_OMEGA, (int)node, phi.data(), (size_t)nv * nv) == RN_E_ARG);
}

int main(int argc, char **argv) {
    if (argc < 3) { std::cerr << "usage: test_host <loaders|engine|controller|fbe|nama|closedloop|warmstart|nullspace|sharded> <fixture dir> [args] [ops=auto|dense|structured]\n"; return 2; }
    const string mode = argv[1], dir = argv[2];
    // last argument "ops=...": the operator mode every controller of this run is constructed with (default: what the class surface gives a
    // caller who says nothing -- the configuration file's key, absent in the reference's files: auto)
    if (argc > 3 && string(argv[argc - 1]).rfind("ops=", 0) == 0) {
        const string m = string(argv[argc - 1]).substr(4);
        g_ops = m == "dense" ? RN_OPS_DENSE : (m == "structured" ? RN_OPS_STRUCTURED : (m == "auto" ? RN_OPS_AUTO : -2));
        if (g_ops == -2) { std::cerr << "unknown operator mode " << m << "\n"; return 2; }
        argc--;
    }
    try {
        if (mode == "loaders") testLoaders(dir);
        else if (mode == "engine") testEngine(dir);
        else if (mode == "controller") {
            TestSmpcController t(dir + "/controllerConfig.json");
            t.getForecaster()->predictDemand(1);   // timeInst = 1, Testing.cu:500-502
            t.getForecaster()->predictPrices(1);
            t.run(dir);
        } else if (mode == "fbe" || mode == "nama") {
            TestFbeNamaController t(dir + (mode == "fbe" ? "/controllerFbeConfig.json" : "/controllerNamaConfig.json"));
            t.getForecaster()->predictDemand(1);   // timeInst = 1, Testing.cu:540-542
            t.getForecaster()->predictPrices(1);
            t.run(dir);
        } else if (mode == "closedloop") testClosedLoop(dir);
        else if (mode == "closedloop_dump") dumpClosedLoop(dir, argc > 3 ? (uint_t)std::atoi(argv[3]) : 3, argc > 4 && std::atoi(argv[4]) != 0);
        else if (mode == "opsmode") testOperatorMode(dir);
        else if (mode == "nullspace") testNullSpace(dir);
        else if (mode == "warmstart") testWarmStart(dir);
        else if (mode == "sharded") testSharded(dir, argc > 3 ? std::atoi(argv[3]) : 2);
        else { std::cerr << "unknown mode\n"; return 2; }
    } catch (const std::exception &e) {
        std::cerr << "EXCEPTION: " << e.what() << "\n";
        return 3;
    }
    if (g_failures) { std::cerr << g_failures << " check(s) failed\n"; return 1; }
    std::cout << mode << ": all checks passed" << (g_ops >= 0 ? (g_ops == RN_OPS_DENSE ? " [dense]" : (g_ops == RN_OPS_STRUCTURED ? " [structured]" : " [auto]")) : "") << "\n";
    return 0;
}
