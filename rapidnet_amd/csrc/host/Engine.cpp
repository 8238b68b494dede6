// Engine.cpp -- see Engine.hpp.
#include "Engine.hpp"

#include "NullSpace.hpp"

void Engine::check(int rc, const char *what) {
    if (rc != RN_OK) throw std::runtime_error(string(what) + ": " + rn_last_error(ctx));
}

Engine::Engine(SmpcConfiguration *smpcConfig, int precision, int device, int operatorMode) : ctx(nullptr) {
    ptrMySmpcConfig = smpcConfig;
    ptrMyNetwork = new DwnNetwork(smpcConfig->getPathToNetwork());          // never deleted by the reference either
    ptrMyScenarioTree = new ScenarioTree(smpcConfig->getPathToScenarioTree());
    create(precision, device, operatorMode);
}

Engine::Engine(DwnNetwork *network, ScenarioTree *scenarioTree, SmpcConfiguration *smpcConfig, int precision, int device, int operatorMode) : ctx(nullptr) {
    ptrMyNetwork = network; ptrMyScenarioTree = scenarioTree; ptrMySmpcConfig = smpcConfig;
    create(precision, device, operatorMode);
}

Engine::Engine(SmpcConfiguration *smpcConfig, int precision, int device, int rank, int nranks, const void *id128, int cutStage, int operatorMode) : ctx(nullptr) {
    ptrMySmpcConfig = smpcConfig;
    ptrMyNetwork = new DwnNetwork(smpcConfig->getPathToNetwork());
    ptrMyScenarioTree = new ScenarioTree(smpcConfig->getPathToScenarioTree());
    create(precision, device, operatorMode, rank, nranks, id128, cutStage);
}

Engine::Engine(DwnNetwork *network, ScenarioTree *scenarioTree, SmpcConfiguration *smpcConfig, int precision, int device, int rank, int nranks,
               const void *id128, int cutStage, int operatorMode) : ctx(nullptr) {
    ptrMyNetwork = network; ptrMyScenarioTree = scenarioTree; ptrMySmpcConfig = smpcConfig;
    create(precision, device, operatorMode, rank, nranks, id128, cutStage);
}

uint_t Engine::getNumLocalNodes() {
    int info[7];
    check(rn_shard_info(ctx, info), "rn_shard_info");
    return (uint_t)info[5];
}
std::vector<int> Engine::getGlobalNodes() {
    std::vector<int> g(getNumLocalNodes());
    check(rn_shard_global_nodes(ctx, g.data(), g.size()), "rn_shard_global_nodes");
    return g;
}

void Engine::setOperatorMode(int mode) { check(rn_set_operator_mode(ctx, mode), "rn_set_operator_mode"); }
int Engine::getOperatorMode() {
    int active = RN_OPS_DENSE;
    check(rn_get_operator_mode(ctx, nullptr, &active), "rn_get_operator_mode");
    return active;
}
void Engine::setOperator(int op, uint_t node, const real_t *host, size_t n) { check(rn_set_operator(ctx, op, node, host, n), "rn_set_operator"); }

void Engine::create(int precision, int device, int operatorMode, int rank, int nranks, const void *id128, int cutStage) {
    myRank = rank; numRanks = nranks;
    priceUncertaintyFlag = true; demandUncertaintyFlag = true;
    const string alg = ptrMySmpcConfig->getOptimisationAlgorithm();   // Engine.cu:151-163
    globalFbeFlag = (alg == "globalFbeAlgorithm");
    namaFlag = (alg == "namaAlgorithm");
    apgFlag = !globalFbeFlag && !namaFlag;
    _ASSERT(ptrMyNetwork->getNumTanks() == ptrMySmpcConfig->getNX() && ptrMyNetwork->getNumControls() == ptrMySmpcConfig->getNU());
    _ASSERT(ptrMyNetwork->getNumDemands() == ptrMySmpcConfig->getND());
    rn_dims d;
    d.nx = ptrMyNetwork->getNumTanks(); d.nu = ptrMyNetwork->getNumControls(); d.nv = ptrMySmpcConfig->getNV();
    d.nd = ptrMyNetwork->getNumDemands(); d.N = ptrMyScenarioTree->getPredHorizon(); d.K = ptrMyScenarioTree->getNumScenarios();
    d.nodes = ptrMyScenarioTree->getNumNodes(); d.nNonLeafNodes = ptrMyScenarioTree->getNumNonleafNodes();
    rn_tree t;
    t.stages = ptrMyScenarioTree->getStageNodes(); t.nodesPerStage = ptrMyScenarioTree->getNodesPerStage();
    t.nodesPerStageCumul = ptrMyScenarioTree->getNodesPerStageCumul(); t.ancestor = ptrMyScenarioTree->getAncestorArray();
    t.nChildren = ptrMyScenarioTree->getNumChildren(); t.nChildrenCumul = ptrMyScenarioTree->getNumChildrenCumul();
    t.probNode = ptrMyScenarioTree->getProbArray();
    // one entry point for both cases: nranks == 1 is a plain rn_create + rn_set_tree_errors
    const int rc = rn_create_sharded(&d, &t, ptrMyScenarioTree->getErrorDemandArray(), ptrMyScenarioTree->getErrorPriceArray(), precision, device,
                                     rank, nranks, cutStage, id128, &ctx);
    if (rc != RN_OK) throw std::runtime_error(string("rn_create_sharded: ") + rn_last_error(nullptr));
    check(rn_set_parameters(ctx, ptrMySmpcConfig->getStepSize(), ptrMySmpcConfig->getPenaltyState(), ptrMySmpcConfig->getPenaltySafety()),
          "rn_set_parameters");
    if (operatorMode < 0) {   // the configuration's optional key (absent: auto)
        const string m = ptrMySmpcConfig->getOperatorMode();
        operatorMode = m == "dense" ? RN_OPS_DENSE : (m == "structured" ? RN_OPS_STRUCTURED : RN_OPS_AUTO);
    }
    check(rn_set_operator_mode(ctx, operatorMode), "rn_set_operator_mode");
    // SmpcController::allocateApgAlgorithm (SmpcController.cu:124-151): per-iteration storage for maxIterations, allocated once
    check(rn_reserve_iterations(ctx, (int)ptrMySmpcConfig->getMaxIterations()), "rn_reserve_iterations");
    if (!apgFlag)   // SmpcController::allocateGlobalFbeAlgorithm / allocateNamaAlgorithm / allocateLbfgsBuffer (SmpcController.cu:234-330)
        check(rn_set_algorithm(ctx, globalFbeFlag ? RN_ALG_GLOBAL_FBE : RN_ALG_NAMA, (int)ptrMySmpcConfig->getLbfgsBufferSize()), "rn_set_algorithm");
}

Engine::~Engine() { if (ctx) rn_destroy(ctx); }

void Engine::factorStep() {
    rn_system s;
    s.matB = ptrMyNetwork->getMatB(); s.matGd = ptrMyNetwork->getMatGd();
    s.matL = getMatL(); s.matLhat = getMatLhat();
    s.costW = ptrMySmpcConfig->getCostW(); s.matDiagPrecnd = ptrMySmpcConfig->getMatPrcndDiag();
    s.vecXmin = ptrMyNetwork->getXmin(); s.vecXmax = ptrMyNetwork->getXmax(); s.vecXsafe = ptrMyNetwork->getXsafe();
    s.vecUmin = ptrMyNetwork->getUmin(); s.vecUmax = ptrMyNetwork->getUmax(); s.costAlpha1 = ptrMyNetwork->getAlpha();
    check(rn_factor_step(ctx, &s), "rn_factor_step");
}
void Engine::updateStateControl(real_t *currentX, real_t *prevU, real_t *prevDemand) {
    check(rn_update_state_control(ctx, currentX, prevU, prevDemand), "rn_update_state_control");
}
void Engine::eliminateInputDistubanceCoupling(real_t *nominalDemand, real_t *nominalPrices) {
    check(rn_eliminate_input_disturbance_coupling(ctx, nominalDemand, nominalPrices), "rn_eliminate_input_disturbance_coupling");
}
void Engine::setPriceUncertaintyFlag(bool f) {
    priceUncertaintyFlag = f;
    check(rn_set_uncertainty(ctx, demandUncertaintyFlag, priceUncertaintyFlag, ptrMySmpcConfig->getWeightEconomical()), "rn_set_uncertainty");
}
void Engine::setDemandUncertaintyFlag(bool f) {
    demandUncertaintyFlag = f;
    check(rn_set_uncertainty(ctx, demandUncertaintyFlag, priceUncertaintyFlag, ptrMySmpcConfig->getWeightEconomical()), "rn_set_uncertainty");
}
size_t Engine::getBufferSize(int id) { return rn_buffer_size(ctx, id); }
void Engine::getBuffer(int id, real_t *host) { check(rn_get(ctx, id, host, rn_buffer_size(ctx, id)), "rn_get"); }
void Engine::setBuffer(int id, const real_t *host) { check(rn_set(ctx, id, host, rn_buffer_size(ctx, id)), "rn_set"); }
void Engine::getBufferRange(int id, size_t first, size_t n, real_t *host) { check(rn_get_range(ctx, id, first, n, host), "rn_get_range"); }
void Engine::setBufferRange(int id, size_t first, size_t n, const real_t *host) { check(rn_set_range(ctx, id, first, n, host), "rn_set_range"); }
void Engine::getOperator(int op, uint_t node, real_t *host, size_t n) { check(rn_get_operator(ctx, op, node, host, n), "rn_get_operator"); }
void *Engine::getDevicePointer(int bufferId, size_t *n) {
    void *p = nullptr;
    size_t cnt = 0;
    int prec = 0;
    check(rn_device_pointer(ctx, bufferId, &p, &cnt, &prec), "rn_device_pointer");
    if (n) *n = cnt;
    return p;
}
int Engine::getDevicePrecision() {
    void *p = nullptr;
    size_t cnt = 0;
    int prec = 0;
    check(rn_device_pointer(ctx, RN_BUF_UHAT, &p, &cnt, &prec), "rn_device_pointer");
    return prec;
}

void Engine::calculateMatLandMatLhat() {
    const int ne = ptrMyNetwork->getNumMixNodes(), nu = ptrMyNetwork->getNumControls(), nd = ptrMyNetwork->getNumDemands();
    const int rank = computeNullSpaceAndParticular(ptrMyNetwork->getMatE(), ptrMyNetwork->getMatEd(), ne, nu, nd, computedL, computedLhat);
    if (nu - rank != (int)ptrMySmpcConfig->getNV())
        throw std::logic_error("calculateMatLandMatLhat: null space of E has dimension " + std::to_string(nu - rank) + ", the configuration says nv = " +
                               std::to_string(ptrMySmpcConfig->getNV()));
    useComputedL = true;
}
void Engine::setWarmStart(bool on) { check(rn_set_warm_start(ctx, on ? 1 : 0), "rn_set_warm_start"); }
