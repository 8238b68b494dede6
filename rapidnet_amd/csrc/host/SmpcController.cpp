// SmpcController.cpp -- see SmpcController.hpp.
#include "SmpcController.hpp"

#include <cmath>

void SmpcController::check(int rc, const char *what) {
    if (rc != RN_OK) throw std::runtime_error(string(what) + ": " + rn_last_error(ptrMyEngine->getContext()));
}

SmpcController::SmpcController(Forecaster *f, Engine *e, SmpcConfiguration *c)
    : ptrMyEngine(e), ptrMyForecaster(f), ptrMySmpcConfig(c), factorStepFlag(false), simulatorFlag(true), ownsObjects(false) {
    stepSize = c->getStepSize();
    vecPrimalInfs.assign(c->getMaxIterations() + 1, 0.0);
    vecValueFbe.assign(c->getMaxIterations() + 1, 0.0); vecTau.assign(c->getMaxIterations() + 1, 0.0);
    lastControl.assign(c->getNU(), 0.0);
    economicKpi = smoothKpi = safeKpi = networkKpi = 0;
}

SmpcController::SmpcController(string pathToConfigFile, int operatorMode) : factorStepFlag(false), simulatorFlag(true), ownsObjects(true) {
    ptrMySmpcConfig = new SmpcConfiguration(pathToConfigFile);          // SmpcController.cu:78-81
    ptrMyForecaster = new Forecaster(ptrMySmpcConfig->getPathToForecaster());
    ptrMyEngine = new Engine(ptrMySmpcConfig, RN_F64, 0, operatorMode);
    stepSize = ptrMySmpcConfig->getStepSize();
    vecPrimalInfs.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0);
    vecValueFbe.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0); vecTau.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0);
    lastControl.assign(ptrMySmpcConfig->getNU(), 0.0);
    economicKpi = smoothKpi = safeKpi = networkKpi = 0;
}

SmpcController::SmpcController(string pathToConfigFile, int rank, int nranks, const void *id128, int device, int precision, int cutStage, int operatorMode)
    : factorStepFlag(false), simulatorFlag(true), ownsObjects(true) {
    ptrMySmpcConfig = new SmpcConfiguration(pathToConfigFile);
    ptrMyForecaster = new Forecaster(ptrMySmpcConfig->getPathToForecaster());
    ptrMyEngine = new Engine(ptrMySmpcConfig, precision, device, rank, nranks, id128, cutStage, operatorMode);
    stepSize = ptrMySmpcConfig->getStepSize();
    vecPrimalInfs.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0);
    vecValueFbe.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0); vecTau.assign(ptrMySmpcConfig->getMaxIterations() + 1, 0.0);
    lastControl.assign(ptrMySmpcConfig->getNU(), 0.0);
    economicKpi = smoothKpi = safeKpi = networkKpi = 0;
}

// the reference never deletes the objects its path-constructor news (SmpcController.cu:2091-2102); callers that want
// them gone delete them through the getters, exactly as the reference's tests do (Testing.cu:515-520)
SmpcController::~SmpcController() {}

void SmpcController::initialiseSmpcController() {
    factorStepFlag = true;
    ptrMyEngine->factorStep();
    ptrMyEngine->updateStateControl(ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(), ptrMySmpcConfig->getPrevDemand());
    ptrMyEngine->eliminateInputDistubanceCoupling(ptrMyForecaster->getNominalDemand(), ptrMyForecaster->getNominalPrices());
}

void SmpcController::controllerSmpc() {
    if (!factorStepFlag) { ptrMyEngine->factorStep(); factorStepFlag = true; }
    ptrMyEngine->updateStateControl(ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(), ptrMySmpcConfig->getPrevDemand());
    ptrMyEngine->eliminateInputDistubanceCoupling(ptrMyForecaster->getNominalDemand(), ptrMyForecaster->getNominalPrices());
    algorithmApg();
}

// The reference's leak check (SmpcController.cu:1612-1623, :1641, :1657-1664): free device memory before and after the control
// step; any difference prints "RUNTIME ERROR: MEMORY LEAKS" and makes controlAction return 0.  rn_device_memory_info gives the
// device-wide figure the reference compares (cudaMemGetInfo) and the bytes this controller's context holds.  The device-wide
// figure also moves when ANOTHER context of this process allocates (several sharded controllers on one device in one process):
// it is compared only while this context is the process's only one; the context's own bytes are compared always.
bool SmpcController::deviceMemoryChanged(const size_t before[4]) {
    size_t after[4];
    check(rn_device_memory_info(ptrMyEngine->getContext(), after), "rn_device_memory_info");
    const bool alone = before[3] == 1 && after[3] == 1;
    if (after[2] != before[2] || (alone && after[0] != before[0])) {
        std::cout << "RUNTIME ERROR: MEMORY LEAKS" << std::endl;
        return true;
    }
    return false;
}

uint_t SmpcController::controlAction(real_t *u) {
    if (!factorStepFlag) { ptrMyEngine->factorStep(); factorStepFlag = true; }
    size_t before[4];
    check(rn_device_memory_info(ptrMyEngine->getContext(), before), "rn_device_memory_info");
    const int rc = rn_control_action(ptrMyEngine->getContext(), ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(),
                                     ptrMySmpcConfig->getPrevDemand(), ptrMyForecaster->getNominalDemand(),
                                     ptrMyForecaster->getNominalPrices(), ptrMySmpcConfig->getMaxIterations(), 0, u);
    if (rc != RN_OK) { std::cerr << "controlAction: " << rn_last_error(ptrMyEngine->getContext()) << std::endl; return 0; }
    if (deviceMemoryChanged(before)) return 0;
    return 1;   // devControlAction (lastControl) is only written by the stream overload (:1647), as in the reference
}

uint_t SmpcController::controlAction(std::fstream &out) {
    if (!out.is_open()) return 0;
    if (!factorStepFlag) { ptrMyEngine->factorStep(); factorStepFlag = true; }
    const uint_t nu = ptrMySmpcConfig->getNU();
    std::vector<real_t> u(nu);
    size_t before[4];
    check(rn_device_memory_info(ptrMyEngine->getContext(), before), "rn_device_memory_info");
    const int rc = rn_control_action(ptrMyEngine->getContext(), ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(),
                                     ptrMySmpcConfig->getPrevDemand(), ptrMyForecaster->getNominalDemand(),
                                     ptrMyForecaster->getNominalPrices(), ptrMySmpcConfig->getMaxIterations(), 1 /* projectionBox :1649 */,
                                     u.data());
    if (rc != RN_OK) { std::cerr << "controlAction: " << rn_last_error(ptrMyEngine->getContext()) << std::endl; return 0; }
    out << "\"control\" : [";                                   // same (not quite JSON) format as :1651-1658
    for (uint_t i = 0; i < nu; i++) out << u[i] << ", ";
    out << "]" << std::endl;
    lastControl = u;
    if (deviceMemoryChanged(before)) return 0;
    return 1;
}

// SmpcController::moveForewardInTime (SmpcController.cu:1679-1716).
// In-built simulator: the plant is advanced with the (projected) control of the last controlAction(fstream&), the KPIs
// are accumulated with the still-unshifted previous control, then state / previous control / previous demand are shifted.
// Default: the reference as written -- its "x = x + w" (:1695) adds the disturbance of node 0 to devVecX, the x iterate of
// node 0, not to the state update, so the simulated plant is x+ = x + B u and that output buffer changes; reproduced
// here, buffer included.  setSimulatorDisturbance(true): x+ = x + e_0 + B u, the plant of DwnNetwork.cuh:41-57.
// External simulator (simulatorFlag == false): the three vectors are re-read from the configuration file.
void SmpcController::moveForewardInTime() {
    if (!simulatorFlag) {
        ptrMySmpcConfig->setCurrentState();
        ptrMySmpcConfig->setPreviousControl();
        ptrMySmpcConfig->setPreviousDemand();
        return;
    }
    DwnNetwork *net = ptrMyEngine->getDwnNetwork();
    const uint_t nx = net->getNumTanks(), nu = net->getNumControls(), nd = net->getNumDemands();
    std::vector<real_t> e(nx);                                  // only node 0's rows travel (rn_get_range / rn_set_range)
    ptrMyEngine->getBufferRange(RN_BUF_E, 0, nx, e.data());
    std::vector<real_t> next(ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getCurrentX() + nx);
    if (simulatorDisturbance) {
        for (uint_t i = 0; i < nx; i++) next[i] += e[i];
    } else {
        std::vector<real_t> x0(nx);
        ptrMyEngine->getBufferRange(RN_BUF_X, 0, nx, x0.data());
        for (uint_t i = 0; i < nx; i++) x0[i] += e[i];
        ptrMyEngine->setBufferRange(RN_BUF_X, 0, nx, x0.data());
    }
    const real_t *B = net->getMatB();
    for (uint_t j = 0; j < nu; j++)                             // column sweep, like the gemv it stands for
        for (uint_t i = 0; i < nx; i++) next[i] += B[i + (size_t)j * nx] * lastControl[j];
    updateKpi(next.data(), lastControl.data());
    ptrMySmpcConfig->setCurrentState(next.data());
    ptrMySmpcConfig->setPreviousControl(lastControl.data());
    std::vector<real_t> demand(ptrMyForecaster->getNominalDemand(), ptrMyForecaster->getNominalDemand() + nd);
    ptrMySmpcConfig->setpreviousdemand(demand.data());
}

// Key performance indicators of one simulated step, accumulated over the closed loop (SmpcController.cu:1778-1813):
//   economic += w_e * sum_j (alpha1_j + alphahat_j) |u_j|        smooth += || u_prev - u ||^2
//   safety   += sum_i max(0, xs_i - x_i)                          network += sum_i |x_i|
// alphahat = the forecaster's nominal prices of the current stage, u_prev = the configuration's previous control.
void SmpcController::updateKpi(real_t *state, real_t *control) {
    DwnNetwork *net = ptrMyEngine->getDwnNetwork();
    const uint_t nTanks = ptrMySmpcConfig->getNX(), nInputs = ptrMySmpcConfig->getNU();
    const real_t *price0 = net->getAlpha(), *priceNow = ptrMyForecaster->getNominalPrices(), *uBefore = ptrMySmpcConfig->getPrevU();
    const real_t wEco = ptrMySmpcConfig->getWeightEconomical();
    real_t pumping = 0, variation = 0;
    for (uint_t j = 0; j < nInputs; j++) {
        pumping = pumping + wEco * (price0[j] + priceNow[j]) * std::fabs(control[j]);
        const real_t step = uBefore[j] - control[j];
        variation = variation + step * step;
    }
    const real_t *reserve = net->getXsafe();
    real_t shortfall = 0, stored = 0;
    for (uint_t i = 0; i < nTanks; i++) {
        real_t below = state[i] - reserve[i];
        if (below > 0) below = 0;
        shortfall = shortfall + std::fabs(below);
        stored = stored + std::fabs(state[i]);
    }
    economicKpi = economicKpi + pumping;
    smoothKpi = smoothKpi + variation;
    safeKpi = safeKpi + shortfall;
    networkKpi = networkKpi + stored;
}
// KPI read-outs (SmpcController.cu:1819-1859): the first two are per simulated hour (sampling time 3600 s)
real_t SmpcController::getEconomicKpi(uint_t simulationTime) { const real_t perHour = economicKpi / 3600; return perHour / simulationTime; }
real_t SmpcController::getSmoothKpi(uint_t simulationTime) { const real_t perHour = smoothKpi / 3600; return perHour / simulationTime; }
real_t SmpcController::getNetworkKpi(uint_t simulationTime) {
    const real_t *reserve = getDwnNetwork()->getXsafe();
    real_t reserveTotal = 0;
    for (uint_t i = 0; i < ptrMySmpcConfig->getNX(); i++) reserveTotal = reserveTotal + reserve[i];
    return 100 * simulationTime * reserveTotal / networkKpi;
}
real_t SmpcController::getSafetyKpi(uint_t) { return safeKpi; }

void SmpcController::initialiseAlgorithm() {
    const bool quasiNewton = ptrMyEngine->getGlobalFbeFlag() || ptrMyEngine->getNamaFlag();
    check(quasiNewton ? rn_fbe_reset(ptrMyEngine->getContext()) : rn_apg_reset(ptrMyEngine->getContext()), "initialiseAlgorithm");
}
void SmpcController::initaliseLbfgBuffer() { check(rn_fbe_reset(ptrMyEngine->getContext()), "rn_fbe_reset"); }
void SmpcController::updateLbfgsBuffer() { check(rn_update_lbfgs_buffer(ptrMyEngine->getContext()), "rn_update_lbfgs_buffer"); }
void SmpcController::twoLoopRecursionLbfgs() { check(rn_two_loop_recursion_lbfgs(ptrMyEngine->getContext()), "rn_two_loop_recursion_lbfgs"); }
void SmpcController::dualExtrapolationStep(real_t lambda) { check(rn_dual_extrapolation_step(ptrMyEngine->getContext(), lambda), "rn_dual_extrapolation_step"); }
void SmpcController::solveStep() {
    if (!factorStepFlag) initialiseSmpcController();          // SmpcController.cu:579-582
    check(rn_solve_step(ptrMyEngine->getContext()), "rn_solve_step");
}
void SmpcController::proximalFunG() { check(rn_proximal_fun_g(ptrMyEngine->getContext()), "rn_proximal_fun_g"); }
void SmpcController::computeFixedPointResidual() { check(rn_compute_fixed_point_residual(ptrMyEngine->getContext()), "rn_compute_fixed_point_residual"); }
void SmpcController::dualUpdate() { check(rn_dual_update(ptrMyEngine->getContext()), "rn_dual_update"); }
real_t SmpcController::updatePrimalInfeasibity() {
    double v = 0;
    check(rn_update_primal_infeasibility(ptrMyEngine->getContext(), &v), "rn_update_primal_infeasibility");
    return v;
}
uint_t SmpcController::algorithmGlobalFbe() {
    if (!ptrMyEngine->getGlobalFbeFlag()) throw std::logic_error("algorithmGlobalFbe: the configuration selects another algorithm");
    if (!factorStepFlag) initialiseSmpcController();
    check(rn_algorithm_fbe_nama(ptrMyEngine->getContext(), ptrMySmpcConfig->getMaxIterations(), vecPrimalInfs.data(), vecValueFbe.data(), vecTau.data()),
          "rn_algorithm_fbe_nama");
    return 1;
}
uint_t SmpcController::algorithmNama() {
    if (!ptrMyEngine->getNamaFlag()) throw std::logic_error("algorithmNama: the configuration selects another algorithm");
    if (!factorStepFlag) initialiseSmpcController();
    check(rn_algorithm_fbe_nama(ptrMyEngine->getContext(), ptrMySmpcConfig->getMaxIterations(), vecPrimalInfs.data(), vecValueFbe.data(), vecTau.data()),
          "rn_algorithm_fbe_nama");
    return 1;
}
void SmpcController::computeHessianOracalGlobalFbe() {
    if (!factorStepFlag) initialiseSmpcController();          // SmpcController.cu:899-902
    check(rn_compute_hessian_oracle(ptrMyEngine->getContext()), "rn_compute_hessian_oracle");
}
void SmpcController::updateFixedPointResidualNamaAlgorithm() { check(rn_update_fixed_point_residual_nama(ptrMyEngine->getContext()), "rn_update_fixed_point_residual_nama"); }
void SmpcController::computeGradientFbe() { check(rn_compute_gradient_fbe(ptrMyEngine->getContext()), "rn_compute_gradient_fbe"); }
void SmpcController::computeLbfgsDirection() { check(rn_compute_lbfgs_direction(ptrMyEngine->getContext()), "rn_compute_lbfgs_direction"); }
real_t SmpcController::computeLineSearchLbfgsUpdate(real_t valueFbeY) {
    double tau = 0;
    check(rn_line_search_lbfgs_update(ptrMyEngine->getContext(), valueFbeY, &tau), "rn_line_search_lbfgs_update");
    return tau;
}
real_t SmpcController::computeLineSearchAmeLbfgsUpdate(real_t valueFbeYvar) {
    double tau = 0;
    check(rn_line_search_ame_lbfgs_update(ptrMyEngine->getContext(), valueFbeYvar, &tau), "rn_line_search_ame_lbfgs_update");
    return tau;
}
real_t SmpcController::computeValueFbe() {
    if (!factorStepFlag) initialiseSmpcController();
    double v = 0;
    check(rn_compute_value_fbe(ptrMyEngine->getContext(), &v), "rn_compute_value_fbe");
    return v;
}
uint_t SmpcController::algorithmApg() {
    check(rn_algorithm_apg(ptrMyEngine->getContext(), ptrMySmpcConfig->getMaxIterations(), vecPrimalInfs.data()), "rn_algorithm_apg");
    return 1;
}
