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

SmpcController::SmpcController(string pathToConfigFile) : factorStepFlag(false), simulatorFlag(true), ownsObjects(true) {
    ptrMySmpcConfig = new SmpcConfiguration(pathToConfigFile);          // SmpcController.cu:78-81
    ptrMyForecaster = new Forecaster(ptrMySmpcConfig->getPathToForecaster());
    ptrMyEngine = new Engine(ptrMySmpcConfig);
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

uint_t SmpcController::controlAction(real_t *u) {
    if (!factorStepFlag) { ptrMyEngine->factorStep(); factorStepFlag = true; }
    const int rc = rn_control_action(ptrMyEngine->getContext(), ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(),
                                     ptrMySmpcConfig->getPrevDemand(), ptrMyForecaster->getNominalDemand(),
                                     ptrMyForecaster->getNominalPrices(), ptrMySmpcConfig->getMaxIterations(), 0, u);
    if (rc != RN_OK) { std::cerr << "controlAction: " << rn_last_error(ptrMyEngine->getContext()) << std::endl; return 0; }
    for (uint_t i = 0; i < ptrMySmpcConfig->getNU(); i++) lastControl[i] = u[i];
    return 1;
}

uint_t SmpcController::controlAction(std::fstream &out) {
    if (!out.is_open()) return 0;
    if (!factorStepFlag) { ptrMyEngine->factorStep(); factorStepFlag = true; }
    const uint_t nu = ptrMySmpcConfig->getNU();
    std::vector<real_t> u(nu);
    const int rc = rn_control_action(ptrMyEngine->getContext(), ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getPrevU(),
                                     ptrMySmpcConfig->getPrevDemand(), ptrMyForecaster->getNominalDemand(),
                                     ptrMyForecaster->getNominalPrices(), ptrMySmpcConfig->getMaxIterations(), 1 /* projectionBox :1649 */,
                                     u.data());
    if (rc != RN_OK) { std::cerr << "controlAction: " << rn_last_error(ptrMyEngine->getContext()) << std::endl; return 0; }
    out << "\"control\" : [";                                   // same (not quite JSON) format as :1651-1658
    for (uint_t i = 0; i < nu; i++) out << u[i] << ", ";
    out << "]" << std::endl;
    lastControl = u;
    return 1;
}

// SmpcController::moveForewardInTime (SmpcController.cu:1679-1716).  In-built simulator: x+ = x + e_root + B u with the
// control returned by the last controlAction, KPIs updated, then state / previous control / previous demand shifted.
// (The reference's version adds e to devVecX instead of the state update, :1695 -- so its simulated plant ignores the
// demand; here the plant equation x+ = x + B u + Gd d of DwnNetwork.cuh:41-57 is applied.)
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
    std::vector<real_t> e(ptrMyEngine->getBufferSize(RN_BUF_E));
    ptrMyEngine->getBuffer(RN_BUF_E, e.data());
    std::vector<real_t> x(ptrMySmpcConfig->getCurrentX(), ptrMySmpcConfig->getCurrentX() + nx);
    const real_t *B = net->getMatB();
    for (uint_t i = 0; i < nx; i++) {
        real_t s = x[i] + e[i];
        for (uint_t j = 0; j < nu; j++) s += B[i + (size_t)j * nx] * lastControl[j];
        x[i] = s;
    }
    updateKpi(x.data(), lastControl.data());                  // uses the still-unshifted previous control (:1706)
    ptrMySmpcConfig->setCurrentState(x.data());
    ptrMySmpcConfig->setPreviousControl(lastControl.data());
    std::vector<real_t> d(ptrMyForecaster->getNominalDemand(), ptrMyForecaster->getNominalDemand() + nd);
    ptrMySmpcConfig->setpreviousdemand(d.data());
}

// SmpcController::updateKpi (SmpcController.cu:1769-1802)
void SmpcController::updateKpi(real_t *state, real_t *control) {
    const uint_t nx = ptrMySmpcConfig->getNX(), nu = ptrMySmpcConfig->getNU();
    const real_t *safeX = ptrMyEngine->getDwnNetwork()->getXsafe();
    const real_t *constantPrice = ptrMyEngine->getDwnNetwork()->getAlpha();
    const real_t *variablePrice = ptrMyForecaster->getNominalPrices();
    const real_t *previousControl = ptrMySmpcConfig->getPrevU();
    const real_t weightEconomic = ptrMySmpcConfig->getWeightEconomical();
    real_t ecoKpi = 0, smKpi = 0, saKpi = 0, netKpi = 0;
    for (uint_t i = 0; i < nu; i++) {
        ecoKpi += weightEconomic * (constantPrice[i] + variablePrice[i]) * std::fabs(control[i]);
        const real_t dU = previousControl[i] - control[i];
        smKpi += dU * dU;
    }
    for (uint_t i = 0; i < nx; i++) {
        real_t level = state[i] - safeX[i];
        if (level > 0) level = 0;
        saKpi += std::fabs(level);
        netKpi += std::fabs(state[i]);
    }
    economicKpi += ecoKpi; smoothKpi += smKpi; safeKpi += saKpi; networkKpi += netKpi;
}
real_t SmpcController::getEconomicKpi(uint_t simulationTime) { return economicKpi / 3600 / simulationTime; }
real_t SmpcController::getSmoothKpi(uint_t simulationTime) { return smoothKpi / 3600 / simulationTime; }
real_t SmpcController::getNetworkKpi(uint_t simulationTime) {
    real_t safeLevelNorm = 0;
    for (uint_t i = 0; i < ptrMySmpcConfig->getNX(); i++) safeLevelNorm += getDwnNetwork()->getXsafe()[i];
    return 100 * simulationTime * safeLevelNorm / networkKpi;
}
real_t SmpcController::getSafetyKpi(uint_t) { return safeKpi; }

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
