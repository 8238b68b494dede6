// SmpcController.hpp -- the reference's SmpcController class surface (src/SmpcController.cuh:42-259) over the C-ABI.
// Public methods keep the reference's names, arguments and return values; the protected step methods (the seam the
// reference's known-answer tests subclass into, TestSmpcController.cuh:80) forward one-to-one to rn_* entry points.
// The reference's protected raw device vectors (devVecXi ...) are replaced by getVector / setVector with RN_BUF_* ids.
#ifndef RAPIDNET_SMPCCONTROLLER_HPP_
#define RAPIDNET_SMPCCONTROLLER_HPP_

#include "Engine.hpp"

class SmpcController {
public:
    SmpcController(Forecaster *myForecaster, Engine *myEngine, SmpcConfiguration *mySmpcConfig);
    // operatorMode: Engine.hpp (RN_OPS_AUTO / _DENSE / _STRUCTURED; -1 = the configuration file's optional "operatorMode" key, absent: auto)
    explicit SmpcController(string pathToConfigFile, int operatorMode = -1);
    // Multi-GPU (new): rank `rank` of `nranks`, see Engine's sharded constructor.  Same configuration file on every rank.
    SmpcController(string pathToConfigFile, int rank, int nranks, const void *ncclUniqueId128, int device = 0, int precision = RN_F64, int cutStage = 0,
                   int operatorMode = -1);
    void initialiseSmpcController();                        // SmpcController.cu:476-487
    void controllerSmpc();                                  // :1593-1599
    uint_t controlAction(real_t *u);                        // :1607-1625  (1 = ok)
    uint_t controlAction(std::fstream &controlOutputJson);  // :1633-1667
    DwnNetwork *getDwnNetwork() { return ptrMyEngine->getDwnNetwork(); }
    ScenarioTree *getScenarioTree() { return ptrMyEngine->getScenarioTree(); }
    SmpcConfiguration *getSmpcConfiguration() { return ptrMySmpcConfig; }
    Forecaster *getForecaster() { return ptrMyForecaster; }
    Engine *getEngine() { return ptrMyEngine; }
    void moveForewardInTime();                              // :1679-1716
    // in-built simulator of moveForewardInTime: false (default) = the reference as written, x+ = x + B u (its disturbance
    // term lands in the x of node 0 instead of the state update, :1695); true = x+ = x + e_0 + B u (DwnNetwork.cuh:41-57)
    void setSimulatorDisturbance(bool on) { simulatorDisturbance = on; }
    void setSimulatorFlag(bool inBuilt) { simulatorFlag = inBuilt; }   // false: external simulator, the state is re-read from the configuration file
    real_t getEconomicKpi(uint_t simulationTime);           // :1808-1811
    real_t getSmoothKpi(uint_t simulationTime);             // :1817-1820
    real_t getNetworkKpi(uint_t simulationTime);            // :1826-1835
    real_t getSafetyKpi(uint_t simulationTime);             // :1841-1843
    void updateKpi(real_t *state, real_t *control);         // :1769-1802
    real_t *getPrimalInfeasibility() { return vecPrimalInfs.data(); }
    real_t *getValueFbe() { return vecValueFbe.data(); }    // :1883
    real_t *getVecTau() { return vecTau.data(); }
    ~SmpcController();

protected:
    void initialiseAlgorithm();                             // :420-450  zero the iterates of the selected algorithm
    void initialiseAlgorithmSpecificData() {}               // :494-528  rebuilds device pointer tables: nothing to do here
    void initaliseLbfgBuffer();                             // :453-468
    void updateLbfgsBuffer();                               // :1103
    void twoLoopRecursionLbfgs();                           // :1175
    void dualExtrapolationStep(real_t lambda);              // :535
    void solveStep();                                       // :563
    void proximalFunG();                                    // :759
    void dualUpdate();                                      // :854
    uint_t algorithmApg();                                  // :1500
    void computeFixedPointResidual();                       // :839
    real_t updatePrimalInfeasibity();                       // :1480
    // global FBE / NAMA (selected by "algorithmName" of the controller configuration, Engine.cu:151-163)
    uint_t algorithmGlobalFbe();                            // :1529
    uint_t algorithmNama();                                 // :1559
    void computeHessianOracalGlobalFbe();                   // :884
    void updateFixedPointResidualNamaAlgorithm();           // :1060
    void computeGradientFbe();                              // :1077
    void computeLbfgsDirection();                           // :1234  (updateLbfgsBuffer :1103 + twoLoopRecursionLbfgs :1175)
    real_t computeLineSearchLbfgsUpdate(real_t valueFbeY);  // :1242
    real_t computeLineSearchAmeLbfgsUpdate(real_t valueFbeYvar);   // :1311
    real_t computeValueFbe();                               // :1416
    // lbfgsBufferCol / lbfgsBufferMemory / lbfgsBufferHessian / lbfgsBufferRho and the columns of devLbfgsBufferMatS / MatY
    void getLbfgsState(int &col, int &mem, real_t &H, real_t *rho) { check(rn_lbfgs_state(ptrMyEngine->getContext(), 0, &col, &mem, &H, rho), "rn_lbfgs_state"); }
    void setLbfgsState(int col, int mem, real_t H, real_t *rho) { check(rn_lbfgs_state(ptrMyEngine->getContext(), 1, &col, &mem, &H, rho), "rn_lbfgs_state"); }
    void getLbfgsColumn(int which, int col, real_t *host, size_t n) { check(rn_lbfgs_column(ptrMyEngine->getContext(), 0, which, col, host, n), "rn_lbfgs_column"); }
    void setLbfgsColumn(int which, int col, const real_t *host, size_t n) { check(rn_lbfgs_column(ptrMyEngine->getContext(), 1, which, col, const_cast<real_t *>(host), n), "rn_lbfgs_column"); }
    void getVector(int bufferId, real_t *host) { ptrMyEngine->getBuffer(bufferId, host); }
    void setVector(int bufferId, const real_t *host) { ptrMyEngine->setBuffer(bufferId, host); }
    void check(int rc, const char *what);
    bool deviceMemoryChanged(const size_t before[4]);       // the leak check of :1612-1623 (prints the reference's message)

    Engine *ptrMyEngine;
    Forecaster *ptrMyForecaster;
    SmpcConfiguration *ptrMySmpcConfig;
    real_t stepSize;
    bool factorStepFlag, simulatorFlag, ownsObjects, simulatorDisturbance = false;
    std::vector<real_t> vecPrimalInfs, vecValueFbe, vecTau, lastControl;
    real_t economicKpi, smoothKpi, safeKpi, networkKpi;
};

#endif
