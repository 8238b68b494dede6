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
    explicit SmpcController(string pathToConfigFile);
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
    real_t getEconomicKpi(uint_t simulationTime);           // :1808-1811
    real_t getSmoothKpi(uint_t simulationTime);             // :1817-1820
    real_t getNetworkKpi(uint_t simulationTime);            // :1826-1835
    real_t getSafetyKpi(uint_t simulationTime);             // :1841-1843
    void updateKpi(real_t *state, real_t *control);         // :1769-1802
    real_t *getPrimalInfeasibility() { return vecPrimalInfs.data(); }
    ~SmpcController();

protected:
    void dualExtrapolationStep(real_t lambda);              // :535
    void solveStep();                                       // :563
    void proximalFunG();                                    // :759
    void dualUpdate();                                      // :854
    uint_t algorithmApg();                                  // :1500
    void computeFixedPointResidual();                       // :839
    real_t updatePrimalInfeasibity();                       // :1480
    void getVector(int bufferId, real_t *host) { ptrMyEngine->getBuffer(bufferId, host); }
    void setVector(int bufferId, const real_t *host) { ptrMyEngine->setBuffer(bufferId, host); }
    void check(int rc, const char *what);

    Engine *ptrMyEngine;
    Forecaster *ptrMyForecaster;
    SmpcConfiguration *ptrMySmpcConfig;
    real_t stepSize;
    bool factorStepFlag, simulatorFlag, ownsObjects;
    std::vector<real_t> vecPrimalInfs, lastControl;
    real_t economicKpi, smoothKpi, safeKpi, networkKpi;
};

#endif
