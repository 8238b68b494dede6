// DataModel.hpp -- host data model with the reference's class surface:
//   DwnNetwork        (reference src/DwnNetwork.cuh:67-143)
//   ScenarioTree      (reference src/ScenarioTree.cuh:64-154)
//   Forecaster        (reference src/Forecaster.cuh:57-95)
//   SmpcConfiguration (reference src/SmpcConfiguration.cuh:60-181)
// Same constructors (path to a JSON file in the reference's schema), same getter names, borrowed raw pointers.
// Differences, all deliberate: values are parsed as double (the reference truncates with GetFloat()); arrays are
// sized from the JSON (the reference under-allocates nodesPerStage / nodesPerStageCumul, ScenarioTree.cu:66-75);
// a missing file throws std::runtime_error from every class (the reference exit(100)s, only ScenarioTree throws);
// SmpcConfiguration::setPreviousDemand() writes prevDemand (the reference writes prevU, SmpcConfiguration.cu:290).
#ifndef RAPIDNET_DATAMODEL_HPP_
#define RAPIDNET_DATAMODEL_HPP_

#include <vector>

#include "Configuration.h"
#include "JsonLite.hpp"

class DwnNetwork {
public:
    explicit DwnNetwork(string pathToFile);
    uint_t getNumTanks() { return nTanks; }
    uint_t getNumControls() { return nControl; }
    uint_t getNumDemands() { return nDemand; }
    uint_t getNumMixNodes() { return nMixNodes; }
    real_t *getMatA() { return matA.data(); }
    real_t *getMatB() { return matB.data(); }
    real_t *getMatGd() { return matGd.data(); }
    real_t *getMatE() { return matE.data(); }
    real_t *getMatEd() { return matEd.data(); }
    real_t *getXmin() { return vecXmin.data(); }
    real_t *getXmax() { return vecXmax.data(); }
    real_t *getXsafe() { return vecXsafe.data(); }
    real_t *getUmin() { return vecUmin.data(); }
    real_t *getUmax() { return vecUmax.data(); }
    real_t *getAlpha() { return vecCostAlpha1.data(); }
    ~DwnNetwork() {}

private:
    uint_t nTanks, nControl, nDemand, nMixNodes;
    std::vector<real_t> matA, matB, matGd, matE, matEd, vecXmin, vecXmax, vecXsafe, vecUmin, vecUmax, vecCostAlpha1;
};

class ScenarioTree {
public:
    explicit ScenarioTree(string pathToFileName);
    uint_t getPredHorizon() { return nPredHorizon; }
    uint_t getNumScenarios() { return nScenario; }
    uint_t getNumNodes() { return nNodes; }
    uint_t getNumChildrenTot() { return nChildrenTot; }
    uint_t getNumNonleafNodes() { return nNonleafNodes; }
    uint_t getFinalBranchNode();   // ScenarioTree.cu:147-156
    uint_t getFinalBranchStage();  // ScenarioTree.cu:158-167
    uint_t *getStageNodes() { return stageArray.data(); }
    uint_t *getNodesPerStage() { return nodesPerStage.data(); }
    uint_t *getNodesPerStageCumul() { return nodesPerStageCumul.data(); }
    uint_t *getLeaveArray() { return leaveArray.data(); }
    uint_t *getChildArray() { return childArray.data(); }
    uint_t *getAncestorArray() { return ancestorArray.data(); }
    uint_t *getNumChildren() { return nChildArray.data(); }
    uint_t *getNumChildrenCumul() { return nChildCumulArray.data(); }
    real_t *getProbArray() { return probNodeArray.data(); }
    real_t *getErrorDemandArray() { return errorDemandArray.data(); }
    real_t *getErrorPriceArray() { return errorPriceArray.data(); }
    uint_t getDimDemand() { return dimDemand; }
    uint_t getDimPrice() { return dimPrice; }
    ~ScenarioTree() {}

private:
    uint_t nPredHorizon, nScenario, nNodes, nChildrenTot, nNonleafNodes, dimDemand, dimPrice;
    std::vector<uint_t> stageArray, nodesPerStage, nodesPerStageCumul, leaveArray, childArray, ancestorArray, nChildArray,
        nChildCumulArray;
    std::vector<real_t> probNodeArray, errorDemandArray, errorPriceArray;
};

class Forecaster {
public:
    explicit Forecaster(string pathToFile);
    uint_t getPredHorizon() { return nPredHorizon; }
    uint_t getSimHorizon() { return simHorizon; }
    uint_t getDimDemand() { return dimDemand; }
    uint_t getDimPrice() { return dimPrices; }
    real_t *getNominalDemand() { return nominalDemand.data(); }
    real_t *getNominalPrices() { return nominalPrice.data(); }
    // members 4+2t / 5+2t of the file, in file order (Forecaster.cu:93-119); return 1 on success, 0 past the end
    virtual uint_t predictDemand(uint_t simTime);
    virtual uint_t predictPrices(uint_t simTime);
    virtual ~Forecaster() {}

private:
    uint_t simHorizon, nPredHorizon, dimDemand, dimPrices;
    std::vector<real_t> nominalDemand, nominalPrice;
    jsonlite::Document jsonDocument;
};

class SmpcConfiguration {
public:
    explicit SmpcConfiguration(string pathToFile);
    uint_t getNX() { return NX; }
    uint_t getNU() { return NU; }
    uint_t getND() { return ND; }
    uint_t getNV() { return NV; }
    uint_t getLbfgsBufferSize() { return lbfgsBufferSize; }
    real_t *getMatL() { return matL.data(); }
    real_t *getMatLhat() { return matLhat.data(); }
    real_t *getMatPrcndDiag() { return matDiagPrecnd.data(); }
    real_t *getCostW() { return matCostW.data(); }
    real_t *getCurrentX() { return currentX.data(); }
    real_t *getPrevU() { return prevU.data(); }
    real_t *getPrevDemand() { return prevDemand.data(); }
    real_t getPenaltyState() { return penaltyStateX; }
    real_t getPenaltySafety() { return penaltySafetyX; }
    uint_t getMaxIterations() { return maxIteration; }
    real_t getStepSize() { return stepSize; }
    string getPathToControllerConfig() { return pathToConfiguration; }
    string getPathToNetwork() { return pathToNetwork; }
    string getPathToScenarioTree() { return pathToScenarioTree; }
    string getPathToForecaster() { return pathToForecaster; }
    real_t getWeightEconomical() { return weightPrice; }
    string getOptimisationAlgorithm() { return algorithmName; }
    // "operatorMode" (optional key, not in the reference's files): "auto" (default), "dense" or "structured" -- Engine.hpp, setOperatorMode
    string getOperatorMode() { return operatorMode; }
    void setCurrentState();     // re-read from the configuration file (SmpcConfiguration.cu:240-256)
    void setPreviousControl();  // :261-277
    void setPreviousDemand();   // :283-299
    void setCurrentState(real_t *state);
    void setPreviousControl(real_t *control);
    void setpreviousdemand(real_t *demand);
    ~SmpcConfiguration() {}

private:
    uint_t NX, NU, ND, NV, lbfgsBufferSize, maxIteration;
    std::vector<real_t> matL, matLhat, matCostW, matDiagPrecnd, currentX, prevU, prevDemand;
    real_t penaltyStateX, penaltySafetyX, stepSize, weightPrice, weightSmooth, weightSafety;
    string pathToConfiguration, pathToNetwork, pathToScenarioTree, pathToForecaster, algorithmName, operatorMode;
};

#endif
