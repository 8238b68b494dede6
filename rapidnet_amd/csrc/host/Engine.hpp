// Engine.hpp -- the reference's Engine class surface (src/Engine.cuh:49-358) over the C-ABI of include/rapidnet.h.
// The reference's Engine owns ~60 raw device arrays and hands out device pointers; here the device state lives behind
// an opaque rn_ctx and the getters copy to the host on demand (node-major, the reference's layout):
//   reference getter returning a device pointer        here
//   getVecUhat() / getVecBeta() / getVecE() ...         getBuffer(RN_BUF_UHAT, host) ...
//   getMatPhi() / getPtrMatPhi()[node] ...              getOperator(RN_OP_PHI, node, host) ...
#ifndef RAPIDNET_ENGINE_HPP_
#define RAPIDNET_ENGINE_HPP_

#include "../../../include/rapidnet.h"
#include "DataModel.hpp"

class Engine {
public:
    // reference: Engine(SmpcConfiguration*) news its own network and tree from the paths in the configuration
    // (Engine.cu:126-132).  precision: RN_F64 (default) or RN_F32; device: HIP device ordinal.
    explicit Engine(SmpcConfiguration *smpcConfig, int precision = RN_F64, int device = 0);
    Engine(DwnNetwork *network, ScenarioTree *scenarioTree, SmpcConfiguration *smpcConfig, int precision = RN_F64, int device = 0);
    void eliminateInputDistubanceCoupling(real_t *nominalDemand, real_t *nominalPrices);  // Engine.cu:1147
    void updateStateControl(real_t *currentX, real_t *prevU, real_t *prevDemand);        // Engine.cu:1300
    void factorStep();                                                                    // Engine.cu:671
    // Engine::calculateMatLandMatLhat (Engine.cu:466-669): L = null(E), Lhat = -pinv(E) Ed from the network's E, Ed.
    // By default the factor step uses matL / matLhat of the controller configuration (they are already there,
    // SmpcConfiguration.cu:59-68); after this call it uses the computed ones.  Any orthonormal basis of null(E) gives
    // the same x, u and duals.
    void calculateMatLandMatLhat();
    real_t *getMatL() { return useComputedL ? computedL.data() : ptrMySmpcConfig->getMatL(); }
    real_t *getMatLhat() { return useComputedL ? computedLhat.data() : ptrMySmpcConfig->getMatLhat(); }
    void setWarmStart(bool on);
    ScenarioTree *getScenarioTree() { return ptrMyScenarioTree; }
    DwnNetwork *getDwnNetwork() { return ptrMyNetwork; }
    SmpcConfiguration *getSmpcConfiguration() { return ptrMySmpcConfig; }
    bool getPriceUncertainty() { return priceUncertaintyFlag; }
    bool getDemandUncertantiy() { return demandUncertaintyFlag; }
    bool getApgFlag() { return apgFlag; }
    bool getGlobalFbeFlag() { return globalFbeFlag; }
    bool getNamaFlag() { return namaFlag; }
    void setPriceUncertaintyFlag(bool inputFlag);
    void setDemandUncertaintyFlag(bool inputFlag);
    // device state access (replaces the raw device-pointer getters)
    rn_ctx *getContext() { return ctx; }
    size_t getBufferSize(int bufferId);
    void getBuffer(int bufferId, real_t *host);
    void setBuffer(int bufferId, const real_t *host);
    void getOperator(int opId, uint_t node, real_t *host, size_t n);
    ~Engine();

private:
    void create(int precision, int device);
    void check(int rc, const char *what);
    DwnNetwork *ptrMyNetwork;
    ScenarioTree *ptrMyScenarioTree;
    SmpcConfiguration *ptrMySmpcConfig;
    rn_ctx *ctx;
    bool priceUncertaintyFlag, demandUncertaintyFlag, apgFlag, globalFbeFlag, namaFlag;
    bool useComputedL = false;
    std::vector<real_t> computedL, computedLhat;
};

#endif
