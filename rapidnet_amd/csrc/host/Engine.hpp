// Engine.hpp -- the reference's Engine class surface (src/Engine.cuh:49-358) over the C-ABI of include/rapidnet.h.
// The reference's Engine owns ~60 raw device arrays and hands out device pointers; here the device state lives behind
// an opaque rn_ctx and the getters copy to the host on demand (node-major, the reference's layout):
//   reference getter returning a device pointer        here
//   getVecUhat() / getVecBeta() / getVecE() ...         getBuffer(RN_BUF_UHAT, host) ...  -- or, for the arrays the library keeps in the
//                                                       reference's own node-major layout, the raw device pointer under the
//                                                       reference's name: getVecUhat(), getVecBeta(), getVecE(), getPriceAlpha()
//                                                       (element type = the engine's precision: getDevicePrecision())
//   getMatPhi() / getPtrMatPhi()[node] ...              getOperator(RN_OP_PHI, node, host) ...  (the blocks are stored interleaved: DESIGN.md section 4)
#ifndef RAPIDNET_ENGINE_HPP_
#define RAPIDNET_ENGINE_HPP_

#include "../../../include/rapidnet.h"
#include "DataModel.hpp"

class Engine {
public:
    // reference: Engine(SmpcConfiguration*) news its own network and tree from the paths in the configuration
    // (Engine.cu:126-132).  precision: RN_F64 (default) or RN_F32; device: HIP device ordinal.
    // operatorMode: how the per-node operator blocks of the factor step (Engine.cu:166-189, :721-745) are kept.  RN_OPS_AUTO: never
    // materialised while they are the factor step's own -- block = shared matrix x stage diagonal x power of p_i, applied as shared-operator
    // products: identical iterates, 32 instead of 335 ms per 500-iteration control step on the 493-scenario tree -- and dense from the
    // first setOperator() on; RN_OPS_DENSE: the reference's storage (one dense block per node, streamed every iteration); RN_OPS_STRUCTURED:
    // never any block.  -1 (default): the configuration file's optional "operatorMode" key ("auto" | "dense" | "structured"; absent: auto).
    explicit Engine(SmpcConfiguration *smpcConfig, int precision = RN_F64, int device = 0, int operatorMode = -1);
    Engine(DwnNetwork *network, ScenarioTree *scenarioTree, SmpcConfiguration *smpcConfig, int precision = RN_F64, int device = 0, int operatorMode = -1);
    // Multi-GPU (new; the reference is single-GPU): rank `rank` of `nranks`, one process and one GPU per rank.  The engine is
    // given the FULL network / tree / configuration exactly as above; the library keeps this rank's subtrees plus the
    // replicated crown (rn_create_sharded: partition below `cutStage`, 0 = the most balanced cut), creates the RCCL
    // communicator from `ncclUniqueId128` (rn_comm_unique_id on rank 0, distributed by the caller; NULL: no communicator, the
    // exchange is a test's job) and all-reduces the cut parents' children sums once per APG iteration.  Everything else --
    // factorStep, updateStateControl, eliminateInputDistubanceCoupling, the controller -- is called as on one GPU; node-major
    // buffers are local (getNumLocalNodes() nodes; getGlobalNodes() maps them to nodes of the full tree).
    Engine(SmpcConfiguration *smpcConfig, int precision, int device, int rank, int nranks, const void *ncclUniqueId128, int cutStage = 0, int operatorMode = -1);
    Engine(DwnNetwork *network, ScenarioTree *scenarioTree, SmpcConfiguration *smpcConfig, int precision, int device, int rank, int nranks,
           const void *ncclUniqueId128, int cutStage = 0, int operatorMode = -1);
    void setOperatorMode(int mode);                  // RN_OPS_AUTO / _DENSE / _STRUCTURED; before factorStep()
    int getOperatorMode();                           // what the engine runs: RN_OPS_DENSE or RN_OPS_STRUCTURED
    // a per-node block handed in by the caller (the reference: write through getMatPhi() / getPtrMatPhi()[node] ..., Engine.cuh:170-230);
    // RN_OP_PHI, _PSI, _D, _F, col-major nv x (2nx | nu); after factorStep()
    void setOperator(int opId, uint_t node, const real_t *host, size_t n);
    int getRank() { return myRank; }
    int getNumRanks() { return numRanks; }
    uint_t getNumLocalNodes();                       // nodes this rank holds (= the tree's node count on one GPU)
    std::vector<int> getGlobalNodes();               // index in the full tree of every local node
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
    void getBufferRange(int bufferId, size_t first, size_t n, real_t *host);         // elements [first, first + n) only
    void setBufferRange(int bufferId, size_t first, size_t n, const real_t *host);
    void getOperator(int opId, uint_t node, real_t *host, size_t n);
    // Raw device pointers under the reference's names (Engine.cuh:108-318) for the arrays whose device layout IS the reference's:
    // [node][dim], valid until the engine is destroyed, contents as of the last completed call (rn_synchronize / rn_stream).  The
    // element type is double for RN_F64 engines (the default) and float for RN_F32 ones -- the reference's real_t is float, this
    // host API's is double, so the pointers are untyped and getDevicePrecision() says which.
    int getDevicePrecision();                           // RN_F64 or RN_F32
    void *getDevicePointer(int bufferId, size_t *n = nullptr);      // RN_BUF_X, _U, _V, _UHAT, _E, _BETA, _ALPHA, _XMIN ... _UMAX (others: std::runtime_error)
    void *getVecUhat() { return getDevicePointer(RN_BUF_UHAT); }   // Engine.cuh: getVecUhat
    void *getVecBeta() { return getDevicePointer(RN_BUF_BETA); }   //             getVecBeta
    void *getVecE() { return getDevicePointer(RN_BUF_E); }         //             getVecE
    void *getPriceAlpha() { return getDevicePointer(RN_BUF_ALPHA); }   //         getPriceAlpha
    // the scaled bounds (Engine.cuh:294-314; after factorStep()): node-major copies in the reference's layout, made by the first call of any of the five
    void *getSysXmin() { return getDevicePointer(RN_BUF_XMIN); }   //             getSysXmin
    void *getSysXmax() { return getDevicePointer(RN_BUF_XMAX); }   //             getSysXmax
    void *getSysXs() { return getDevicePointer(RN_BUF_XS); }       //             getSysXs
    void *getSysUmin() { return getDevicePointer(RN_BUF_UMIN); }   //             getSysUmin
    void *getSysUmax() { return getDevicePointer(RN_BUF_UMAX); }   //             getSysUmax
    void *getVecX() { return getDevicePointer(RN_BUF_X); }         // SmpcController.cuh: devVecX (protected there; public here for GPU-side consumers)
    void *getVecU() { return getDevicePointer(RN_BUF_U); }         //                     devVecU
    void *getVecV() { return getDevicePointer(RN_BUF_V); }         //                     devVecV
    ~Engine();

private:
    void create(int precision, int device, int operatorMode, int rank = 0, int nranks = 1, const void *id128 = nullptr, int cutStage = 0);
    int myRank = 0, numRanks = 1;
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
