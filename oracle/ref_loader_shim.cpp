// ref_loader_shim.cpp -- TEST INFRASTRUCTURE.  C accessors around the REFERENCE's own host loader classes
// (DwnNetwork, ScenarioTree, Forecaster, SmpcConfiguration).  Those four classes are plain C++ despite their .cu
// suffix, so oracle/build_ref.sh compiles them where they lie under /root/reference/src (nothing is copied) together
// with this shim into oracle/_ref/libref_loaders.so.  Used by tests/test_ref_loaders.py to check that the JSON files
// this repo reads and writes go through the reference's parsers unchanged.
#include <cstring>

#include "DwnNetwork.cuh"
#include "Forecaster.cuh"
#include "ScenarioTree.cuh"
#include "SmpcConfiguration.cuh"

extern "C" {

void *ref_network_new(const char *path) { return new DwnNetwork(path); }
void ref_network_dims(void *h, int *out) {
    DwnNetwork *n = (DwnNetwork *)h;
    out[0] = n->getNumTanks(); out[1] = n->getNumControls(); out[2] = n->getNumDemands(); out[3] = n->getNumMixNodes();
}
const float *ref_network_array(void *h, const char *name) {
    DwnNetwork *n = (DwnNetwork *)h;
    if (!strcmp(name, "matA")) return n->getMatA();
    if (!strcmp(name, "matB")) return n->getMatB();
    if (!strcmp(name, "matGd")) return n->getMatGd();
    if (!strcmp(name, "matE")) return n->getMatE();
    if (!strcmp(name, "matEd")) return n->getMatEd();
    if (!strcmp(name, "vecXmin")) return n->getXmin();
    if (!strcmp(name, "vecXmax")) return n->getXmax();
    if (!strcmp(name, "vecXsafe")) return n->getXsafe();
    if (!strcmp(name, "vecUmin")) return n->getUmin();
    if (!strcmp(name, "vecUmax")) return n->getUmax();
    if (!strcmp(name, "costAlpha1")) return n->getAlpha();
    return 0;
}

void *ref_tree_new(const char *path) { return new ScenarioTree(path); }
void ref_tree_dims(void *h, int *out) {
    ScenarioTree *t = (ScenarioTree *)h;
    out[0] = t->getPredHorizon(); out[1] = t->getNumScenarios(); out[2] = t->getNumNodes(); out[3] = t->getNumNonleafNodes();
    out[4] = t->getNumChildrenTot(); out[5] = t->getFinalBranchNode(); out[6] = t->getFinalBranchStage();
}
const int *ref_tree_int_array(void *h, const char *name) {
    ScenarioTree *t = (ScenarioTree *)h;
    if (!strcmp(name, "stages")) return t->getStageNodes();
    if (!strcmp(name, "nodesPerStage")) return t->getNodesPerStage();
    if (!strcmp(name, "nodesPerStageCumul")) return t->getNodesPerStageCumul();
    if (!strcmp(name, "leaves")) return t->getLeaveArray();
    if (!strcmp(name, "children")) return t->getChildArray();
    if (!strcmp(name, "ancestor")) return t->getAncestorArray();
    if (!strcmp(name, "nChildren")) return t->getNumChildren();
    if (!strcmp(name, "nChildrenCumul")) return t->getNumChildrenCumul();
    return 0;
}
const float *ref_tree_array(void *h, const char *name) {
    ScenarioTree *t = (ScenarioTree *)h;
    if (!strcmp(name, "probNode")) return t->getProbArray();
    if (!strcmp(name, "errorDemandNode")) return t->getErrorDemandArray();
    if (!strcmp(name, "errorPriceNode")) return t->getErrorPriceArray();
    return 0;
}

void *ref_config_new(const char *path) { return new SmpcConfiguration(path); }
void ref_config_dims(void *h, int *out) {
    SmpcConfiguration *c = (SmpcConfiguration *)h;
    out[0] = c->getNX(); out[1] = c->getNU(); out[2] = c->getND(); out[3] = c->getNV(); out[4] = c->getMaxIterations();
}
void ref_config_scalars(void *h, float *out) {
    SmpcConfiguration *c = (SmpcConfiguration *)h;
    out[0] = c->getStepSize(); out[1] = c->getPenaltyState(); out[2] = c->getPenaltySafety();
}
const float *ref_config_array(void *h, const char *name) {
    SmpcConfiguration *c = (SmpcConfiguration *)h;
    if (!strcmp(name, "matL")) return c->getMatL();
    if (!strcmp(name, "matLhat")) return c->getMatLhat();
    if (!strcmp(name, "costW")) return c->getCostW();
    if (!strcmp(name, "matDiagPrecnd")) return c->getMatPrcndDiag();
    if (!strcmp(name, "currentX")) return c->getCurrentX();
    if (!strcmp(name, "prevU")) return c->getPrevU();
    if (!strcmp(name, "prevDemand")) return c->getPrevDemand();
    return 0;
}
const char *ref_config_string(void *h, const char *name) {
    static std::string s;
    SmpcConfiguration *c = (SmpcConfiguration *)h;
    if (!strcmp(name, "pathToNetwork")) s = c->getPathToNetwork();
    else if (!strcmp(name, "pathToScenarioTree")) s = c->getPathToScenarioTree();
    else if (!strcmp(name, "pathToForecaster")) s = c->getPathToForecaster();
    else if (!strcmp(name, "algorithmName")) s = c->getOptimisationAlgorithm();
    else s = "";
    return s.c_str();
}

void *ref_forecaster_new(const char *path) { return new Forecaster(path); }
void ref_forecaster_dims(void *h, int *out) {
    Forecaster *f = (Forecaster *)h;
    out[0] = f->getPredHorizon(); out[1] = f->getSimHorizon(); out[2] = f->getDimDemand(); out[3] = f->getDimPrice();
}
int ref_forecaster_predict(void *h, int t) { Forecaster *f = (Forecaster *)h; return f->predictDemand(t) && f->predictPrices(t); }
const float *ref_forecaster_demand(void *h) { return ((Forecaster *)h)->getNominalDemand(); }
const float *ref_forecaster_prices(void *h) { return ((Forecaster *)h)->getNominalPrices(); }
}
