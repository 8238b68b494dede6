// DataModel.cpp -- JSON loaders of the host data model (see DataModel.hpp for the reference map).
#include "DataModel.hpp"

namespace {
uint_t scalarInt(const jsonlite::Document &d, const char *key) {
    const jsonlite::Value &a = d[key];
    _ASSERT(a.IsArray() || a.kind == jsonlite::Value::NUMBER);
    _ASSERT(a.Size() >= 1);
    return (uint_t)a[0];
}
real_t scalarReal(const jsonlite::Document &d, const char *key) {
    const jsonlite::Value &a = d[key];
    _ASSERT(a.Size() >= 1);
    return a[0];
}
void readReal(const jsonlite::Document &d, const char *key, size_t expected, std::vector<real_t> &out) {
    const jsonlite::Value &a = d[key];
    _ASSERT(a.IsArray());
    if (a.Size() != expected)
        throw std::logic_error(string("JSON member \"") + key + "\" has " + std::to_string(a.Size()) + " entries, expected " +
                               std::to_string(expected));
    out = a.arr;
}
void readInt(const jsonlite::Document &d, const char *key, std::vector<uint_t> &out) {
    const jsonlite::Value &a = d[key];
    _ASSERT(a.IsArray());
    out.resize(a.Size());
    for (size_t i = 0; i < a.Size(); i++) out[i] = (uint_t)a[i];
}
}  // namespace

// ---- DwnNetwork (reference src/DwnNetwork.cu:30-118) -------------------------------------------------------------------
DwnNetwork::DwnNetwork(string pathToFile) {
    jsonlite::Document doc(pathToFile);
    nTanks = scalarInt(doc, "nx");
    nControl = scalarInt(doc, "nu");
    nDemand = scalarInt(doc, "nd");
    nMixNodes = scalarInt(doc, "ne");
    const size_t nx = nTanks, nu = nControl, nd = nDemand, ne = nMixNodes;
    readReal(doc, "matA", nx * nx, matA);
    readReal(doc, "matB", nx * nu, matB);
    readReal(doc, "matGd", nx * nd, matGd);
    readReal(doc, "matE", ne * nu, matE);
    readReal(doc, "matEd", ne * nd, matEd);
    readReal(doc, "vecXmin", nx, vecXmin);
    readReal(doc, "vecXmax", nx, vecXmax);
    readReal(doc, "vecXsafe", nx, vecXsafe);
    readReal(doc, "vecUmin", nu, vecUmin);
    readReal(doc, "vecUmax", nu, vecUmax);
    readReal(doc, "costAlpha1", nu, vecCostAlpha1);
}

// ---- ScenarioTree (reference src/ScenarioTree.cu:32-128) ---------------------------------------------------------------
ScenarioTree::ScenarioTree(string pathToFileName) {
    jsonlite::Document doc(pathToFileName);
    nPredHorizon = scalarInt(doc, "N");
    nScenario = scalarInt(doc, "K");
    nNodes = scalarInt(doc, "nodes");
    nNonleafNodes = scalarInt(doc, "nNonLeafNodes");
    nChildrenTot = scalarInt(doc, "nChildrenTot");
    dimDemand = scalarInt(doc, "dimDemand");
    dimPrice = scalarInt(doc, "dimPrice");
    readInt(doc, "stages", stageArray);
    readInt(doc, "nodesPerStage", nodesPerStage);
    readInt(doc, "nodesPerStageCumul", nodesPerStageCumul);
    readInt(doc, "leaves", leaveArray);
    readInt(doc, "children", childArray);
    readInt(doc, "ancestor", ancestorArray);
    readInt(doc, "nChildren", nChildArray);
    readInt(doc, "nChildrenCumul", nChildCumulArray);
    readReal(doc, "probNode", (size_t)nNodes, probNodeArray);
    readReal(doc, "errorDemandNode", (size_t)nNodes * dimDemand, errorDemandArray);
    readReal(doc, "errorPriceNode", (size_t)nNodes * dimPrice, errorPriceArray);
    _ASSERT((uint_t)stageArray.size() == nNodes && (uint_t)ancestorArray.size() == nNodes);
    _ASSERT((uint_t)nodesPerStage.size() >= nPredHorizon + 1 && (uint_t)nodesPerStageCumul.size() >= nPredHorizon + 2);
    _ASSERT((uint_t)nChildCumulArray.size() == nNodes && (uint_t)nChildArray.size() == nNonleafNodes);
    _ASSERT((uint_t)leaveArray.size() == nScenario);
}
uint_t ScenarioTree::getFinalBranchNode() {
    for (uint_t i = 0; i < nPredHorizon - 1; i++)
        if (nodesPerStage[i] == nodesPerStage[i + 1]) return nodesPerStageCumul[i + 1];
    return 0;
}
uint_t ScenarioTree::getFinalBranchStage() {
    for (uint_t i = 0; i < nPredHorizon - 1; i++)
        if (nodesPerStage[i] == nodesPerStage[i + 1]) return i;
    return 0;
}

// ---- Forecaster (reference src/Forecaster.cu:26-119) -------------------------------------------------------------------
Forecaster::Forecaster(string pathToFile) : jsonDocument(pathToFile) {
    nPredHorizon = scalarInt(jsonDocument, "N");
    simHorizon = scalarInt(jsonDocument, "simHorizon");
    dimDemand = scalarInt(jsonDocument, "dimDemand");
    dimPrices = scalarInt(jsonDocument, "dimPrices");
    nominalDemand.assign((size_t)dimDemand * nPredHorizon, 0.0);
    nominalPrice.assign((size_t)dimPrices * nPredHorizon, 0.0);
}
uint_t Forecaster::predictDemand(uint_t simTime) {
    const size_t idx = 4 + 2 * (size_t)simTime;
    if (idx >= jsonDocument.MemberCount()) return 0;
    const jsonlite::Value &v = jsonDocument.MemberValue(idx);
    _ASSERT(v.Size() <= nominalDemand.size());
    for (size_t i = 0; i < v.Size(); i++) nominalDemand[i] = v[i];
    return 1;
}
uint_t Forecaster::predictPrices(uint_t simTime) {
    const size_t idx = 5 + 2 * (size_t)simTime;
    if (idx >= jsonDocument.MemberCount()) return 0;
    const jsonlite::Value &v = jsonDocument.MemberValue(idx);
    _ASSERT(v.Size() <= nominalPrice.size());
    for (size_t i = 0; i < v.Size(); i++) nominalPrice[i] = v[i];
    return 1;
}

// ---- SmpcConfiguration (reference src/SmpcConfiguration.cu:28-134) -----------------------------------------------------
SmpcConfiguration::SmpcConfiguration(string pathToFile) {
    jsonlite::Document doc(pathToFile);
    weightPrice = 1; weightSmooth = 1; weightSafety = 1;
    NX = scalarInt(doc, "nx");
    NU = scalarInt(doc, "nu");
    ND = scalarInt(doc, "nd");
    NV = scalarInt(doc, "nv");
    const uint_t N = scalarInt(doc, "N");
    readReal(doc, "matL", (size_t)NU * NV, matL);
    readReal(doc, "matLhat", (size_t)NU * ND, matLhat);
    readReal(doc, "costW", (size_t)NU * NU, matCostW);
    for (size_t i = 0; i < matCostW.size(); i++) matCostW[i] *= weightSmooth;
    penaltyStateX = scalarReal(doc, "penaltyStateX");
    penaltySafetyX = weightSafety * scalarReal(doc, "penaltySafetyX");
    readReal(doc, "matDiagPrecnd", (size_t)(NU + 2 * NX) * N, matDiagPrecnd);
    readReal(doc, "currentX", NX, currentX);
    readReal(doc, "prevU", NU, prevU);
    readReal(doc, "prevDemand", ND, prevDemand);
    stepSize = scalarReal(doc, "stepSize");
    maxIteration = scalarInt(doc, "maxIterations");
    _ASSERT(doc["pathToNetwork"].IsString() && doc["pathToScenarioTree"].IsString() && doc["pathToForecaster"].IsString());
    pathToNetwork = doc["pathToNetwork"].str;
    pathToScenarioTree = doc["pathToScenarioTree"].str;
    pathToForecaster = doc["pathToForecaster"].str;
    _ASSERT(doc["algorithmName"].IsString());
    algorithmName = doc["algorithmName"].str;
    // optional key of this implementation (absent in the reference's files = "auto"): how the engine stores the factor step's operators
    operatorMode = "auto";
    if (doc.HasMember("operatorMode")) {
        _ASSERT(doc["operatorMode"].IsString());
        operatorMode = doc["operatorMode"].str;
        if (operatorMode != "auto" && operatorMode != "dense" && operatorMode != "structured")
            throw std::logic_error("controller configuration: operatorMode must be \"auto\", \"dense\" or \"structured\" (got \"" + operatorMode + "\")");
    }
    lbfgsBufferSize = scalarInt(doc, "lbfgsBufferSize");
    pathToConfiguration = pathToFile;
    // relative paths in the configuration are relative to the configuration file's directory when they do not
    // resolve from the working directory (the reference requires the binary to run from a fixed directory)
    const size_t slash = pathToFile.find_last_of('/');
    const string dir = slash == string::npos ? string(".") : pathToFile.substr(0, slash);
    string *paths[] = {&pathToNetwork, &pathToScenarioTree, &pathToForecaster};
    for (string *p : paths) {
        if (p->empty() || (*p)[0] == '/') continue;
        std::ifstream probe(p->c_str());
        if (probe.good()) continue;
        const size_t s2 = p->find_last_of('/');
        const string base = s2 == string::npos ? *p : p->substr(s2 + 1);
        std::ifstream probe2((dir + "/" + *p).c_str());
        *p = probe2.good() ? dir + "/" + *p : dir + "/" + base;
    }
}
void SmpcConfiguration::setCurrentState(real_t *state) { for (uint_t i = 0; i < NX; i++) currentX[i] = state[i]; }
void SmpcConfiguration::setPreviousControl(real_t *control) { for (uint_t i = 0; i < NU; i++) prevU[i] = control[i]; }
void SmpcConfiguration::setpreviousdemand(real_t *demand) { for (uint_t i = 0; i < ND; i++) prevDemand[i] = demand[i]; }
void SmpcConfiguration::setCurrentState() { jsonlite::Document doc(pathToConfiguration); readReal(doc, "currentX", NX, currentX); }
void SmpcConfiguration::setPreviousControl() { jsonlite::Document doc(pathToConfiguration); readReal(doc, "prevU", NU, prevU); }
void SmpcConfiguration::setPreviousDemand() { jsonlite::Document doc(pathToConfiguration); readReal(doc, "prevDemand", ND, prevDemand); }
