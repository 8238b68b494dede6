// partition.hpp -- subtree partitioner behind the C-ABI (rn_partition_create / rn_create_sharded, include/rapidnet.h).
//
// New capability: the reference is single-GPU.  The scenario tree (ScenarioTree.cuh:92-154
// conventions) is cut below stage c: nodes of stages < c (the crown) are replicated on every rank, the subtrees rooted at
// stage c are dealt round-robin by position within the stage.  A rank's nodes keep their breadth-first order, so the local
// tree is again stage-contiguous with contiguous children and goes through rn_create unchanged.  Host-only code.
#pragma once
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rapidnet.h"

namespace rn {

struct PartitionData {
    std::vector<int> stages, nodesPerStage, nodesPerStageCumul, ancestor, nChildren, nChildrenCumul, globalNode;
    std::vector<double> probNode, errD, errP, momE, momP;
};

inline int default_cut_stage(const rn_dims *d, const rn_tree *t) {
    if (!d || !t || !t->nodesPerStage || d->N < 1) return RN_E_ARG;
    for (int c = 1; c < d->N; c++)
        if (t->nodesPerStage[c] == d->K) return c;
    return d->N - 1 > 1 ? d->N - 1 : 1;
}

// returns RN_OK or RN_E_ARG with a message in `err`
inline int build_partition(const rn_dims *d, const rn_tree *t, const double *errD, const double *errP, int rank, int nranks, int cut,
                           rn_partition *out, std::string &err) {
    if (!d || !t || !out) { err = "rn_partition_create: null argument"; return RN_E_ARG; }
    if (!t->stages || !t->nodesPerStageCumul || !t->ancestor || !t->probNode) { err = "rn_partition_create: null tree array"; return RN_E_ARG; }
    const int N = d->N, nodes = d->nodes, nd = d->nd, nu = d->nu;
    if (nranks < 1 || rank < 0 || rank >= nranks) { err = "rn_partition_create: bad rank"; return RN_E_ARG; }
    if (cut <= 0) cut = default_cut_stage(d, t);
    if (cut < 1 || cut >= N) { err = "rn_partition_create: the cut stage must lie in [1, N-1] (a horizon of 1 cannot be sharded)"; return RN_E_ARG; }
    const int *cum = t->nodesPerStageCumul;
    if (nodes < 1 || cum[0] != 0 || cum[N] != nodes) { err = "rn_partition_create: nodesPerStageCumul inconsistent with nodes"; return RN_E_ARG; }
    // the whole tree is validated BEFORE anything is indexed through it (this is a public entry point: a malformed `ancestor` or
    // `stages` must end in RN_E_ARG, not in an out-of-bounds access): stage boundaries ascending, nodes numbered stage by stage,
    // node 0 the only root, every other node's ancestor a node of the previous stage
    for (int k = 0; k < N; k++)
        if (cum[k + 1] <= cum[k]) { err = "rn_partition_create: nodesPerStageCumul must be strictly ascending (every stage has a node)"; return RN_E_ARG; }
    if (cum[1] != 1 || t->ancestor[0] != 0 || t->stages[0] != 0) { err = "rn_partition_create: node 0 must be the only root"; return RN_E_ARG; }
    for (int k = 1; k < N; k++)
        for (int i = cum[k]; i < cum[k + 1]; i++) {
            if (t->stages[i] != k) { err = "rn_partition_create: nodes are not numbered stage by stage"; return RN_E_ARG; }
            const int par = t->ancestor[i] - 1;
            if (par < cum[k - 1] || par >= cum[k]) { err = "rn_partition_create: ancestor must be a node of the previous stage"; return RN_E_ARG; }
        }
    // owner of every node: -1 = replicated crown node; subtree roots by position, descendants inherit (parents come first)
    std::vector<int> owner(nodes, -1);
    for (int i = cum[cut]; i < cum[cut + 1]; i++) owner[i] = (i - cum[cut]) % nranks;
    for (int i = cum[cut + 1]; i < nodes; i++) owner[i] = owner[t->ancestor[i] - 1];
    PartitionData *p = new PartitionData();
    std::vector<int> newId(nodes, -1);
    for (int i = 0; i < nodes; i++)
        if (owner[i] == -1 || owner[i] == rank) { newId[i] = (int)p->globalNode.size(); p->globalNode.push_back(i); }
    const int ln = (int)p->globalNode.size();
    p->stages.resize(ln); p->ancestor.resize(ln); p->probNode.resize(ln);
    p->nodesPerStage.assign(N + 1, 0);
    std::vector<int> childCount(ln, 0);
    for (int l = 0; l < ln; l++) {
        const int g = p->globalNode[l];
        p->stages[l] = t->stages[g];
        if (p->stages[l] < 0 || p->stages[l] >= N) { delete p; err = "rn_partition_create: stage out of range"; return RN_E_ARG; }
        p->nodesPerStage[p->stages[l]]++;
        p->probNode[l] = t->probNode[g];
        const int par = t->ancestor[g] - 1;
        p->ancestor[l] = par >= 0 ? newId[par] + 1 : 0;
        if (par >= 0) childCount[newId[par]]++;
    }
    for (int k = 0; k < N; k++)
        if (p->nodesPerStage[k] == 0) {
            delete p;
            err = "rn_partition_create: rank " + std::to_string(rank) + " of " + std::to_string(nranks) + " would own no node of stage " +
                  std::to_string(k) + " (more ranks than subtrees at the cut stage)";
            return RN_E_ARG;
        }
    p->nodesPerStageCumul.assign(N + 2, 0);
    for (int k = 0; k < N; k++) p->nodesPerStageCumul[k + 1] = p->nodesPerStageCumul[k] + p->nodesPerStage[k];
    p->nodesPerStageCumul[N + 1] = ln;
    p->nChildrenCumul.resize(ln);
    int run = 0, nonLeaf = 0;
    for (int l = 0; l < ln; l++) {
        run += childCount[l];
        p->nChildrenCumul[l] = run;
        if (childCount[l] > 0) { p->nChildren.push_back(childCount[l]); nonLeaf++; }
    }
    if (errD) {
        p->errD.resize((size_t)ln * nd);
        for (int l = 0; l < ln; l++) std::memcpy(&p->errD[(size_t)l * nd], errD + (size_t)p->globalNode[l] * nd, sizeof(double) * nd);
    }
    if (errP) {
        p->errP.resize((size_t)ln * nu);
        for (int l = 0; l < ln; l++) std::memcpy(&p->errP[(size_t)l * nu], errP + (size_t)p->globalNode[l] * nu, sizeof(double) * nu);
    }
    // children moments of the cut parents over the FULL tree, children in ascending order
    const int first = cum[cut - 1], nPar = cum[cut] - cum[cut - 1];
    p->momP.assign(nPar, 0.0);
    if (errD) p->momE.assign((size_t)nPar * nd, 0.0);
    for (int c = cum[cut]; c < cum[cut + 1]; c++) {
        const int par = t->ancestor[c] - 1 - first;
        if (par < 0 || par >= nPar) { delete p; err = "rn_partition_create: ancestor must be a node of the previous stage"; return RN_E_ARG; }
        const double pc = t->probNode[c];
        p->momP[par] += pc;
        if (errD) for (int j = 0; j < nd; j++) p->momE[(size_t)par * nd + j] += pc * errD[(size_t)c * nd + j];
    }
    rn_partition r;
    r.dims = *d;
    r.dims.nodes = ln; r.dims.K = p->nodesPerStage[N - 1]; r.dims.nNonLeafNodes = nonLeaf;
    r.tree.stages = p->stages.data(); r.tree.nodesPerStage = p->nodesPerStage.data(); r.tree.nodesPerStageCumul = p->nodesPerStageCumul.data();
    r.tree.ancestor = p->ancestor.data(); r.tree.nChildren = p->nChildren.data(); r.tree.nChildrenCumul = p->nChildrenCumul.data();
    r.tree.probNode = p->probNode.data();
    r.globalNode = p->globalNode.data();
    r.errorDemandNode = errD ? p->errD.data() : nullptr;
    r.errorPriceNode = errP ? p->errP.data() : nullptr;
    r.rank = rank; r.nranks = nranks; r.cutStage = cut; r.nCutParents = nPar;
    r.momE = errD ? p->momE.data() : nullptr;
    r.momP = p->momP.data();
    r.owner = p;
    *out = r;
    return RN_OK;
}

}  // namespace rn
