// AddressSanitizer / UBSan run of the subtree partitioner (rapidnet_amd/csrc/partition.hpp: host-only code behind
// rn_partition_create / rn_create_sharded) over random stage-contiguous trees -- uniform and per-node child counts -- for
// every cut stage and several rank counts, with the invariants a local tree must keep.  Built and run by
// tests/test_host_sanitized.py (g++ -fsanitize=address,undefined); exit code 0 = clean.
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../rapidnet_amd/csrc/partition.hpp"

struct Tree {
    rn_dims d;
    std::vector<int> stages, nps, cum, anc, nch, nchCum;
    std::vector<double> prob, errD, errP;
    rn_tree t;
};

static Tree make_tree(std::mt19937 &rng, int N, int maxKids, bool ragged, int nd, int nu) {
    Tree T;
    std::vector<std::vector<int>> level(1, std::vector<int>(1, -1));   // parents of the nodes of each stage
    std::vector<int> first(1, 0);
    int nodes = 1;
    const int branchStages = 1 + rng() % 3;
    for (int k = 1; k < N; k++) {
        std::vector<int> par;
        const int uniformKids = k <= branchStages ? 1 + (int)(rng() % maxKids) : 1;
        for (int i = 0; i < (int)level[k - 1].size(); i++) {
            const int kids = (ragged && k <= branchStages) ? 1 + (int)(rng() % maxKids) : uniformKids;
            for (int c = 0; c < kids; c++) par.push_back(first[k - 1] + i);
        }
        first.push_back(nodes);
        nodes += (int)par.size();
        level.push_back(par);
    }
    T.nps.assign(N + 1, 0);
    T.cum.assign(N + 2, 0);
    for (int k = 0; k < N; k++) {
        T.nps[k] = (int)level[k].size();
        T.cum[k + 1] = T.cum[k] + T.nps[k];
        for (int i = 0; i < T.nps[k]; i++) { T.stages.push_back(k); T.anc.push_back(level[k][i] + 1); }
    }
    T.cum[N + 1] = nodes;
    std::vector<int> kids(nodes, 0);
    for (int i = 1; i < nodes; i++) kids[T.anc[i] - 1]++;
    int run = 0, nonLeaf = 0;
    for (int i = 0; i < nodes; i++) {
        run += kids[i];
        T.nchCum.push_back(run);
        if (kids[i]) { T.nch.push_back(kids[i]); nonLeaf++; }
    }
    T.prob.assign(nodes, 1.0);
    for (int i = 1; i < nodes; i++) T.prob[i] = T.prob[T.anc[i] - 1] / kids[T.anc[i] - 1];
    std::uniform_real_distribution<double> u(-1, 1);
    T.errD.resize((size_t)nodes * nd);
    T.errP.resize((size_t)nodes * nu);
    for (auto &v : T.errD) v = u(rng);
    for (auto &v : T.errP) v = u(rng);
    T.d = rn_dims{3, nu, nu - 1, nd, N, T.nps[N - 1], nodes, nonLeaf};
    T.t = rn_tree{T.stages.data(), T.nps.data(), T.cum.data(), T.anc.data(), T.nch.data(), T.nchCum.data(), T.prob.data()};
    return T;
}

#define CHECK(c)                                                                                                        \
    do {                                                                                                                \
        if (!(c)) { std::fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #c); std::exit(2); }                \
    } while (0)

int main() {
    std::mt19937 rng(20260103);
    long parts = 0, refused = 0;
    for (int trial = 0; trial < 300; trial++) {
        const int N = 2 + rng() % 6, nd = 1 + rng() % 3, nu = 2 + rng() % 3;
        Tree T = make_tree(rng, N, 4, trial % 2 == 1, nd, nu);
        for (int cut = 0; cut < N; cut++)
            for (int W : {1, 2, 3, 5, 8}) {
                std::vector<int> seen(T.d.nodes, 0);
                bool anyRefused = false;
                for (int r = 0; r < W; r++) {
                    rn_partition P;
                    std::string err;
                    const bool withErr = (trial + r) % 3 != 0;
                    const int rc = rn::build_partition(&T.d, &T.t, withErr ? T.errD.data() : nullptr, withErr ? T.errP.data() : nullptr, r, W, cut, &P, err);
                    if (rc != RN_OK) { CHECK(!err.empty()); anyRefused = true; refused++; continue; }
                    parts++;
                    const int ln = P.dims.nodes, c = P.cutStage;
                    CHECK(c >= 1 && c < N && P.tree.nodesPerStageCumul[N] == ln && P.tree.nodesPerStageCumul[N + 1] == ln);
                    CHECK(P.dims.K == P.tree.nodesPerStage[N - 1]);
                    for (int l = 0; l < ln; l++) {
                        const int g = P.globalNode[l];
                        CHECK(g >= 0 && g < T.d.nodes && (l == 0 || P.globalNode[l - 1] < g));     // breadth-first order kept
                        CHECK(P.tree.stages[l] == T.stages[g]);
                        const int par = P.tree.ancestor[l];
                        CHECK(l == 0 ? par == 0 : (par >= 1 && par <= l && P.globalNode[par - 1] == T.anc[g] - 1));
                        CHECK(l == 0 || P.tree.ancestor[l - 1] <= par);                                // children contiguous
                        if (T.stages[g] >= c) seen[g]++;
                        if (withErr) for (int j = 0; j < nd; j++) CHECK(P.errorDemandNode[(size_t)l * nd + j] == T.errD[(size_t)g * nd + j]);
                    }
                    CHECK(P.nCutParents == T.nps[c - 1]);
                    double pm = 0;
                    for (int i = 0; i < P.nCutParents; i++) pm += P.momP[i];
                    double ps = 0;
                    for (int i = T.cum[c]; i < T.cum[c + 1]; i++) ps += T.prob[i];
                    CHECK(std::abs(pm - ps) < 1e-12);
                    delete static_cast<rn::PartitionData *>(P.owner);
                }
                if (!anyRefused)
                    for (int g = 0; g < T.d.nodes; g++) CHECK(seen[g] == (T.stages[g] >= (cut <= 0 ? rn::default_cut_stage(&T.d, &T.t) : cut) ? 1 : 0));
            }
    }
    // malformed trees through the public entry point: every one must be refused with a message, none may be indexed through
    // (the sanitizers watch): ancestors beyond the tree, forward-pointing, in the wrong stage, negative; stages out of order;
    // an empty stage; a second root
    long malformed = 0;
    for (int trial = 0; trial < 200; trial++) {
        const int N = 3 + rng() % 4;
        Tree T = make_tree(rng, N, 3, trial % 2 == 1, 2, 3);
        if (T.d.nodes < 4) continue;
        const int victim = 1 + (int)(rng() % (T.d.nodes - 1)), kind = trial % 8;
        const int k = T.stages[victim];
        switch (kind) {
            case 0: T.anc[victim] = T.d.nodes + 1 + (int)(rng() % 1000); break;          // beyond the tree
            case 1: T.anc[victim] = victim + 1; break;                                   // itself
            case 2: T.anc[victim] = T.d.nodes; break;                                    // forward-pointing (last node)
            case 3: T.anc[victim] = -5; break;                                           // negative
            case 4: T.anc[victim] = 0; break;                                            // a second root
            case 5: T.stages[victim] = k + 1; break;                                     // not numbered stage by stage
            case 6: T.cum[k + 1] = T.cum[k]; break;                                      // an empty stage
            default: if (k >= 2) T.anc[victim] = T.cum[k - 2] + 1; else T.anc[victim] = T.d.nodes + 7; break;   // two stages up
        }
        for (int cut = 0; cut < N; cut++)
            for (int W : {1, 2, 3}) {
                rn_partition P;
                std::string err;
                const int rc = rn::build_partition(&T.d, &T.t, T.errD.data(), T.errP.data(), W - 1, W, cut, &P, err);
                CHECK(rc == RN_E_ARG && !err.empty());
                malformed++;
            }
    }
    std::printf("partitions built %ld, refused %ld, malformed trees refused %ld\n", parts, refused, malformed);
    return parts > 1000 && malformed > 1000 ? 0 : 3;
}
