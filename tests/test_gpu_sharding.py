"""GPU tests of the multi-GPU sharding path on ONE GPU: (a) two shard contexts on the same device with the exchange
emulated through the rn_debug_* hooks, (b) a one-rank RCCL communicator so that the real ncclAllReduce calls (cut
payload + prox distances) run on the solver's stream."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, partition, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert np.isfinite(a).all()
    return float(np.abs(a - b).max() / np.abs(b).max())


def lambdas(n):
    th0, th1, out = 1.0, 1.0, []
    for _ in range(n):
        out.append(th1 * (1 / th0 - 1))
        th0, th1 = th1, 0.5 * (np.sqrt(th1 ** 4 + 4 * th1 ** 2) - th1 ** 2)
    return out


@pytest.mark.parametrize("name,cut,world,structured", [("medium", 2, 2, False), ("medium", 1, 2, False), ("small", 3, 2, False),
                                                        ("medium", 2, 3, False), ("medium", 2, 2, True),
                                                        # non-uniform trees: subtrees of unequal size per rank (cut at 1 and 2), cut at the chain stage
                                                        ("ragged", 1, 2, False), ("ragged", 1, 3, False), ("ragged", 2, 2, False), ("ragged", 2, 3, True),
                                                        ("ragged", 3, 4, False), ("ragged2", 1, 2, False), ("ragged2", 3, 3, False)])
def test_shards_on_one_gpu_match_full_tree(name, cut, world, structured):
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], alias_operators=(name != "ragged2"))
    o.initialise(dh, ah)
    iters = 8
    o.apg(iters)
    moments = partition.cut_children_moments(p["tree"], cut)
    shards, ids = [], []
    for r in range(world):
        lt, gids = partition.local_tree(p["tree"], r, world, cut)
        s = capi.Solver(p["network"], lt, p["config"], structured=structured)
        s.commInit(r, world, None)
        s.setCutStage(cut, moments)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        shards.append(s); ids.append(gids)
    n_par = p["tree"]["nodesPerStage"][cut - 1]
    n_cut = n_par * (o.nv + 2 * o.nx)
    for lam in lambdas(iters):
        for s in shards:
            s.dualExtrapolationStep(lam)
            s.debugSweepPhase(1)
        total = sum(s.debugCutBuffer(n_cut) for s in shards)       # the all-reduce
        for s in shards:
            s.debugCutBuffer(n_cut, total)
            s.debugSweepPhase(2)
            s.proximalFunG(); s.computeFixedPointResidual(); s.dualUpdate()
    # beta of the replicated crown must equal the full tree's (children moments)
    crown = p["tree"]["nodesPerStageCumul"][cut]
    for s in shards:
        assert relmax(s.get(capi.BUF_BETA)[: crown * o.nv], o.get("beta")[: crown * o.nv]) < 1e-12
    for bid, nm, dim in ((capi.BUF_X, "x", o.nx), (capi.BUF_U, "u", o.nu), (capi.BUF_UPD_XI, "updXi", 2 * o.nx),
                         (capi.BUF_UPD_PSI, "updPsi", o.nu), (capi.BUF_V, "v", o.nv)):
        full = partition.scatter_to_global([s.get(bid) for s in shards], ids, o.nodes, dim)
        assert relmax(full, o.get(nm)) < REL_TOL, nm


def test_single_rank_rccl_cut_path():
    """nranks = 1 with a real RCCL communicator: k_cut_partial_sums + ncclAllReduce + presummed crown, and the
    all-reduced prox distances, must reproduce the plain single-GPU solve (and the soft-constraint branch)."""
    for kw, optimistic in (({}, True), ({}, False), ({"penalty_x": 20.0, "penalty_xs": 5.0}, True),
                           ({"penalty_x": 20.0, "penalty_xs": 5.0}, False)):
        # optimistic: one collective per iteration + verification; with the small penalties the thresholds trip and the
        # batch is replayed through the exact path from its checkpoint -- same result either way
        p = synth.make_problem("medium", **kw)
        dh, ah = synth.forecast_at(p["forecast"], 0)
        o = Oracle(p["network"], p["tree"], p["config"])
        o.initialise(dh, ah)
        ohist = o.apg(12)
        cut = partition.default_cut_stage(p["tree"])
        s = capi.Solver(p["network"], p["tree"], p["config"])
        s.commInit(0, 1, capi.comm_unique_id())
        s.setCutStage(cut, partition.cut_children_moments(p["tree"], cut))
        s.setExchangeMode(optimistic)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        hist = np.concatenate([s.apgIterate(5), s.apgIterate(7)])   # two batches: checkpoints, tails and theta carry over
        for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"),
                        (capi.BUF_DUAL_XI, "dualXi")):
            assert relmax(s.get(bid), o.get(nm)) < REL_TOL, nm
        assert np.abs(hist - ohist).max() <= 1e-9 * np.abs(ohist).max()
