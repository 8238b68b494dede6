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


@pytest.mark.parametrize("name,kw,replayed", [("medium", {}, 0), ("ragged", {}, 0), ("medium", {"penalty_x": 20.0, "penalty_xs": 5.0}, 1)])
def test_the_exchange_sets_itself_up_and_chooses_over_a_real_communicator(name, kw, replayed):
    """RN_EXCHANGE_AUTO over the library's own RCCL communicator, as far as one GPU can show it: a one-rank communicator made AFTER the cut
    stage is known -- rn_comm_init then allocates the rank's inbox, gathers the IPC handles over the communicator (a byte-wise SUM
    all-reduce), wires the inbox and agrees that everybody could -- and the first batch times the real ncclAllReduce against the one-shot
    exchange on the context's own iterations.  Whatever it picks: the oracle's iterates, the counters of an untuned run, and the getters
    untouched by an explicit second tune (incl. a batch whose soft-constraint thresholds trip and is replayed)."""
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    ohist = o.apg(41)
    cut = partition.default_cut_stage(p["tree"])
    runs = {}
    for fixed in (None, capi.EXCHANGE_COLLECTIVE, capi.EXCHANGE_ONESHOT):
        s = capi.Solver(p["network"], p["tree"], p["config"])
        s.setCutStage(cut, partition.cut_children_moments(p["tree"], cut))
        if fixed == capi.EXCHANGE_COLLECTIVE:
            s.setExchangeTransport(fixed)                  # before the communicator: nothing of the one-shot transport is set up
        s.commInit(0, 1, capi.comm_unique_id())
        if fixed == capi.EXCHANGE_ONESHOT:
            s.setExchangeTransport(fixed)                  # the inboxes were wired by rn_comm_init
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        h = [s.apgIterate(20), s.apgIterate(1)]
        info = s.exchangeAutotune(0)
        if fixed is None:
            assert info["tunes"] == 1 and info["candidates"] == 3 and info["iterations"] == 20 and info["collective_us"] > 0 and info["oneshot_us"] > 0, info
            assert info["transport"] == (1 if info["oneshot_us"] < info["collective_us"] else 0), info
            state = (capi.BUF_XI, capi.BUF_PSI, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_ACC_XI, capi.BUF_ACC_PSI)
            before = [s.get(b) for b in state]
            again = s.exchangeAutotune(12)
            assert again["tunes"] == 2 and all(np.array_equal(a, s.get(b)) for a, b in zip(before, state))
        else:
            assert info["tunes"] == 0 and info["transport"] == fixed, info
        h.append(s.apgIterate(20))
        runs[fixed] = (np.concatenate(h), s.counters(), [s.get(b) for b in (capi.BUF_X, capi.BUF_U, capi.BUF_UPD_XI, capi.BUF_UPD_PSI)])
        s.close()
    for fixed, (h, c, vecs) in runs.items():
        assert np.abs(h - ohist).max() <= 1e-9 * np.abs(ohist).max(), fixed
        for v, nm in zip(vecs, ("x", "u", "updXi", "updPsi")):
            assert relmax(v, o.get(nm)) < REL_TOL, (fixed, nm)
        assert c == runs[capi.EXCHANGE_COLLECTIVE][1] and c["replayed"] >= replayed, (fixed, c)
        assert np.array_equal(h, runs[capi.EXCHANGE_COLLECTIVE][0])            # one rank: the same sums whichever way they travel
