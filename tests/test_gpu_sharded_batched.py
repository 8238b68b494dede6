"""The DEVICE-RESIDENT sharded path with several ranks.

rn_apg_iterate on a sharded context runs, per batch: checkpoint, one all-reduce per iteration whose 2-element tail carries
the previous iteration's dist^2, the closing tail all-reduce, the verdict vote and -- if a soft-constraint threshold
tripped -- a replay of the batch through the exact two-collective path.  A one-GPU box cannot host several RCCL ranks
("Duplicate GPU detected"), so the ranks are contexts of ONE process, one host thread each, created through
rn_create_sharded (the partitioner behind the C-ABI) and joined to the library's in-process stand-in for the communicator
(rn_debug_local_group_join: called exactly where ncclAllReduce would be).  The reassembled iterates must equal the unsharded
solve and the CPU oracle."""
import threading

import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, partition, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9
VECS = ((capi.BUF_X, "x", "nx"), (capi.BUF_U, "u", "nu"), (capi.BUF_V, "v", "nv"), (capi.BUF_UPD_XI, "updXi", "2nx"),
        (capi.BUF_UPD_PSI, "updPsi", "nu"), (capi.BUF_DUAL_XI, "dualXi", "2nx"), (capi.BUF_RES_PSI, "resPsi", "nu"))


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


class Ranks:
    """`world` shard contexts of one problem on one GPU, joined to an in-process group; run(fn) calls fn(solver) on every
    rank from a thread of its own (the collectives inside rn_apg_iterate rendezvous across the threads)."""

    def __init__(self, p, world, cut=0, structured=False, precision="f64", optimistic=True, knobs=None):
        self.group = capi.local_group_create(world)
        self.shards = []
        for r in range(world):
            s = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=world, cut_stage=cut, structured=structured, precision=precision, knobs=knobs)
            s.joinLocalGroup(self.group, r)
            s.setExchangeMode(optimistic)
            self.shards.append(s)
        self.nodes = self.shards[0].full_nodes

    def run(self, fn):
        out, errs = [None] * len(self.shards), []

        def work(i):
            try:
                out[i] = fn(self.shards[i])
            except Exception as e:   # noqa: BLE001 -- reported below, with the rank
                errs.append((i, e))

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(self.shards))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        return out

    def gathered(self, bid, dim):
        return partition.scatter_to_global([s.get(bid) for s in self.shards], [s.global_nodes for s in self.shards], self.nodes, dim)

    def close(self):
        for s in self.shards:
            s.close()
        capi.local_group_destroy(self.group)


def dims_of(s):
    return {"nx": s.nx, "nu": s.nu, "nv": s.nv, "2nx": 2 * s.nx}


@pytest.mark.parametrize("name,world,cut,structured,kw,trips", [
    ("medium", 2, 0, False, {}, False), ("medium", 4, 1, False, {}, False), ("medium", 3, 2, True, {}, False),
    ("small", 2, 3, False, {}, False), ("ragged", 3, 1, False, {}, False), ("ragged", 2, 2, False, {}, False),
    # small penalties: the tree-global distances exceed gamma / lambda, the optimistic batch is replayed through the exact path
    ("medium", 4, 0, False, {"penalty_x": 20.0, "penalty_xs": 5.0}, True),
    ("medium", 2, 1, True, {"penalty_x": 20.0, "penalty_xs": 5.0}, True),
])
def test_batched_sharded_solve_matches_oracle(name, world, cut, structured, kw, trips):
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    ohist = o.apg(24)
    rk = Ranks(p, world, cut, structured)
    try:
        def solve(s):
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
            h1 = s.apgIterate(20)                   # two batches: checkpoint, tails and theta carry over
            h2 = s.apgIterate(4)
            return s.counters(), np.concatenate([h1, h2])

        res = rk.run(solve)
        counters = [r[0] for r in res]
        assert all(c == counters[0] for c in counters), counters          # the ranks took the same path through every batch
        assert (counters[0]["replayed"] >= 1) == trips, counters
        # vecPrimalInfs (SmpcController.cu:1480-1496, :1521) is tree-global on EVERY rank: one MAX all-reduce per batch
        for _, hist in res:
            assert np.array_equal(hist, res[0][1])
            assert np.abs(hist - ohist).max() <= 1e-9 * np.abs(ohist).max()
        d = dims_of(rk.shards[0])
        for bid, nm, dm in VECS:
            assert relmax(rk.gathered(bid, d[dm]), o.get(nm)) < REL_TOL, nm
        # ... and the same values follow from the ranks' arg-max parts (rn_get_history_parts stays rank-local)
        parts = np.stack([s.historyParts(0, 24) for s in rk.shards])      # [rank, it, (absXi, valXi, absPsi, valPsi)]
        ix, ip = parts[:, :, 0].argmax(0), parts[:, :, 2].argmax(0)
        vx, vp = parts[ix, np.arange(24), 1], parts[ip, np.arange(24), 3]
        assert np.abs(np.maximum(vx, vp) - ohist).max() <= 1e-9 * np.abs(ohist).max()
        # the replicated crown is bit-identical on every rank
        crown = p["tree"]["nodesPerStageCumul"][rk.shards[0].shardInfo()["cut_stage"]]
        for bid, _, dm in VECS:
            ref = rk.shards[0].get(bid)[: crown * d[dm]]
            for s in rk.shards[1:]:
                assert np.array_equal(s.get(bid)[: crown * d[dm]], ref)
    finally:
        rk.close()


def test_barcelona493_batched_over_8_ranks():
    """BASELINE.json configs[3] through the path `bench.py --gpus 8` times: rn_apg_iterate(20) on 8 shard contexts."""
    p = synth.make_problem("barcelona493")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    full = capi.Solver(p["network"], p["tree"], p["config"])
    full.initialiseSmpcController(dh, ah)
    full.algorithmApg(20)
    rk = Ranks(p, 8)
    try:
        info = rk.shards[3].shardInfo()
        assert info["cut_stage"] == 2 and info["cut_parents"] == 17 and info["nranks"] == 8 and info["full_nodes"] == 10864
        assert info["comm_ranks"] == 0           # no RCCL communicator on this box: the stand-in carries the exchange

        def solve(s):
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
            s.apgIterate(20, history=False)
            return s.counters()

        counters = rk.run(solve)
        assert all(c == {"optimistic": 1, "exact": 0, "replayed": 0, "hold": 0} for c in counters), counters
        d = dims_of(full)
        for bid, nm, dm in VECS:
            assert relmax(rk.gathered(bid, d[dm]), full.get(bid)) < REL_TOL, nm
    finally:
        rk.close()
        full.close()


def test_a_missing_rank_fails_the_others_instead_of_hanging_them(monkeypatch):
    """A rank that never issues its collective must not leave its peers waiting for ever: the group's barrier gives up after
    its timeout (120 s by default, 2 s here) and the waiting rank's call returns RN_E_COMM."""
    import time

    monkeypatch.setenv("RAPIDNET_GROUP_TIMEOUT_S", "2")
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    rk = Ranks(p, 2)
    try:
        for s in rk.shards:
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
        t0 = time.time()
        with pytest.raises(capi.RapidNetError, match="all-reduce callback failed"):
            rk.shards[0].apgIterate(20, history=False)          # rank 1 never arrives
        assert time.time() - t0 < 30
    finally:
        rk.close()


@pytest.mark.parametrize("world,cut", [(2, 0), (4, 0), (2, 2)])
def test_wide_network_fp32_sharded(world, cut):
    """BASELINE.json configs[4] is the wide network in fp32 ON 8 GPUs: the same network and precision on a 16-scenario tree
    with a three-stage crown (7 crown nodes replicated; 512 lanes per chain in k_up_chain_cut, tile-kernel products,
    stage-by-stage crown), sharded over 2 and 4 ranks below stage 3 (and below stage 2: subtrees of two chains), through
    the device-resident path, against the fp32 oracle of the whole tree and the unsharded HIP solve."""
    synth.CONFIGS.setdefault("wide16", (4, 200, 360, 280, 54, 24, [4, 2, 2]))
    p = synth.make_problem("wide16")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], precision="f32")
    o.initialise(dh, ah)
    o.apg(20)
    full = capi.Solver(p["network"], p["tree"], p["config"], precision="f32")
    full.initialiseSmpcController(dh, ah)
    full.algorithmApg(20)
    rk = Ranks(p, world, cut, precision="f32")
    try:
        def solve(s):
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
            s.apgIterate(20, history=False)
            return s.counters()

        counters = rk.run(solve)
        assert all(c == {"optimistic": 1, "exact": 0, "replayed": 0, "hold": 0} for c in counters), counters
        d = dims_of(full)
        for bid, nm, dm in VECS:
            got = rk.gathered(bid, d[dm])
            assert relmax(got, o.get(nm)) < 2e-4, ("oracle", nm)
            assert relmax(got, full.get(bid)) < 2e-5, ("unsharded", nm)       # same kernels, another summation order at the cut
    finally:
        rk.close()
        full.close()
