"""BASELINE.json configs[3] and configs[4] through the C-ABI on the GPU.

configs[3]: the 493-scenario Barcelona tree sharded by subtree over 8 ranks, cut below stage 2 (what `bench.py --gpus 8`
runs).  A one-GPU box cannot host 8 RCCL ranks, so the 8 rank-local contexts live on ONE device and the per-iteration
all-reduce of the cut parents' children sums is done by the test (rn_debug_sweep_phase / rn_debug_cut_buffer); the
reassembled iterates must equal the unsharded HIP solve (1e-9 after 20 iterations) and the CPU oracle (2 iterations: the
oracle needs ~25 s and ~6 GB for its factor step at this size).  The real ncclAllReduce path is run on the full tree with a
one-rank communicator in both exchange modes.

configs[4]: the wide network (200 states, 360 inputs, nv = 306) on the 4 096-scenario tree in fp32, 160 GB of per-node
blocks on one MI355X.  No CPU oracle can hold that, so (i) the same network is checked against the fp32 oracle on a
16-scenario tree with a three-stage crown (same kernels and template instantiations: 3 slots per lane in k_stream_gemv,
tile-kernel fallback of the shared products), and (ii) at full size the dense-block path and the structured path -- two
independent implementations of the same operator -- must agree, the dual-gradient map must be affine in the dual, and a
repeated sweep must be bitwise identical.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, partition, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9
FP32_TOL = 2e-4


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape
    assert np.isfinite(a).all() and np.isfinite(b).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def lambdas(n):
    th0, th1, out = 1.0, 1.0, []
    for _ in range(n):
        out.append(th1 * (1 / th0 - 1))
        th0, th1 = th1, 0.5 * (np.sqrt(th1 ** 4 + 4 * th1 ** 2) - th1 ** 2)
    return out


# ---- configs[3] --------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def barcelona493():
    p = synth.make_problem("barcelona493")
    return p, synth.forecast_at(p["forecast"], 0)


VECS = ((capi.BUF_X, "x", "nx"), (capi.BUF_U, "u", "nu"), (capi.BUF_V, "v", "nv"), (capi.BUF_UPD_XI, "updXi", "2nx"),
        (capi.BUF_UPD_PSI, "updPsi", "nu"), (capi.BUF_DUAL_XI, "dualXi", "2nx"), (capi.BUF_RES_PSI, "resPsi", "nu"))


def test_barcelona493_sharded_over_8_ranks(barcelona493):
    p, (dh, ah) = barcelona493
    world, iters, oracle_iters = 8, 20, 2
    cut = partition.default_cut_stage(p["tree"])
    assert cut == 2 and p["tree"]["nodesPerStage"][cut] == 493            # below stage 2: 493 subtrees, 61 or 62 per rank
    moments = partition.cut_children_moments(p["tree"], cut)
    full = capi.Solver(p["network"], p["tree"], p["config"])
    full.initialiseSmpcController(dh, ah)
    nx, nu, nv, nodes = full.nx, full.nu, full.nv, full.nodes
    dims = {"nx": nx, "nu": nu, "nv": nv, "2nx": 2 * nx}
    shards, ids = [], []
    for r in range(world):
        lt, gids = partition.local_tree(p["tree"], r, world, cut)
        assert lt["nodesPerStage"][cut] in (61, 62)
        s = capi.Solver(p["network"], lt, p["config"])
        s.commInit(r, world, None)
        s.setCutStage(cut, moments)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        shards.append(s); ids.append(gids)
    # every node of the tree is owned by exactly one rank or replicated on all of them (the 18 crown nodes)
    count = np.zeros(nodes, int)
    for g in ids:
        count[g] += 1
    crown = p["tree"]["nodesPerStageCumul"][cut]
    assert (count[:crown] == world).all() and (count[crown:] == 1).all()
    n_cut = p["tree"]["nodesPerStage"][cut - 1] * (nv + 2 * nx)

    def gathered(bid, dim):
        return partition.scatter_to_global([s.get(bid) for s in shards], ids, nodes, dim)

    oracle_state = None
    for it, lam in enumerate(lambdas(iters)):
        for s in shards:
            s.dualExtrapolationStep(lam)
            s.debugSweepPhase(1)
        total = sum(s.debugCutBuffer(n_cut) for s in shards)             # the one all-reduce of the iteration
        for s in shards:
            s.debugCutBuffer(n_cut, total)
            s.debugSweepPhase(2)
            s.proximalFunG(); s.computeFixedPointResidual(); s.dualUpdate()
        if it + 1 == oracle_iters:
            oracle_state = {nm: gathered(bid, dims[d]) for bid, nm, d in VECS}
    # (1) against the CPU oracle after 2 iterations
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    o.apg(oracle_iters)
    for s in shards:   # beta of the replicated crown comes from the children moments
        assert relmax(s.get(capi.BUF_BETA)[: crown * nv], o.get("beta")[: crown * nv]) < 1e-12
    for _, nm, _ in VECS:
        assert relmax(oracle_state[nm], o.get(nm)) < REL_TOL, ("oracle", nm)
    del o
    # (2) against the unsharded device-resident solve after 20 iterations
    full.algorithmApg(iters)
    for bid, nm, d in VECS:
        assert relmax(gathered(bid, dims[d]), full.get(bid)) < REL_TOL, ("unsharded", nm)
    # the replicated crown is identical on every rank
    for s in shards[1:]:
        assert np.array_equal(s.get(capi.BUF_U)[: crown * nu], shards[0].get(capi.BUF_U)[: crown * nu])
    for s in shards:
        s.close()
    full.close()


@pytest.mark.parametrize("optimistic", [True, False])
def test_barcelona493_one_rank_rccl_exchange(barcelona493, optimistic):
    """The library's own ncclAllReduce on the solver's stream (k_cut_partial_sums -> all-reduce -> presummed crown step in
    k_gemm_vlv) on the full 493-scenario tree, one-rank communicator, both exchange modes."""
    p, (dh, ah) = barcelona493
    cut = partition.default_cut_stage(p["tree"])
    full = capi.Solver(p["network"], p["tree"], p["config"])
    full.initialiseSmpcController(dh, ah)
    hist = full.algorithmApg(24)
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.commInit(0, 1, capi.comm_unique_id())
    s.setCutStage(cut, partition.cut_children_moments(p["tree"], cut))
    s.setExchangeMode(optimistic)
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    h = np.concatenate([s.apgIterate(17), s.apgIterate(7)])               # two batches: checkpoint, payload tail, theta carry over
    for bid, nm, _ in VECS:
        assert relmax(s.get(bid), full.get(bid)) < REL_TOL, nm
    assert np.abs(h - hist).max() <= 1e-9 * np.abs(hist).max()
    s.close(); full.close()


# ---- configs[4] --------------------------------------------------------------------------------------------------
synth.CONFIGS.setdefault("wide16", (4, 200, 360, 280, 54, 24, [4, 2, 2]))   # the wide network, 16 scenarios, three-stage crown


@pytest.mark.parametrize("structured", [False, True])
def test_wide_network_fp32_against_the_oracle(structured):
    p = synth.make_problem("wide16")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], precision="f32")
    o.initialise(dh, ah)
    s = capi.Solver(p["network"], p["tree"], p["config"], precision="f32", structured=structured)
    s.initialiseSmpcController(dh, ah)
    for bid, nm in ((capi.BUF_UHAT, "uhat"), (capi.BUF_E, "e"), (capi.BUF_BETA, "beta"), (capi.BUF_XMAX, "xmax"), (capi.BUF_UMAX, "umax")):
        assert relmax(s.get(bid), o.get(nm)) < 2e-5, nm
    hist, ohist = s.algorithmApg(10), o.apg(10)
    for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"),
                    (capi.BUF_PRIMAL_XI, "primalXi"), (capi.BUF_DUAL_XI, "dualXi"), (capi.BUF_RES_PSI, "resPsi")):
        assert relmax(s.get(bid), o.get(nm)) < FP32_TOL, nm
    assert np.abs(hist - ohist).max() <= FP32_TOL * np.abs(ohist).max()
    s.close()


def test_wide4096_fp32_full_size():
    p = synth.make_problem("wide4096")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    d = capi.Solver(p["network"], p["tree"], p["config"], precision="f32")              # 160 GB of per-node blocks
    d.initialiseSmpcController(dh, ah)
    assert d.nodes == 86289 and d.nx == 200 and d.K == 4096
    # (o) anchored to the oracle at FULL size: the CPU oracle cannot hold 160 GB of blocks, but the factor step and the affine terms
    # are node-local formulas -- tests/numpy_engine.py evaluates them node by node (pinned to the oracle on small trees,
    # tests/test_numpy_engine.py): operator blocks at 10 sampled nodes incl. the root, both ends of the crown and the last node,
    # the scaled bounds of those nodes, and uhat / e / alpha / beta of ALL nodes (Engine.cu:671-774, 1147-1298)
    from numpy_engine import NumpyEngine

    ne = NumpyEngine(p["network"], p["tree"], p["config"])
    nv, nx, nu = d.nv, d.nx, d.nu
    sample = sorted({0, 1, 16, 17, 272, 273, 4368, 4369, 40000, d.nodes - 1})
    for node in sample:
        ops = ne.operators(node)
        for op, key in ((capi.OP_PHI, "Phi"), (capi.OP_D, "D"), (capi.OP_PSI, "Psi"), (capi.OP_F, "Ftil")):
            assert relmax(d.getOperator(op, node), ops[key].ravel(order="F")) < 2e-6, (key, node)      # fp32 storage of fp64-evaluated blocks
    b = ne.bounds_of(sample)
    for bid, key, dim in ((capi.BUF_XMIN, "xmin", nx), (capi.BUF_XMAX, "xmax", nx), (capi.BUF_XS, "xs", nx), (capi.BUF_UMIN, "umin", nu),
                          (capi.BUF_UMAX, "umax", nu)):
        got = d.get(bid).reshape(d.nodes, dim)[sample]
        assert relmax(got, b[key]) < 2e-6, key
    a = ne.affine(dh, ah)
    worst = {}
    for bid, key in ((capi.BUF_UHAT, "uhat"), (capi.BUF_E, "e"), (capi.BUF_ALPHA, "alpha"), (capi.BUF_BETA, "beta")):
        worst[key] = relmax(d.get(bid), a[key].ravel())
    print("wide4096 fp32, full size, affine terms vs the fp64 numpy evaluation:", {k: "%.1e" % v for k, v in worst.items()})
    assert max(worst[k] for k in ("uhat", "e", "alpha")) < 2e-5 and worst["beta"] < 1e-4, worst       # fp32 sums over nd = 280 / nu = 360 terms
    del ne, a, b
    st = capi.Solver(p["network"], p["tree"], p["config"], precision="f32", structured=True)
    st.initialiseSmpcController(dh, ah)
    # (i) two implementations of the operator, 10 device-resident iterations
    hd, hs = d.algorithmApg(10), st.algorithmApg(10)
    for bid in (capi.BUF_X, capi.BUF_U, capi.BUF_V, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_PRIMAL_XI, capi.BUF_DUAL_XI, capi.BUF_RES_PSI):
        assert relmax(d.get(bid), st.get(bid)) < FP32_TOL, bid
    assert np.abs(hd - hs).max() <= FP32_TOL * np.abs(hd).max()
    st.close()
    # (ii) Hx(w) - Hx(0) is linear in the dual; (iii) a repeated sweep is bitwise identical
    rng = np.random.default_rng(5)
    nxi, nps = d.nodes * 2 * d.nx, d.nodes * d.nu

    def hx(xi, psi):
        d.set(capi.BUF_ACC_XI, xi); d.set(capi.BUF_ACC_PSI, psi)
        d.solveStep()
        return np.concatenate([d.get(capi.BUF_PRIMAL_XI), d.get(capi.BUF_PRIMAL_PSI)])

    a = (rng.standard_normal(nxi).astype(np.float32) * 30, rng.standard_normal(nps).astype(np.float32) * 30)
    b = (rng.standard_normal(nxi).astype(np.float32) * 30, rng.standard_normal(nps).astype(np.float32) * 30)
    h0 = hx(np.zeros(nxi), np.zeros(nps))
    ha, hb = hx(*a), hx(*b)
    cxi = (2.0 * a[0].astype(np.float64) - 0.5 * b[0]).astype(np.float32)
    cpsi = (2.0 * a[1].astype(np.float64) - 0.5 * b[1]).astype(np.float32)
    hab = hx(cxi, cpsi)
    lin = 2.0 * (ha - h0) - 0.5 * (hb - h0) + h0
    assert relmax(hab, lin) < FP32_TOL
    assert np.array_equal(hx(cxi, cpsi), hab)
    d.close()
