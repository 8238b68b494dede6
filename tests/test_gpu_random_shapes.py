"""Seeded random network / tree shapes through the HIP path against the CPU oracle.

The streaming kernel chooses its span (columns per span, slots per thread) from the operator-column size, the slab
products their tile counts from nv / nu / nx, and the chain kernels their prefetch batches from N: a sweep over odd
sizes exercises the combinations that the named configs of rapidnet_amd.synth do not hit (columns that are / are not
whole 16-byte slots or 128-byte lines, nv just below / above a multiple of 16, trees whose crown is 1, 2 or 3 stages).
Same tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, partition, synth

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9
FP32_TOL = 2e-4


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def random_config(seed):
    rng = np.random.default_rng(9000 + seed)
    nx = int(rng.integers(2, 40))
    nu = int(nx + rng.integers(1, 40))          # nu > nx keeps the network generator's incidence structure valid
    ne = int(rng.integers(1, max(2, min(nu - 1, nx))))
    nd = int(rng.integers(2, 30))
    N = int(rng.integers(4, 14))
    depth = int(rng.integers(1, min(3, N - 2) + 1))   # branching stops before the last stages (the shape the reference supports)
    branching = [int(rng.integers(2, 5)) for _ in range(depth)]
    return (100 + seed, nx, nu, nd, ne, N, branching)


VARIANTS = [(s, prec, structured) for s in range(12) for prec, structured in ((("f64", False),) if s % 3 else (("f64", False), ("f64", True), ("f32", False)))]


@pytest.mark.parametrize("seed,precision,structured", VARIANTS)
def test_random_shape_matches_oracle(seed, precision, structured):
    name = "_random_%d" % seed
    synth.CONFIGS[name] = random_config(seed)
    try:
        p = synth.make_problem(name)
    finally:
        del synth.CONFIGS[name]
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision)
    o.initialise(dh, ah)
    s = capi.Solver(p["network"], p["tree"], p["config"], precision=precision, structured=structured)
    s.initialiseSmpcController(dh, ah)
    tol = REL_TOL if precision == "f64" else FP32_TOL
    iters = 15
    hist, ohist = s.algorithmApg(iters), o.apg(iters)
    for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"),
                    (capi.BUF_PRIMAL_XI, "primalXi"), (capi.BUF_PRIMAL_PSI, "primalPsi"), (capi.BUF_DUAL_XI, "dualXi"),
                    (capi.BUF_RES_XI, "resXi"), (capi.BUF_RES_PSI, "resPsi")):
        assert relmax(s.get(bid), o.get(nm)) < tol, (nm, synth_shape(p))
    assert np.abs(hist - ohist).max() <= (1e-9 if precision == "f64" else 1e-3) * max(np.abs(ohist).max(), 1.0)
    s.close()


def synth_shape(p):
    return {k: p["tree"][k][0] for k in ("N", "K", "nodes")}


@pytest.mark.parametrize("seed", [1, 2, 3, 8, 11])
def test_random_shape_sharded_single_rank_rccl(seed):
    """The same shapes through the sharded code path (cut-sums launch with the folded bookkeeping, real ncclAllReduce on a
    one-rank communicator, presummed exchange stage, crown nodes dealt to the chain workgroups), optimistic and exact."""
    name = "_random_%d" % seed
    synth.CONFIGS[name] = random_config(seed)
    try:
        p = synth.make_problem(name)
    finally:
        del synth.CONFIGS[name]
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    ohist = o.apg(11)
    cut = partition.default_cut_stage(p["tree"])
    for optimistic in (True, False):
        s = capi.Solver(p["network"], p["tree"], p["config"])
        s.commInit(0, 1, capi.comm_unique_id())
        s.setCutStage(cut, partition.cut_children_moments(p["tree"], cut))
        s.setExchangeMode(optimistic)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        hist = np.concatenate([s.apgIterate(4), s.apgIterate(7)])
        for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi")):
            assert relmax(s.get(bid), o.get(nm)) < REL_TOL, (nm, optimistic)
        assert np.abs(hist - ohist).max() <= 1e-9 * max(np.abs(ohist).max(), 1.0)
        s.close()
