"""Exchange mode 2 (rn_set_exchange_mode): device-resident batches that do not store the accelerated dual between their
iterations (the dual update without its w_next stream).  The sweep and the next dual update derive
w = (1 + l) y_t - l y_{t-1} from the two iterates with the roundings the stored vector would have, so the results must be
BITWISE those of mode 1 -- iterates, history, and every buffer a caller can read after a batch (the last iteration of a batch
stores both w_t and w_{t+1})."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, synth
from tests.test_gpu_sharded_batched import Ranks

pytestmark = pytest.mark.gpu
BUFS = (capi.BUF_X, capi.BUF_U, capi.BUF_V, capi.BUF_XI, capi.BUF_PSI, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_ACC_XI, capi.BUF_ACC_PSI,
        capi.BUF_PRIMAL_XI, capi.BUF_PRIMAL_PSI, capi.BUF_DUAL_XI, capi.BUF_DUAL_PSI, capi.BUF_RES_XI, capi.BUF_RES_PSI)


def solve(p, mode, batches, structured=False, precision="f64"):
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    s.initialiseSmpcController(dh, ah)
    s.setExchangeMode(mode)
    s.apgReset()
    hist = []
    for n in batches:
        hist += list(s.apgIterate(n))
    out = {b: s.get(b) for b in BUFS}
    counters = s.counters()
    s.close()
    return np.array(hist), out, counters


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("medium", True, "f64"), ("ragged", False, "f64"),
                                                       ("barcelona31", False, "f64"), ("medium", False, "f32"), ("odd", False, "f64")])
def test_lazy_batches_are_bitwise_the_stored_ones(name, structured, precision):
    p = synth.make_problem(name)
    batches = (20, 1, 16, 33, 5)      # optimistic batches of several lengths, a one-iteration batch and an exact (short) one in between
    h1, o1, c1 = solve(p, 1, batches, structured, precision)
    h2, o2, c2 = solve(p, 2, batches, structured, precision)
    assert c1 == c2
    assert np.array_equal(h1, h2)
    for b in BUFS:
        assert np.array_equal(o1[b], o2[b]), b


def test_lazy_batches_match_the_oracle_and_survive_a_replay():
    # small penalties: the distances exceed gamma / lambda, the optimistic batch is replayed through the exact path
    p = synth.make_problem("small", penalty_x=20.0, penalty_xs=5.0)
    h1, o1, c1 = solve(p, 1, (20, 20))
    h2, o2, c2 = solve(p, 2, (20, 20))
    assert c2["replayed"] > 0 and c1 == c2
    assert np.array_equal(h1, h2)
    for b in BUFS:
        assert np.array_equal(o1[b], o2[b]), b
    o = Oracle(p["network"], p["tree"], p["config"])
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o.initialise(dh, ah)
    ref = o.apg(40)
    assert np.abs(h2 - np.asarray(ref)).max() <= 1e-9 * np.abs(np.asarray(ref)).max()


@pytest.mark.parametrize("world,cut", [(2, 0), (4, 1)])
def test_lazy_batches_sharded(world, cut):
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    res = []
    for mode in (1, 2):
        rk = Ranks(p, world, cut)
        for s in rk.shards:
            s.setExchangeMode(mode)

        def fn(s):
            s.initialiseSmpcController(dh, ah)
            s.apgReset()
            s.apgIterate(20, history=False)
            s.apgIterate(25, history=False)
            return True

        rk.run(fn)
        res.append({b: rk.gathered(b, d) for b, d in ((capi.BUF_X, rk.shards[0].nx), (capi.BUF_UPD_XI, 2 * rk.shards[0].nx), (capi.BUF_ACC_XI, 2 * rk.shards[0].nx),
                                                        (capi.BUF_ACC_PSI, rk.shards[0].nu))})
        rk.close()
    for b in res[0]:
        assert np.array_equal(res[0][b], res[1][b]), b
