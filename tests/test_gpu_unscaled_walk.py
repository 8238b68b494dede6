"""k_down_chain<T, UNSC> + k_dual_stage<..., SCALE> (round 5; RN_KNOB_UNSCALED_WALK, default on): in the inner iterations of an optimistic batch the forward walk
leaves the PRIMAL values in the Hx buffer and the dual update applies sqrt(p_i) d_k -- the factor it holds for the bounds anyway -- so the walk
requests no preconditioner entries.  The product is the same two roundings in either kernel: iterates, histories and batch counters must be the
scaled walk's bit for bit -- single GPU and sharded, dense and structured, fp64 and fp32, incl. a batch whose soft-constraint thresholds trip and
which is replayed through the exact path."""
import numpy as np
import pytest

from rapidnet_amd import synth
from test_gpu_fused_walk_dual import BUFS, run
from test_gpu_sharded_batched import VECS, Ranks, dims_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,structured,precision,kw", [("medium", False, "f64", {}), ("medium", True, "f64", {}), ("ragged", False, "f64", {}), ("small2", False, "f64", {}),
                                                          ("barcelona31", False, "f64", {}), ("medium", False, "f32", {}), ("late", False, "f64", {}),
                                                          ("barcelona31_infeasible", False, "f64", {"penalty_x": 20.0, "penalty_xs": 5.0})])
def test_unscaled_walk_is_bitwise_the_scaled_one(monkeypatch, name, structured, precision, kw):
    p = synth.make_problem(name, **kw)
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "0")      # (the fused walk + dual update has no Hx hand-off to scale)
    h0, o0, c0 = run(p, structured, precision, knobs={"unscaled_walk": 0})
    h1, o1, c1 = run(p, structured, precision, knobs={"unscaled_walk": 1})
    assert c0 == c1, (c0, c1)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b


@pytest.mark.parametrize("name,world,structured,kw", [("medium", 2, False, {}), ("medium", 4, True, {}), ("ragged", 3, False, {}),
                                                      ("medium", 3, False, {"penalty_x": 20.0, "penalty_xs": 5.0})])
def test_unscaled_walk_sharded(monkeypatch, name, world, structured, kw):
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    out = []
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "0")
    for on in (0, 1):
        rk = Ranks(p, world, 0, structured, knobs={"unscaled_walk": on})
        try:
            def solve(s):
                s.initialiseSmpcController(dh, ah)
                s.apgReset()
                return s.counters(), np.concatenate([s.apgIterate(20), s.apgIterate(5)])

            res = rk.run(solve)
            d = dims_of(rk.shards[0])
            out.append((res[0][0], res[0][1], [rk.gathered(bid, d[dm]) for bid, _, dm in VECS]))
        finally:
            rk.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a, b)
