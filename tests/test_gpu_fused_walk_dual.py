"""k_down_chain_dual (round 5, rn_set_fused_walk_dual / RAPIDNET_FUSE_DOWN_DUAL=1): the forward walk and the fused dual update of the nodes it has just walked in
ONE launch, Hx kept in LDS in between.  Element by element it is dual_elem, the arithmetic of k_dual_stage: iterates, histories and
batch counters must be the unfused path's bit for bit -- single GPU (the first descendant chain of a crown node writes it) and
sharded (crown nodes written by workgroups of their own), dense and structured, fp64 and fp32, incl. a batch whose soft-constraint
thresholds trip and which is replayed through the exact path."""
import numpy as np
import pytest

from rapidnet_amd import capi, synth
from test_gpu_sharded_batched import VECS, Ranks, dims_of

pytestmark = pytest.mark.gpu
BUFS = (capi.BUF_X, capi.BUF_U, capi.BUF_V, capi.BUF_XI, capi.BUF_PSI, capi.BUF_UPD_XI, capi.BUF_UPD_PSI, capi.BUF_ACC_XI, capi.BUF_ACC_PSI,
        capi.BUF_PRIMAL_XI, capi.BUF_PRIMAL_PSI, capi.BUF_DUAL_XI, capi.BUF_DUAL_PSI, capi.BUF_RES_XI, capi.BUF_RES_PSI)


def run(p, structured, precision, batches=(20, 17, 3), knobs=None):
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision, knobs=knobs)
    s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
    s.apgReset()
    hist = np.concatenate([s.apgIterate(n) for n in batches])      # batches of >= 16 take the optimistic (fusable) path, the last one the exact path
    out = {b: s.get(b) for b in BUFS}
    c = s.counters()
    s.close()
    return hist, out, c


@pytest.mark.parametrize("name,structured,precision,kw", [("medium", False, "f64", {}), ("medium", True, "f64", {}), ("ragged", False, "f64", {}), ("small", False, "f64", {}),
                                                          ("barcelona31", False, "f64", {}), ("medium", False, "f32", {}), ("widecrown", False, "f64", {}),
                                                          ("barcelona31_infeasible", False, "f64", {"penalty_x": 20.0, "penalty_xs": 5.0})])
def test_fused_walk_and_dual_update_is_bitwise_the_two_launches(monkeypatch, name, structured, precision, kw):
    p = synth.make_problem(name, **kw)
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "0")
    h0, o0, c0 = run(p, structured, precision)
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "1")
    h1, o1, c1 = run(p, structured, precision)
    assert c0 == c1, (c0, c1)
    assert np.array_equal(h0, h1)
    for b in BUFS:
        assert np.array_equal(o0[b], o1[b]), b


@pytest.mark.parametrize("name,world,structured,kw", [("medium", 2, False, {}), ("medium", 4, True, {}), ("ragged", 3, False, {}),
                                                      ("medium", 3, False, {"penalty_x": 20.0, "penalty_xs": 5.0})])
def test_fused_walk_and_dual_update_sharded(monkeypatch, name, world, structured, kw):
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    out = []
    for fused in ("0", "1"):
        monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", fused)
        rk = Ranks(p, world, 0, structured)
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


@pytest.mark.parametrize("name,structured,precision", [("medium", False, "f64"), ("ragged", True, "f64"), ("barcelona31", False, "f64"), ("small", False, "f32"),
                                                       ("late", False, "f64")])
def test_several_workgroups_per_chain_are_bitwise_the_two_launches(monkeypatch, name, structured, precision):
    """k_down_chain_dual with P workgroups per chain (trees and shards with fewer chains than CUs: each walks the chain as far as its own rows reach and
    updates the dual of those rows only; part 0 takes the crown rows, the last part stores the primal iterates): P = 1, 2, 3, the chain length (one
    row per workgroup; requests beyond it are clamped) and the library's own choice -- iterates, histories, counters bit for bit those of the two launches."""
    p = synth.make_problem(name)
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "0")
    h0, o0, c0 = run(p, structured, precision)
    monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", "1")
    for split in (1, 2, 3, 1000, None):
        h1, o1, c1 = run(p, structured, precision, knobs=None if split is None else {"fuse_split": split})
        assert c0 == c1, (split, c0, c1)
        assert np.array_equal(h0, h1), split
        for b in BUFS:
            assert np.array_equal(o0[b], o1[b]), (split, b)


@pytest.mark.parametrize("name,world,structured,split", [("medium", 2, False, 3), ("ragged", 3, True, 2), ("medium", 4, False, 1000)])
def test_several_workgroups_per_chain_sharded(monkeypatch, name, world, structured, split):
    """the same on shards (crown nodes written by workgroups of their own behind the K x P chain workgroups)"""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    out = []
    for fused, knobs in (("0", None), ("1", {"fuse_split": split})):
        monkeypatch.setenv("RAPIDNET_FUSE_DOWN_DUAL", fused)
        rk = Ranks(p, world, 0, structured, knobs=knobs)
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


def test_the_switch_can_be_flipped_between_batches():
    """rn_set_fused_walk_dual on a live context: batches with and without the fusion alternate and the iterates are those of a context
    that never fused (what bench.py's same-context A/B relies on)."""
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    ref = capi.Solver(p["network"], p["tree"], p["config"])
    ref.initialiseSmpcController(dh, ah)
    ref.apgReset()
    h0 = np.concatenate([ref.apgIterate(20) for _ in range(4)])
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    h1 = []
    for on in (1, 0, 1, 0):
        s.setFusedWalkDual(on)
        h1.append(s.apgIterate(20))
    assert np.array_equal(h0, np.concatenate(h1))
    for b in BUFS:
        assert np.array_equal(ref.get(b), s.get(b)), b
    ref.close(); s.close()


def test_more_chains_than_partial_slots_take_the_two_launches(monkeypatch):
    """k_down_chain_dual leaves one partial per workgroup (= per chain) in a buffer of 16 384 entries: a tree with more chains than
    that must fall back to the two-launch form instead of writing past the buffer (round-5 advisor finding).  16 510 chains of a tiny
    network, the fused form forced, under the guard (red zones around every buffer): same bits as the unfused run, no red-zone byte
    touched, and the oracle's iterates."""
    from oracle.oracle import Oracle

    synth.CONFIGS.setdefault("manychains", (24, 3, 6, 4, 2, 4, [130, 127]))
    p = synth.make_problem("manychains")
    monkeypatch.setenv("RAPIDNET_GUARD", "1")
    outs = []
    for fused in (0, 1):
        s = capi.Solver(p["network"], p["tree"], p["config"])
        s.setFusedWalkDual(fused)
        s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
        s.apgReset()
        hist = np.concatenate([s.apgIterate(20), s.apgIterate(16)])
        outs.append((hist, {b: s.get(b) for b in BUFS}))
        assert s.guardCheck() == 0
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    for b in BUFS:
        assert np.array_equal(outs[0][1][b], outs[1][1][b]), b
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(*synth.forecast_at(p["forecast"], 0))
    oh = o.apg(36)
    assert np.abs(outs[1][0] - oh).max() <= 1e-9 * np.abs(oh).max()
    for b, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi")):
        ref = o.get(nm)
        assert np.abs(outs[1][1][b] - ref).max() <= 1e-9 * np.abs(ref).max(), nm
