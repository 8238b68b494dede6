"""The chain-fused form of the sweep's helper path (rapidnet_amd/csrc/chain_kernels.hpp: k_chain_sweep + k_cut_partial_sums + k_crown_small,
the crown's offsets added inside k_dual_stage or by k_hx_finish; rn_set_sweep_form) against the six-launch form and against the CPU oracle.
The two forms compute the same sums in a different association (the forward sums of a chain start at zero and the crown's contribution is
added at the end; the factor -1/(2p) of the v product goes onto its right-hand side), so they agree to rounding, not bitwise; both are
within 1e-9 of the oracle.  Within the chain-fused form every consumer of Hx forms it with ONE expression (cf_primal), so its own paths --
optimistic batches, the exact path, a replayed batch, step-wise calls -- agree bit for bit."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu
BUFS = ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"),
        (capi.BUF_XI, "xi"), (capi.BUF_PSI, "psi"), (capi.BUF_PRIMAL_XI, "primalXi"), (capi.BUF_PRIMAL_PSI, "primalPsi"),
        (capi.BUF_DUAL_XI, "dualXi"), (capi.BUF_DUAL_PSI, "dualPsi"))


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def solve(p, form, structured, precision, batches, dh, ah, expect=None):
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision)
    try:
        active = s.setSweepForm(form)
        if expect is not None:
            assert active == expect, (form, active)
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        hist = np.concatenate([s.apgIterate(n) for n in batches])
        return {nm: s.get(bid) for bid, nm in BUFS}, hist, s.counters(), active
    finally:
        s.close()


# chains from stage 1 (barcelona31, fan, horizon2: one node per chain, tall: 290 rows per operator column), 2 (tiny, small2 with unequal children
# counts, medium), 3 (small, ragged: unequal counts), 4 (ragged2) and 5 (late: single-child crown nodes); an odd ny (small, odd, ragged ...: the flat
# dual update, so Hx is finished by k_hx_finish); structured operators; fp32
@pytest.mark.parametrize("name,structured,precision", [("tiny", False, "f64"), ("horizon2", False, "f64"), ("fan", False, "f64"), ("small2", False, "f64"),
                                                       ("odd", False, "f64"), ("small", False, "f64"), ("ragged", False, "f64"), ("ragged2", False, "f64"),
                                                       ("late", False, "f64"), ("tall", False, "f64"),
                                                       ("medium", False, "f64"), ("medium", True, "f64"), ("barcelona31", False, "f64"),
                                                       ("barcelona31", True, "f64"), ("medium", False, "f32"), ("barcelona31", False, "f32")])
def test_chain_fused_sweep_matches_the_six_launch_form_and_the_oracle(name, structured, precision):
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"], precision=precision, alias_operators=name not in ("horizon2", "late", "ragged2"))
    o.initialise(dh, ah)
    iters = 44 if precision == "f64" else 36
    ohist = o.apg(iters)
    out = {form: solve(p, form, structured, precision, (iters - 20, 17, 3), dh, ah, expect=form) for form in (1, 0)}
    tol_pair, tol_oracle = (1e-11, 1e-9) if precision == "f64" else (2e-4, 2e-4)
    for _, nm in BUFS:
        assert relmax(out[1][0][nm], out[0][0][nm]) < tol_pair, nm
        assert relmax(out[1][0][nm], o.get(nm)) < tol_oracle, nm
    assert np.abs(out[1][1] - ohist).max() <= tol_oracle * np.abs(ohist).max()
    assert out[1][2] == out[0][2]


@pytest.mark.parametrize("name,structured,precision,env", [("medium", False, "f64", {"RAPIDNET_CF_REG": "0"}), ("small2", False, "f64", {"RAPIDNET_CF_REG": "0"}),
                                                           ("barcelona31", False, "f64", {"RAPIDNET_CF_REG": "0"}), ("barcelona31", True, "f32", {"RAPIDNET_CF_REG": "0"})])
def test_the_kernels_that_stream_their_operator_fragments_from_l2(monkeypatch, name, structured, precision, env):
    """operators of at most 40 / 28 k-steps take the register-resident kernels (k_chain_sweep_reg, k_crown_small_reg); larger ones ("tall" above)
    and RAPIDNET_CF_REG=0 the ones whose MFMA loops stream the A fragments from L2 (k_chain_sweep, k_crown_small)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    test_chain_fused_sweep_matches_the_six_launch_form_and_the_oracle(name, structured, precision)


def test_shapes_outside_the_conditions_take_the_six_launch_form():
    # one chain from the root (no crown), also with N = 1; a crown of 127 nodes; more components per node than a workgroup has threads
    for name in ("toy", "horizon1", "deep", "widecrown"):
        p = synth.make_problem(name)
        s = capi.Solver(p["network"], p["tree"], p["config"])
        try:
            assert s.setSweepForm(1) == 0, name
        finally:
            s.close()


@pytest.mark.parametrize("name,kw", [("medium", {}), ("barcelona31_infeasible", {"penalty_x": 20.0, "penalty_xs": 5.0})])
def test_every_path_of_the_chain_fused_form_forms_hx_with_the_same_bits(name, kw):
    """optimistic batches (k_dual_stage adds the offsets), short batches (the exact path: k_hx_finish in front of the dual update and its fix-up
    pass) and -- second problem -- batches whose soft-constraint thresholds trip and are replayed: the same iterates bit for bit"""
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    a = solve(p, 1, False, "f64", (20, 20), dh, ah, expect=1)
    b = solve(p, 1, False, "f64", (5,) * 8, dh, ah, expect=1)
    assert np.array_equal(a[1], b[1])
    for _, nm in BUFS:
        assert np.array_equal(a[0][nm], b[0][nm]), nm
    if kw:
        assert a[2]["replayed"] >= 1, a[2]


def test_the_form_can_be_switched_between_batches():
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    o.apg(60)
    s = capi.Solver(p["network"], p["tree"], p["config"])
    try:
        s.initialiseSmpcController(dh, ah)
        s.apgReset()
        for k in range(3):
            assert s.setSweepForm(k % 2) == k % 2
            s.apgIterate(20)
        for bid, nm in BUFS:
            assert relmax(s.get(bid), o.get(nm)) < 1e-9, nm
    finally:
        s.close()
