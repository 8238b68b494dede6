"""GPU parity of the global-FBE and NAMA outer loops through the C-ABI.

Two anchors, as for the APG path:
  * the reference's own known-answer vectors (smpcFbeTest.json / smpcNamaTest.json), fed through the HIP path exactly
    as Testing::testSmpcFbeController / testSmpcNamaController feed the reference (src/test/Testing.cu:536-590,
    TestSmpcController.cu:403-1040) -- tolerances are those of the 7-digit prints (see tests/test_oracle_fbe_nama.py);
  * the CPU oracle on seeded synthetic problems, step by step and over whole loops, at 1e-9 of each vector's scale.
The line searches branch on computed values (value <= previous value, skip rule of the L-BFGS update): the loops are
compared through the accepted step lengths tau (must be identical) as well as through the iterates.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, forecast_at
from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9


def relmax(a, b):
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    assert a.shape == b.shape
    assert np.isfinite(a).all() and np.isfinite(b).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def rel_err(a, b, floor=1.0):
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    assert a.size == b.size
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


ALGS = ["globalFbeAlgorithm", "namaAlgorithm"]


# ---------------------------------------------------------------------------------------------------------------
# the reference's known-answer vectors through the HIP path
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=ALGS)
def case(request, ref_fixture):
    s = capi.Solver(ref_fixture["network"], ref_fixture["tree"], ref_fixture["config"])
    dh, ah = forecast_at(ref_fixture["forecast"], 1)  # timeInst = 1, Testing.cu:540-542
    s.initialiseSmpcController(dh, ah)
    s.setAlgorithm(request.param, 5)
    key = "smpc_fbe" if request.param == "globalFbeAlgorithm" else "smpc_nama"
    return request.param, s, ref_fixture[key]


def test_fixture_hessian_oracle(case):
    name, s, f = case
    fbe = name == "globalFbeAlgorithm"
    s.set(capi.BUF_LBFGS_CUR_YVEC_XI if fbe else capi.BUF_RES_XI, f["fixedPointResidualXi"])
    s.set(capi.BUF_LBFGS_CUR_YVEC_PSI if fbe else capi.BUF_RES_PSI, f["fixedPointResidualPsi"])
    s.computeHessianOracalGlobalFbe()
    kx, ku = ("fbeHessianDirXdir", "fbeHessianDirUdir") if fbe else ("ameFixedPointDirXdir", "ameFixedPointDirUdir")
    assert relmax(s.get(capi.BUF_UDIR), f[ku]) < 1e-6
    assert relmax(s.get(capi.BUF_XDIR), f[kx]) < 1e-6


def test_fixture_gradient_and_nama_residual(case):
    name, s, f = case
    s.set(capi.BUF_RES_XI, f["fixedPointResidualXi"]); s.set(capi.BUF_RES_PSI, f["fixedPointResidualPsi"])
    if name == "globalFbeAlgorithm":
        s.computeGradientFbe()
        assert rel_err(s.get(capi.BUF_LBFGS_CUR_YVEC_XI), f["fbeGradXi"]) < 2e-6
        assert rel_err(s.get(capi.BUF_LBFGS_CUR_YVEC_PSI), f["fbeGradPsi"]) < 2e-6
    else:
        s.updateFixedPointResidualNamaAlgorithm()
        assert rel_err(s.get(capi.BUF_LBFGS_CUR_YVEC_XI), f["lbfgsCurrentYvecXi"]) < 1e-6
        assert rel_err(s.get(capi.BUF_LBFGS_CUR_YVEC_PSI), f["lbfgsCurrentYvecPsi"]) < 1e-6


def test_fixture_value_fbe(case):
    name, s, f = case
    s.set(capi.BUF_RES_XI, f["fixedPointResidualXi"]); s.set(capi.BUF_RES_PSI, f["fixedPointResidualPsi"])
    s.set(capi.BUF_ACC_XI, f["acceleXi"]); s.set(capi.BUF_ACC_PSI, f["accelePsi"])
    s.set(capi.BUF_U, f["U"])
    assert abs(s.computeValueFbe() / f["fbeObjDual"][0] - 1) < 2e-6


def test_fixture_lbfgs_direction(case):
    name, s, f = case
    n = s.nodes * (2 * s.nx + s.nu)
    s.set(capi.BUF_PREV_XI, f["xi"]); s.set(capi.BUF_PREV_PSI, f["psi"])
    s.set(capi.BUF_XI, f["acceleXi"]); s.set(capi.BUF_PSI, f["accelePsi"])
    s.set(capi.BUF_LBFGS_CUR_YVEC_XI, f["lbfgsCurrentYvecXi"]); s.set(capi.BUF_LBFGS_CUR_YVEC_PSI, f["lbfgsCurrentYvecPsi"])
    s.set(capi.BUF_LBFGS_PREV_YVEC_XI, f["lbfgsPreviousYvecXi"]); s.set(capi.BUF_LBFGS_PREV_YVEC_PSI, f["lbfgsPreviousYvecPsi"])
    S, Y = np.array(f["matS"]).reshape(5, n), np.array(f["matY"]).reshape(5, n)
    for c in range(5):
        s.lbfgsColumn(0, c, S[c]); s.lbfgsColumn(1, c, Y[c])
    inv = np.array(f["vecInvRho"], float)
    rho = np.zeros(6); rho[:5] = np.where(inv != 0, 1 / np.where(inv != 0, inv, 1), 0)
    s.lbfgsState(int(f["colLbfgs"][0]), int(f["memLbfgs"][0]), float(f["H"][0]), rho)
    s.computeLbfgsDirection()
    col, mem, H, rho2 = s.lbfgsState()
    assert col == int(f["updateColLbfgs"][0]) and mem == int(f["updateMemLbfgs"][0])
    assert abs(H / f["updateH"][0] - 1) < 2e-6
    assert rel_err(1 / rho2[:5], f["updateVecInvRho"], floor=1e-300) < 2e-6
    S2 = np.concatenate([s.lbfgsColumn(0, c) for c in range(5)])
    Y2 = np.concatenate([s.lbfgsColumn(1, c) for c in range(5)])
    assert rel_err(S2, f["updateMatS"]) < 1e-4 and rel_err(Y2, f["updateMatY"]) < 1e-4
    scale = np.abs(np.array(f["lbfgsDirXi"])).max()
    assert np.abs(s.get(capi.BUF_LBFGS_DIR_XI) - f["lbfgsDirXi"]).max() < 1e-4 * scale
    assert np.abs(s.get(capi.BUF_LBFGS_DIR_PSI) - f["lbfgsDirPsi"]).max() < 1e-4 * scale


def test_fixture_line_search(case):
    name, s, f = case
    fbe = name == "globalFbeAlgorithm"
    s.set(capi.BUF_RES_XI, f["fixedPointResidualXi"]); s.set(capi.BUF_RES_PSI, f["fixedPointResidualPsi"])
    s.set(capi.BUF_ACC_XI, f["acceleXi"]); s.set(capi.BUF_ACC_PSI, f["accelePsi"])
    s.set(capi.BUF_X, f["X"]); s.set(capi.BUF_U, f["U"])
    s.set(capi.BUF_LBFGS_DIR_XI, f["lbfgsDirXi"]); s.set(capi.BUF_LBFGS_DIR_PSI, f["lbfgsDirPsi"])
    if fbe:
        s.set(capi.BUF_LBFGS_CUR_YVEC_XI, f["fbeGradXi"]); s.set(capi.BUF_LBFGS_CUR_YVEC_PSI, f["fbeGradPsi"])
    s.set(capi.BUF_PRIMAL_XI, f["primalX"]); s.set(capi.BUF_PRIMAL_PSI, f["primalU"])
    val = s.computeValueFbe()
    tau = s.computeLineSearchLbfgsUpdate(val) if fbe else s.computeLineSearchAmeLbfgsUpdate(val)
    assert abs(val / f["fbeObjDual"][0] - 1) < 2e-6
    assert abs(tau - f["tau"][0]) < 1e-12
    assert rel_err(s.get(capi.BUF_ACC_XI), f["updateXi"]) < 1e-5
    assert rel_err(s.get(capi.BUF_ACC_PSI), f["updatePsi"]) < 1e-5
    assert rel_err(s.get(capi.BUF_RES_XI), f["updateResidualXi"]) < 2e-4
    assert rel_err(s.get(capi.BUF_RES_PSI), f["updateResidualPsi"]) < 2e-4


def test_fixture_dual_update(case):
    name, s, f = case
    s.set(capi.BUF_XI, f["acceleXi"]); s.set(capi.BUF_PSI, f["accelePsi"])
    s.set(capi.BUF_ACC_XI, f["updateXi"]); s.set(capi.BUF_ACC_PSI, f["updatePsi"])
    s.set(capi.BUF_RES_XI, f["updateResidualXi"]); s.set(capi.BUF_RES_PSI, f["updateResidualPsi"])
    s.set(capi.BUF_LBFGS_CUR_YVEC_XI, f["lbfgsCurrentYvecXi"]); s.set(capi.BUF_LBFGS_CUR_YVEC_PSI, f["lbfgsCurrentYvecPsi"])
    s.dualUpdate()
    assert rel_err(s.get(capi.BUF_XI), f["finalUpdateXi"]) < 2e-6
    assert rel_err(s.get(capi.BUF_PSI), f["finalUpdatePsi"]) < 2e-6
    assert np.array_equal(s.get(capi.BUF_LBFGS_PREV_YVEC_XI), np.array(f["lbfgsCurrentYvecXi"], float))
    assert np.array_equal(s.get(capi.BUF_PREV_XI), np.array(f["acceleXi"], float))
    assert np.array_equal(s.get(capi.BUF_PREV_PSI), np.array(f["accelePsi"], float))
    assert np.array_equal(s.get(capi.BUF_ACC_XI), s.get(capi.BUF_XI))


# ---------------------------------------------------------------------------------------------------------------
# HIP path vs oracle on seeded problems
# ---------------------------------------------------------------------------------------------------------------
FBE_PAIRS = [(capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_XI, "xi"), (capi.BUF_PSI, "psi"), (capi.BUF_ACC_XI, "accXi"),
             (capi.BUF_ACC_PSI, "accPsi"), (capi.BUF_PRIMAL_XI, "primalXi"), (capi.BUF_PRIMAL_PSI, "primalPsi"),
             (capi.BUF_DUAL_XI, "dualXi"), (capi.BUF_DUAL_PSI, "dualPsi"), (capi.BUF_RES_XI, "resXi"), (capi.BUF_RES_PSI, "resPsi"),
             (capi.BUF_PREV_XI, "prevXi"), (capi.BUF_PREV_PSI, "prevPsi"), (capi.BUF_LBFGS_DIR_XI, "dirXi"),
             (capi.BUF_LBFGS_DIR_PSI, "dirPsi"), (capi.BUF_XDIR, "xdir"), (capi.BUF_UDIR, "udir"),
             (capi.BUF_PRIMAL_XI_DIR, "primalXiDir"), (capi.BUF_PRIMAL_PSI_DIR, "primalPsiDir")]


def cur_names(alg):
    return ("gradXi", "gradPsi", "prevGradXi", "prevGradPsi") if alg == "globalFbeAlgorithm" else ("curResXi", "curResPsi", "prevResXi", "prevResPsi")


def compare_fbe(s, o, alg, tol, what):
    worst = {}
    for bid, name in FBE_PAIRS:
        worst[name] = relmax(s.get(bid), o.get(name))
    cx, cp, px, pp = cur_names(alg)
    for bid, name in ((capi.BUF_LBFGS_CUR_YVEC_XI, cx), (capi.BUF_LBFGS_CUR_YVEC_PSI, cp), (capi.BUF_LBFGS_PREV_YVEC_XI, px),
                      (capi.BUF_LBFGS_PREV_YVEC_PSI, pp)):
        worst[name] = relmax(s.get(bid), o.get(name))
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, "%s mismatch vs oracle: %s" % (what, bad)


def make_pair(name, alg, structured=False, m=5, **kw):
    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.set_algorithm(alg, m)
    o.initialise(dh, ah)
    o.fbe_reset()
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured)
    s.initialiseSmpcController(dh, ah)
    s.setAlgorithm(alg, m)
    return p, o, s


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("name,structured", [("tiny", False), ("small", False), ("small", True), ("odd", False), ("medium", True)])
def test_loop_matches_oracle(name, structured, alg):
    """algorithmGlobalFbe / algorithmNama: same step lengths, same primal infeasibilities, same iterates."""
    p, o, s = make_pair(name, alg, structured)
    iters = 12
    ho, vo, to = o.fbe_nama(iters)
    hs, vs, ts = (s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama)(iters)
    assert np.array_equal(ts, to), (ts, to)
    assert relmax(vs, vo) < REL_TOL
    assert relmax(hs, ho) < 1e-7   # a signed arg-max entry: ties between equal |entries| may resolve to either sign
    compare_fbe(s, o, alg, 1e-8, "%s %s after %d iterations" % (alg, name, iters))
    assert s.lbfgsState()[:2] == o.lbfgs_state()[:2]
    # the line searches evaluated their trials in batches of candidates (two passes + one read-back per batch), none trial by trial
    c = s.fbeCounters()
    assert c["searches"] >= 1 and c["sequential"] == 0 and c["searches"] <= c["batches"] <= 3 * c["searches"], c


@pytest.mark.parametrize("alg", ALGS)
def test_loop_matches_oracle_at_the_headline_dimensions(alg):
    """the 31-scenario Barcelona-sized tree (nx = 63, nu = 114, nv = 97: the headline's per-node dimensions, so the streaming kernel runs
    its two-slots-per-thread instantiation -- with two right-hand sides in NAMA -- and k_value_mfma its eight-tile, 32-k-step form
    with W in registers) against the oracle: same step lengths, same values, same iterates"""
    p, o, s = make_pair("barcelona31", alg)
    iters = 8
    ho, vo, to = o.fbe_nama(iters)
    hs, vs, ts = (s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama)(iters)
    assert np.array_equal(ts, to), (ts, to)
    assert relmax(vs, vo) < REL_TOL
    assert relmax(hs, ho) < 1e-7
    compare_fbe(s, o, alg, 1e-8, "%s barcelona31 after %d iterations" % (alg, iters))
    c = s.fbeCounters()
    assert c["sequential"] == 0 and (alg != "namaAlgorithm" or c["sweep_pairs"] == iters - 1), c


@pytest.mark.parametrize("alg", ALGS)
def test_algorithm_selected_before_the_factor_step(alg):
    """the order of the C++ host (Engine::create selects the algorithm right after rn_create, the factor step comes later): everything
    the loops derive from the uploaded system -- k_value_mfma's padded copy of W among it -- must be (re)built by the factor step"""
    p = synth.make_problem("small")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.set_algorithm(alg, 5); o.initialise(dh, ah); o.fbe_reset()
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.setAlgorithm(alg, 5)
    s.initialiseSmpcController(dh, ah)
    ho, vo, to = o.fbe_nama(8)
    hs, vs, ts = (s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama)(8)
    assert np.array_equal(ts, to) and relmax(vs, vo) < REL_TOL
    compare_fbe(s, o, alg, 1e-8, "algorithm selected before the factor step")


@pytest.mark.parametrize("name,precision", [("medium", "f64"), ("odd", "f64"), ("small", "f32")])
def test_nama_pair_of_hessian_sweeps_is_bitwise_the_two_sweeps(name, precision):
    """NAMA's two Hessian oracles of an iteration (SmpcController.cu:1331, :1341-1345) in ONE pass over the operator blocks
    (k_stream_gemv with two right-hand sides): every buffer and every accepted step as with the two sweeps one after the other."""
    p = synth.make_problem(name)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    runs = []
    for pair in (1, 0):
        s = capi.Solver(p["network"], p["tree"], p["config"], precision=precision, knobs={"nama_pair": pair})
        s.initialiseSmpcController(dh, ah)
        s.setAlgorithm("namaAlgorithm", 5)
        h, v, t = s.algorithmNama(10)
        bufs = [s.get(b) for b, _ in FBE_PAIRS] + [s.get(capi.BUF_X), s.get(capi.BUF_U)]
        runs.append((h, v, t, bufs, s.fbeCounters()))
        s.close()
    (h1, v1, t1, b1, c1), (h0, v0, t0, b0, c0) = runs
    assert c1["sweep_pairs"] >= 8 and c0["sweep_pairs"] == 0, (c1, c0)
    assert np.array_equal(t1, t0) and np.array_equal(v1, v0) and np.array_equal(h1, h0)
    for x, y in zip(b1, b0):
        assert np.array_equal(x, y)


def test_nama_on_a_cut_context_without_a_communicator_takes_the_two_sweeps():
    """A one-rank context with a cut stage set and no communicator (what the FBE entry points explicitly allow) keeps ONE buffer for
    the cut parents' sums: the pair of Hessian sweeps -- whose two helper chains run side by side on two streams -- would both write
    and read it (round 4's advisor finding).  Such a context takes the two sweeps one after the other and its loop is the plain
    context's: same accepted steps, same values, same iterates, twice in a row."""
    from rapidnet_amd import partition

    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    cut = partition.default_cut_stage(p["tree"])
    runs = []
    for kind in ("plain", "cut", "cut"):
        s = capi.Solver(p["network"], p["tree"], p["config"])
        if kind == "cut":
            s.commInit(0, 1, None)
            s.setCutStage(cut, partition.cut_children_moments(p["tree"], cut))
        s.initialiseSmpcController(dh, ah)
        s.setAlgorithm("namaAlgorithm", 5)
        h, v, t = s.algorithmNama(10)
        runs.append((h, v, t, s.get(capi.BUF_X), s.get(capi.BUF_U), s.fbeCounters()))
        s.close()
    plain, cut1, cut2 = runs
    assert plain[5]["sweep_pairs"] >= 8 and cut1[5]["sweep_pairs"] == 0, (plain[5], cut1[5])
    for a, b in zip(cut1[:5], cut2[:5]):
        assert np.array_equal(a, b)                      # deterministic
    assert np.array_equal(plain[2], cut1[2])             # the same accepted steps
    assert relmax(cut1[1], plain[1]) < 1e-12 and relmax(cut1[3], plain[3]) < 1e-10 and relmax(cut1[4], plain[4]) < 1e-10


@pytest.mark.parametrize("alg", ALGS)
def test_steps_match_oracle(alg):
    """every step of one iteration, each fed by the previous one, compared after each call"""
    p, o, s = make_pair("small", alg)
    fbe = alg == "globalFbeAlgorithm"
    for it in range(4):
        o.solve_step(); s.solveStep()
        o.prox(); s.proximalFunG()
        o.residual(); s.computeFixedPointResidual()
        if fbe:
            o.gradient_fbe(); s.computeGradientFbe()
        else:
            o.nama_residual(); s.updateFixedPointResidualNamaAlgorithm()
        compare_fbe(s, o, alg, REL_TOL, "it %d gradient" % it)
        if it > 0:
            vo, vs = o.value_fbe(), s.computeValueFbe()
            assert abs(vs - vo) <= REL_TOL * abs(vo)
            o.lbfgs_direction(); s.computeLbfgsDirection()
            compare_fbe(s, o, alg, REL_TOL, "it %d lbfgs direction" % it)
            co, mo, Ho = o.lbfgs_state()
            cs, ms, Hs, rho = s.lbfgsState()
            assert (cs, ms) == (co, mo) and abs(Hs - Ho) <= REL_TOL * abs(Ho)
            assert relmax(rho, o.get("rho")) < REL_TOL
            n = s.nodes * (2 * s.nx + s.nu)
            nxi = s.nodes * 2 * s.nx
            for which, nm in ((0, "matS"), (1, "matY")):
                got = s.lbfgsColumn(which, cs)
                ref = o.get(nm)[cs * n:(cs + 1) * n]
                assert relmax(got[:nxi], ref[:nxi]) < REL_TOL and relmax(got[nxi:], ref[nxi:]) < REL_TOL
            to = o.line_search_fbe(vo) if fbe else o.line_search_ame(vo)
            ts = s.computeLineSearchLbfgsUpdate(vs) if fbe else s.computeLineSearchAmeLbfgsUpdate(vs)
            assert ts == to
            compare_fbe(s, o, alg, REL_TOL, "it %d line search" % it)
        o.dual_update(); s.dualUpdate()
        compare_fbe(s, o, alg, REL_TOL, "it %d dual update" % it)
        assert abs(s.updatePrimalInfeasibity() - o.primal_infeasibility()) <= 1e-9 * max(1.0, abs(o.primal_infeasibility()))


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("mfma", ["1", "0"])
def test_batched_line_search_is_bitwise_the_sequential_one(alg, mfma):
    """the candidates of a line search evaluated in batches (two passes + one read-back) against the trials one by one: the same
    accepted steps, values and buffers to the last bit -- with the value's primal terms on the matrix cores (k_value_mfma) and on
    the vector ALUs (k_value_terms / k_ls_value)"""
    p = synth.make_problem("medium")
    dh, ah = synth.forecast_at(p["forecast"], 0)
    runs = []
    for seq in (0, 1):
        s = capi.Solver(p["network"], p["tree"], p["config"], knobs={"value_mfma": int(mfma), "ls_sequential": seq})
        s.initialiseSmpcController(dh, ah)
        s.setAlgorithm(alg, 5)
        h, v, t = (s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama)(10)
        runs.append((h, v, t, [s.get(b) for b, _ in FBE_PAIRS], s.fbeCounters()))
        s.close()
    (h1, v1, t1, b1, c1), (h0, v0, t0, b0, c0) = runs
    assert c1["sequential"] == 0 and c1["batches"] >= c1["searches"] >= 1 and c0["sequential"] == c0["searches"] >= 1, (c1, c0)
    assert np.array_equal(t1, t0) and np.array_equal(v1, v0) and np.array_equal(h1, h0)
    for x, y in zip(b1, b0):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("kind", ["positive", "tiny"])
def test_line_search_direction_rule(alg, kind):
    """the two early exits of the line searches (SmpcController.cu:1262-1270, :1370-1379): a direction of positive slope leaves tau = 1
    and applies nothing, a slope below 1e-4 in magnitude gives tau = 0.  The HIP path reads the slope together with the values of the
    first candidate batch -- the state afterwards must be the oracle's all the same."""
    p, o, s = make_pair("small", alg)
    fbe = alg == "globalFbeAlgorithm"
    for it in range(2):
        o.solve_step(); s.solveStep()
        o.prox(); s.proximalFunG()
        o.residual(); s.computeFixedPointResidual()
        if fbe:
            o.gradient_fbe(); s.computeGradientFbe()
        else:
            o.nama_residual(); s.updateFixedPointResidualNamaAlgorithm()
        if it > 0:
            vo, vs = o.value_fbe(), s.computeValueFbe()
            o.lbfgs_direction(); s.computeLbfgsDirection()
            # FBE: slope = <grad, dir>; NAMA: slope = -<res, dir>
            yx, yp = (o.get("gradXi"), o.get("gradPsi")) if fbe else (-o.get("resXi"), -o.get("resPsi"))
            scale = 1.0 if kind == "positive" else -1e-7 / max(float(np.dot(yx, yx) + np.dot(yp, yp)), 1e-300)
            for name, bid, v in (("dirXi", capi.BUF_LBFGS_DIR_XI, scale * yx), ("dirPsi", capi.BUF_LBFGS_DIR_PSI, scale * yp)):
                o.set(name, v); s.set(bid, v)
            to = o.line_search_fbe(vo) if fbe else o.line_search_ame(vo)
            ts = s.computeLineSearchLbfgsUpdate(vs) if fbe else s.computeLineSearchAmeLbfgsUpdate(vs)
            assert to == (1.0 if kind == "positive" else 0.0) and ts == to
            compare_fbe(s, o, alg, REL_TOL, "line search with a %s slope" % kind)
            assert s.fbeCounters()["searches"] == 0
        o.dual_update(); s.dualUpdate()


@pytest.mark.parametrize("alg", ALGS)
def test_soft_constraint_branch_value(alg):
    """penalties small enough that the soft branch trips: the g terms of the FBE value are exercised"""
    p, o, s = make_pair("small", alg, penalty_x=2.0, penalty_xs=1.0)
    iters = 12
    ho, vo, to = o.fbe_nama(iters)
    hs, vs, ts = (s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama)(iters)
    dx, ds = o.dist()
    assert dx > 2.0 / float(p["config"]["stepSize"][0]) or ds > 1.0 / float(p["config"]["stepSize"][0]), "branch not exercised"
    assert np.array_equal(ts, to)
    assert relmax(vs, vo) < REL_TOL
    compare_fbe(s, o, alg, 1e-8, "soft branch")
    assert s.fbeCounters()["sequential"] >= 1      # a candidate's prox tripped the branch: those searches ran trial by trial


def test_fbe_api_errors():
    p = synth.make_problem("tiny")
    s = capi.Solver(p["network"], p["tree"], p["config"])
    with pytest.raises(capi.RapidNetError):
        s.computeHessianOracalGlobalFbe()          # no algorithm selected
    with pytest.raises(capi.RapidNetError):
        s.setAlgorithm("namaAlgorithm", 0)         # bad buffer size
    s.setAlgorithm("namaAlgorithm", 3)
    with pytest.raises(capi.RapidNetError):
        s.computeValueFbe()                        # before the factor step / affine terms
    with pytest.raises(capi.RapidNetError):
        s.setAlgorithm("globalFbeAlgorithm", 4)    # buffer size cannot change
    s.setAlgorithm("proximalAlgorithm")


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("name,world,cut,structured,kw", [("medium", 2, 0, False, {}), ("medium", 3, 1, False, {}), ("small", 2, 0, True, {}),
                                                          ("medium", 2, 0, False, {"penalty_x": 2.0, "penalty_xs": 1.0})])
def test_sharded_loop_matches_oracle(name, world, cut, structured, kw, alg):
    """The quasi-Newton loops on a SHARDED tree (missing item 3): `world` rank-local contexts (rn_create_sharded, one
    thread each, the in-process stand-in for the communicator), every dot product / value term / prox distance all-reduced
    over the ranks with the replicated crown counted once.  All ranks must take the same skip and line-search decisions (the
    same tau sequence as the oracle of the whole tree) and the reassembled iterates must match it."""
    import threading

    from rapidnet_amd import partition

    p = synth.make_problem(name, **kw)
    dh, ah = synth.forecast_at(p["forecast"], 0)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.set_algorithm(alg, 5)
    o.initialise(dh, ah)
    o.fbe_reset()
    iters = 10
    ho, vo, to = o.fbe_nama(iters)
    group = capi.local_group_create(world)
    shards = []
    for r in range(world):
        sh = capi.Solver(p["network"], p["tree"], p["config"], rank=r, nranks=world, cut_stage=cut, structured=structured)
        sh.joinLocalGroup(group, r)
        shards.append(sh)
    out, errs = [None] * world, []

    def work(i):
        try:
            sh = shards[i]
            sh.initialiseSmpcController(dh, ah)
            sh.setAlgorithm(alg, 5)
            out[i] = (sh.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else sh.algorithmNama)(iters)
        except Exception as e:   # noqa: BLE001
            errs.append((i, e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for hs, vs, tau in out:
        assert np.array_equal(tau, to), (tau, to)                   # every rank: the oracle's step lengths
        assert relmax(vs, vo) < REL_TOL
        assert relmax(hs, ho) < 1e-7                                # tree-global primal infeasibility on every rank
    if kw:
        dx, ds = o.dist()
        assert dx > kw["penalty_x"] / float(p["config"]["stepSize"][0]) or ds > kw["penalty_xs"] / float(p["config"]["stepSize"][0]), "branch not exercised"
    nodes = shards[0].full_nodes
    dims = {"x": o.nx, "u": o.nu, "xi": 2 * o.nx, "psi": o.nu, "accXi": 2 * o.nx, "dualXi": 2 * o.nx, "dirXi": 2 * o.nx, "dirPsi": o.nu, "resPsi": o.nu}
    for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_XI, "xi"), (capi.BUF_PSI, "psi"), (capi.BUF_ACC_XI, "accXi"), (capi.BUF_DUAL_XI, "dualXi"),
                    (capi.BUF_LBFGS_DIR_XI, "dirXi"), (capi.BUF_LBFGS_DIR_PSI, "dirPsi"), (capi.BUF_RES_PSI, "resPsi")):
        full = partition.scatter_to_global([sh.get(bid) for sh in shards], [sh.global_nodes for sh in shards], nodes, dims[nm])
        assert relmax(full, o.get(nm)) < 1e-8, nm
    assert all(sh.lbfgsState()[:2] == o.lbfgs_state()[:2] for sh in shards)
    for sh in shards:
        sh.close()
    capi.local_group_destroy(group)
