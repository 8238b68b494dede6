"""Pins the CPU oracle (oracle/apg_oracle.c) to the reference's own golden vectors.

Mirrors Testing::testEngineTesting (src/test/Testing.cu:340-477) and the TestSmpcController known-answer
tests (src/test/TestSmpcController.cu:114-420): same inputs (forecast time index 1), same keys of
engineTest.json / smpcTest.json.  The fixtures were printed with 7 significant digits by MATLAB, so the
achievable agreement is ~1e-6 relative; the reference's own tolerances are 1e-2 abs (engine) and
1e-1 abs / 0.1 % rel (controller) -- ours are far tighter.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, forecast_at


def rel_err(a, b):
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    assert a.size == b.size
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1.0)).max())


@pytest.fixture(scope="module")
def oracle(ref_fixture):
    o = Oracle(ref_fixture["network"], ref_fixture["tree"], ref_fixture["config"])
    dh, ah = forecast_at(ref_fixture["forecast"], 1)  # timeInst = 1, Testing.cu:365-367
    o.initialise(dh, ah)
    return o


def test_dims(oracle):
    assert (oracle.nx, oracle.nu, oracle.nd, oracle.nv) == (3, 6, 4, 4)
    assert (oracle.N, oracle.K, oracle.nodes) == (24, 6, 136)
    assert oracle.final_branch_node == 10  # ScenarioTree::getFinalBranchNode


def test_affine_terms(oracle, ref_fixture):
    eng = ref_fixture["engine"]
    assert rel_err(oracle.get("uhat"), eng["uHat"]) < 2e-6
    assert rel_err(oracle.get("e"), eng["vecE"]) < 2e-6
    assert rel_err(oracle.get("beta"), eng["beta"]) < 2e-6
    assert rel_err(oracle.get("alpha"), eng["costAlpha"]) < 2e-6
    assert rel_err(oracle.get("L"), eng["matL"]) < 1e-7


def test_factor_step_operators(oracle, ref_fixture):
    eng = ref_fixture["engine"]
    nx, nu, nv = oracle.nx, oracle.nu, oracle.nv
    sn = np.array(eng["scenarioNodes"], int) - 1  # 1-based node ids along one scenario

    def along(name, dim, cnt=None):
        b = oracle.get(name).reshape(-1, dim)
        return b[sn if cnt is None else sn[:cnt]].ravel()

    assert rel_err(along("sysF", 2 * nx * nx), eng["sysF"]) < 1e-6
    assert rel_err(along("sysG", nu * nu), eng["sysG"]) < 1e-6
    for key, name, dim in (("xmin", "xmin", nx), ("xmax", "xmax", nx), ("xs", "xs", nx), ("umin", "umin", nu),
                           ("umax", "umax", nu)):
        assert rel_err(along(name, dim), eng[key]) < 2e-6, key
    fbs = 2  # getFinalBranchStage(): omega/g/Theta are checked for the first 2 path nodes only (Testing.cu:432-443)
    assert rel_err(along("Omega", nv * nv, fbs), eng["omega"][: fbs * nv * nv]) < 2e-6
    assert rel_err(along("Theta", nv * nx, fbs), eng["Theta"][: fbs * nv * nx]) < 2e-6
    assert rel_err(np.tile(oracle.get("Gtil"), fbs), eng["g"][: fbs * nv * nx]) < 1e-7
    assert rel_err(along("D", 2 * nx * nv), eng["d"]) < 1e-6
    assert rel_err(along("Ftil", nu * nv), eng["f"]) < 1e-6
    assert rel_err(along("Phi", 2 * nx * nv), eng["Phi"]) < 1e-6
    assert rel_err(along("Psi", nu * nv), eng["Psi"]) < 1e-6


def test_extrapolation(oracle, ref_fixture):
    s = ref_fixture["smpc"]
    oracle.set("xi", s["xi"]); oracle.set("psi", s["psi"])
    oracle.set("updXi", s["updateXi"]); oracle.set("updPsi", s["updatePsi"])
    th = s["theta"]
    oracle.extrapolate(th[1] * (1 / th[0] - 1))  # TestSmpcController.cu:146
    assert rel_err(oracle.get("accXi"), s["acceleXi"]) < 2e-6
    assert rel_err(oracle.get("accPsi"), s["accelePsi"]) < 2e-6
    assert rel_err(oracle.get("xi"), s["finalXi"]) < 1e-7
    assert rel_err(oracle.get("psi"), s["finalPsi"]) < 1e-7


def test_solve_step_prox_residual_update(oracle, ref_fixture):
    s = ref_fixture["smpc"]
    oracle.set("accXi", s["acceleXi"]); oracle.set("accPsi", s["accelePsi"])
    oracle.solve_step()
    assert rel_err(oracle.get("x"), s["X"]) < 2e-6
    assert rel_err(oracle.get("u"), s["U"]) < 2e-6
    assert rel_err(oracle.get("v"), s["tempV"]) < 1e-5  # unused by the reference's test; L's basis matches here
    assert rel_err(oracle.get("primalXi"), s["primalX"]) < 2e-6
    assert rel_err(oracle.get("primalPsi"), s["primalU"]) < 2e-6
    oracle.prox()
    # t = Hx + w/lambda with lambda = 1e-4 amplifies the 7-digit print error of w
    assert rel_err(oracle.get("dualXi"), s["dualX"]) < 1e-4
    assert rel_err(oracle.get("dualPsi"), s["dualU"]) < 2e-4
    dx, ds = oracle.dist()
    assert dx < 1e6 / 1e-4 and ds < 1e4 / 1e-4  # soft-constraint branch not taken on the fixture
    oracle.residual()
    assert rel_err(oracle.get("resXi"), s["fixedPointResidualXi"]) < 1e-5
    assert rel_err(oracle.get("resPsi"), s["fixedPointResidualPsi"]) < 1e-5
    assert rel_err(oracle.get("resXi"), s["primalInfsXi"]) < 1e-5
    oracle.dual_update()
    assert rel_err(oracle.get("updXi"), s["finalUpdateXi"]) < 5e-6
    assert rel_err(oracle.get("updPsi"), s["finalUpdatePsi"]) < 5e-6


def test_steps_from_fixture_inputs(ref_fixture):
    """Each step fed with the fixture's own inputs (as the reference's tests do), not with our previous output."""
    s = ref_fixture["smpc"]
    o = Oracle(ref_fixture["network"], ref_fixture["tree"], ref_fixture["config"])
    dh, ah = forecast_at(ref_fixture["forecast"], 1)
    o.initialise(dh, ah)
    o.set("primalXi", s["primalX"]); o.set("primalPsi", s["primalU"])
    o.set("dualXi", s["dualX"]); o.set("dualPsi", s["dualU"])
    o.residual()  # a difference of two 7-digit prints: cancellation limits agreement to ~1e-5
    assert rel_err(o.get("resXi"), s["fixedPointResidualXi"]) < 1e-5
    assert rel_err(o.get("resPsi"), s["fixedPointResidualPsi"]) < 1e-5
    o.set("accXi", s["acceleXi"]); o.set("accPsi", s["accelePsi"])
    o.set("resXi", s["fixedPointResidualXi"]); o.set("resPsi", s["fixedPointResidualPsi"])
    o.dual_update()
    assert rel_err(o.get("updXi"), s["finalUpdateXi"]) < 2e-6
    assert rel_err(o.get("updPsi"), s["finalUpdatePsi"]) < 2e-6


def test_f32_oracle_agrees_with_f64(ref_fixture):
    """The reference ran in fp32 (Configuration.h:31); the fp32 build of the oracle stays within fp32 noise."""
    dh, ah = forecast_at(ref_fixture["forecast"], 1)
    res = []
    for prec in ("f64", "f32"):
        o = Oracle(ref_fixture["network"], ref_fixture["tree"], ref_fixture["config"], precision=prec)
        o.initialise(dh, ah)
        o.apg(3)
        res.append((o.get("x"), o.get("u")))
    assert rel_err(res[1][0], res[0][0]) < 1e-3
    assert rel_err(res[1][1], res[0][1]) < 1e-3
