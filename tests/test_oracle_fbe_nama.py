"""Pins the oracle's global-FBE / NAMA restatement (oracle/apg_oracle.c, second half) to the reference's golden vectors.

Python twin of Testing::testSmpcFbeController / testSmpcNamaController (src/test/Testing.cu:536-590) and the
TestSmpcController methods they call (src/test/TestSmpcController.cu:403-1040): same keys of smpcFbeTest.json /
smpcNamaTest.json, same inputs loaded into the same buffers, in the same order.  controllerFbeConfig.json and
controllerNamaConfig.json differ from controllerConfig.json only in "algorithmName", so the committed
controllerConfig.json is used with the algorithm set explicitly.  The reference's tolerances are 1e-1 abs / 0.1 % rel;
the vectors carry 7 significant digits.
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, forecast_at


def rel_err(a, b, floor=1.0):
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    assert a.size == b.size
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def scaled_err(a, b):
    """max |a - b| over the vector's own scale: entries that are small through cancellation of 7-digit inputs of
    magnitude ~1e4 cannot be resolved to 1e-6 of themselves."""
    a = np.asarray(a, float).ravel()
    b = np.asarray(b, float).ravel()
    assert a.size == b.size
    return float(np.abs(a - b).max() / np.abs(b).max())


def make(ref_fixture, algorithm):
    o = Oracle(ref_fixture["network"], ref_fixture["tree"], ref_fixture["config"])
    o.set_algorithm(algorithm, 5)
    dh, ah = forecast_at(ref_fixture["forecast"], 1)  # timeInst = 1, Testing.cu:540-542
    o.initialise(dh, ah)
    o.fbe_reset()
    return o


@pytest.fixture(scope="module", params=["globalFbeAlgorithm", "namaAlgorithm"])
def case(request, ref_fixture):
    key = "smpc_fbe" if request.param == "globalFbeAlgorithm" else "smpc_nama"
    return request.param, make(ref_fixture, request.param), ref_fixture[key]


def test_hessian_oracle(case):
    """testHessianOracalGlobalFbe (:403-452): direction = the fixed-point residual -> Xdir, Udir."""
    name, o, s = case
    fbe = name == "globalFbeAlgorithm"
    o.set("gradXi" if fbe else "resXi", s["fixedPointResidualXi"])
    o.set("gradPsi" if fbe else "resPsi", s["fixedPointResidualPsi"])
    o.hessian_oracle()
    kx, ku = ("fbeHessianDirXdir", "fbeHessianDirUdir") if fbe else ("ameFixedPointDirXdir", "ameFixedPointDirUdir")
    assert scaled_err(o.get("udir"), s[ku]) < 1e-6
    assert scaled_err(o.get("xdir"), s[kx]) < 1e-6


def test_fbe_gradient(case):
    """testFbeGradient (:458-497)"""
    name, o, s = case
    if name != "globalFbeAlgorithm":
        pytest.skip("FBE only (Testing.cu:548)")
    o.set("resXi", s["fixedPointResidualXi"]); o.set("resPsi", s["fixedPointResidualPsi"])
    o.gradient_fbe()
    assert rel_err(o.get("gradXi"), s["fbeGradXi"]) < 2e-6
    assert rel_err(o.get("gradPsi"), s["fbeGradPsi"]) < 2e-6


def test_value_fbe(case):
    """testValueFbe (:683-745)"""
    name, o, s = case
    o.set("resXi", s["fixedPointResidualXi"]); o.set("resPsi", s["fixedPointResidualPsi"])
    o.set("accXi", s["acceleXi"]); o.set("accPsi", s["accelePsi"])
    o.set("u", s["U"])
    assert abs(o.value_fbe() / s["fbeObjDual"][0] - 1) < 2e-6


def test_nama_residual(case):
    """testUpdateFixedPointResidualNamaAlgorithm (:633-677)"""
    name, o, s = case
    if name != "namaAlgorithm":
        pytest.skip("NAMA only (Testing.cu:580)")
    o.set("resXi", s["fixedPointResidualXi"]); o.set("resPsi", s["fixedPointResidualPsi"])
    o.nama_residual()
    assert rel_err(o.get("curResXi"), s["lbfgsCurrentYvecXi"]) < 1e-6
    assert rel_err(o.get("curResPsi"), s["lbfgsCurrentYvecPsi"]) < 1e-6


def load_lbfgs(o, s, fbe):
    n = o.nodes * (2 * o.nx + o.nu)
    cur, prv = ("grad", "prevGrad") if fbe else ("curRes", "prevRes")
    o.set("prevXi", s["xi"]); o.set("prevPsi", s["psi"])
    o.set("xi", s["acceleXi"]); o.set("psi", s["accelePsi"])
    o.set(cur + "Xi", s["lbfgsCurrentYvecXi"]); o.set(cur + "Psi", s["lbfgsCurrentYvecPsi"])
    o.set(prv + "Xi", s["lbfgsPreviousYvecXi"]); o.set(prv + "Psi", s["lbfgsPreviousYvecPsi"])
    o.buf("matS")[: 5 * n] = s["matS"]
    o.buf("matY")[: 5 * n] = s["matY"]
    inv = np.array(s["vecInvRho"], float)
    o.buf("rho")[:5] = np.where(inv != 0, 1 / np.where(inv != 0, inv, 1), 0)
    o.lbfgs_state(int(s["colLbfgs"][0]), int(s["memLbfgs"][0]), float(s["H"][0]))


def test_lbfgs_direction(case):
    """testLbfgsDirection (:503-627)"""
    name, o, s = case
    fbe = name == "globalFbeAlgorithm"
    n = o.nodes * (2 * o.nx + o.nu)
    load_lbfgs(o, s, fbe)
    o.lbfgs_direction()
    col, mem, H = o.lbfgs_state()
    assert col == int(s["updateColLbfgs"][0]) and mem == int(s["updateMemLbfgs"][0])
    assert abs(H / s["updateH"][0] - 1) < 2e-6
    inv = np.array(s["updateVecInvRho"], float)
    assert rel_err(1 / o.get("rho")[:5], inv, floor=1e-300) < 2e-6
    # the new column is a difference of two 7-digit vectors of magnitude ~1e3
    assert rel_err(o.get("matS")[: 5 * n], s["updateMatS"]) < 1e-4
    assert rel_err(o.get("matY")[: 5 * n], s["updateMatY"]) < 1e-4
    # the two-loop recursion mixes five 7-digit columns of norm ~1e3 with weights ~1e0: 1e-4 of the direction's scale
    scale = np.abs(np.array(s["lbfgsDirXi"])).max()
    assert np.abs(o.get("dirXi") - s["lbfgsDirXi"]).max() < 1e-4 * scale
    assert np.abs(o.get("dirPsi") - s["lbfgsDirPsi"]).max() < 1e-4 * scale


def test_line_search(case):
    """testFbeLineSearch (:841-935) / testAmeLineSearch (:748-838)"""
    name, o, s = case
    fbe = name == "globalFbeAlgorithm"
    o.set("resXi", s["fixedPointResidualXi"]); o.set("resPsi", s["fixedPointResidualPsi"])
    o.set("accXi", s["acceleXi"]); o.set("accPsi", s["accelePsi"])
    o.set("x", s["X"]); o.set("u", s["U"])
    o.set("dirXi", s["lbfgsDirXi"]); o.set("dirPsi", s["lbfgsDirPsi"])
    if fbe:
        o.set("gradXi", s["fbeGradXi"]); o.set("gradPsi", s["fbeGradPsi"])
    o.set("primalXi", s["primalX"]); o.set("primalPsi", s["primalU"])
    val = o.value_fbe()
    tau = o.line_search_fbe(val) if fbe else o.line_search_ame(val)
    assert abs(val / s["fbeObjDual"][0] - 1) < 2e-6
    assert abs(tau - s["tau"][0]) < 1e-12
    assert rel_err(o.get("accXi"), s["updateXi"]) < 1e-5
    assert rel_err(o.get("accPsi"), s["updatePsi"]) < 1e-5
    # residual = lambda^-1-amplified difference of 7-digit inputs, as in test_oracle_reference_fixtures
    assert rel_err(o.get("resXi"), s["updateResidualXi"]) < 2e-4
    assert rel_err(o.get("resPsi"), s["updateResidualPsi"]) < 2e-4


def test_fbe_dual_update(case):
    """testFbeDualUpdate (:938-1040)"""
    name, o, s = case
    fbe = name == "globalFbeAlgorithm"
    cur, prv = ("grad", "prevGrad") if fbe else ("curRes", "prevRes")
    o.set("xi", s["acceleXi"]); o.set("psi", s["accelePsi"])
    o.set("accXi", s["updateXi"]); o.set("accPsi", s["updatePsi"])
    o.set("resXi", s["updateResidualXi"]); o.set("resPsi", s["updateResidualPsi"])
    o.set(cur + "Xi", s["lbfgsCurrentYvecXi"]); o.set(cur + "Psi", s["lbfgsCurrentYvecPsi"])
    o.dual_update()
    assert rel_err(o.get("xi"), s["finalUpdateXi"]) < 2e-6
    assert rel_err(o.get("psi"), s["finalUpdatePsi"]) < 2e-6
    assert np.array_equal(o.get(prv + "Xi"), np.array(s["lbfgsCurrentYvecXi"], float))
    assert np.array_equal(o.get("prevXi"), np.array(s["acceleXi"], float))
    assert np.array_equal(o.get("prevPsi"), np.array(s["accelePsi"], float))
    assert np.array_equal(o.get("accXi"), o.get("xi")) and np.array_equal(o.get("accPsi"), o.get("psi"))
