"""Long-run parity on REFERENCE-HELD Barcelona data: tests/golden/reference_barcelona30/, built by
tests/golden/make_reference_instance.py from /root/reference/src/paser/dataSource/controllerConfig32.json (matL, matLhat,
costW, matDiagPrecnd, prices, currentX, prevDemand, stepSize 1e-4, maxIterations 500) and scenarioTree65.json (K = 30,
667 nodes), with only the network and the nominal forecast synthesised (the reference does not hold them).

CPU: the oracle reproduces the committed iterates; the fixture loads through this repository's loaders with the dimensions
the reference's files state.  GPU: the HIP path against the committed iterates and the live oracle -- 1e-8 directly up to
100 iterations.  Beyond that the REAL preconditioner and tree do not tame the iteration: on this data two runs of the oracle
itself whose beta differs by one part in 1e13 are 4e-10 apart after 200 iterations and 1e-4 after 500, whatever the step
size (0.25 ... 1 x the held 1e-4), the bounds or the size of the tree errors -- the iterates are still far from converged
after 500 iterations (primal infeasibility 30-70 of 1700) and the active set keeps changing.  So at 200 ... 500 iterations the
bound is the oracle's own sensitivity (x 20; measured ratios are at most 3), as for the infeasible synthetic data; 1e-8 is asserted
directly up to 200 iterations; the table is printed."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.oracle import Oracle, forecast_at, load_json

DIR = os.path.join(GOLDEN, "reference_barcelona30")
CHECKPOINTS = (1, 10, 100, 500)
NAMES = ("x", "u", "updXi", "updPsi", "dualXi")


def problem():
    return {k: load_json(os.path.join(DIR, f)) for k, f in (("network", "network.json"), ("tree", "scenarioTree.json"),
                                                             ("config", "controllerConfig.json"), ("forecast", "forecastor.json"))}


def rel(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / np.abs(b).max())


def oracle_checkpoints(p, dh, ah, perturb, checkpoints, names=NAMES):
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    if perturb:
        o.set("beta", o.get("beta") * (1.0 + perturb))
    o.apg_reset()
    th, done, out, hist = [1.0, 1.0], 0, [], []
    for k in checkpoints:
        for _ in range(k - done):
            th = o.apg_continue(1, th)
            hist.append(o.primal_infeasibility())
        done = k
        out.append({n: o.get(n) for n in names})
    return out, np.array(hist)


def test_fixture_is_the_reference_held_data():
    p = problem()
    c, t, n = p["config"], p["tree"], p["network"]
    assert [int(c[k][0]) for k in ("nx", "nu", "ne", "nv", "nd", "N")] == [63, 114, 17, 97, 88, 24]
    assert c["stepSize"][0] == 1e-4 and c["maxIterations"][0] == 500 and c["penaltyStateX"][0] == 1e10 and c["penaltySafetyX"][0] == 1e7
    assert [int(t[k][0]) for k in ("N", "K", "nodes", "dimDemand", "dimPrice")] == [24, 30, 667, 88, 114] and t["nChildren"][:7] == [6, 5, 5, 5, 5, 5, 5]
    nu, nv, nd, ne = 114, 97, 88, 17
    L = np.array(c["matL"]).reshape(nu, nv, order="F")
    Lhat = np.array(c["matLhat"]).reshape(nu, nd, order="F")
    E = np.array(n["matE"]).reshape(ne, nu, order="F")
    Ed = np.array(n["matEd"]).reshape(ne, nd, order="F")
    assert np.abs(E @ L).max() < 1e-12 and np.abs(E @ Lhat + Ed).max() < 1e-9 * np.abs(Ed).max()   # L = null(E), E Lhat = -Ed
    if os.path.isdir("/root/reference"):   # the held files, value for value
        import json

        src = json.load(open("/root/reference/src/paser/dataSource/controllerConfig32.json"))
        for k in ("matL", "matLhat", "costW", "matDiagPrecnd", "currentX", "prevDemand", "stepSize", "costAlpha1"):
            ref = src[k]
            got = c[k] if k != "costAlpha1" else n[k]
            assert np.array_equal(np.asarray(ref, float), np.asarray(got, float)), k
        tsrc = json.load(open("/root/reference/src/paser/dataSource/scenarioTree65.json"))
        for k in tsrc:
            assert np.array_equal(np.asarray(tsrc[k], float), np.asarray(t[k], float)), k


def test_oracle_reproduces_the_committed_iterates():
    p = problem()
    g = np.load(os.path.join(DIR, "iterates.npz"))
    st = int(g["stride"][0])
    dh, ah = forecast_at(p["forecast"], 0)
    out, hist = oracle_checkpoints(p, dh, ah, 0.0, CHECKPOINTS)
    for k, o in zip(CHECKPOINTS, out):
        tol = 1e-10 if k <= 100 else 1e-6      # same sources, same flags: normally identical; 500 iterations amplify any difference (docstring)
        for nm in NAMES:
            assert rel(o[nm][::st], g["%s_%d" % (nm, k)]) < tol, (k, nm)
    assert rel(hist[:100], g["hist"][:100]) < 1e-10


@pytest.mark.gpu
def test_hip_path_on_reference_held_data(capsys):
    from rapidnet_amd import capi

    p = problem()
    g = np.load(os.path.join(DIR, "iterates.npz"))
    st = int(g["stride"][0])
    dh, ah = forecast_at(p["forecast"], 0)
    cks = (1, 10, 50, 100, 200, 300, 400, 500)
    from conftest import run_pair
    (base, ohist), (pert, _) = run_pair(lambda: oracle_checkpoints(p, dh, ah, 0.0, cks), lambda: oracle_checkpoints(p, dh, ah, 1e-13, cks))
    bids = {"x": capi.BUF_X, "u": capi.BUF_U, "updXi": capi.BUF_UPD_XI, "updPsi": capi.BUF_UPD_PSI, "dualXi": capi.BUF_DUAL_XI}
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    done, rows, hist = 0, [], []
    for i, k in enumerate(cks):
        hist.append(s.apgIterate(k - done)); done = k
        e_gpu = max(rel(s.get(bids[n]), base[i][n]) for n in NAMES)
        e_self = max(rel(pert[i][n], base[i][n]) for n in NAMES)
        rows.append((k, e_gpu, e_self))
        if k <= 200:
            assert e_gpu < 1e-8, (k, e_gpu)                      # north_star's tolerance, directly (measured: 6.5e-11 at 200)
        else:                                                    # measured ratios are <= 3: a factor of 20 leaves one digit, not two
            assert e_gpu < max(1e-8, 20 * e_self), (k, e_gpu, e_self)
        if k in CHECKPOINTS and k <= 100:                            # and against the committed vectors
            for n in NAMES:
                assert rel(s.get(bids[n])[::st], g["%s_%d" % (n, k)]) < 1e-8, (k, n)
    hist = np.concatenate(hist)
    assert np.abs(hist[:100] - ohist[:100]).max() <= 1e-8 * np.abs(ohist[:100]).max()
    with capsys.disabled():
        print("\\n[reference-held Barcelona data, K = 30, stepSize 1e-4] iterations: HIP-vs-oracle max rel. error | oracle-vs-perturbed-oracle (beta * (1 + 1e-13))")
        for r in rows:
            print("    %4d: %.2e | %.2e" % r)
