#!/usr/bin/env python3
"""Builds tests/golden/reference_barcelona30/: a Barcelona-dimension problem (nx = 63, nu = 114, nd = 88, ne = 17, nv = 97,
N = 24) on REFERENCE-HELD data, for long-run parity at the reference's own settings (stepSize 1e-4, maxIterations 500).

From /root/reference/src/paser/dataSource/ (data files the reference holds; read here, committed as fixture data):
  controllerConfig32.json   matL, matLhat, costW, matDiagPrecnd, costAlpha1, costAlpha2, currentX, prevDemand, stepSize,
                            maxIterations, penaltyStateX, penaltySafetyX
  scenarioTree65.json       the 6 x 5 scenario tree: N = 24, K = 30, 667 nodes, real demand / price errors, probabilities
Synthesised, because the reference does not hold them (/root/reference/.MISSING_LARGE_BLOBS: systemData/network.json and
forecaster/nominalForecast.json are missing):
  network.json     E = an orthonormal basis of null(L')' and Ed = -E Lhat (the only E, Ed consistent with the held L, Lhat up to a
                   change of basis, which the solve does not see); B, Gd sparse incidence matrices (every actuator fills one
                   tank, a third also drain one; 30 of the 88 demands are drawn from tanks) with entries +-dt, dt chosen so that
                   the held stepSize is 0.9 / (Lipschitz constant of the dual gradient); xmin = 0, xsafe = 0.6 currentX,
                   xmax = 2 currentX; umin / umax by rapidnet_amd.synth.make_feasible (a strictly feasible policy exists)
  forecastor.json  nominal demand = the held prevDemand with a daily profile; nominal prices = the held costAlpha2 (24 x 114)
  prevU            the steady-state control of prevDemand (the held prevU violates the mass balance E u + Ed d = 0 by 4e6)
The oracle's iterates after k = 1, 10, 100, 500 APG iterations are stored with the inputs (iterates.npz: every 11th entry of
x, u, y+ (xi and psi parts) and z, and the whole primal-infeasibility history).

    python tests/golden/make_reference_instance.py      # needs /root/reference; deterministic
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from rapidnet_amd import synth  # noqa: E402

SRC = "/root/reference/src/paser/dataSource"
OUT = os.path.join(ROOT, "tests", "golden", "reference_barcelona30")
CHECKPOINTS = (1, 10, 100, 500)
STRIDE = 11


def build():
    c = json.load(open(os.path.join(SRC, "controllerConfig32.json")))
    tree = json.load(open(os.path.join(SRC, "scenarioTree65.json")))
    nx, nu, ne, nv, nd, N = (int(c[k][0]) for k in ("nx", "nu", "ne", "nv", "nd", "N"))
    assert (nx, nu, ne, nv, nd, N) == (63, 114, 17, 97, 88, 24) and int(tree["N"][0]) == N and int(tree["dimDemand"][0]) == nd
    rng = np.random.default_rng(20260165)
    L = np.array(c["matL"], float).reshape(nu, nv, order="F")
    Lhat = np.array(c["matLhat"], float).reshape(nu, nd, order="F")
    # E: orthonormal rows spanning the complement of span(L); Ed = -E Lhat
    E = np.linalg.svd(L.T, full_matrices=True)[2][nv:]
    Ed = -E @ Lhat
    B = np.zeros((nx, nu))
    for j in range(nu):
        B[rng.integers(nx), j] = 1.0
        if rng.random() < 0.35:
            t = rng.integers(nx)
            if B[t, j] == 0:
                B[t, j] = -1.0
    for i in range(nx):
        if not B[i].any():
            B[i, rng.integers(nu)] = 1.0
    Gd = np.zeros((nx, nd))
    for j in rng.permutation(nd)[:30]:
        Gd[rng.integers(nx), j] = -1.0
    x0 = np.array(c["currentX"], float)
    col = lambda M: np.asarray(M).ravel(order="F").tolist()

    def network_for(dt):
        return {"nx": [nx], "nu": [nu], "ne": [ne], "nd": [nd], "matA": col(np.eye(nx)), "matB": col(dt * B), "matGd": col(dt * Gd),
                "matE": col(E), "matEd": col(Ed), "vecXmin": [0.0] * nx, "vecXmax": (2.0 * x0).tolist(), "vecXsafe": (0.6 * x0).tolist(),
                "vecUmin": [0.0] * nu, "vecUmax": [1.0] * nu, "costAlpha1": list(c["costAlpha1"])}

    config = {k: c[k] for k in ("nx", "nu", "ne", "nv", "nd", "N", "matL", "matLhat", "costW", "matDiagPrecnd", "currentX", "prevDemand", "stepSize",
                                "maxIterations", "penaltyStateX", "penaltySafetyX")}
    config.update({"prevU": [0.0] * nu, "pathToNetwork": "network.json", "pathToScenarioTree": "scenarioTree.json", "pathToForecaster": "forecastor.json",
                   "algorithmName": "proximalAlgorithm", "lbfgsBufferSize": [5]})
    step = float(c["stepSize"][0])
    # dt by bisection (log scale) on step * Lipschitz(dt) = 0.9; the estimate is a 60-step power iteration from below, hence the margin
    lo, hi = 1e-3, 1e3
    for _ in range(40):
        dt = float(np.sqrt(lo * hi))
        lip = synth.lipschitz_estimate(network_for(dt), tree, config, iters=60)
        lo, hi = (dt, hi) if step * lip < 0.9 else (lo, dt)
    dt = float(np.round(lo, 4))
    network = network_for(dt)
    lip = synth.lipschitz_estimate(network, tree, config, iters=200)
    assert step * lip < 0.95, (dt, step * lip)
    pd = np.array(c["prevDemand"], float)
    alpha2 = np.array(c["costAlpha2"], float).reshape(N, nu)
    sim = 2
    hours = np.arange(N + sim)
    phase = rng.uniform(0, 24, nd)
    dhat_all = pd[None, :] * (1 + 0.3 * np.sin(2 * np.pi * (hours[:, None] + phase[None, :]) / 24))
    ahat_all = alpha2[hours % N]
    forecast = {"N": [N], "simHorizon": [sim], "dimDemand": [nd], "dimPrices": [nu]}
    for t in range(sim):
        forecast["timeIdDemand%d" % t] = dhat_all[t:t + N].ravel().tolist()
        forecast["timeIdPrice%d" % t] = ahat_all[t:t + N].ravel().tolist()
    problem = synth.make_feasible({"network": network, "tree": tree, "config": config, "forecast": forecast})
    return problem, {"dt": dt, "step_times_lipschitz": step * lip}


def main():
    problem, info = build()
    synth.write_problem(problem, OUT)
    # relative paths in the committed configuration (the tests rewrite them when they hand the files to the C++ loaders)
    cfg = json.load(open(os.path.join(OUT, "controllerConfig.json")))
    cfg.update({"pathToNetwork": "network.json", "pathToScenarioTree": "scenarioTree.json", "pathToForecaster": "forecastor.json"})
    json.dump(cfg, open(os.path.join(OUT, "controllerConfig.json"), "w"))
    dh, ah = synth.forecast_at(problem["forecast"], 0)
    out = {"dt": np.array([info["dt"]]), "step_times_lipschitz": np.array([info["step_times_lipschitz"]])}
    o = Oracle(problem["network"], problem["tree"], problem["config"])
    o.initialise(dh, ah)
    o.apg_reset()
    th, done, hist = [1.0, 1.0], 0, []
    for k in CHECKPOINTS:
        for _ in range(k - done):
            th = o.apg_continue(1, th)
            hist.append(o.primal_infeasibility())
        done = k
        for nm in ("x", "u", "updXi", "updPsi", "dualXi"):
            out["%s_%d" % (nm, k)] = o.get(nm)[::STRIDE]
    out["hist"] = np.array(hist)
    out["stride"] = np.array([STRIDE])
    np.savez_compressed(os.path.join(OUT, "iterates.npz"), **out)
    print("dt = %g, stepSize * Lipschitz = %.3f, primal infeasibility after 1 / 10 / 100 / 500 iterations: %s" % (
        info["dt"], info["step_times_lipschitz"], [float("%.4g" % hist[k - 1]) for k in CHECKPOINTS]))
    print({k: os.path.getsize(os.path.join(OUT, k)) for k in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    main()
