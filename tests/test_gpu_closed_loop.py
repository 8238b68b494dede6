"""Closed loop around the solve on the GPU against the CPU oracle: controlAction(fstream&) ->
moveForewardInTime (in-built simulator) -> next step, three steps, through the C++ host classes over the C-ABI
(tests/cpp/test_host.cpp closedloop_dump) -- current state, previous control, previous demand, the root control before the
projection, the x iterate of node 0 (which the reference's simulator modifies, SmpcController.cu:1695) and all four KPIs at
1e-9; both plant modes.  Plus the dual warm start across control steps against the oracle started from the same duals."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REF_FIXTURE
from oracle.oracle import Oracle, forecast_at, load_json
from rapidnet_amd import build, capi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-9


def relmax(a, b):
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    assert a.shape == b.shape and np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def _dump(directory, steps, disturbance):
    if not os.path.exists(build.TEST_HOST):
        build.build_host()
    r = subprocess.run([build.TEST_HOST, "closedloop_dump", directory, str(steps), "1" if disturbance else "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return [json.loads(l[len("CLSTEP "):]) for l in r.stdout.splitlines() if l.startswith("CLSTEP ")]


def _oracle_loop(directory, steps, disturbance):
    net, tree, cfg, fc = (load_json(os.path.join(directory, f)) for f in ("network.json", "scenarioTree.json", "controllerConfig.json", "forecastor.json"))
    o = Oracle(net, tree, cfg)
    o.factor_step()
    o.update_state_control()
    out = []
    for t in range(steps):
        dh, ah = forecast_at(fc, t)
        o.control_action(dh, ah, project=True)                 # maxIterations of the configuration, like the reference
        u_raw = o.get("u")[: o.nu].copy()
        x, u, d = o.move_forward(dh, ah, weight_economical=1.0, plant_mode=1 if disturbance else 0)
        out.append({"u_root_unprojected": u_raw, "x": x, "prevU": u, "prevD": d, "x_node0": o.get("x")[: o.nx].copy(), "kpi": np.array(o.kpis(t + 1))})
    return out


def _compare(directory, steps, disturbance):
    got, want = _dump(directory, steps, disturbance), _oracle_loop(directory, steps, disturbance)
    assert len(got) == steps
    for t, (g, w) in enumerate(zip(got, want)):
        for key in ("u_root_unprojected", "x", "prevU", "prevD", "x_node0"):
            assert relmax(g[key], w[key]) < TOL, (t, key)
        for j, name in enumerate(("economic", "smooth", "network", "safety")):
            assert abs(g["kpi"][j] - w["kpi"][j]) <= TOL * max(abs(w["kpi"][j]), 1e-300), (t, name, g["kpi"][j], w["kpi"][j])


@pytest.mark.parametrize("disturbance", [False, True])
def test_closed_loop_on_the_reference_fixture(tmp_path, disturbance):
    """3-tank fixture, the reference's own controller settings (stepSize 1e-4, 500 iterations per control step); the
    forecaster file holds the horizons of time instants 0 and 1, so two steps."""
    d = str(tmp_path / "fixture")
    shutil.copytree(REF_FIXTURE, d)
    cfg = load_json(os.path.join(d, "controllerConfig.json"))
    for k, f in (("pathToNetwork", "network.json"), ("pathToScenarioTree", "scenarioTree.json"), ("pathToForecaster", "forecastor.json")):
        cfg[k] = os.path.join(d, f)
    json.dump(cfg, open(os.path.join(d, "controllerConfig.json"), "w"))
    _compare(d, 2, disturbance)


@pytest.mark.parametrize("disturbance", [False, True])
def test_closed_loop_on_a_synthetic_problem(tmp_path, disturbance):
    p = synth.make_problem("small", max_iterations=60, sim_horizon=3)
    synth.write_problem(p, str(tmp_path))
    _compare(str(tmp_path), 3, disturbance)


def test_warm_start_against_the_oracle_started_from_the_same_duals():
    """rn_set_warm_start: the next control step keeps y+ of the previous one (y := y+, momentum restarted, theta = {1,1});
    the reference cold-starts (SmpcController.cu:1509).  Oracle: same duals installed by hand, then the plain APG loop."""
    p = synth.make_problem("small")
    dh0, ah0 = synth.forecast_at(p["forecast"], 0)
    dh1, ah1 = synth.forecast_at(p["forecast"], 1)
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.factorStep()
    s.setWarmStart(True)
    u0 = s.controlAction(dh0, ah0, maxIterations=30)
    x1 = np.asarray(p["config"]["currentX"], float) * 0.97
    u1 = s.controlAction(dh1, ah1, currentX=x1, prevU=u0, maxIterations=25)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.factor_step()
    o.update_state_control()
    o.eliminate(dh0, ah0)
    o.apg(30)
    assert relmax(u0, o.get("u")[: o.nu]) < TOL
    o.set("xi", o.get("updXi")); o.set("psi", o.get("updPsi"))      # y := y+ ; y+ kept
    o.update_state_control(x1, u0)
    o.eliminate(dh1, ah1)
    o.apg_continue(25, [1.0, 1.0])
    assert relmax(u1, o.get("u")[: o.nu]) < TOL
    for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"), (capi.BUF_XI, "xi")):
        assert relmax(s.get(bid), o.get(nm)) < TOL, nm
    # and a cold start differs (the warm start is doing something)
    c = capi.Solver(p["network"], p["tree"], p["config"])
    c.factorStep()
    uc = c.controlAction(dh1, ah1, currentX=x1, prevU=u0, maxIterations=25)
    assert relmax(uc, u1) > 1e-6
