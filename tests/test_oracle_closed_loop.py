"""The oracle's closed-loop restatement (controlAction, moveForewardInTime, updateKpi, KPI getters:
/root/reference/src/SmpcController.cu:1607-1716, 1778-1859) against an independent numpy computation of the same
formulas on the reference's 3-tank fixture.  The reference holds no golden vectors for these functions (SURVEY.md
section 4: "no end-to-end test"), so this is what pins the checker before the GPU tests use it."""
import numpy as np
import pytest

from oracle.oracle import Oracle, forecast_at


@pytest.mark.parametrize("plant_mode", [0, 1])
def test_closed_loop_formulas(ref_fixture, plant_mode):
    f = ref_fixture
    net, cfg = f["network"], f["config"]
    nx, nu, nd = net["nx"][0], net["nu"][0], net["nd"][0]
    B = np.array(net["matB"], float).reshape(nx, nu, order="F")
    xs, a1 = np.array(net["vecXsafe"], float), np.array(net["costAlpha1"], float)
    o = Oracle(net, f["tree"], cfg)
    o.factor_step()
    o.update_state_control()
    x, up = np.array(cfg["currentX"], float), np.array(cfg["prevU"], float)
    eco = smooth = safe = netk = 0.0
    for t in range(3):
        dh, ah = forecast_at(f["forecast"], t)
        u = o.control_action(dh, ah, max_iterations=40, project=True)
        # projectionBox with the scaled bounds of node 0 (SmpcController.cu:1649)
        raw = o.get("u")[:nu]
        assert np.array_equal(u, np.clip(raw, o.get("umin")[:nu], o.get("umax")[:nu]))
        e0, x_node0 = o.get("e")[:nx].copy(), o.get("x")[:nx].copy()
        xn, un, dn = o.move_forward(dh, ah, weight_economical=1.0, plant_mode=plant_mode)
        expect = x + B @ u + (e0 if plant_mode == 1 else 0.0)
        assert np.allclose(xn, expect, rtol=1e-14, atol=0)
        if plant_mode == 0:     # the reference's slip: the disturbance lands in the x iterate of node 0 (:1695)
            assert np.allclose(o.get("x")[:nx], x_node0 + e0, rtol=1e-15)
        else:
            assert np.array_equal(o.get("x")[:nx], x_node0)
        assert np.array_equal(un, u) and np.array_equal(dn, dh[:nd])
        eco += float(np.sum((a1 + ah[:nu]) * np.abs(u)))
        smooth += float(np.sum((up - u) ** 2))
        safe += float(np.sum(np.maximum(0.0, xs - xn)))
        netk += float(np.sum(np.abs(xn)))
        k = o.kpis(t + 1)
        assert np.allclose(k, (eco / 3600 / (t + 1), smooth / 3600 / (t + 1), 100 * (t + 1) * xs.sum() / netk, safe), rtol=1e-12)
        x, up = xn, un
        # the shifted triple is what the next control step starts from
        assert np.array_equal(o.get("curX"), xn) and np.array_equal(o.get("prevU"), un) and np.array_equal(o.get("prevD"), dn)


def test_control_action_without_projection_leaves_the_stored_control_alone(ref_fixture):
    """controlAction(real_t*) returns devVecU as it is and does not touch devControlAction (SmpcController.cu:1607-1626)."""
    f = ref_fixture
    o = Oracle(f["network"], f["tree"], f["config"])
    o.factor_step()
    o.update_state_control()
    dh, ah = forecast_at(f["forecast"], 0)
    u1 = o.control_action(dh, ah, max_iterations=20, project=True)
    stored = o.get("controlAction").copy()
    assert np.array_equal(stored, u1)
    u2 = o.control_action(dh, ah, max_iterations=25, project=False)
    assert np.array_equal(u2, o.get("u")[: o.nu])
    assert np.array_equal(o.get("controlAction"), stored)
