"""Device-buffer guard mode ("race detection / sanitizers"; no GPU address sanitizer exists on this pool).

With RAPIDNET_GUARD=1 every device buffer of a context sits between two 128 KiB red zones; red zones and payloads of the
floating-point buffers start as NaN.  A kernel that reads outside its buffer -- the streaming kernel's clamped re-reads of the
last slot, the slab products' prefetches past the last group, the zero-padded operator columns are all meant to stay INSIDE --
or reads something nobody wrote, drags a NaN into the iterates: the parity checks below then fail (`relmax` asserts finiteness).
A kernel that writes outside its buffer changes a red zone: every context is checked when it is destroyed and the process-wide
tally (rn_guard_report) must stay at zero.  The bodies are the parity tests themselves, run once more under the guard."""
import gc

import numpy as np
import pytest

import test_gpu_fbe_nama as fbe
import test_gpu_parity as par
import test_gpu_sharded_batched as shb
from rapidnet_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def guard(monkeypatch):
    monkeypatch.setenv("RAPIDNET_GUARD", "1")
    gc.collect()
    before = capi.guard_report()
    box = {}
    yield box
    gc.collect()
    after = capi.guard_report()
    assert after[0] - before[0] >= box.get("contexts", 1), "no context was created (and checked) under the guard: %s -> %s" % (before, after)
    assert after[1] == before[1], "a kernel wrote outside its buffer: %d red-zone bytes overwritten" % (after[1] - before[1])


def test_guard_mode_detects_an_overwritten_red_zone(monkeypatch):
    """The detector itself: payloads start as NaN (a buffer nobody has written reads back as NaN), a clean solve leaves every
    red zone intact, and a deliberate write behind a buffer (rn_debug_guard_poke: 24 bytes right behind the payload of the
    context's first buffer) is counted by rn_guard_check and by the tally rn_destroy keeps."""
    monkeypatch.setenv("RAPIDNET_GUARD", "1")
    p = synth.make_problem("tiny")
    s2 = capi.Solver(p["network"], p["tree"], p["config"])
    assert np.isnan(s2.get(capi.BUF_BETA)).all()
    s2.close()
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
    s.algorithmApg(20)
    assert s.guardCheck() == 0
    assert np.isfinite(s.get(capi.BUF_X)).all()
    s.debugGuardPoke(24)
    assert s.guardCheck() == 24
    before = capi.guard_report()
    s.close()
    after = capi.guard_report()
    assert after[0] == before[0] + 1 and after[1] == before[1] + 24
    monkeypatch.delenv("RAPIDNET_GUARD")
    s3 = capi.Solver(p["network"], p["tree"], p["config"])          # outside guard mode: nothing to check, nothing counted
    assert s3.guardCheck() == 0
    s3.close()
    assert capi.guard_report() == after


@pytest.mark.parametrize("name", ["toy", "odd", "medium"])
def test_stepwise_under_guard(guard, name):
    par.test_factor_step_and_affine_terms(name)
    par.test_stepwise_known_answer(name)
    guard["contexts"] = 2


@pytest.mark.parametrize("name,iters", [("small", 40), ("odd", 40), ("medium", 25)])
def test_apg_dense_and_structured_under_guard(guard, name, iters):
    par.test_apg_iterates_match_oracle(name, iters)
    par.test_structured_operator_mode(name, iters)
    guard["contexts"] = 2


@pytest.mark.parametrize("structured", [False, True])
@pytest.mark.parametrize("name,alias", [("widecrown", True), ("ragged", True), ("horizon1", False), ("fan", True), ("late", False)])
def test_edge_shapes_under_guard(guard, name, alias, structured):
    par.test_edge_tree_shapes(name, alias, structured)


def test_fp32_and_soft_branch_under_guard(guard):
    par.test_fp32_path()
    par.test_structured_fp32_and_soft_branch()
    par.test_soft_constraint_branch()
    guard["contexts"] = 4


@pytest.mark.parametrize("name,world,cut,structured,kw,trips", [("medium", 3, 2, False, {}, False), ("ragged", 3, 1, False, {}, False),
                                                                ("medium", 2, 1, True, {"penalty_x": 20.0, "penalty_xs": 5.0}, True)])
def test_sharded_batches_under_guard(guard, name, world, cut, structured, kw, trips):
    shb.test_batched_sharded_solve_matches_oracle(name, world, cut, structured, kw, trips)
    guard["contexts"] = world


@pytest.mark.parametrize("alg", fbe.ALGS)
def test_fbe_nama_loops_under_guard(guard, alg):
    fbe.test_loop_matches_oracle("small", False, alg)
    fbe.test_loop_matches_oracle("odd", False, alg)          # odd ny: the streaming kernel's two right-hand sides (NAMA) and k_value_mfma on ragged tiles
    fbe.test_loop_matches_oracle("medium", True, alg)
    fbe.test_line_search_direction_rule(alg, "positive")
    guard["contexts"] = 4


def test_barcelona31_under_guard(guard):
    """The full-size K = 31 Barcelona config: the kernel instantiations and launch shapes of the headline workload (G = 8 spans,
    two slots per thread, pipelined slab products, k_dual_stage tiles)."""
    par.test_barcelona31_full_size()
