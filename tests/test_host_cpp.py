"""Runs the C++ tests of the host classes (tests/cpp/test_host.cpp), which mirror the reference's own harness
(src/test/Testing.cu, src/test/TestSmpcController.cu) on the reference's fixture files."""
import os
import subprocess

import pytest

from conftest import REF_FIXTURE, ROOT
from rapidnet_amd import build


def _run(mode, directory=REF_FIXTURE, *extra):
    exe = build.TEST_HOST
    if not os.path.exists(exe):
        build.build_host()
    r = subprocess.run([exe, mode, directory] + [str(e) for e in extra], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "test_host %s failed (rc %d):\n%s\n%s" % (mode, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return r.stdout


def test_loaders_cpp():
    out = _run("loaders")
    assert "finalBranchNode 10 finalBranchStage 2" in out


def test_loaders_cpp_on_synthetic_files(tmp_path):
    from rapidnet_amd import synth

    synth.write_problem(synth.make_problem("small"), str(tmp_path))
    _run("loaders", str(tmp_path))


# Every known-answer mode runs in BOTH operator modes (round 6): "dense" is the reference's storage model, "auto" what a drop-in
# SmpcController(path) gets when the configuration file says nothing -- the structured form, never a per-node block -- and the
# reference's own vectors must come out of either.
OPS = ["ops=dense", "ops=auto"]


@pytest.mark.gpu
@pytest.mark.parametrize("ops", OPS + ["ops=structured"])
def test_engine_cpp(ops):
    _run("engine", REF_FIXTURE, ops)


@pytest.mark.gpu
@pytest.mark.parametrize("ops", OPS)
def test_controller_known_answers_cpp(ops):
    out = _run("controller", REF_FIXTURE, ops)
    assert "all checks passed [%s]" % ops[4:] in out


@pytest.mark.gpu
@pytest.mark.parametrize("ops", OPS)
@pytest.mark.parametrize("mode", ["fbe", "nama"])
def test_fbe_nama_known_answers_cpp(mode, ops):
    """Testing::testSmpcFbeController / testSmpcNamaController re-run on the reference's own vectors."""
    out = _run(mode, REF_FIXTURE, ops)
    assert "all checks passed" in out


@pytest.mark.gpu
@pytest.mark.parametrize("ops", OPS)
def test_closed_loop_cpp(ops):
    _run("closedloop", REF_FIXTURE, ops)


@pytest.mark.gpu
def test_default_of_the_class_surface_is_the_structured_form_cpp():
    """a drop-in SmpcController(path) on the reference's own configuration file (no operatorMode key) runs the exact fast path"""
    out = _run("opsmode")
    assert "default: auto -> structured" in out and "after setOperator: dense" in out


@pytest.mark.gpu
def test_null_space_basis_invariance_cpp():
    out = _run("nullspace")
    assert "null-space basis invariance" in out


@pytest.mark.gpu
def test_warm_start_cpp():
    _run("warmstart")


@pytest.mark.gpu
@pytest.mark.parametrize("ops", OPS)
@pytest.mark.parametrize("name,world,kw,replayed", [("medium", 2, {}, 0), ("medium", 8, {}, 0), ("ragged", 3, {}, 0),
                                                    ("medium", 4, {"penalty_x": 20.0, "penalty_xs": 5.0}, 1)])
def test_sharded_controllers_cpp(tmp_path, name, world, kw, replayed, ops):
    """Multi-GPU through the C++ class surface: `world` SmpcController(path, rank, world, id) objects of
    one process, one thread each, against the unsharded controller -- see testSharded in tests/cpp/test_host.cpp.  With the
    small penalties the soft-constraint thresholds trip, so the optimistic batch is replayed through the exact
    two-collective path on all ranks together."""
    from rapidnet_amd import synth

    synth.write_problem(synth.make_problem(name, max_iterations=40, **kw), str(tmp_path))
    out = _run("sharded", str(tmp_path), world, ops)
    assert "sharded: %d ranks" % world in out and "all checks passed" in out
    # one algorithmApg + three control steps: all optimistic, or one replayed batch followed by the back-off's exact batches
    assert ("optimistic/exact/replayed %s" % ("1/3/1" if replayed else "4/0/0")) in out, out
