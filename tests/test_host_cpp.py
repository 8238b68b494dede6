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


@pytest.mark.gpu
def test_engine_cpp():
    _run("engine")


@pytest.mark.gpu
def test_controller_known_answers_cpp():
    _run("controller")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fbe", "nama"])
def test_fbe_nama_known_answers_cpp(mode):
    """Testing::testSmpcFbeController / testSmpcNamaController re-run on the reference's own vectors."""
    out = _run(mode)
    assert "all checks passed" in out


@pytest.mark.gpu
def test_closed_loop_cpp():
    _run("closedloop")


@pytest.mark.gpu
def test_null_space_basis_invariance_cpp():
    out = _run("nullspace")
    assert "null-space basis invariance" in out


@pytest.mark.gpu
def test_warm_start_cpp():
    _run("warmstart")


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,kw,replayed", [("medium", 2, {}, 0), ("medium", 8, {}, 0), ("ragged", 3, {}, 0),
                                                    ("medium", 4, {"penalty_x": 20.0, "penalty_xs": 5.0}, 1)])
def test_sharded_controllers_cpp(tmp_path, name, world, kw, replayed):
    """Multi-GPU through the C++ class surface: `world` SmpcController(path, rank, world, id) objects of
    one process, one thread each, against the unsharded controller -- see testSharded in tests/cpp/test_host.cpp.  With the
    small penalties the soft-constraint thresholds trip, so the optimistic batch is replayed through the exact
    two-collective path on all ranks together."""
    from rapidnet_amd import synth

    synth.write_problem(synth.make_problem(name, max_iterations=40, **kw), str(tmp_path))
    out = _run("sharded", str(tmp_path), world)
    assert "sharded: %d ranks" % world in out and "all checks passed" in out
    # one algorithmApg + three control steps: all optimistic, or one replayed batch followed by the back-off's exact batches
    assert ("optimistic/exact/replayed %s" % ("1/3/1" if replayed else "4/0/0")) in out, out
