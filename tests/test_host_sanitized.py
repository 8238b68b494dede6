"""The subtree partitioner behind rn_partition_create / rn_create_sharded (rapidnet_amd/csrc/partition.hpp, host-only C++) under
AddressSanitizer and UBSan on the CPU: 300 random trees (uniform and per-node child counts), every cut stage, 1-8 ranks,
with the invariants of a local tree (breadth-first order, contiguous children, every subtree node owned exactly once,
children moments).  GPU sanitizers are not available on the pool; this is the part of the sharding code that is plain C++."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_partitioner_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "partition_sanitize")
    src = os.path.join(ROOT, "tests", "cpp", "partition_sanitize.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, src], check=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "partitions built" in out.stdout


def _san(tmp_path, name, sources, extra=()):
    exe = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread", "-o", exe] + sources + list(extra),
                   check=True, timeout=900)
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_null_space_routine_is_clean_under_asan_and_ubsan(tmp_path):
    """Engine::calculateMatLandMatLhat on the host (one-sided Jacobi SVD): 200 random E, every fifth rank-deficient; E L = 0,
    L'L = I, E Lhat = -Ed."""
    exe = _san(tmp_path, "host_sanitize", [os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp"), os.path.join(ROOT, "rapidnet_amd", "csrc", "host", "NullSpace.cpp")])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "null-space runs 200" in out.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_json_loaders_are_clean_under_asan_and_ubsan(tmp_path):
    """DwnNetwork / ScenarioTree / Forecaster / SmpcConfiguration (the reference's loader classes re-built on JsonLite) parse the
    reference's own fixture files under the sanitizers: the host sources compiled in, linked against the C-ABI library (which is
    only loaded, never called, by this mode)."""
    from rapidnet_amd import build

    build.build_hip()
    hdir = os.path.join(ROOT, "rapidnet_amd", "csrc", "host")
    srcs = [os.path.join(ROOT, "tests", "cpp", "test_host.cpp")] + [os.path.join(hdir, f) for f in ("DataModel.cpp", "Engine.cpp", "SmpcController.cpp", "NullSpace.cpp")]
    lib = os.path.join(ROOT, "rapidnet_amd")
    exe = _san(tmp_path, "test_host_asan", srcs, ["-L" + lib, "-lrapidnet_hip", "-Wl,-rpath," + lib])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")      # the HIP runtime's own start-up allocations are not ours to judge
    out = subprocess.run([exe, "loaders", os.path.join(ROOT, "tests", "golden", "reference_fixture")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "all checks passed" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
