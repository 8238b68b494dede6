"""Cross-check of the JSON data formats against the REFERENCE's own loader classes (oracle/_ref/libref_loaders.so,
built by oracle/build_ref.sh from /root/reference/src in place): the reference fixture AND the synthetic problem
files written by rapidnet_amd.synth must read back identically (to fp32, the reference parses with GetFloat()).

The reference's ScenarioTree constructor writes past two of its heap blocks (it allocates N / N+1 entries for
nodesPerStage / nodesPerStageCumul but the JSON carries N+1 / N+2, ScenarioTree.cu:66-75), so its code is only ever
run in a CHILD process that leaves through os._exit(): `python tests/test_ref_loaders.py <directory>`, with a
heap-padding preload (oracle/malloc_pad.c, built by oracle/build_ref.sh) so that the overrun cannot reach a chunk header."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from rapidnet_amd import synth  # noqa: E402
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_loaders.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref not built (needs /root/reference)")


def _lib():
    lib = C.CDLL(LIB)
    for f in ("ref_network_new", "ref_tree_new", "ref_config_new", "ref_forecaster_new"):
        getattr(lib, f).restype = C.c_void_p
        getattr(lib, f).argtypes = [C.c_char_p]
    for f in ("ref_network_array", "ref_tree_array", "ref_config_array"):
        getattr(lib, f).restype = C.POINTER(C.c_float)
        getattr(lib, f).argtypes = [C.c_void_p, C.c_char_p]
    lib.ref_tree_int_array.restype = C.POINTER(C.c_int)
    lib.ref_tree_int_array.argtypes = [C.c_void_p, C.c_char_p]
    lib.ref_config_string.restype = C.c_char_p
    lib.ref_config_string.argtypes = [C.c_void_p, C.c_char_p]
    for f in ("ref_network_dims", "ref_tree_dims", "ref_config_dims", "ref_forecaster_dims", "ref_config_scalars"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_void_p]
    lib.ref_forecaster_predict.argtypes = [C.c_void_p, C.c_int]
    for f in ("ref_forecaster_demand", "ref_forecaster_prices"):
        getattr(lib, f).restype = C.POINTER(C.c_float)
        getattr(lib, f).argtypes = [C.c_void_p]
    return lib


def _close(ptr, expected):
    e = np.asarray(expected, float)
    got = np.ctypeslib.as_array(ptr, shape=(e.size,)).astype(float)
    return np.allclose(got, e, rtol=2e-7, atol=1e-30)


def _check_dir(d):
    lib = _lib()
    net, tree, cfg, fc = (json.load(open(os.path.join(d, f))) for f in ("network.json", "scenarioTree.json", "controllerConfig.json", "forecastor.json"))
    dims = (C.c_int * 8)()
    h = lib.ref_network_new(os.path.join(d, "network.json").encode())
    lib.ref_network_dims(h, dims)
    assert list(dims[:4]) == [net["nx"][0], net["nu"][0], net["nd"][0], net["ne"][0]]
    for k in ("matA", "matB", "matGd", "matE", "matEd", "vecXmin", "vecXmax", "vecXsafe", "vecUmin", "vecUmax", "costAlpha1"):
        assert _close(lib.ref_network_array(h, k.encode()), net[k]), k
    h = lib.ref_tree_new(os.path.join(d, "scenarioTree.json").encode())
    lib.ref_tree_dims(h, dims)
    assert list(dims[:5]) == [tree["N"][0], tree["K"][0], tree["nodes"][0], tree["nNonLeafNodes"][0], tree["nChildrenTot"][0]]
    for k in ("stages", "leaves", "children", "ancestor", "nChildren", "nChildrenCumul"):
        got = np.ctypeslib.as_array(lib.ref_tree_int_array(h, k.encode()), shape=(len(tree[k]),))
        assert np.array_equal(got, np.asarray(tree[k], int)), k
    for k in ("probNode", "errorDemandNode", "errorPriceNode"):
        assert _close(lib.ref_tree_array(h, k.encode()), tree[k]), k
    h = lib.ref_config_new(os.path.join(d, "controllerConfig.json").encode())
    lib.ref_config_dims(h, dims)
    assert list(dims[:5]) == [cfg["nx"][0], cfg["nu"][0], cfg["nd"][0], cfg["nv"][0], cfg["maxIterations"][0]]
    sc = (C.c_float * 3)()
    lib.ref_config_scalars(h, sc)
    assert np.allclose(list(sc), [cfg["stepSize"][0], cfg["penaltyStateX"][0], cfg["penaltySafetyX"][0]], rtol=2e-7)
    for k in ("matL", "matLhat", "costW", "matDiagPrecnd", "currentX", "prevU", "prevDemand"):
        assert _close(lib.ref_config_array(h, k.encode()), cfg[k]), k
    assert lib.ref_config_string(h, b"algorithmName").decode() == cfg["algorithmName"]
    h = lib.ref_forecaster_new(os.path.join(d, "forecastor.json").encode())
    lib.ref_forecaster_dims(h, dims)
    assert list(dims[:4]) == [fc["N"][0], fc["simHorizon"][0], fc["dimDemand"][0], fc["dimPrices"][0]]
    keys = list(fc.keys())
    for t in range(2):
        assert lib.ref_forecaster_predict(h, t) == 1
        assert _close(lib.ref_forecaster_demand(h), fc[keys[4 + 2 * t]])
        assert _close(lib.ref_forecaster_prices(h), fc[keys[5 + 2 * t]])
        dh, ah = synth.forecast_at(fc, t)     # this repo's member-order rule is the reference's
        assert _close(lib.ref_forecaster_demand(h), dh) and _close(lib.ref_forecaster_prices(h), ah)


PAD = os.path.join(ROOT, "oracle", "_ref", "libmalloc_pad.so")


def _check_in_child(directory):
    # the child preloads oracle/malloc_pad.c: every heap request is padded, so the reference constructor's 4-byte overrun
    # (ScenarioTree.cu:66-75) lands in padding instead of the next chunk's header whatever the heap layout is
    env = dict(os.environ)
    if os.path.exists(PAD):
        env["LD_PRELOAD"] = PAD + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), directory], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "REF_LOADERS_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_reference_fixture_through_reference_loaders():
    _check_in_child(os.path.join(ROOT, "tests", "golden", "reference_fixture"))


@pytest.mark.parametrize("name", ["toy", "small", "medium"])
def test_synthetic_files_through_reference_loaders(tmp_path, name):
    p = synth.make_problem(name)
    synth.write_problem(p, str(tmp_path))
    _check_in_child(str(tmp_path))


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    _check_dir(sys.argv[1])
    print("REF_LOADERS_OK", flush=True)
    os._exit(0)   # skip interpreter teardown: the reference's loaders have corrupted the heap by now
