import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FIXTURE = os.path.join(GOLDEN, "reference_fixture")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref_fixture():
    """The reference's own test data files (src/test/testDataFiles/*.json), committed as data."""
    from oracle.oracle import load_json

    names = {"network": "network.json", "tree": "scenarioTree.json", "config": "controllerConfig.json",
             "forecast": "forecastor.json", "engine": "engineTest.json", "smpc": "smpcTest.json",
             "smpc_fbe": "smpcFbeTest.json", "smpc_nama": "smpcNamaTest.json"}
    return {k: load_json(os.path.join(REF_FIXTURE, v)) for k, v in names.items()}


# ---- progress lines while a long GPU test runs -----------------------------------------------------------------------------------
# A test that drives bench.py end to end says nothing for minutes (its child's output is captured); the GPU box takes a run that
# writes nothing for 7 minutes to be hung.  During `-m gpu` sessions a background thread therefore reports once a minute which test
# is running and for how long -- to the real stderr and to gpurun_out/pytest_progress.log.  It stops reporting on a test that has
# been running for 12 minutes: a real hang must still look like one.
_progress = {"test": None, "t0": 0.0, "stop": None}


def pytest_runtest_logstart(nodeid, location):
    import time

    _progress["test"], _progress["t0"] = nodeid, time.time()


def pytest_sessionstart(session):
    import threading
    import time

    if "gpu" not in (session.config.getoption("-m") or "") or "not gpu" in (session.config.getoption("-m") or ""):
        return
    stop = threading.Event()
    _progress["stop"] = stop
    path = os.path.join(ROOT, "gpurun_out", "pytest_progress.log")

    def run():
        while not stop.wait(60.0):
            t = _progress["test"]
            if t is None or time.time() - _progress["t0"] > 720.0:
                continue
            line = "[pytest progress] %s running for %.0f s\n" % (t, time.time() - _progress["t0"])
            try:
                sys.__stderr__.write(line)
                sys.__stderr__.flush()
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "a") as f:
                    f.write(line)
            except OSError:
                pass

    threading.Thread(target=run, daemon=True).start()


def pytest_sessionfinish(session, exitstatus):
    if _progress["stop"] is not None:
        _progress["stop"].set()


# ---- the full-size oracle run, behind the other tests ----------------------------------------------------------------------------
# tests/test_gpu_fullsize.py::test_500_iterations_against_the_oracle_at_full_size needs the CPU oracle's 500 iterations on the
# 10 864-node tree: ~0.2 s each on one host core, 100 s in which the GPU has nothing to do.  When that test is part of the session,
# the oracle's run starts on a background thread as soon as the tests are collected (its C code runs without the GIL; every Oracle
# instance owns its state) and the test collects the snapshots -- iterates after 2, 100 and 500 iterations, the affine terms, two
# operators at four nodes, the primal-infeasibility history -- when it gets there.  Same oracle, same inputs, same comparisons.
class _FullSizeOracle:
    NAME = "barcelona493"
    VECS = ("x", "u", "v", "updXi", "updPsi", "dualXi", "resPsi", "primalPsi", "accPsi")

    def __init__(self):
        self.thread, self.snap, self.error, self.problem = None, None, None, None

    def start(self):
        import threading

        from oracle.oracle import Oracle
        from rapidnet_amd import synth

        p = synth.make_problem(self.NAME)
        dh, ah = synth.forecast_at(p["forecast"], 0)
        self.problem = (p, (dh, ah))
        o = Oracle(p["network"], p["tree"], p["config"])      # created on the main thread (the aliasing switch of the C file is process-wide)

        def run():
            try:
                o.initialise(dh, ah)
                snap = {"static": {nm: o.get(nm) for nm in ("uhat", "e", "beta", "xmax", "umax")}, "dims": (o.nv, o.nx, o.nu, o.nodes)}
                nv, nx, nu = o.nv, o.nx, o.nu
                phi, ftil = o.get("Phi").reshape(-1, nv * 2 * nx), o.get("Ftil").reshape(-1, nv * nu)
                snap["ops"] = {n: (phi[n].copy(), ftil[n].copy()) for n in (0, 5, 4000, o.nodes - 1)}
                del phi, ftil
                snap["hist2"] = o.apg(2)
                snap[2] = {nm: o.get(nm) for nm in self.VECS}
                o.apg_reset()
                th, hist, done = [1.0, 1.0], [], 0
                for total in (100, 500):
                    for _ in range(total - done):
                        th = o.apg_continue(1, th)
                        hist.append(o.primal_infeasibility())
                    done = total
                    snap[total] = {nm: o.get(nm) for nm in self.VECS}
                snap["hist"] = hist
                self.snap = snap
            except BaseException as e:   # noqa: BLE001 -- handed to the test
                self.error = e

        self.thread = threading.Thread(target=run, daemon=True)
        self.thread.start()

    def result(self):
        if self.thread is None:
            self.start()
        self.thread.join()
        if self.error is not None:
            raise self.error
        return self.problem, self.snap


_fullsize_oracle = _FullSizeOracle()


def pytest_collection_modifyitems(session, config, items):
    """the test that collects the full-size oracle's 100 s background run goes last: by then the run has finished behind the other tests
    (in file order it came up after ~90 s and waited for the rest)"""
    last = [it for it in items if "test_500_iterations_against_the_oracle_at_full_size" in it.nodeid]
    if last:
        items[:] = [it for it in items if it not in last] + last


def pytest_collection_finish(session):
    if any("test_500_iterations_against_the_oracle_at_full_size" in item.nodeid for item in session.items) and not session.config.option.collectonly:
        _fullsize_oracle.start()


@pytest.fixture(scope="session")
def fullsize_oracle():
    return _fullsize_oracle.result()


def run_pair(fa, fb):
    """two CPU-oracle runs side by side (the oracle's C code runs without the GIL, every Oracle instance owns its state): the base run and the
    perturbed run of the long-run parity tests, which otherwise keep the GPU box waiting for twice their time"""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:
        a, b = ex.submit(fa), ex.submit(fb)
        return a.result(), b.result()
