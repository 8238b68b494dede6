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
