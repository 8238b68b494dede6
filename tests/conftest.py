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
