"""Runs the given test files in THIS process with RAPIDNET_GUARD=1 and prints the process-wide guard tally afterwards
(contexts checked at destruction, red-zone bytes found overwritten):  python tools/guard_suite.py tests/test_gpu_sharded_batched.py ..."""
import gc
import os
import sys

os.environ["RAPIDNET_GUARD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402

rc = pytest.main(["-q", "-x", "--timeout", "900"] + sys.argv[1:])
gc.collect()
from rapidnet_amd import capi  # noqa: E402

rep = capi.guard_report()
print("guard report: %d contexts checked, %d red-zone bytes overwritten" % (rep[0], rep[1]))
sys.exit(int(rc) or (1 if rep[1] else 0))
