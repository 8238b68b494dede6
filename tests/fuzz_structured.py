"""One-off fuzz of the structured operator mode against the CPU oracle on random shapes (tests/test_gpu_random_shapes.py's generator): optimistic
batches (the chain walk riding in the fused launch), a short exact batch, a second control step (the constants of the control step refreshed), fp64.
    python tests/fuzz_structured.py [first_seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle  # noqa: E402  (test infrastructure: this tool is a test)
from rapidnet_amd import capi, synth  # noqa: E402
from test_gpu_random_shapes import random_config, relmax  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
worst = 0.0
for seed in range(first, first + n):
    name = "_fuzz_%d" % seed
    synth.CONFIGS[name] = random_config(seed)
    try:
        p = synth.make_problem(name)
    finally:
        del synth.CONFIGS[name]
    dh, ah = synth.forecast_at(p["forecast"], 0)
    dh1, ah1 = synth.forecast_at(p["forecast"], 1)
    o = Oracle(p["network"], p["tree"], p["config"])
    o.initialise(dh, ah)
    s = capi.Solver(p["network"], p["tree"], p["config"], operator_mode="structured")
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    h = np.concatenate([s.apgIterate(20), s.apgIterate(3), s.apgIterate(17)])
    ho = o.apg(40)
    errs = {nm: relmax(s.get(bid), o.get(nm)) for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"), (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"))}
    e1 = max(errs.values()); eh = float(np.abs(h - ho).max() / max(np.abs(ho).max(), 1.0))
    # second control step: new forecast, fresh iteration
    o.eliminate(dh1, ah1)
    s.eliminateInputDistubanceCoupling(dh1, ah1)
    s.apgReset()
    h2 = s.apgIterate(32)
    ho2 = o.apg(32)
    errs2 = {nm: relmax(s.get(bid), o.get(nm)) for bid, nm in ((capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_V, "v"))}
    e2 = max(errs2.values())
    shape = {k: p["tree"][k][0] for k in ("N", "K", "nodes")}
    print("seed %3d %s nx %d nu %d: step 1 %.1e (hist %.1e)  step 2 %.1e" % (seed, shape, s.nx, s.nu, e1, eh, e2), flush=True)
    worst = max(worst, e1, e2, eh)
    s.close()
print("worst %.2e" % worst)
sys.exit(0 if worst < 1e-8 else 1)
