"""Times the global-FBE / NAMA loops (rn_algorithm_fbe_nama) on a synthetic workload; prints one JSON line per run.

usage: python tools/time_fbe_nama.py [workload] [iterations] [f64|f32] [knob=value ...]     (default: barcelona493 30 f64; knobs: rapidnet_amd.capi.KNOBS,
the library's test hook rn_debug_set_knob -- e.g. nama_pair=0, value_mfma=0, ls_sequential=1 for the A/B of tools/ab_fbe.sh)
Not a bench.py line (the headline metric is APG iterations/s); the numbers go into DESIGN.md.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from rapidnet_amd import capi, synth

name = sys.argv[1] if len(sys.argv) > 1 else "barcelona493"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
precision = sys.argv[3] if len(sys.argv) > 3 else "f64"
knobs = {k: int(v) for k, _, v in (a.partition("=") for a in sys.argv[4:])}
p = synth.make_problem(name)
dh, ah = synth.forecast_at(p["forecast"], 0)
for structured in (False, True):
    for alg in ("proximalAlgorithm", "globalFbeAlgorithm", "namaAlgorithm"):
        s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured, precision=precision, knobs=knobs or None)
        s.initialiseSmpcController(dh, ah)
        if alg == "proximalAlgorithm":
            s.algorithmApg(5); s.synchronize()
            t = time.perf_counter(); h = s.algorithmApg(iters); s.synchronize(); dt = time.perf_counter() - t
            extra = {}
        else:
            s.setAlgorithm(alg, 5)
            run = s.algorithmGlobalFbe if alg == "globalFbeAlgorithm" else s.algorithmNama
            run(3)
            t = time.perf_counter(); h, v, tau = run(iters); dt = time.perf_counter() - t
            trials = [1 + {1.0: 0}.get(x, 0) for x in tau]
            extra = {"tau": [float(x) for x in tau[:12]], "value_first_last": [float(v[0]), float(v[-1])], "line_searches": s.fbeCounters()}
        print(json.dumps({"workload": name, "precision": precision, "structured": structured, "algorithm": alg, "iterations": iters,
                          "ms_per_iteration": 1e3 * dt / iters, "primal_inf_first_last": [float(h[0]), float(h[-1])], **extra}))
        s.close()
