import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from rapidnet_amd import capi, synth
for name, st in (("barcelona31", False), ("barcelona493", True), ("barcelona493", False)):
    p = synth.make_problem(name); dh, ah = synth.forecast_at(p["forecast"], 0)
    s = capi.Solver(p["network"], p["tree"], p["config"], structured=st); s.initialiseSmpcController(dh, ah); s.apgReset()
    for _ in range(15): s.apgIterate(20, history=False)
    s.synchronize(); r = []
    for _ in range(15):
        t0 = time.perf_counter(); s.apgIterate(20, history=False); s.synchronize(); r.append(1e3 * (time.perf_counter() - t0) / 20)
    print(name, "structured" if st else "dense", "ms/it in batches of 20: median %.5f min %.5f" % (float(np.median(r)), min(r))); s.close()
