import sys, time, json
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from rapidnet_amd import capi, synth
p = synth.make_problem("barcelona493")
dh, ah = synth.forecast_at(p["forecast"], 0)
s = capi.Solver(p["network"], p["tree"], p["config"], structured=True)
s.initialiseSmpcController(dh, ah); s.apgReset()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    s.apgIterate(20, history=False); s.synchronize()
res = {0: [], 1: []}
for r in range(5):
    for f in (0, 1):
        s.setFusedWalkDual(f)
        s.apgIterate(40, history=False); s.synchronize()
        t0 = time.perf_counter(); s.apgIterate(200, history=False); s.synchronize()
        res[f].append(1e3 * (time.perf_counter() - t0) / 200)
print(json.dumps({k: [float(np.median(v)), min(v), max(v)] for k, v in res.items()}))
