"""Does the iteration time depend on WHERE the operator blocks were allocated?  Contexts created one after the other in one process
(each timed over 300 iterations), then with the previous context kept alive (so the next one gets other physical memory).
    python3 tools/placement_probe.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rapidnet_amd import capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
p = synth.make_problem("barcelona493")
dh, ah = synth.forecast_at(p["forecast"], 0)


def timed(s):
    s.apgReset(); s.apgIterate(100, history=False); s.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); s.apgIterate(300, history=False); s.synchronize()
        best = min(best, 1e3 * (time.perf_counter() - t) / 300)
    return best


for keep in (False, True):
    held = []
    for i in range(rounds):
        s = capi.Solver(p["network"], p["tree"], p["config"])
        s.initialiseSmpcController(dh, ah)
        print("keep_previous=%s context %d: %.4f ms per iteration" % (keep, i, timed(s)), flush=True)
        if keep:
            held.append(s)
        else:
            s.close()
    for s in held:
        s.close()
