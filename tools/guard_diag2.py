"""bisect: which first call of a fresh context's life produces NaNs under the guard"""
import os
import sys

os.environ["RAPIDNET_GUARD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from rapidnet_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "barcelona31"
BUFS = [(getattr(capi, k), k[4:]) for k in dir(capi) if k.startswith("BUF_") and getattr(capi, k) < capi.BUF_PREV_XI]
p = synth.make_problem(name)
dh, ah = synth.forecast_at(p["forecast"], 0)


def report(s, what):
    bad = []
    for bid, nm in sorted(BUFS):
        a = s.get(bid)
        n = int(np.isnan(a).sum())
        if n:
            bad.append("%s %d/%d" % (nm, n, a.size))
    print("%-44s %s" % (what, "; ".join(bad) if bad else "clean"), flush=True)


def fresh():
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(dh, ah)
    return s


s = fresh(); s.apgReset(); s.solveStep(); report(s, "initialise, solveStep"); s.close()
s = fresh(); s.synchronize(); s.apgReset(); s.apgIterate(1); report(s, "initialise, sync, apgIterate(1)"); s.close()
s = fresh(); s.apgReset(); s.apgIterate(1); report(s, "initialise, apgIterate(1)"); s.close()
s = fresh(); s.apgReset(); s.apgIterate(2); report(s, "initialise, apgIterate(2)"); s.close()
s = fresh(); s.apgReset(); s.solveStep(); s.apgReset(); s.apgIterate(2); report(s, "initialise, solveStep, apgIterate(2)"); s.close()
s = fresh(); s.get(capi.BUF_X); s.apgReset(); s.apgIterate(2); report(s, "initialise, get(X), apgIterate(2)"); s.close()
s = fresh(); s.get(capi.BUF_XI); s.apgReset(); s.apgIterate(2); report(s, "initialise, get(XI) [k_pack], apgIterate(2)"); s.close()
s = fresh(); s.setExchangeMode(0); s.apgReset(); s.apgIterate(20); report(s, "initialise, exact mode, apgIterate(20)"); s.close()
s = fresh(); s.apgReset(); s.apgIterate(20); report(s, "initialise, apgIterate(20) optimistic"); s.close()
