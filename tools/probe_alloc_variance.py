#!/usr/bin/env python3
"""Does the streaming kernel's time depend on WHICH allocation holds the operator blocks?  Creates the solver several times
in one process (freeing it in between, or keeping the previous ones alive so that the next allocation lands elsewhere) and
prints the hipEvent time of k_stream_gemv for each.   python tools/probe_alloc_variance.py [rounds] [keep]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rapidnet_amd import capi, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
keep = len(sys.argv) > 2 and sys.argv[2] == "keep"
p = synth.make_problem("barcelona493")
dh, ah = synth.forecast_at(p["forecast"], 0)
alive = []
for r in range(rounds):
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(dh, ah)
    s.apgReset()
    s.apgIterate(10, history=False)
    s.profileEnable(1); s.profileReset()
    s.apgIterate(40, history=False)
    ms, n = s.profileRead()
    s.profileEnable(0)
    print("round %d (%s): k_stream_gemv %.1f us  rest %.1f us" % (r, "kept" if keep else "freed", 1e3 * ms[0] / n[0], 1e3 * (ms[1] / n[1] + ms[2] / n[2] + ms[3] / n[3])), flush=True)
    if keep and len(alive) < 8:
        alive.append(s)
    else:
        s.close()
