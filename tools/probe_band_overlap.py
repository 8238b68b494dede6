"""Upper bound of what a chain-banded two-stream schedule of ONE context could gain (round 5, review item 4): the 493 chains of the bench
workload as TWO independent contexts of 238 and 255 chains (17 x 14 and 17 x 15 trees, same network), each on its own stream, driven
by two host threads at once -- one context's helper launches then overlap the other's streaming kernel with NO cross-stream wait at
all, which a banded schedule of one context cannot avoid (the crown couples the bands once per iteration).  Prints one JSON line:
ms per iteration of the whole tree, of the halves run one after the other, and of the halves run concurrently.

usage: python tools/probe_band_overlap.py [iterations]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from rapidnet_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
synth.CONFIGS["half_a"] = (2, 63, 114, 88, 17, 24, [17, 14])
synth.CONFIGS["half_b"] = (2, 63, 114, 88, 17, 24, [17, 15])


def make(name):
    p = synth.make_problem(name, feasible=False)
    s = capi.Solver(p["network"], p["tree"], p["config"])
    s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
    s.apgReset()
    for _ in range(4):
        s.apgIterate(20, history=False)
    s.synchronize()
    return s


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        out.append(1e3 * (time.perf_counter() - t0) / n)
    return sorted(out)[len(out) // 2]


full = make("barcelona493")
t_full = timed(lambda: (full.apgIterate(n, history=False), full.synchronize()))
full.close()
a, b = make("half_a"), make("half_b")
t_a = timed(lambda: (a.apgIterate(n, history=False), a.synchronize()))
t_b = timed(lambda: (b.apgIterate(n, history=False), b.synchronize()))


def both():
    th = [threading.Thread(target=lambda s=s: (s.apgIterate(n, history=False), s.synchronize())) for s in (a, b)]
    for t in th:
        t.start()
    for t in th:
        t.join()


t_both = timed(both)
print(json.dumps({"iterations": n, "ms_per_iteration": {"whole_tree_493_chains": t_full, "half_238_chains_alone": t_a, "half_255_chains_alone": t_b,
                                                         "halves_one_after_the_other": t_a + t_b, "halves_concurrently_two_streams": t_both},
                  "gain_of_perfect_overlap_vs_whole_tree": 1.0 - t_both / t_full,
                  "note": "two independent contexts: no coupling between the halves, so this is an upper bound for a banded schedule of one context"}))
