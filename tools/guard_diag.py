"""Where does a NaN come from under RAPIDNET_GUARD=1?  Prints the NaN count of every caller-visible buffer after each step
of one APG iteration and after short batches: python tools/guard_diag.py <workload> [precision] [structured]"""
import os
import sys

os.environ["RAPIDNET_GUARD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from rapidnet_amd import capi, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "barcelona31"
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
structured = len(sys.argv) > 3 and sys.argv[3] == "1"
BUFS = [(getattr(capi, k), k[4:]) for k in dir(capi) if k.startswith("BUF_") and getattr(capi, k) < capi.BUF_PREV_XI]


def report(s, what):
    bad = []
    for bid, nm in sorted(BUFS):
        a = s.get(bid)
        n = int(np.isnan(a).sum())
        if n:
            first = int(np.flatnonzero(np.isnan(a))[0])
            bad.append("%s %d/%d (first %d)" % (nm, n, a.size, first))
    print("%-28s %s" % (what, "; ".join(bad) if bad else "clean"), flush=True)


p = synth.make_problem(name)
dh, ah = synth.forecast_at(p["forecast"], 0)
s = capi.Solver(p["network"], p["tree"], p["config"], precision=prec, structured=structured)
print(name, prec, "structured" if structured else "dense", "nodes", s.nodes, s.kernelInfo())
s.factorStep(); report(s, "factorStep")
s.updateStateControl(); s.eliminateInputDistubanceCoupling(dh, ah); report(s, "affine terms")
s.apgReset(); report(s, "apgReset")
s.dualExtrapolationStep(0.0); report(s, "extrapolate")
s.solveStep(); report(s, "solveStep")
s.proximalFunG(); report(s, "prox")
s.computeFixedPointResidual(); report(s, "residual")
s.dualUpdate(); report(s, "dualUpdate")
s.apgReset()
s.apgIterate(1); report(s, "apgIterate(1) exact")
s.apgIterate(3); report(s, "apgIterate(3) exact")
s.apgReset()
s.apgIterate(20); report(s, "apgIterate(20) optimistic")
print("red zones overwritten:", s.guardCheck())
# the sequence of the parity tests: a fresh context, initialise, algorithmApg(n) straight away
for n in (5, 2, 8, 16):
    s2 = capi.Solver(p["network"], p["tree"], p["config"], precision=prec, structured=structured)
    s2.initialiseSmpcController(dh, ah)
    h = s2.algorithmApg(n)
    report(s2, "fresh: algorithmApg(%d)" % n)
    print("   history finite:", bool(np.isfinite(h).all()), " red zones:", s2.guardCheck(), flush=True)
    s2.close()
