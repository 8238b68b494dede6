"""diagnostic: HIP vs oracle on `medium` with both soft-constraint thresholds tripping from the first iteration"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle.oracle import Oracle
from rapidnet_amd import capi, synth

PAIRS = [(capi.BUF_X, "x"), (capi.BUF_U, "u"), (capi.BUF_XI, "xi"), (capi.BUF_PSI, "psi"), (capi.BUF_ACC_XI, "accXi"), (capi.BUF_ACC_PSI, "accPsi"),
         (capi.BUF_UPD_XI, "updXi"), (capi.BUF_UPD_PSI, "updPsi"), (capi.BUF_PRIMAL_PSI, "primalPsi"), (capi.BUF_DUAL_XI, "dualXi"), (capi.BUF_DUAL_PSI, "dualPsi"),
         (capi.BUF_RES_XI, "resXi"), (capi.BUF_RES_PSI, "resPsi")]
px, pxs = float(sys.argv[1]), float(sys.argv[2])
p = synth.make_problem("medium", penalty_x=px, penalty_xs=pxs)
dh, ah = synth.forecast_at(p["forecast"], 0)
for mode in (0, 1):
    for n in (1, 2, 3, 5, 16, 30):
        o = Oracle(p["network"], p["tree"], p["config"]); o.initialise(dh, ah)
        s = capi.Solver(p["network"], p["tree"], p["config"]); s.initialiseSmpcController(dh, ah)
        s.setExchangeMode(mode)
        hs, ho = s.algorithmApg(n), o.apg(n)
        bad = {}
        for bid, nm in PAIRS:
            a, b = s.get(bid), o.get(nm)
            bad[nm] = (float(np.abs(a - b).max()), float(np.abs(b).max()))
        worst = {k: "%.2e/%.2e" % v for k, v in bad.items() if v[0] > 1e-9 * max(v[1], 1e-300)}
        print("mode", mode, "n", n, "hist", float(np.abs(hs - ho).max() / np.abs(ho).max()), s.counters(), worst, flush=True)
        s.close()
