"""One global-FBE (or NAMA) run for a kernel-level profile: rocprofv3 --kernel-trace --stats -- python3 tools/profile_fbe.py [workload] [iterations] [structured 0|1] [fbe|nama]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from rapidnet_amd import capi, synth

name = sys.argv[1] if len(sys.argv) > 1 else "barcelona493"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
structured = len(sys.argv) > 3 and sys.argv[3] == "1"
p = synth.make_problem(name)
dh, ah = synth.forecast_at(p["forecast"], 0)
s = capi.Solver(p["network"], p["tree"], p["config"], structured=structured)
s.initialiseSmpcController(dh, ah)
nama = len(sys.argv) > 4 and sys.argv[4] == "nama"
s.setAlgorithm("namaAlgorithm" if nama else "globalFbeAlgorithm", 5)
run = s.algorithmNama if nama else s.algorithmGlobalFbe
run(3)
t = time.perf_counter()
h, v, tau = run(iters)
print("ms per iteration %.4f, tau %s" % (1e3 * (time.perf_counter() - t) / iters, [float(x) for x in tau[:12]]))
s.close()
