"""Does the read bandwidth of a 4 GB buffer depend on which physical memory it got?  rn_measure_hbm (a flat reader over a scratch buffer it
allocates) called again and again while 4 GiB pieces of the device are being held by earlier allocations, so that every call's buffer lands elsewhere.
    python3 tools/placement_read_probe.py [pieces]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from rapidnet_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
p = synth.make_problem("medium")
s = capi.Solver(p["network"], p["tree"], p["config"])
s.initialiseSmpcController(*synth.forecast_at(p["forecast"], 0))
held = []
for i in range(n):
    r = [s.measureHbm(4 << 30, 3)[0] for _ in range(3)]
    print("pieces held %2d: read %s GB/s" % (i, " ".join("%.0f" % x for x in r)), flush=True)
    held.append(torch.empty(4 << 30, dtype=torch.uint8, device="cuda"))
s.close()
