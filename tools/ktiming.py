#!/usr/bin/env python3
"""Phase stamps inside the instrumented kernels (a -DRN_KTIMING build of the library, see kernels.hpp RN_KT).

    python tools/ktiming.py [emulate_world]      (builds rapidnet_amd/librapidnet_hip_kt.so if missing)
Prints, for workgroups 0, 1, 2 and the last one, the time between consecutive stamps in microseconds (100 MHz clock)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "rapidnet_amd", "librapidnet_hip_kt.so")
from rapidnet_amd import build  # noqa: E402

build.build_hip(defines=["RN_KTIMING=1"], out=LIB)      # (re)built when its sources have changed
os.environ["RAPIDNET_LIB"] = LIB
import numpy as np  # noqa: E402
from rapidnet_amd import capi, partition, synth  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 0
problem = synth.make_problem("barcelona493")
tree = problem["tree"]
dh, ah = synth.forecast_at(problem["forecast"], 0)
s_args = dict(precision="f64")
if W > 0:
    cut = partition.default_cut_stage(problem["tree"])
    tree, _ = partition.local_tree(problem["tree"], 0, W, cut)
s = capi.Solver(problem["network"], tree, problem["config"], **s_args)
if W > 0:
    s.commInit(0, 1, capi.comm_unique_id())
    s.setCutStage(cut, partition.cut_children_moments(problem["tree"], cut))
if W > 0 and len(sys.argv) > 2 and sys.argv[2] == "oneshot":     # the one-shot exchange (the rank writes to and reads from its own inbox)
    s.peerInboxConnect([s.peerInboxCreate()])
    s.setExchangeTransport(1)
s.initialiseSmpcController(dh, ah)
s.apgReset()
s.apgIterate(20, history=False)
s.synchronize()
buf = (C.c_ulonglong * 128)()
assert s.lib.rn_debug_ktiming(buf) == 0
t = np.array(list(buf), dtype=np.float64).reshape(8, 16)
base = t[:4, 0].min()
for b, name in enumerate(("wg0", "wg1", "wg2", "last")):
    row = t[b, :6]
    print("%-5s start +%6.2f us | phases (us):" % (name, (row[0] - base) / 100.0), " ".join("%6.2f" % ((row[i + 1] - row[i]) / 100.0) for i in range(5)),
          "| total %6.2f" % ((row[5] - row[0]) / 100.0))
print("phases: exchange-stage step | root step | slab load | product v | product [Lv;BLv]")
for b, name in enumerate(("wg0", "wg1", "wg2", "last")):
    r = t[b]
    print("%-5s wave 0, product v: aux issue %5.2f  mfma loop %5.2f  epilogue %5.2f | product Lv: aux %5.2f  mfma loop %5.2f  epilogue %5.2f   (last pass of the wave)" % (
        name, (r[7] - r[6]) / 100, (r[8] - r[7]) / 100, (r[9] - r[8]) / 100, (r[11] - r[10]) / 100, (r[12] - r[11]) / 100, (r[13] - r[12]) / 100))
s.close()
