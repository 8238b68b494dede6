#!/usr/bin/env python3
"""Phase stamps inside k_chain_sweep_reg / k_crown_small_reg (a -DRN_KTIMING build, chain_kernels.hpp CF_KT / CR_KT): python tools/ktiming_cf.py [config]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "rapidnet_amd", "librapidnet_hip_kt.so")
from rapidnet_amd import build  # noqa: E402

build.build_hip(defines=["RN_KTIMING=1"], out=LIB)
os.environ["RAPIDNET_LIB"] = LIB
import numpy as np  # noqa: E402
from rapidnet_amd import capi, synth  # noqa: E402

problem = synth.make_problem(sys.argv[1] if len(sys.argv) > 1 else "barcelona493")
dh, ah = synth.forecast_at(problem["forecast"], 0)
s = capi.Solver(problem["network"], problem["tree"], problem["config"])
assert s.setSweepForm(1) == 1
s.initialiseSmpcController(dh, ah)
s.apgReset()
for _ in range(3):
    s.apgIterate(20, history=False)
    s.synchronize()
    buf = (C.c_ulonglong * 128)()
    assert s.lib.rn_debug_ktiming(buf) == 0
    t = np.array(list(buf), dtype=np.float64).reshape(8, 16)
    cn = ["A loads issued", "phase A sums", "barrier", "v mfma", "aL issue + v epilogue", "barrier", "Lv mfma + epilogue", "D loads + barrier", "D sums + stores"]
    base = min(x for x in t[:4, 0] if x > 0)
    for b, name in enumerate(("wg0", "wg1", "wg2", "last")):
        row = t[b, :10]
        if row[0] <= 0:
            continue
        print("chain %-5s start +%6.2f us: " % (name, (row[0] - base) / 100.0) + "  ".join("%s %.2f" % (cn[i], (row[i + 1] - row[i]) / 100.0) for i in range(9)) + "  | total %.2f us" % ((row[9] - row[0]) / 100.0))
    rn = ["A/aux loads issued", "up: cut stage", "up: upper stages", "zero pad", "barrier", "v mfma", "aL + v epi + barrier", "Lv mfma + epi", "down preloads", "barrier", "down pass", "offset rows"]
    row = t[4, :12]
    print("crown: " + "  ".join("%s %.2f" % (rn[i], (row[i + 1] - row[i]) / 100.0) for i in range(11)) + "  | total %.2f us" % ((row[11] - row[0]) / 100.0))
    print()
s.close()
