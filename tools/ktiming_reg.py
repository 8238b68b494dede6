#!/usr/bin/env python3
"""Phase stamps inside k_gemm_vlv_reg / k_gemm_vlv_reg8 (a -DRN_KTIMING build, see kernels.hpp RN_KT): python tools/ktiming_reg.py [1|2]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "rapidnet_amd", "librapidnet_hip_kt.so")
from rapidnet_amd import build  # noqa: E402

build.build_hip(defines=["RN_KTIMING=1"], out=LIB)
os.environ["RAPIDNET_LIB"] = LIB
os.environ["RAPIDNET_SLAB_REG"] = sys.argv[1] if len(sys.argv) > 1 else "1"      # 1: k_gemm_vlv_reg, 2: k_gemm_vlv_reg8
import numpy as np  # noqa: E402
from rapidnet_amd import capi, synth  # noqa: E402

problem = synth.make_problem("barcelona493")
dh, ah = synth.forecast_at(problem["forecast"], 0)
s = capi.Solver(problem["network"], problem["tree"], problem["config"])
s.initialiseSmpcController(dh, ah)
s.apgReset()
s.apgIterate(20, history=False)
s.synchronize()
buf = (C.c_ulonglong * 128)()
assert s.lib.rn_debug_ktiming(buf) == 0
t = np.array(list(buf), dtype=np.float64).reshape(8, 16)
base = t[:4, 0].min()
names = ["prologue(A tiles, first slab, barrier)"] + ["slab%d %s" % (i, w) for i in range(3) for w in ("V mfma", "V epi+barrier", "LV mfma", "LV epi+stage+barrier")]
for b, name in enumerate(("wg0", "wg1", "wg2", "last")):
    row = t[b, :14]
    print("%-5s start +%6.2f us" % (name, (row[0] - base) / 100.0))
    for i in range(13):
        if row[i + 1] > row[i] > 0:
            print("        %-40s %6.2f us" % (names[i], (row[i + 1] - row[i]) / 100.0))
    last = max(r for r in row if r > 0)
    print("        total %6.2f us" % ((last - row[0]) / 100.0))
s.close()
