"""Medians per launch of the SQ counters of tools/collect_sq.sh for the kernels that issue MFMAs -> profiles/r05_sq_counters.json.

mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock): the counter sums the cycles in which a
SIMD's matrix pipe was busy over all SIMDs of the chip (MI355X_MICROARCH.md: it counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* count
quad-cycles); the clock is taken as SQ_BUSY_CYCLES' own rate when that counter is per-chip-cycle consistent, else 2.4 GHz -- both
are printed.  mfma_floor_us = the busy cycles spread evenly over the 1 024 SIMDs at 2.4 GHz: what the launch would take if the
matrix pipes were its only limit."""
import csv
import glob
import json
import os
import statistics
import sys

WANT = ("k_gemm_vlv", "k_gemm_prep_m2", "k_value_mfma", "k_gemm_slab", "k_gemm_shared")


def short(name):
    n = name.replace("void rn::", "").split("(")[0]
    return n


def load(directory):
    counters, durs = {}, {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if not k.startswith(WANT):
                continue
            d = counters.setdefault(k, {}).setdefault(row["Counter_Name"], {})
            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if k.startswith(WANT):
                durs.setdefault(k, []).append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
    return counters, durs


out = {"source": "tools/collect_sq.sh: two `rocprofv3 --kernel-trace --pmc ...` passes (counters only) of one bench.py run (dense headline, structured mode, "
                 "FBE / NAMA loops; barcelona493 fp64); medians per launch.  Durations are those of the counter-collecting runs (launches are serialised "
                 "and somewhat longer than in a plain run).",
       "clock_GHz_assumed": 2.4, "simds": 1024, "kernels": {}}
for directory in sys.argv[1:]:
    counters, durs = load(directory)
    for k, cs in counters.items():
        e = out["kernels"].setdefault(k, {})
        for c, vals in cs.items():
            e[c] = statistics.median(vals.values())
        if k in durs and "duration_us_median" not in e:
            e["duration_us_median"] = statistics.median(durs[k])
            e["launches"] = len(durs[k])
for k, e in out["kernels"].items():
    busy, dur = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("duration_us_median")
    if busy is not None and dur:
        e["mfma_floor_us"] = busy / 1024.0 / 2400.0
        e["mfma_busy_fraction"] = e["mfma_floor_us"] / dur
    if e.get("SQ_WAVE_CYCLES") and e.get("SQ_WAIT_ANY") is not None:
        e["wait_fraction_of_wave_cycles"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1, sort_keys=True))
