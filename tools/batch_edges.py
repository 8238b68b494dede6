"""Timeline around the batch boundaries of a bench.py run (rocprofv3 --kernel-trace csv): which launches, and which gaps, sit between the last
iteration of one optimistic batch and the first of the next.   python tools/batch_edges.py <kernel_trace.csv> [n_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
# the last batch_open of the trace: print from 6 launches before it
idx = [i for i, r in enumerate(rows) if "k_batch_open" in r["Kernel_Name"]]
if not idx:
    sys.exit("no k_batch_open in the trace")
for which in idx[-2:-1] or idx[-1:]:
    i0 = max(0, which - 8)
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = None
    for r in rows[i0:i0 + n]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("%9.1f us  gap %7.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
        prev_end = e
