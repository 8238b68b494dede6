"""register / LDS / scratch usage of the library's kernels: python tools/kres.py [substring ...] (compiles with -Rpass-analysis)"""
import re
import subprocess
import sys

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from rapidnet_amd import build

build.build_hip(force=True, verbose=True, defines=[d for d in sys.argv[1:] if d.startswith("RN_")], out="/tmp/kres.so")
out = build.build_hip.remarks
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
    for key in ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:", "LDS Size [bytes/block]:", "SGPRs:"):
        if key in line and cur:
            rows[cur][key.split()[0].rstrip(":")] = line.split(key)[-1].strip().split()[0]
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
want = [a for a in sys.argv[1:] if not a.startswith("RN_")]
for n, d in zip(names, dem):
    if not want or any(w in d for w in want):
        print("%-120s %s" % (d[:120], rows[n]))
