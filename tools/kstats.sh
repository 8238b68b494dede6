#!/bin/bash
# rocprofv3 kernel stats of one bench.py run: bash tools/kstats.sh <tag> [bench.py args]  -> gpurun_out/ks_<tag>/ + a printed summary
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
rm -rf gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -o k -- python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 --profile-steps 0 --dense-only --repeats 0 --other-configs "" "$@" > gpurun_out/ks_$tag.json 2> gpurun_out/ks_$tag.err || { tail -5 gpurun_out/ks_$tag.err; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$tag/**/k_kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:12]:
    print("   %-70s calls %6s avg %9.2f us  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
import shutil; shutil.copy(f[0], "gpurun_out/ks_$tag.csv")
PY
