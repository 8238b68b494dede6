#!/bin/bash
# rocprofv3 kernel stats of a 1/W shard run (bench.py --emulate-world W [extra flags]):  bash tools/kstats_shard.sh 8 "--one-shot" tag
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-8}; EXTRA=$2; TAG=${3:-shard}
O=gpurun_out/ks_$TAG; rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 bench.py --emulate-world $W $EXTRA --no-cpu-baseline --steps 200 --warmup 20 --profile-steps 0 --repeats 0 --other-configs "" > /dev/null 2>&1 || exit 1
cp $(find $O -name k_kernel_stats.csv | head -1) gpurun_out/kstats_$TAG.csv; rm -rf $O
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kstats_$TAG.csv")))
for r in rows[:12]:
    print("   %-100s calls %6s avg %9.2f us  pct %5s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
