#!/bin/bash
# same-context A/B of the two sweep forms + rocprofv3 kernel stats of the same command: bash tools/ab_sweep_form.sh <tag> [ab_sweep_form.py args]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=$1; shift
python3 tools/ab_sweep_form.py "$@" --profile > gpurun_out/abform_$tag.jsonl 2> gpurun_out/abform_$tag.err || { tail -5 gpurun_out/abform_$tag.err; exit 1; }
cat gpurun_out/abform_$tag.jsonl
rm -rf gpurun_out/abform_ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abform_ks_$tag -o k -- python3 tools/ab_sweep_form.py "$@" --rounds 2 > /dev/null 2> gpurun_out/abform_ks_$tag.err || { tail -5 gpurun_out/abform_ks_$tag.err; exit 1; }
python3 - <<PY
import csv,glob,shutil
f=glob.glob("gpurun_out/abform_ks_$tag/**/k_kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("   %-90s calls %6s avg %9.2f us  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
shutil.copy(f[0], "gpurun_out/abform_$tag.csv")
PY
