#!/bin/bash
# Per-rank timing of the sharded code path on ONE GPU: rank 0's shard of a W-rank partition, one-rank RCCL communicator
# (bench.py --emulate-world W).  The collective's real latency is missing; everything else is what a rank executes.
#   bash tools/emulate_shards.sh "1 2 4 8"   -> gpurun_out/emu_w<W>.json + rocprof kernel stats gpurun_out/emu_w<W>/
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for W in ${1:-1 2 4 8}; do
  python3 bench.py --emulate-world $W --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/emu_w$W.json 2> gpurun_out/emu_w$W.err || exit 1
  rm -rf gpurun_out/emu_w$W
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/emu_w$W -o k -- python3 bench.py --emulate-world $W --no-cpu-baseline --steps 200 --warmup 20 --profile-steps 0 > /dev/null 2>&1 || exit 1
  python3 - <<PY
import json,csv,glob
d=json.loads([l for l in open("gpurun_out/emu_w$W.json") if l.startswith("{")][-1])
print("W=$W nodes", d["config"].get("local_nodes"), "it/s %.1f  ms/it %.4f" % (d["value"], d["ms_per_step"]), {k: round(v["avg_us"],1) for k,v in d["kernel_classes"].items()})
f=glob.glob("gpurun_out/emu_w$W/**/k_kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print("   %-90s calls %6s avg %9.2f us  pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
