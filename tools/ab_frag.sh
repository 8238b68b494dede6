#!/bin/bash
# A/B of the knob slab_frag (A operands of the slab products from the fragment-ordered operator copies) by kernel time under rocprofv3:
#   bash tools/ab_frag.sh   -> whole tree dense / structured, 31-scenario tree, 1/8 shard, wide fp32 network: the slab kernels' average durations, FRAG = 0 | 1
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # tag, bench args...
  tag=$1; shift
  for f in 0 1; do
    rm -rf gpurun_out/abfrag_${tag}_$f
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abfrag_${tag}_$f -o k -- python3 bench.py --no-cpu-baseline --no-traffic --profile-steps 0 --repeats 0 --other-configs "" --knob slab_frag=$f "$@" > gpurun_out/abfrag_${tag}_$f.json 2> gpurun_out/abfrag_${tag}_$f.err || { tail -3 gpurun_out/abfrag_${tag}_$f.err; exit 1; }
  done
  python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for f in (0, 1):
    rows = list(csv.DictReader(open(glob.glob("gpurun_out/abfrag_%s_%d/**/k_kernel_stats.csv" % (tag, f), recursive=True)[0])))
    for r in rows:
        if "k_gemm" in r["Name"]:
            out.setdefault(r["Name"].split("(")[0].replace("void rn::", ""), {})[f] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    d = json.loads([l for l in open("gpurun_out/abfrag_%s_%d.json" % (tag, f)) if l.startswith("{")][-1])
    out.setdefault("ms_per_step", {})[f] = (d["ms_per_step"], d["steps"])
print(tag)
for k, v in out.items():
    if 0 in v and 1 in v:
        print("   %-50s FRAG=0 %9.3f | FRAG=1 %9.3f   (%d calls)" % (k[:50], v[0][0], v[1][0], v[1][1]))
PY
}
run whole_dense --steps 200 --warmup 20 --dense-only
run whole_structured --steps 200 --warmup 20 --structured
run b31 --workload barcelona31 --steps 400 --warmup 40 --dense-only
run shard8 --emulate-world 8 --steps 200 --warmup 20
run wide4096_f32 --workload wide4096 --precision f32 --steps 10 --warmup 3 --dense-only
