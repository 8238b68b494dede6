#!/bin/bash
# how often is the contract's timed region (region 1) slower than the regions behind it?  bash tools/first_region.sh <runs> [bench args]
n=${1:-6}; shift
mkdir -p gpurun_out/fr
for i in $(seq 1 $n); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs "" --no-shard-ceiling --no-quasi-newton "$@" 2>/dev/null | grep '"metric"' > gpurun_out/fr/r$i.json
  python - gpurun_out/fr/r$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); t=d["timing_spread"]
print("region1 %.4f ms  median %.4f  min %.4f  max %.4f | structured region1 %s" % (d["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_min"], t["ms_per_step_max"],
      ("%.4f median %.4f" % (d["structured_mode"]["ms_per_step"], d["structured_mode"]["timing_spread"]["ms_per_step_median"])) if "structured_mode" in d and "ms_per_step" in d["structured_mode"] else "-"))
PY
done
