#!/bin/bash
# fused forward walk + dual update: workgroups per chain (rn_debug_set_knob fuse_split), interleaved on one box
#   bash tools/ab_split.sh "<values>" <rounds> <steps> [bench args]     value 0 = round 6's first rule (one workgroup per chain, only where the chains fill the chip), -1 = the library's choice
vals=$1; rounds=${2:-3}; steps=${3:-300}; shift 3
mkdir -p gpurun_out/abs
for r in $(seq 1 $rounds); do
  for v in $vals; do
    python bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 0 --dense-only --repeats 2 --other-configs "" --no-shard-ceiling --no-quasi-newton --knob fuse_split=$v "$@" 2>/dev/null | grep '"metric"' > gpurun_out/abs/v${v}_$r.json
  done
done
python - "$rounds" $vals <<'PY'
import json,sys,statistics as st
rounds=int(sys.argv[1])
for v in sys.argv[2:]:
    ds=[json.load(open("gpurun_out/abs/v%s_%d.json"%(v,r))) for r in range(1,rounds+1)]
    ms=[d["timing_spread"]["ms_per_step_median"] for d in ds]
    print("fuse_split=%-3s ms/step median %.4f (min %.4f max %.4f)"%(v,st.median(ms),min(ms),max(ms)))
PY
