#!/bin/bash
# interleaved same-box A/B of two BUILDS of the library: the tree's own and rapidnet_amd/librapidnet_hip_<variant>.so (e.g. the previous commit's,
# built with `python -c "from rapidnet_amd import build; build.build_hip(out='rapidnet_amd/librapidnet_hip_<variant>.so')"` in a checkout of it)
#   bash tools/ab_lib.sh <variant> <rounds> <steps> [bench args]
var=$1; rounds=${2:-3}; steps=${3:-300}; shift 3
mkdir -p gpurun_out/abl
for r in $(seq 1 $rounds); do
  for v in default $var; do
    if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$PWD/rapidnet_amd/librapidnet_hip_$v.so; fi
    python bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 0 --dense-only --repeats 2 --other-configs "" --no-shard-ceiling --no-quasi-newton "$@" 2>/dev/null | grep '"metric"' > gpurun_out/abl/${v}_$r.json
  done
done
unset RAPIDNET_LIB
python - "$rounds" default $var <<'PY'
import json,sys,statistics as st
rounds=int(sys.argv[1])
for v in sys.argv[2:]:
    ds=[json.load(open("gpurun_out/abl/%s_%d.json"%(v,r))) for r in range(1,rounds+1)]
    ms=[d["timing_spread"]["ms_per_step_median"] for d in ds]
    print("%-10s ms/step median %.4f (min %.4f max %.4f)"%(v,st.median(ms),min(ms),max(ms)))
PY
