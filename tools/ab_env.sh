#!/bin/bash
# interleaved A/B of an environment switch of librapidnet_hip (same binary):
#   bash tools/ab_env.sh <VAR> <rounds> <steps> [extra bench.py args]   -> per-value median ms/step and per-class times
var=$1; rounds=${2:-4}; steps=${3:-200}; shift 3
mkdir -p gpurun_out/abe
for r in $(seq 1 $rounds); do
  for v in 1 0; do
    env $var=$v python bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 40 --dense-only --repeats 2 --other-configs "" "$@" 2>/dev/null | grep '"metric"' > gpurun_out/abe/v${v}_$r.json
  done
done
python - "$var" "$rounds" <<'PY'
import json,sys,statistics as st
var,rounds=sys.argv[1],int(sys.argv[2])
for v in (1,0):
    ds=[json.load(open("gpurun_out/abe/v%d_%d.json"%(v,r))) for r in range(1,rounds+1)]
    ms=[d["timing_spread"]["ms_per_step_median"] for d in ds]
    cls=lambda k:[d["kernel_classes"][k]["avg_us"] for d in ds]
    k0=[k for k in ds[0]["kernel_classes"] if k.startswith("stream") or k.startswith("struct")][0]
    print("%s=%d ms/step median %.4f (min %.4f max %.4f) | %s %.1f rest %.1f dual %.1f"%(var,v,st.median(ms),min(ms),max(ms),k0,st.median(cls(k0)),st.median(cls("recursion+shared_gemms")),st.median(cls("dual_update"))))
PY
