#!/bin/bash
# interleaved A/B of a rn_debug_set_knob choice (same binary):  bash tools/ab_knob.sh <knob> <valueA> <valueB> <rounds> <steps> [extra bench.py args]
knob=$1; va=$2; vb=$3; rounds=${4:-3}; steps=${5:-300}; shift 5
mkdir -p gpurun_out/abk
for r in $(seq 1 $rounds); do
  for v in $va $vb; do
    python bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 40 --dense-only --repeats 2 --other-configs "" --knob $knob=$v "$@" 2>/dev/null | grep '"metric"' > gpurun_out/abk/v${v}_$r.json
  done
done
python - "$knob" "$rounds" "$va" "$vb" <<'PY'
import json,sys,statistics as st
knob,rounds=sys.argv[1],int(sys.argv[2])
for v in sys.argv[3:5]:
    ds=[json.load(open("gpurun_out/abk/v%s_%d.json"%(v,r))) for r in range(1,rounds+1)]
    ms=[d["timing_spread"]["ms_per_step_median"] for d in ds]
    cls=lambda k:[d["kernel_classes"][k]["avg_us"] for d in ds]
    k0=[k for k in ds[0]["kernel_classes"] if k.startswith("stream") or k.startswith("struct")][0]
    print("%s=%s ms/step median %.4f (min %.4f max %.4f) | %s %.1f rest %.1f dual %.1f"%(knob,v,st.median(ms),min(ms),max(ms),k0,st.median(cls(k0)),st.median(cls("recursion+shared_gemms")),st.median(cls("dual_update"))))
PY
