#!/bin/bash
# interleaved A/B of tuning variants of librapidnet_hip (see tools/sweep_variants.sh): <rounds> <steps> variant...
# prints per-variant median ms/step and median per-class times over the rounds (boxes drift by a few % within a call)
rounds=${1:-5}; steps=${2:-60}; shift 2
mkdir -p gpurun_out/ab
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$PWD/rapidnet_amd/librapidnet_hip_$v.so; fi
    python bench.py --steps $steps --warmup 10 --no-cpu-baseline --profile-steps 30 --dense-only 2>/dev/null | grep '"metric"' > gpurun_out/ab/${v}_$r.json
  done
done
python - "$rounds" "$@" <<'PY'
import json,sys,statistics as st
rounds=int(sys.argv[1]); vs=sys.argv[2:]
for v in vs:
    ds=[json.load(open("gpurun_out/ab/%s_%d.json"%(v,r))) for r in range(1,rounds+1)]
    ms=[d["ms_per_step"] for d in ds]
    cls=lambda k:[d["kernel_classes"][k]["avg_us"] for d in ds]
    print("%-10s ms/step median %.4f (min %.4f max %.4f) | stream %.1f rest %.1f dual %.1f"%(v,st.median(ms),min(ms),max(ms),st.median(cls("stream_gemv")),st.median(cls("recursion+shared_gemms")),st.median(cls("dual_update"))))
PY
