#!/bin/bash
# like sweep_variants.sh but for the structured mode (where the helper kernels dominate)
steps=${1:-200}; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$PWD/rapidnet_amd/librapidnet_hip_$v.so; fi
  python bench.py --steps $steps --warmup 10 --no-cpu-baseline --profile-steps 30 --structured 2>/dev/null | grep '"metric"' > gpurun_out/sweeps_$v.json
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open("gpurun_out/sweeps_%s.json"%v))
k=d["kernel_classes"]
print("%-10s it/s %7.1f us %.1f | %s"%(v,d["value"],1e3*d["ms_per_step"]," ".join("%s %.1f"%(n[:12],c["avg_us"]) for n,c in k.items())))
PY
done
