#!/bin/bash
# runs bench.py once per tuning variant of librapidnet_hip (built by rapidnet_amd.build.build_hip(defines=...))
# usage: tools/sweep_variants.sh <steps> variant1 variant2 ...   ("default" = the shipped library)
steps=${1:-100}; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$PWD/rapidnet_amd/librapidnet_hip_$v.so; fi
  python bench.py --steps $steps --warmup 10 --no-cpu-baseline --profile-steps 30 --dense-only 2>/dev/null | grep '"metric"' > gpurun_out/sweep_$v.json
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open("gpurun_out/sweep_%s.json"%v))
k=d["kernel_classes"]
print("%-10s it/s %7.1f  ms %.4f | stream %.1f us (%.0f GB/s) rest %.1f dual %.1f (%.0f GB/s) book %.1f"%(v,d["value"],d["ms_per_step"],k["stream_gemv"]["avg_us"],d["roofline"]["achieved"],k["recursion+shared_gemms"]["avg_us"],k["dual_update"]["avg_us"],d["roofline"]["dual_update"]["achieved"],k["bookkeeping"]["avg_us"]))
PY
done
