# timing-only ablation of the fused dual update without its w_next store (build: rapidnet_amd.build.build_hip(defines=("RN_DUAL_ABL=8",), out=.../librapidnet_hip_now.so)); results are WRONG in that build
set -e
for v in default now default now; do
  if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$PWD/rapidnet_amd/librapidnet_hip_$v.so; fi
  for w in barcelona493 wide4096; do
    python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 40 --dense-only --repeats 0 --other-configs "" --no-traffic 2>/dev/null | grep '"metric"' > gpurun_out/abl_${v}_$w.json
    python - $v $w <<'PY'
import json,sys
v,w=sys.argv[1:]
d=json.load(open("gpurun_out/abl_%s_%s.json"%(v,w)))
k=d["kernel_classes"]; du=d["roofline"]["dual_update"]
print("%-8s %-13s ms %.4f | stream %.1f us  dual %.2f us (5-stream bytes %.1f MB)"%(v,w,d["ms_per_step"],k["stream_gemv"]["avg_us"],du["avg_launch_us"],du["algorithmic_bytes_per_launch"]/1e6), flush=True)
PY
  done
done
