timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s11_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/s11_pytest.log
for v in default old default old; do
  if [ "$v" = default ]; then unset RAPIDNET_LIB; else export RAPIDNET_LIB=$GRAFT_REPO_ROOT/rapidnet_amd/librapidnet_hip_$v.so; fi
  python bench.py --workload wide256 --precision f32 --steps 40 --warmup 5 --no-cpu-baseline --dense-only --profile-steps 20 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide256 $v', round(d['value'],1), 'stream us', round(d['kernel_classes']['stream_gemv']['avg_us'],1), round(d['roofline']['achieved']), 'GB/s')"
done
unset RAPIDNET_LIB
python bench.py > gpurun_out/s11_bench.json 2> gpurun_out/s11_bench.err; tail -c 3000 gpurun_out/s11_bench.json | head -c 1200
