timeout -k 10 300 python -m pytest tests/test_gpu_sharding.py -m gpu -x -q > gpurun_out/s7_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/s7_pytest.log
python tools/ktiming.py 8 2>/dev/null | tail -5
python bench.py --emulate-world 8 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('W=8', d['value'], d['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernel_classes'].items()})"
