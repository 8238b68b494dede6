cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s12_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/s12_pytest.log
bash tools/collect_traffic.sh > gpurun_out/traffic.log 2>&1; tail -1 gpurun_out/traffic.log | cut -c1-600
python bench.py > gpurun_out/s12_bench.json 2> gpurun_out/s12_bench.err; echo "bench rc=$?"
rm -rf gpurun_out/r01_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_stats -o k -- python3 bench.py --no-cpu-baseline > gpurun_out/s12_bench_prof.json 2>/dev/null; echo "rocprof rc=$?"
python bench.py --workload wide256 --precision f32 --steps 40 --warmup 5 --no-cpu-baseline --dense-only --profile-steps 20 2>/dev/null | grep '"metric"' > gpurun_out/s12_wide256.json
for W in 2 4 8; do python bench.py --emulate-world $W --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '"metric"' > gpurun_out/s12_emu_w$W.json; done
python - <<'PY'
import json
d=json.load(open("gpurun_out/s12_bench.json"))
print("bench", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["structured_mode"]["value"], d["cpu_baseline"]["value"])
for W in (2,4,8):
    e=json.load(open("gpurun_out/s12_emu_w%d.json"%W)); print("emu", W, e["value"], e["ms_per_step"])
w=json.load(open("gpurun_out/s12_wide256.json")); print("wide256", w["value"], w["roofline"]["achieved"])
PY
