for i in 1 2; do
for m in "--no-traffic" ""; do
python bench.py --no-cpu-baseline --other-configs "" --dense-only $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode[$m]', round(d['value'],1), 'stream', round(d['roofline']['avg_launch_us'],1), 'spread', round(d['timing_spread']['ms_per_step_min'],4), round(d['timing_spread']['ms_per_step_max'],4), 'measured', d['roofline']['traffic_source']['measured_in_this_run'])"
done; done
