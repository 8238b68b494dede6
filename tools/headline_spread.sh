#!/bin/bash
# the headline of N fresh processes on one box (what a context's placement in physical memory is worth): bash tools/headline_spread.sh [N]
N=${1:-6}
for i in $(seq 1 $N); do
  python3 bench.py --no-cpu-baseline --no-traffic --other-configs "" --dense-only --repeats 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('process $i: %.1f it/s  %.4f ms  k_stream_gemv %.1f us = %.3f of peak  read probe %.0f GB/s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('measured_read_ceiling') or 0))
"
done
