#!/bin/bash
# does the PMC traffic measurement (two rocprofv3 child runs BEFORE the process touches the GPU) change the headline of the run that follows? same box, alternating
for r in 1 2 3; do
  for t in "" "--no-traffic"; do
    python3 bench.py --no-cpu-baseline --other-configs "" --dense-only --repeats 2 $t 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r %-12s %.1f it/s  %.4f ms  k_stream_gemv %.1f us' % ('$t' or 'with traffic', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us']))
"
  done
done
