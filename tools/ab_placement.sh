#!/bin/bash
# headline with the placement probe of the operator blocks (RAPIDNET_PLACEMENT_TRIES=4, default) and without (=1); fresh processes, same box
for round in 1 2 3; do
for t in 4 1; do
  RAPIDNET_PLACEMENT_TRIES=$t python3 bench.py --no-cpu-baseline --no-traffic --other-configs "" --dense-only --repeats 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $round tries=$t  %.1f it/s  %.4f ms  stream %.1f us = %.3f of peak  placement %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d.get('placement', {}).get('stream_us')))
"
done
done
