#!/bin/bash
# same-box A/B of an environment switch on the whole tree and on a 1/8 shard:  bash tools/ab_env3.sh VAR [on] [off]
VAR=$1; ON=${2:-1}; OFF=${3:-0}
for r in 1 2 3; do
  for v in $ON $OFF; do
    for extra in "" "--emulate-world 8"; do
      env $VAR=$v python3 bench.py $extra --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 40 --dense-only --repeats 2 --other-configs "" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r $VAR=$v %-18s ms/it %.4f' % ('$extra' or 'whole tree', d['ms_per_step']), {k: round(v['avg_us'],1) for k,v in d['kernel_classes'].items()})"
    done
  done
done
