#!/bin/bash
# the streaming kernel of a 1/8 shard against the number of blocks dealt to two workgroups at the end of the launch (same box)
for round in 1 2; do
for r in 0 102 144 186; do
  if [ $r = 0 ]; then export RAPIDNET_STREAM_SPLIT_R=; unset RAPIDNET_STREAM_SPLIT_R; else export RAPIDNET_STREAM_SPLIT_R=$r; fi
  python3 bench.py --emulate-world 8 --no-cpu-baseline --steps 300 --warmup 20 --repeats 2 --other-configs "" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_classes']
print('round $round split r=%-4s ms/it %.4f  stream %.2f us  rest %.2f us' % ('$r' if '$r' != '0' else 'dflt', d['ms_per_step'], k['stream_gemv']['avg_us'], k['recursion+shared_gemms']['avg_us'] + k['dual_update']['avg_us']))
"
done
done
