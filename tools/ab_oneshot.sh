#!/bin/bash
# per-rank time of a 1/W shard (bench.py --emulate-world W) with the one-rank RCCL all-reduce vs the one-shot exchange
# (the rank writes to / reads from its own inbox): interleaved rounds on one box
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-8}; R=${2:-3}
for r in $(seq 1 $R); do
  for mode in rccl oneshot; do
    flag=""; [ $mode = oneshot ] && flag="--one-shot"
    python3 bench.py --emulate-world $W $flag --no-cpu-baseline --steps 200 --warmup 20 --other-configs "" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('W=$W round $r %-8s ms/it %.4f (median %.4f)  classes' % ('$mode', d['ms_per_step'], d['timing_spread']['ms_per_step_median']), {k: round(v['avg_us'],1) for k,v in d['kernel_classes'].items()})"
  done
done
