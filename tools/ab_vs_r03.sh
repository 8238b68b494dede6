#!/bin/bash
# same-box comparison of this tree with the round-3 tree (`git archive e0a0e60 bench.py rapidnet_amd include oracle` unpacked at _r03x/, its library built there): interleaved runs of
# both trees' own bench.py on the whole tree and on a 1/8 shard
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for tree in r03 now; do
    D=$R; [ $tree = r03 ] && D=$R/_r03x
    for extra in "" "--emulate-world 8"; do
      ( cd $D && python3 bench.py $extra --steps 100 --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 40 --dense-only --repeats 2 --other-configs "" 2>/dev/null ) | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r %-4s %-18s ms/it %.4f median %.4f' % ('$tree', '$extra' or 'whole tree', d['ms_per_step'], d['timing_spread']['ms_per_step_median']), {k: round(v['avg_us'],1) for k,v in d['kernel_classes'].items()})"
    done
  done
done
