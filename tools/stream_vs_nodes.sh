#!/bin/bash
# k_stream_gemv time against the number of node blocks: rank 0's shard of a W-rank partition for several W (bench.py --emulate-world)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for W in ${1:-6 7 8 9 10 11 12 13 14 16}; do
  python3 bench.py --emulate-world $W --no-cpu-baseline --steps 100 --warmup 20 --repeats 2 --other-configs "" 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
n=d['local_nodes']; us=d['kernel_classes']['stream_gemv']['avg_us']
print('W=%2d nodes %5d rounds %.2f  k_stream_gemv %7.2f us  ns/node %.2f  ms/it %.4f' % ($W, n, n/256.0, us, 1e3*us/n, d['ms_per_step']))"
done
