#!/bin/bash
# does running under rocprofv3 --kernel-trace change the kernels' durations?  plain / traced / plain / traced on one box
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/abr
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); k=d['kernel_classes']
print('$2', round(d['value'],1), 'ms', round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in k.items()})"; }
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --other-configs "" --dense-only --no-traffic > gpurun_out/abr/p$i.json 2>/dev/null; show gpurun_out/abr/p$i.json plain
  rm -rf gpurun_out/abr/t$i; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abr/t$i -o k -- python3 bench.py --no-cpu-baseline --other-configs "" --dense-only --no-traffic > gpurun_out/abr/t$i.json 2>/dev/null; show gpurun_out/abr/t$i.json traced
done
