#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
echo "== comm diag"; timeout -k 10 150 python tools/comm_diag.py 2>&1 | tee $O/comm_diag.log | tail -15
echo "== new tests"; timeout -k 10 300 python -m pytest tests/test_gpu_slab_kernels.py -x -q -m gpu -k "lds" 2>&1 | tee $O/call2_tests.log | tail -15 || exit 1
echo "== band overlap probe"; timeout -k 10 300 python tools/probe_band_overlap.py 300 2> $O/band_probe.err | tee $O/band_probe.json || { tail -5 $O/band_probe.err; }
echo "== LDS A/B"; bash tools/ab_lds.sh 2 || exit 1
echo "== N=2 rehearsal (oversubscribed)"; timeout -k 10 420 python bench.py --gpus 2 --steps 10 --warmup 2 --profile-steps 10 --repeats 1 --cpu-iterations 3 --allow-oversubscribe --no-traffic --time-budget 300 > $O/n2.json 2> $O/n2.err; echo "rc=$?"; grep "supervisor" $O/n2.err | tail -20; tail -c 400 $O/n2.json
echo "== bench"; bash tools/collect_r05.sh bench
