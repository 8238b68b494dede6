#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
echo "== FBE / NAMA fp32 (no spills any more)"; timeout -k 10 200 python tools/time_fbe_nama.py barcelona493 40 f32 2> $O/fbe_f32.err | tee $O/fbe_nama_fp32.jsonl | cut -c1-200
echo "== full GPU suite"; bash tools/collect_r05.sh tests || exit 1
grep -A12 "slowest" $O/tests.log | head -14
echo "== bench + kernel stats"; bash tools/collect_r05.sh bench kstats
