#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_fused_walk_dual.py -x -q -m gpu 2>&1 | tail -12 || exit 1
echo "== whole tree"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 | tee $O/ab_fuse_whole.txt
echo "== 1/8 shard"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 --emulate-world 8 | tee $O/ab_fuse_shard8.txt
