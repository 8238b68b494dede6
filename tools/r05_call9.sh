#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05
timeout -k 10 200 python tools/ktiming_reg.py 2 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" > gpurun_out/r05/ktiming_reg8.txt; head -20 gpurun_out/r05/ktiming_reg8.txt
VARIANTS="0 2" bash tools/ab_slab.sh 2
