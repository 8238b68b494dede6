#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
echo "== chain layout probe"; timeout -k 10 120 tools/probes/probe_chain_layout 2>&1 | tee $O/chain_layout_probe.txt
echo "== full GPU suite with durations"; bash tools/collect_r05.sh tests
grep -A90 "slowest" $O/tests.log | head -100
