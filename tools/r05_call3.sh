#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== SQ counters, L2-fed slab kernels"; RAPIDNET_SLAB_LDS=0 bash tools/collect_sq.sh l2 || exit 1
echo "== SQ counters, LDS-staged slab kernels"; RAPIDNET_SLAB_LDS=1 bash tools/collect_sq.sh lds || exit 1
