#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VARIANTS="${VARIANTS:-0 2}" bash tools/ab_slab.sh ${ROUNDS:-3}
