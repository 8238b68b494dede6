#!/bin/bash
# HBM traffic of the solver's kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 passes (with --kernel-trace only), bench.py itself after "--".  Run on the GPU box:
#   bash tools/collect_traffic.sh            -> gpurun_out/traffic.json  (copy to profiles/traffic.json)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS="bench.py --steps 20 --warmup 2 --no-cpu-baseline --profile-steps 0 --dense-only"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 $ARGS > /dev/null 2>&1
python3 tools/summarize_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/traffic.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/traffic.json"))
print({k: v for k, v in d.items() if k != "raw"})
PY
