#!/bin/bash
# round 5, on the GPU box: SQ counters of the kernels that use the matrix cores (review item 2: "rocprof evidence of matrix-unit use on the
# kernels that ship").  Two rocprofv3 --pmc passes (counters only, with --kernel-trace for the durations; the program directly behind --)
# of ONE bench.py run that executes the dense headline (k_gemm_vlv*), the structured mode (k_gemm_prep_m2*, k_gemm_vlv*) and the
# quasi-Newton loops (k_value_mfma).   bash tools/collect_sq.sh [tag]  ->  gpurun_out/r05/sq_counters[_tag].json
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r05; mkdir -p $O
TAG=${1:+_$1}
export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-traffic --other-configs= --profile-steps 0 --repeats 0 --no-shard-ceiling"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/r05_sq$i
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/r05_sq$i -o sq -- python3 $O/../../bench.py $ARGS > /dev/null 2> $O/sq_pass$i.err) || { tail -8 $O/sq_pass$i.err; exit 1; }
done
python3 tools/summarize_sq.py /tmp/r05_sq1 /tmp/r05_sq2 > $O/sq_counters$TAG.json && python3 - $O/sq_counters$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    print("%-60s %8.1f us  mfma busy %5.1f %%  (%s launches)" % (k[:60], v["duration_us_median"], 100 * (v.get("mfma_busy_fraction") or 0), v["launches"]))
PY
