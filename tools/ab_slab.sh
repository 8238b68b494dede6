#!/bin/bash
# round 5, on the GPU box: a slab-product variant (here: the register-resident persistent k_gemm_vlv_reg, RAPIDNET_SLAB_REG) against the L2-fed kernels, same box, interleaved.
# bash tools/ab_reg.sh [rounds]  ->  gpurun_out/r05/ab_reg.txt (+ kernel stats of both under gpurun_out/r05/)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
R=${1:-3}
timeout -k 10 300 python -m pytest tests/test_gpu_slab_kernels.py -x -q -k "register" > $O/ab_reg_tests.log 2>&1 || { tail -30 $O/ab_reg_tests.log; exit 1; }
tail -2 $O/ab_reg_tests.log
: > $O/ab_reg.txt
for r in $(seq 1 $R); do
  for v in ${VARIANTS:-0 1}; do
    RAPIDNET_SLAB_REG=$v timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-traffic --other-configs "" --profile-steps 40 --repeats 3 \
        --no-shard-ceiling --no-quasi-newton > $O/ab_reg_$v.json 2> $O/ab_reg_$v.err || { tail -5 $O/ab_reg_$v.err; exit 1; }
    python - $v $r $O/ab_reg_$v.json >> $O/ab_reg.txt <<'PY'
import json, sys
v, r, f = sys.argv[1:]
d = json.loads(open(f).read().strip().splitlines()[-1])
kc, st = d["kernel_classes"], d["structured_mode"]
print("round %s SLAB_REG=%s dense: %.4f ms/it (median %.4f) helpers %.1f us stream %.1f us | structured: %.4f ms/it = %.0f it/s prep+m2 %.1f us helpers %.1f us" % (
    r, v, d["ms_per_step"], d["timing_spread"]["ms_per_step_median"], kc["recursion+shared_gemms"]["avg_us"], kc["stream_gemv"]["avg_us"],
    st["ms_per_step"], st["value"], st["kernel_classes"]["struct_prep+gemm_m2"]["avg_us"], st["kernel_classes"]["recursion+shared_gemms"]["avg_us"]))
PY
    tail -1 $O/ab_reg.txt
  done
done
for v in ${VARIANTS:-0 1}; do
  (cd /tmp && rm -rf /tmp/r05_ks$v && RAPIDNET_SLAB_REG=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_ks$v -o ks -- python3 "$OLDPWD/bench.py" --steps 200 --warmup 20 \
      --no-cpu-baseline --no-traffic --other-configs "" --profile-steps 0 --repeats 0 --no-shard-ceiling --no-quasi-newton > /dev/null 2> "$OLDPWD/$O/ab_reg_ks$v.err") || { tail -5 $O/ab_reg_ks$v.err; exit 1; }
  cp $(find /tmp/r05_ks$v -name "*kernel_stats.csv" | head -1) $O/ab_reg_kernel_stats_$v.csv
  echo "== kernel stats, RAPIDNET_SLAB_REG=$v" >> $O/ab_reg.txt; head -14 $O/ab_reg_kernel_stats_$v.csv | cut -c1-200 >> $O/ab_reg.txt
done
tail -32 $O/ab_reg.txt
