#!/bin/bash
# round 5, on the GPU box (gpurun -- bash tools/collect_r05.sh <step> ...): every step writes under gpurun_out/r05/
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    tests)   timeout -k 10 840 python -m pytest tests -x -q -m gpu --durations=80 > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc ;;
    bench)   timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 600 $O/bench.json; [ $rc -eq 0 ] || { tail -20 $O/bench.err; exit $rc; } ;;
    kstats)  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05_ks -o ks -- python3 "$OLDPWD/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --dense-only --other-configs "" --profile-steps 0 --repeats 0 > /dev/null 2> "$OLDPWD/$O/kstats.err") || { tail -5 $O/kstats.err; exit 1; }
             cp $(find /tmp/r05_ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; head -12 $O/kernel_stats.csv ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
