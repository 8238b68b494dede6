#!/bin/bash
# HISTORICAL: the command of record behind profiles/r05_*; it names switches and test files of that round (round 6 turned the tuning
# environment variables into rn_debug_set_knob choices and removed the forms that lost their A/B) -- tools/collect_r06.sh is the current one.
# round 5, on the GPU box (gpurun -- bash tools/collect_r05.sh <step> ...): every step writes under gpurun_out/r05/
# steps: tests bench kstats n2 guard sq probes   (A/B runs of the slab-product variants: tools/ab_slab.sh; in-kernel stamps: tools/ktiming_reg.py)
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
    n2)      # the two-rank rehearsal on ONE GPU (RCCL refuses the duplicate device: agreed fallback exchange) under the budgeted supervisors
             timeout -k 10 420 python bench.py --gpus 2 --steps 10 --warmup 2 --profile-steps 10 --repeats 1 --cpu-iterations 3 --allow-oversubscribe --no-traffic --time-budget 300 > $O/n2.json 2> $O/n2.err; echo "rc=$?"
             grep "supervisor" $O/n2.err | tail -20; tail -c 300 $O/n2.json ;;
    guard)   # whole test files under the buffer guard (RAPIDNET_GUARD=1: red zones + NaN poison) in one process, the round's kernels among them
             timeout -k 10 1000 python tools/guard_suite.py -m gpu tests/test_gpu_slab_kernels.py tests/test_gpu_sharded_batched.py tests/test_gpu_fbe_nama.py tests/test_gpu_lazy_dual.py \
                 tests/test_gpu_random_shapes.py tests/test_gpu_parity.py tests/test_golden_synthetic.py tests/test_gpu_closed_loop.py tests/test_reference_barcelona30.py \
                 tests/test_nonuniform_trees.py tests/test_gpu_fullsize.py tests/test_gpu_oneshot.py tests/test_gpu_comm_timeout.py tests/test_gpu_device_pointer.py tests/test_gpu_fused_walk_dual.py tests/test_gpu_chain_fused.py > $O/guard_suite.log 2>&1; rc=$?
             tail -4 $O/guard_suite.log; [ $rc -eq 0 ] || exit $rc ;;
    sq)      RAPIDNET_SLAB_LDS=0 bash tools/collect_sq.sh l2 || exit 1 ;;
    probes)  timeout -k 10 120 tools/probes/probe_chain_layout 2>&1 | tee $O/chain_layout_probe.txt
             timeout -k 10 300 python tools/probe_band_overlap.py 300 2> $O/band_probe.err | tee $O/band_probe.json
             timeout -k 10 150 python tools/comm_diag.py 2>&1 | tee $O/comm_diag.log | tail -12 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
