#!/bin/bash
# round 6, on the GPU box (gpurun -- bash tools/collect_r06.sh <step> ...): every step writes under gpurun_out/r06/
# steps: tests bench kstats kstats_struct kstats_b31 kstats_shard8 n2 guard ab_fuse
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
ks() {   # ks <tag> <bench args...>: rocprofv3 kernel stats of one bench.py run -> $O/kernel_stats_<tag>.csv
  local tag=$1; shift
  rm -rf /tmp/r06_ks_$tag
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06_ks_$tag -o ks -- python3 "$OLDPWD/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --other-configs "" --profile-steps 0 --repeats 0 --no-shard-ceiling --no-quasi-newton "$@" > "$OLDPWD/$O/ks_$tag.json" 2> "$OLDPWD/$O/ks_$tag.err") || { tail -5 $O/ks_$tag.err; return 1; }
  cp $(find /tmp/r06_ks_$tag -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$tag.csv
  echo "== $tag"; head -11 $O/kernel_stats_$tag.csv | cut -d, -f1-4,6 | cut -c1-150
}
for step in "$@"; do
  case $step in
    tests)   timeout -k 10 840 python -m pytest tests -x -q -m gpu --durations=80 > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc ;;
    bench)   timeout -k 10 560 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 400 $O/bench.json; [ $rc -eq 0 ] || { tail -20 $O/bench.err; exit $rc; } ;;
    kstats)  ks dense --dense-only || exit 1 ;;
    kstats_struct) ks structured --structured || exit 1 ;;
    kstats_b31) ks b31 --workload barcelona31 --dense-only || exit 1; ks b31_structured --workload barcelona31 --structured || exit 1 ;;
    kstats_shard8) ks shard8 --emulate-world 8 --dense-only || exit 1 ;;
    n2)      # the two-rank rehearsal on ONE GPU (RCCL refuses the duplicate device: agreed fallback exchange) under the budgeted supervisors
             timeout -k 10 420 python bench.py --gpus 2 --steps 10 --warmup 2 --profile-steps 10 --repeats 1 --cpu-iterations 3 --allow-oversubscribe --no-traffic --other-configs "" --time-budget 300 > $O/n2.json 2> $O/n2.err; echo "rc=$?"
             grep "supervisor" $O/n2.err | tail -20; tail -c 600 $O/n2.json ;;
    guard)   # whole test files under the buffer guard (RAPIDNET_GUARD=1: red zones + NaN poison) in one process
             timeout -k 10 1000 python tools/guard_suite.py -m gpu tests/test_gpu_slab_kernels.py tests/test_gpu_sharded_batched.py tests/test_gpu_fbe_nama.py \
                 tests/test_gpu_random_shapes.py tests/test_gpu_parity.py tests/test_golden_synthetic.py tests/test_gpu_closed_loop.py tests/test_reference_barcelona30.py \
                 tests/test_nonuniform_trees.py tests/test_gpu_fullsize.py tests/test_gpu_oneshot.py tests/test_gpu_comm_timeout.py tests/test_gpu_device_pointer.py tests/test_gpu_fused_walk_dual.py \
                 tests/test_gpu_unscaled_walk.py tests/test_gpu_operator_mode.py > $O/guard_suite.log 2>&1; rc=$?
             tail -4 $O/guard_suite.log; [ $rc -eq 0 ] || exit $rc ;;
    ab_fuse) # forward walk + dual update: one launch against two, by shape (the rule of fuse_by_shape: unsharded, chains >= CUs)
             { echo "== whole tree, dense"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300
               echo "== whole tree, structured"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 --structured
               echo "== 31-scenario tree, dense"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 --workload barcelona31
               echo "== 31-scenario tree, structured"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 --workload barcelona31 --structured
               for w in 2 4 8; do echo "== 1/$w shard"; bash tools/ab_env.sh RAPIDNET_FUSE_DOWN_DUAL 3 300 --emulate-world $w; done; } 2>&1 | tee $O/ab_fuse_by_shape.txt ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
