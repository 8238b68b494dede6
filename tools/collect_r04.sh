#!/bin/bash
# HISTORICAL: the command of record behind profiles/r04_*; it names switches and test files of that round (round 6 turned the tuning
# environment variables into rn_debug_set_knob choices and removed the forms that lost their A/B) -- tools/collect_r06.sh is the current one.
# Round-4 measurement set (one gpurun call): bench line, rocprof kernel stats of the same command, emulated shards (RCCL one-rank
# vs one-shot), kernel stats of the 1/8 shard for both transports, the two-rank launcher rehearsal, the fault-injection run,
# fp32 on the headline tree, FBE / NAMA timings.  Outputs under gpurun_out/r04/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04; mkdir -p $O
python3 bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err || { echo "bench failed"; tail -5 $O/bench_1gpu.err; exit 1; }
echo "bench done"
rm -rf $O/ks1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -o k -- python3 bench.py --no-cpu-baseline --no-traffic --other-configs "" > $O/bench_1gpu_under_rocprof.json 2> /dev/null || exit 1
cp $(find $O/ks1 -name k_kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/ks1
echo "kstats done"
rm -f $O/emulated_shards.jsonl $O/emulated_shards_oneshot.jsonl
for W in 2 4 8; do
  python3 bench.py --emulate-world $W --no-cpu-baseline --steps 200 --warmup 20 --other-configs "" 2>/dev/null | grep '"metric"' >> $O/emulated_shards.jsonl
  python3 bench.py --emulate-world $W --one-shot --no-cpu-baseline --steps 200 --warmup 20 --other-configs "" 2>/dev/null | grep '"metric"' >> $O/emulated_shards_oneshot.jsonl
done
for mode in rccl oneshot; do
  flag=""; [ $mode = oneshot ] && flag="--one-shot"
  rm -rf $O/ks8; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks8 -o k -- python3 bench.py --emulate-world 8 $flag --no-cpu-baseline --steps 200 --warmup 20 --profile-steps 0 --repeats 0 --other-configs "" > /dev/null 2>&1 || exit 1
  cp $(find $O/ks8 -name k_kernel_stats.csv | head -1) $O/emulated_shard8_${mode}_kernel_stats.csv; rm -rf $O/ks8
done
echo "shards done"
python3 bench.py --gpus 2 --steps 20 --warmup 5 --allow-oversubscribe --cpu-iterations 5 > $O/bench_2ranks_oversubscribed_rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_strict.out 2> $O/bench_2ranks_strict.err; echo "strict rc=$?" | tee -a $O/bench_2ranks_strict.out; tail -3 $O/bench_2ranks_strict.err >> $O/bench_2ranks_strict.out
( time RAPIDNET_BENCH_FAULT=before_comm_init:1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --allow-oversubscribe --no-cpu-baseline --other-configs "" ) > $O/bench_2ranks_fault.out 2> $O/bench_2ranks_fault.err; echo "fault rc=$?" | tee -a $O/bench_2ranks_fault.out; grep -h "injected\|real\|giving up\|rank" $O/bench_2ranks_fault.err | tail -8 >> $O/bench_2ranks_fault.out
python3 bench.py --precision f32 --other-configs "" --cpu-iterations 5 > $O/barcelona493_f32_1gpu.json 2>/dev/null
python3 tools/time_fbe_nama.py barcelona493 40 > $O/fbe_nama_timing.jsonl 2>/dev/null
for alg in fbe nama; do
  rm -rf $O/ksf; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksf -o k -- python3 tools/profile_fbe.py barcelona493 40 0 $alg > $O/${alg}_profile_run.txt 2>/dev/null || exit 1
  cp $(find $O/ksf -name k_kernel_stats.csv | head -1) $O/${alg}_kernel_stats.csv; rm -rf $O/ksf
done
echo "fbe done"
bash tools/stream_vs_nodes.sh "6 7 8 9 10 11 12 16" > $O/stream_vs_nodes.txt 2>/dev/null
echo "all done"
