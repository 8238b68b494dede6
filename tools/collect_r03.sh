#!/bin/bash
# HISTORICAL: the command of record behind profiles/r03_*; it names switches and test files of that round (round 6 turned the tuning
# environment variables into rn_debug_set_knob choices and removed the forms that lost their A/B) -- tools/collect_r06.sh is the current one.
# Round-3 measurement set (one gpurun call): bench line, rocprof kernel stats of the same command, emulated shards, the
# two-rank launcher rehearsal, FBE / NAMA timings, PMC traffic file.  Outputs under gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03; mkdir -p $O
python3 bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err || { echo "bench failed"; tail -5 $O/bench_1gpu.err; exit 1; }
echo "bench done"
rm -rf $O/ks1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -o k -- python3 bench.py --no-cpu-baseline --no-traffic --other-configs "" > $O/bench_1gpu_under_rocprof.json 2> /dev/null || exit 1
cp $(find $O/ks1 -name k_kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/ks1
echo "kstats done"
for W in 2 4 8; do python3 bench.py --emulate-world $W --no-cpu-baseline --steps 200 --warmup 20 --other-configs "" 2>/dev/null | grep '"metric"' >> $O/emulated_shards.jsonl; done
rm -rf $O/ks8; rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks8 -o k -- python3 bench.py --emulate-world 8 --no-cpu-baseline --steps 200 --warmup 20 --profile-steps 0 --repeats 0 --other-configs "" > /dev/null 2>&1 || exit 1
cp $(find $O/ks8 -name k_kernel_stats.csv | head -1) $O/emulated_shard8_kernel_stats.csv; rm -rf $O/ks8
echo "shards done"
python3 bench.py --gpus 2 --steps 20 --warmup 5 --allow-oversubscribe --no-cpu-baseline > $O/bench_2ranks_oversubscribed_rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc=$?"
python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_strict.out 2> $O/bench_2ranks_strict.err; echo "strict rc=$?" | tee -a $O/bench_2ranks_strict.out
python3 tools/time_fbe_nama.py barcelona493 30 > $O/fbe_nama_timing.jsonl 2>/dev/null
echo "fbe done"
bash tools/collect_traffic.sh > /dev/null 2>&1; cp gpurun_out/traffic.json $O/traffic.json
echo "traffic done"
