cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in jb16 jb38; do
export RAPIDNET_LIB=$GRAFT_REPO_ROOT/rapidnet_amd/librapidnet_hip_$v.so
rm -rf gpurun_out/fbe_stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fbe_stats -o k -- python3 tools/time_fbe_nama.py barcelona493 12 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/fbe_stats/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_value_terms" in r["Name"]: print("$v", r["Calls"], float(r["AverageNs"])/1e3)
PY
done
