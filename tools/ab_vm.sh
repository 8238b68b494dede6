#!/bin/bash
# k_value_mfma: grid size sweep (kernel time from rocprofv3 stats), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 256 512 679; do
  rm -rf gpurun_out/prof_vm
  RAPIDNET_VM_BLOCKS=$b rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_vm -- python3 tools/profile_fbe.py barcelona493 20 > gpurun_out/prof_vm.log 2>&1
  f=$(find gpurun_out/prof_vm -name "*kernel_stats.csv" | head -1)
  echo "RAPIDNET_VM_BLOCKS=$b $(grep ms gpurun_out/prof_vm.log | head -1 | cut -c1-30)"
  grep -i "value_mfma" $f | sed 's/.*)",/  k_value_mfma calls,total,avg,%,min,max: /'
done
