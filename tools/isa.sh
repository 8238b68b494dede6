#!/bin/bash
# device ISA of one kernel: bash tools/isa.sh <unit> <mangled-name-prefix> [out.s]   (e.g. k_slab _ZN2rn10k_gemm_vlvIdE)
cd /root/repo/rapidnet_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-unused-function -Wno-pass-failed -o /tmp/k.s $1.hip 2>&1 | grep -v "warning\|^$" | head -5
L=$(grep -n "^$2.*:" /tmp/k.s | head -1 | cut -d: -f1); E=$(awk -v s=$L 'NR>=s && /\.Lfunc_end/{print NR; exit}' /tmp/k.s)
sed -n "${L},${E}p" /tmp/k.s > ${3:-/tmp/one.s}; echo "lines $L-$E -> ${3:-/tmp/one.s}"
grep -n "v_mfma" ${3:-/tmp/one.s} | awk -F: '{print $1}' | tr '\n' ' ' | head -c 500; echo
