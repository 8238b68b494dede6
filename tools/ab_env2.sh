#!/bin/bash
# interleaved comparison of several environment settings of librapidnet_hip (same binary):
#   bash tools/ab_env2.sh <rounds> <steps> "<VAR=val ...>;<VAR=val ...>;..." [extra bench.py args]
rounds=$1; steps=$2; IFS=';' read -ra SETS <<< "$3"; shift 3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/abe2; rm -f gpurun_out/abe2/*.json
for r in $(seq 1 $rounds); do
  for i in "${!SETS[@]}"; do
    env ${SETS[$i]} python3 bench.py --steps $steps --warmup 20 --no-cpu-baseline --no-traffic --profile-steps 40 --dense-only --repeats 2 --other-configs "" "$@" 2>/dev/null | grep '"metric"' > gpurun_out/abe2/s${i}_$r.json
  done
done
python3 - "$rounds" "${SETS[@]}" <<'PY'
import json,sys,statistics as st
rounds=int(sys.argv[1]); sets=sys.argv[2:]
for i,name in enumerate(sets):
    ds=[json.load(open("gpurun_out/abe2/s%d_%d.json"%(i,r))) for r in range(1,rounds+1)]
    ms=[d["timing_spread"]["ms_per_step_median"] for d in ds]
    cls=lambda k:[d["kernel_classes"][k]["avg_us"] for d in ds]
    k0=[k for k in ds[0]["kernel_classes"] if k.startswith("stream") or k.startswith("struct")][0]
    print("%-60s ms/step median %.4f (min %.4f max %.4f) | %s %.1f rest %.1f dual %.1f"%(name,st.median(ms),min(ms),max(ms),k0,st.median(cls(k0)),st.median(cls("recursion+shared_gemms")),st.median(cls("dual_update"))))
PY
