#!/bin/bash
# same-box A/B of the FBE / NAMA loops:  bash tools/ab_fbe.sh KNOB   (KNOB = nama_pair | value_mfma | ls_sequential: rapidnet_amd.capi.KNOBS; 1 against 0)
set -o pipefail
KNOB=${1:-nama_pair}
for r in 1 2; do
  for v in 1 0; do
    echo "round $r $KNOB=$v"
    python3 tools/time_fbe_nama.py barcelona493 40 f64 $KNOB=$v 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        if d.get('algorithm') != 'proximalAlgorithm':
            print('   %-20s structured=%s  %.3f ms/it  %s' % (d.get('algorithm'), d.get('structured'), d.get('ms_per_iteration', float('nan')), d.get('line_searches')))
"
  done
done
