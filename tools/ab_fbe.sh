#!/bin/bash
# same-box A/B of the FBE / NAMA loops:  bash tools/ab_fbe.sh VAR   (VAR = RAPIDNET_NAMA_PAIR | RAPIDNET_VALUE_MFMA | RAPIDNET_LS_SEQUENTIAL ...; on = 1, off = 0)
set -o pipefail
VAR=${1:-RAPIDNET_NAMA_PAIR}
for r in 1 2; do
  for v in 1 0; do
    echo "round $r $VAR=$v"
    env $VAR=$v python3 tools/time_fbe_nama.py barcelona493 40 2>&1 | python3 -c "
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
