#!/bin/bash
# same-box A/B of the FBE / NAMA loops: NAMA's pair of Hessian sweeps in one pass over the blocks (RAPIDNET_NAMA_PAIR) on / off
set -o pipefail
for r in 1 2; do
  for pair in 1 0; do
    echo "round $r RAPIDNET_NAMA_PAIR=$pair"
    RAPIDNET_NAMA_PAIR=$pair python3 tools/time_fbe_nama.py barcelona493 40 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('   %-20s structured=%s  %.3f ms/it  %s' % (d.get('algorithm'), d.get('structured'), d.get('ms_per_iteration', float('nan')), d.get('line_searches')))
"
  done
done
