#!/bin/bash
# Group-size sweep for narrow head x feature shapes: tools/sweep_narrow.sh <workload> "<fwd Ls>" "<bwd Ls>"
w=${1:-products}
for fl in ${2:-256 512}; do for bl in ${3:-64 128 256}; do
  DGNN_FWD_L=$fl DGNN_BWD_L=$bl timeout -k 10 300 python tools/narrow_ab.py $w child 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('$w FWD_L=$fl BWD_L=$bl', d['H'], d['F'], d['fwd_ms'], d['bwd_ms'])
"
done; done
