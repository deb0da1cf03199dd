#!/bin/bash
# A/B of library builds on the headline bench: tools/ab.sh libA.so libB.so ...  ("-" = the in-tree build)
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DGNN_HIP_LIB; else export DGNN_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 150 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$lib', round(d['ms_per_step'],3), 'fwd', r['fwd_kernel_ms'], 'bwd', r['bwd_kernel_ms'])" || exit 1
done
