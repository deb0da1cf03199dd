#!/bin/bash
# A/B of library builds over GAT shapes on the products-shaped graph:
#   DGNN_BENCH_SHAPES=1x16,8x8 tools/ab_shapes.sh - tools/bin/libA.so ...   ("-" = the in-tree build)
# one line per (library, shape): forward / backward call in ms (tools/bench_ops.py shapes)
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset DGNN_HIP_LIB; else export DGNN_HIP_LIB=$PWD/$lib; fi
  timeout -k 10 280 python tools/bench_ops.py shapes 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print('$lib', f\"{d['H']}x{d['F']}\", 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], 'sum', round(d['fwd_ms'] + d['bwd_ms'], 3))
" || exit 1
done
