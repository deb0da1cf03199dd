#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3u; mkdir -p $out; cd $root
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 500 --timeout-method thread > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
: > $out/ab.txt
for odd in 0 1; do
  DGNN_FWD_HT_ODD=$odd DGNN_BENCH_LABELS=scrambled DGNN_BENCH_ITERS=5 DGNN_BENCH_SHAPES=3x16,6x8,5x8,7x8,3x64,6x20,3x41,8x8,4x16 timeout -k 10 400 python tools/bench_ops.py shapes 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('HT_ODD=$odd', str(d['H'])+'x'+str(d['F']), d['fwd_ms'], d['bwd_ms'])
" >> $out/ab.txt || exit 1
done
sort -k2,2 -k1,1 $out/ab.txt
