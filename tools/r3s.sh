#!/bin/bash
# group-size sweep on the final narrow kernels (scrambled labels), second set of shapes
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3s; mkdir -p $out; cd $root
: > $out/sweep2.txt
run() {
  DGNN_BENCH_LABELS=scrambled DGNN_BENCH_ITERS=5 DGNN_BENCH_SHAPES=2x32,4x16,4x32,1x128,1x48,2x16,1x7,4x8,1x47,2x41 timeout -k 10 300 python tools/bench_ops.py shapes 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('$1', str(d['H'])+'x'+str(d['F']), d['fwd_ms'], d['bwd_ms'])
" >> $out/sweep2.txt || exit 1
}
for fl in 128 256 384 512; do DGNN_FWD_L=$fl run "FWD_L=$fl"; done
for bl in 64 128 256; do DGNN_BWD_L=$bl run "BWD_L=$bl"; done
cat $out/sweep2.txt
