#!/bin/bash
# group-size sweep, third set: eight heads and finer steps
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3s; mkdir -p $out; cd $root
: > $out/sweep3.txt
run() {
  DGNN_BENCH_LABELS=scrambled DGNN_BENCH_ITERS=5 DGNN_BENCH_SHAPES=$2 timeout -k 10 300 python tools/bench_ops.py shapes 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('$1', str(d['H'])+'x'+str(d['F']), d['fwd_ms'], d['bwd_ms'])
" >> $out/sweep3.txt || exit 1
}
for fl in 64 96 128 160 192; do DGNN_FWD_L=$fl run "FWD_L=$fl" 8x8,8x4,8x16,6x8; done
for fl in 320 384 448 512 640 768; do DGNN_FWD_L=$fl run "FWD_L=$fl" 1x16,1x32,1x64,2x64,1x128; done
cat $out/sweep3.txt
