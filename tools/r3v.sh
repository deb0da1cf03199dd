#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3v; mkdir -p $out; cd $root
: > $out/gmm.txt
one() { timeout -k 10 300 python bench.py --workload gmm-reddit --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'])" >> $out/gmm.txt; }
one default
for fl in 64 96 192 256 384; do DGNN_FWD_L=$fl one FWD_L=$fl; done
for bl in 64 256; do DGNN_BWD_L=$bl one BWD_L=$bl; done
cat $out/gmm.txt
