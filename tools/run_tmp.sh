#!/bin/bash
set -e -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r2h/ex_gmm; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/examples/train_gmmconv.py --dataset reddit --n-epochs 10 > $out/trace.log 2>&1
grep -v "^W\|^E" $out/trace.log | tail -3 | cut -c1-300
