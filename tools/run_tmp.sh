#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q -k "edgeconv or EdgeConv or dgcnn" --timeout 300 --timeout-method thread 2>&1 | tail -n 3
timeout -k 10 300 python tools/bench_ops.py edgeconv 2>&1 | grep '^{' | tee gpurun_out/r2h/ec_det.txt
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2h/ecprof -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py edgeconv > /dev/null 2>&1
