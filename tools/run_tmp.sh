#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -n 3
export DGNN_BENCH_SHAPES="1x64,1x48,1x16,8x8,4x16,2x32"
for v in 0 1; do
  echo "== PACK_CONST=$v" | tee -a gpurun_out/r2h/packconst_ab.txt
  DGNN_BWD_PACK_CONST=$v timeout -k 10 300 python tools/bench_ops.py shapes 2>&1 | grep '^{' | tee -a gpurun_out/r2h/packconst_ab.txt
  DGNN_BWD_PACK_CONST=$v DGNN_PROFILE_GMM_F=16 timeout -k 10 300 python tools/profile_ops.py gmmsum_reddit 2>&1 | grep "^gmm" | tee -a gpurun_out/r2h/packconst_ab.txt
done
