#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q -k "knn" --timeout 300 --timeout-method thread 2>&1 | tail -n 3
for data in gauss features; do
  echo "== data=$data" | tee -a gpurun_out/r2h/knn_center2.txt
  DGNN_KNN_BENCH_DATA=$data timeout -k 10 300 python tools/knn_bench.py 2>&1 | grep '^{' | tee -a gpurun_out/r2h/knn_center2.txt
done
timeout -k 10 300 python examples/train_edgeconv.py --n-epochs 20 2>&1 | tail -1
timeout -k 10 300 python examples/train_edgeconv.py --n-epochs 20 --batch-size 64 --num-points 1024 --k 40 2>&1 | tail -1
