#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
export DGNN_BENCH_SHAPES="1x40,1x44,1x48,1x52,1x56,1x64,4x40,4x44,4x48,2x44,2x48"
timeout -k 10 500 python tools/bench_ops.py shapes_reddit 2>&1 | tee gpurun_out/r2h/pad_reddit.jsonl
timeout -k 10 500 python tools/bench_ops.py shapes 2>&1 | tee gpurun_out/r2h/pad_products.jsonl
