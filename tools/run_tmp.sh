#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
DGNN_FUZZ_EXTRA=8000 timeout -k 10 1100 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fuzz" --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/r2h/fuzz_long.log | tail -n 12
