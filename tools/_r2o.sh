set -e
mkdir -p gpurun_out/r2o
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_layer.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r2o/parity2.log 2>&1 || { tail -30 gpurun_out/r2o/parity2.log; exit 1; }
tail -3 gpurun_out/r2o/parity2.log
timeout -k 10 300 python tools/bench_ops.py shapes > gpurun_out/r2o/shapes2.log 2>&1
tail -12 gpurun_out/r2o/shapes2.log
