set -e
mkdir -p gpurun_out/r2r
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_layer.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2r/parity.log 2>&1 || { tail -40 gpurun_out/r2r/parity.log; exit 1; }
tail -2 gpurun_out/r2r/parity.log
DGNN_FWD_BATCH4=0 timeout -k 10 300 python tools/bench_ops.py shapes > gpurun_out/r2r/shapes_b0.log 2>&1
timeout -k 10 300 python tools/bench_ops.py shapes > gpurun_out/r2r/shapes_b1.log 2>&1
paste -d'\n' gpurun_out/r2r/shapes_b0.log gpurun_out/r2r/shapes_b1.log | grep '"H"' | cut -c1-110
