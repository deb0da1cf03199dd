set -e
mkdir -p gpurun_out/r2o
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2o/full_gpu.log 2>&1 || { tail -30 gpurun_out/r2o/full_gpu.log; exit 1; }
tail -5 gpurun_out/r2o/full_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
