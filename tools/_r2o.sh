set -e
mkdir -p gpurun_out/r2p
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 200 --timeout-method thread -k "csr2csc or index or transpose or convert or graph" > gpurun_out/r2p/idx_tests.log 2>&1 || { tail -40 gpurun_out/r2p/idx_tests.log; exit 1; }
tail -2 gpurun_out/r2p/idx_tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2p/csr2csc -- python3 $GRAFT_REPO_ROOT/tools/csr2csc_bench.py products > $GRAFT_REPO_ROOT/gpurun_out/r2p/csr2csc.log 2>&1
grep "^{" $GRAFT_REPO_ROOT/gpurun_out/r2p/csr2csc.log
