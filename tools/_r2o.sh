set -e
mkdir -p gpurun_out/r2p
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 200 --timeout-method thread -k "knn" > gpurun_out/r2p/knn_tests.log 2>&1 || { tail -40 gpurun_out/r2p/knn_tests.log; exit 1; }
tail -3 gpurun_out/r2p/knn_tests.log
timeout -k 10 100 python tools/knn_debug.py > gpurun_out/r2p/knn_debug.log 2>&1; tail -4 gpurun_out/r2p/knn_debug.log
timeout -k 10 300 python tools/knn_bench.py > gpurun_out/r2p/knn_bench.log 2>&1 || { tail -20 gpurun_out/r2p/knn_bench.log; exit 1; }
cat gpurun_out/r2p/knn_bench.log
cd /tmp
for d in 64 128; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2p/knn$d -- python3 $GRAFT_REPO_ROOT/tools/knn_prof.py 32 2048 20 $d > $GRAFT_REPO_ROOT/gpurun_out/r2p/knn$d.log 2>&1
done
