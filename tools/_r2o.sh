set -e
mkdir -p gpurun_out/r2p
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_examples.py -m gpu -x -q --timeout 200 --timeout-method thread -k "knn or edgeconv or dgcnn" > gpurun_out/r2p/knn_tests.log 2>&1 || { tail -40 gpurun_out/r2p/knn_tests.log; exit 1; }
tail -3 gpurun_out/r2p/knn_tests.log
timeout -k 10 300 python tools/knn_bench.py > gpurun_out/r2p/knn_bench.log 2>&1
cat gpurun_out/r2p/knn_bench.log
cd examples
rm -f ../gpurun_out/r2p/dgcnn.csv
timeout -k 10 300 python3 train_edgeconv.py --batch-size 32 --num-points 2048 --k 20 --n-epochs 20 --output ../gpurun_out/r2p/dgcnn.csv > ../gpurun_out/r2p/dgcnn.log 2>&1
timeout -k 10 300 python3 train_edgeconv.py --batch-size 64 --num-points 1024 --k 40 --n-epochs 20 --output ../gpurun_out/r2p/dgcnn.csv >> ../gpurun_out/r2p/dgcnn.log 2>&1
cat ../gpurun_out/r2p/dgcnn.csv
cd ..
bash tools/profile.sh r2e edgeconv tools/profile_ops.py edgeconv > gpurun_out/r2p/prof_edgeconv.log 2>&1
echo profiled
