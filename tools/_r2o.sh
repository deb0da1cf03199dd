set -e
mkdir -p gpurun_out/r2o
timeout -k 10 300 python tools/bench_ops.py shapes > gpurun_out/r2o/shapes_products_neg.log 2>&1
tail -12 gpurun_out/r2o/shapes_products_neg.log
timeout -k 10 300 python bench.py --simulate-rank 3/8 --steps 20 --warmup 5 --labels natural > gpurun_out/r2o/simrank.json 2> gpurun_out/r2o/simrank.err
cat gpurun_out/r2o/simrank.json
timeout -k 10 300 python bench.py --workload reddit --steps 10 --warmup 3 --no-other-labels > gpurun_out/r2o/bench_reddit.json 2> gpurun_out/r2o/bench_reddit.err
cat gpurun_out/r2o/bench_reddit.json
bash tools/profile.sh r2d products > gpurun_out/r2o/prof_products.log 2>&1
bash tools/profile.sh r2d reddit > gpurun_out/r2o/prof_reddit.log 2>&1
echo profiles done
