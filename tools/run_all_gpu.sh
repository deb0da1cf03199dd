#!/bin/bash
# The round-end sequence on one GPU box: gpurun -- bash tools/run_all_gpu.sh <tag>
#   full `pytest -m gpu`, smoke(), default bench line, Reddit bench line; logs under gpurun_out/<tag>/
set -e
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread 2>&1 | tee $out/full_gpu.log | grep -E "passed|failed|FAILED|ERROR|outside" | tail -8
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 300 python bench.py > $out/bench_products.json 2> $out/bench_products.err
cat $out/bench_products.json
timeout -k 10 300 python bench.py --workload reddit --no-cpu-baseline > $out/bench_reddit.json 2> $out/bench_reddit.err
cut -c1-400 $out/bench_reddit.json
