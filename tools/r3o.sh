#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3o; mkdir -p $out; cd $root
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_layer.py -x -q -m gpu --timeout 120 --timeout-method thread \
  -k "rectangular or linearity or epilogue" > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
bash tools/run_examples.sh r3o || exit 1
timeout -k 10 300 python bench.py --workload gmm-reddit --no-cpu-baseline > $out/gmm_reddit.json 2> $out/gmm.err || { tail -5 $out/gmm.err; exit 1; }
cut -c1-900 $out/gmm_reddit.json
