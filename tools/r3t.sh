#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3t; mkdir -p $out; cd $root
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 500 --timeout-method thread > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for wl in gmm-reddit gmm-pubmed; do timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$wl', round(d['ms_per_step'],4), d['roofline']['fwd_kernel_ms'], d['roofline']['bwd_kernel_ms'])"; done
DGNN_BENCH_LABELS=scrambled DGNN_BENCH_ITERS=5 timeout -k 10 500 python tools/bench_ops.py shapes > $out/shapes_scrambled.jsonl 2>/dev/null || exit 1
DGNN_BENCH_ITERS=5 timeout -k 10 500 python tools/bench_ops.py shapes > $out/shapes_natural.jsonl 2>/dev/null || exit 1
cat $out/shapes_scrambled.jsonl
for rn in 1/4 3/8; do timeout -k 10 300 python bench.py --simulate-rank $rn --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$rn', round(d['ms_per_step'],3))"; done
