#!/bin/bash
set -e -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_layer.py tests/test_dist.py tests/test_fullsize.py tests/test_zz_no_alloc.py -m gpu -x -v --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/r2h/pytest2.log | tail -n 3
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-labels"
timeout -k 10 300 $B 2>gpurun_out/r2h/bench.err | tee gpurun_out/r2h/bench2.json | cut -c1-400
timeout -k 10 300 $B 2>gpurun_out/r2h/bench.err | python -c "import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print(d['ms_per_step'], r['fwd_call_ms'], r['fwd_kernel_ms'], r['bwd_call_ms'], r['bwd_kernel_ms'])"
timeout -k 10 300 $B --simulate-rank 3/8 2>gpurun_out/r2h/sim.err | python -c "import json,sys;d=json.loads(sys.stdin.read());print('sim 3/8', d['ms_per_step'])"
