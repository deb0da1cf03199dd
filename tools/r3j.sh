#!/bin/bash
# round-3 GPU batch: tolerance twins, simulated ranks (+ kernel trace of 3/8), PMC passes of the wide shapes
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3j; mkdir -p $out
cd $root
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread \
  -k "spmm_sddmm or test_gmm or zz_hip_error" -s > $out/pytest_tol.log 2>&1 || { tail -20 $out/pytest_tol.log; exit 1; }
tail -12 $out/pytest_tol.log; timeout -k 10 200 python tools/tol_probe.py > $out/tol_probe.txt 2>&1
: > $out/simranks.jsonl
for rn in 3/8 0/8 1/4 0/2; do
  timeout -k 10 300 python bench.py --simulate-rank $rn --steps 10 --warmup 3 >> $out/simranks.jsonl 2> $out/sim.err || { tail -5 $out/sim.err; exit 1; }
done
cut -c1-400 $out/simranks.jsonl
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_sim3of8 -- python3 $root/bench.py --simulate-rank 3/8 --steps 5 --warmup 2 > $out/trace_sim.log 2>&1 || { tail -5 $out/trace_sim.log; exit 1; }
cd $root
for hf in "8 32" "4 128" "4 64"; do
  set -- $hf
  DGNN_PROFILE_SQ=1 DGNN_PROFILE_SQ2=1 bash tools/profile.sh r3j gat_$1x$2 tools/pmc_shape.py $1 $2 || exit 1
done

DGNN_BENCH_SHAPES=2x64,4x64,1x64,8x32,4x128 bash tools/ab_shapes.sh - tools/bin/libdgnn_hip_wpe4.so tools/bin/libdgnn_hip_wpe6.so - > $out/ab_wpe.txt 2>&1; cat $out/ab_wpe.txt
echo done
