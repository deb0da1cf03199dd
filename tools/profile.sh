#!/bin/bash
# rocprofv3 passes of the headline benchmark on the GPU box (run through gpurun).
#   gpurun -- bash tools/profile.sh <tag>
# Kernel trace + stats in one run; HBM byte counters in their own runs (one counter each).
set -u
tag=${1:-r1}; root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/prof_$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1 || exit 1
tail -1 $out/bench_trace.log | cut -c1-600
