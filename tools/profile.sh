#!/bin/bash
# rocprofv3 passes of one program on the GPU box (run through gpurun):
#   gpurun -- bash tools/profile.sh <tag> <name> [program args ...]
# default program: bench.py on <name> as workload.  Kernel trace + stats in one run; HBM byte
# counters and the L2 (TCC) hit counters in runs of their own, one --pmc group each, python3
# directly after `--`.  Output: gpurun_out/prof_<tag>_<name>/{trace,fetch,write,tcc}; condensed into
# profiles/ by tools/summarize_profile.py <tag> <name>.
set -u
tag=${1:-r2}; name=${2:-products}; shift 2 || true
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/prof_${tag}_${name}; mkdir -p $out
if [ $# -eq 0 ]; then
  long=(python3 $root/bench.py --workload $name --steps 5 --warmup 2 --no-cpu-baseline --no-other-labels)
  short=(python3 $root/bench.py --workload $name --steps 2 --warmup 1 --no-cpu-baseline --no-other-labels)
else
  prog=$1; shift
  case $prog in /*) ;; *) prog=$root/$prog ;; esac      # the passes run from /tmp
  long=(python3 $prog "$@"); short=(python3 $prog "$@")
fi
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "${long[@]}" > $out/trace.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- "${short[@]}" > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- "${short[@]}" > $out/write.log 2>&1 || exit 1
if [ "${DGNN_PROFILE_TCC:-1}" = "1" ]; then
  timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- "${short[@]}" > $out/tcc.log 2>&1 || exit 1
fi
if [ "${DGNN_PROFILE_SQ:-0}" = "1" ]; then
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/sq -- "${short[@]}" > $out/sq.log 2>&1 || exit 1
fi
if [ "${DGNN_PROFILE_SQ2:-0}" = "1" ]; then     # where the cycles go (a failing counter name must not end the script)
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM --output-format csv -d $out/sq2 -- "${short[@]}" > $out/sq2.log 2>&1 || echo "sq2 pass failed (see sq2.log)"
fi
tail -1 $out/trace.log | cut -c1-700
