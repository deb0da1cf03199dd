#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3l; mkdir -p $out; cd $root
: > $out/simranks.jsonl
for rn in 3/8 1/4 0/2; do
  timeout -k 10 300 python bench.py --simulate-rank $rn --steps 10 --warmup 3 >> $out/simranks.jsonl 2> $out/sim.err || { tail -5 $out/sim.err; exit 1; }
done
DGNN_DIST_CHUNKS=4 DGNN_DIST_CHUNK_SHAPE=geometric timeout -k 10 300 python bench.py --simulate-rank 0/2 --steps 10 --warmup 3 >> $out/simranks.jsonl 2> $out/sim.err || { tail -5 $out/sim.err; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/r3l/simranks.jsonl"):
    d = json.loads(l)
    print(d["simulated"][:22], "ms", round(d["ms_per_step"], 3), "enqueue", d["host_enqueue_ms_per_step"], "plan", d["expected"]["plan"]["chunks"], d["expected"]["plan"]["chunk_shape"], "expected", d["expected"]["ms_per_step"])
PY
timeout -k 10 300 python bench.py --simulate-rank 3/8 --steps 10 --warmup 3 --graph > $out/sim_graph.json 2> $out/sim_graph.err || { tail -8 $out/sim_graph.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r3l/sim_graph.json')); print('graph replay 3/8: ms', round(d['ms_per_step'],3), 'enqueue', d['host_enqueue_ms_per_step'])"
