#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3p; mkdir -p $out; cd $root
: > $out/simranks.jsonl
for rn in 0/2 1/2 1/4 3/4 3/8 6/8; do
  timeout -k 10 300 python bench.py --simulate-rank $rn --steps 10 --warmup 3 >> $out/simranks.jsonl 2> $out/sim.err || { tail -5 $out/sim.err; exit 1; }
done
for rn in 0/2 1/4 3/8; do
  timeout -k 10 300 python bench.py --simulate-rank $rn --strategy rows --steps 10 --warmup 3 >> $out/simranks.jsonl 2> $out/sim.err || { tail -5 $out/sim.err; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/r3p/simranks.jsonl"):
    d = json.loads(l)
    print(d.get("simulated", "force-dist")[:60], "| ms", round(d["ms_per_step"], 3), "| expected", d.get("expected", {}).get("ms_per_step"), "| roof", d.get("roofline", {}).get("frac"), "| peak MB", d["peak_mem_MB"])
PY
