#!/bin/bash
# One rank's share of an N-way partition timed alone, for every rank of N = 8, 4, 2 (bench.py --simulate-rank):
#   gpurun -- bash tools/simulate_ranks.sh <tag>   -> gpurun_out/<tag>/simranks.jsonl
set -e
tag=${1:-sim}; out=gpurun_out/$tag; mkdir -p $out; : > $out/simranks.jsonl
for n in 8 4 2; do
  for ((r = 0; r < n; r++)); do
    timeout -k 10 300 python bench.py --simulate-rank $r/$n --steps 10 --warmup 3 > $out/sim.json 2> $out/sim.err
    python - "$r" "$n" "$out" <<'PY'
import json, sys
r, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
d = json.load(open(f"{out}/sim.json"))
line = {"rank": r, "of": n, "ms_per_step": round(d["ms_per_step"], 3), "labels": "scrambled"}
print(json.dumps(line))
open(f"{out}/simranks.jsonl", "a").write(json.dumps(line) + "\n")
PY
  done
done
