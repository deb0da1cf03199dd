#!/bin/bash
set -u
root=${GRAFT_REPO_ROOT:-$PWD}; out=$root/gpurun_out/r3k; mkdir -p $out; cd $root
timeout -k 10 500 python tools/plan_model.py > $out/plan_model.jsonl 2> $out/plan_model.err || { tail -5 $out/plan_model.err; exit 1; }
cat $out/plan_model.jsonl
DGNN_BENCH_LABELS=scrambled DGNN_BENCH_ITERS=5 timeout -k 10 500 python tools/bench_ops.py shapes > $out/shapes_scrambled.jsonl 2> $out/shapes.err || { tail -5 $out/shapes.err; exit 1; }
cat $out/shapes_scrambled.jsonl
