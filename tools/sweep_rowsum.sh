#!/bin/bash
# Hub-row threshold sweep for the grad_attn_row row sum (DGNN_ROWSUM_LONG).
# Usage: gpurun -- bash tools/sweep_rowsum.sh ; writes gpurun_out/sweep_rowsum.log
set -u
out=gpurun_out/sweep_rowsum.log; mkdir -p gpurun_out; : > $out
for T in 64 128 256 512 1024 4096; do
  line=$(DGNN_ROWSUM_LONG=$T timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric"') || { echo "thr=$T FAILED" | tee -a $out; exit 1; }
  echo "thr=$T $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms/step=%.3f bwd_call=%.3f bwd_kernel=%.3f copy=%.0f" % (d["ms_per_step"], r["bwd_call_ms"], r["bwd_kernel_ms"], r["device_copy_GBps"]))')" | tee -a $out
done
