#!/bin/bash
# Tuning sweep on the GPU box: in-flight depth U (library variants) x group sizes x occupancy.
# Usage: gpurun -- bash tools/sweep.sh ; writes gpurun_out/sweep.log
set -u
out=gpurun_out/sweep.log; mkdir -p gpurun_out; : > $out
V=$PWD/dgnn_amd/csrc/variants
run() { # label, env...
  label=$1; shift
  line=$(env "$@" timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$label $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("ms/step=%.2f fwd=%.2f bwd=%.2f" % (d["ms_per_step"], r["fwd_ms"], r["bwd_ms"]))' 2>/dev/null || echo FAILED)" | tee -a $out
}
for U in 1 2 3; do
  for FL in 512 1024; do
    run "U$U fwd$FL bwd256" DGNN_HIP_LIB=$V/libdgnn_hip_u$U.so DGNN_FWD_L=$FL DGNN_BWD_L=256
  done
  # occupancy limiters: 2 blocks/CU (pad to ~70 KB) and 1 block/CU (pad to ~100 KB)
  run "U$U fwd512 bwd256 pad->2blk" DGNN_HIP_LIB=$V/libdgnn_hip_u$U.so DGNN_FWD_L=512 DGNN_BWD_L=256 DGNN_FWD_LDS_PAD=20000 DGNN_BWD_LDS_PAD=16000
  run "U$U fwd512 bwd256 pad->1blk" DGNN_HIP_LIB=$V/libdgnn_hip_u$U.so DGNN_FWD_L=512 DGNN_BWD_L=256 DGNN_FWD_LDS_PAD=50000 DGNN_BWD_LDS_PAD=45000
done
