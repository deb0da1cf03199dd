#!/usr/bin/env python3
"""Forward of narrow head x feature shapes with the guarded (DGNN_FWD_STEADY=0) and the unguarded
steady-state gather loop, on one workload:  python tools/narrow_ab.py reddit|products"""
import json
import os
import subprocess
import sys

if len(sys.argv) > 2:       # child: one setting
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_ops
    bench_ops.SHAPES = ((1, 64), (1, 48), (1, 16), (2, 32), (4, 16), (8, 8), (4, 64))
    bench_ops.shapes(sys.argv[1])
else:
    base = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdgnn_base.so")
    variants = [("guarded", {"DGNN_FWD_STEADY": "0"}), ("default", {})]
    if os.path.exists(base):
        variants.insert(0, ("base-lib", {"DGNN_HIP_LIB": base}))
    for flag, extra in variants:
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, __file__, sys.argv[1], "child"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                print(flag, d["H"], d["F"], "fwd_ms", d["fwd_ms"], "bwd_ms", d["bwd_ms"], flush=True)
