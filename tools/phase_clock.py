#!/usr/bin/env python3
"""Per-phase wave-cycle split of the forward / backward gather kernels (diagnostic library built by
tools/phase_clock.sh, selected with DGNN_HIP_LIB): for each HxF shape on the products-shaped graph,
the share of all wave-cycles spent in each phase.
forward : A = stage ids + score terms, B = softmax rows, C = gather stream (packed / general)
backward: A = row records into registers, B = column directory, C = per-edge coefficients, D = stream"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    lib = ctypes.CDLL(os.environ["DGNN_HIP_LIB"])
    # the forward and the backward kernels live in separate translation units, each with its own counters
    lib.dgnn_debug_phase_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    lib.dgnn_debug_phase_cycles_bwd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    from dgnn_amd.native import fused_gatconv as fg
    dev = torch.device("cuda:0")
    g = bench.build_graph(os.environ.get("DGNN_BENCH_WORKLOAD", "products"), dev, labels=os.environ.get("DGNN_BENCH_LABELS", "natural"))
    n, nnz = g["n"], g["nnz"]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    buf, bufb = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)()
    for shape in sys.argv[1:] or ["8x8", "1x16", "4x64"]:
        H, F = (int(v) for v in shape.split("x"))
        x = torch.rand((n, H, F), device=dev, generator=gen) - 0.4
        ar = torch.rand((n, H), device=dev, generator=gen) - 0.6
        ac = torch.rand((n, H), device=dev, generator=gen) - 0.4
        go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5
        for _ in range(2):
            out, emax, esum, emask, on, an = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=True)
            fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum,
                            emask, x, ar, ac, go, out_feat=out, out_neg=on, a_neg=an)
        lib.dgnn_debug_phase_cycles(buf, 1)
        lib.dgnn_debug_phase_cycles_bwd(bufb, 1)
        reps = 3
        for _ in range(reps):
            out, emax, esum, emask, on, an = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=True)
            fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum,
                            emask, x, ar, ac, go, out_feat=out, out_neg=on, a_neg=an)
        lib.dgnn_debug_phase_cycles(buf, 1)
        lib.dgnn_debug_phase_cycles_bwd(bufb, 1)
        f, b = [buf[i] / reps for i in range(4)], [bufb[i] / reps for i in range(8, 12)]
        tf, tb = sum(f) or 1, sum(b) or 1
        print(json.dumps({"shape": shape, "fwd_wave_Mcycles": round(tf / 1e6), "fwd_share": {k: round(c / tf, 3) for k, c in zip(("A_stage", "B_softmax", "C_packed", "C_general"), f)},
                          "fwd_cycles_per_edge": round(tf / nnz, 1),
                          "bwd_wave_Mcycles": round(tb / 1e6), "bwd_share": {k: round(c / tb, 3) for k, c in zip(("A_records", "B_columns", "C_coeff", "D_stream"), b)},
                          "bwd_cycles_per_edge": round(tb / nnz, 1)}), flush=True)
        del x, go, out


if __name__ == "__main__":
    main()
