#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_<name>/ (tools/profile.sh) into the committed summaries:
profiles/<tag>_kernel_stats_<name>.csv, profiles/<tag>_pmc_hbm_bytes_<name>.csv and, when the TCC
pass ran, profiles/<tag>_pmc_l2_hit_<name>.csv (hits / (hits + misses) of the L2 per kernel)."""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
name = sys.argv[2] if len(sys.argv) > 2 else "products"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{name}")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)

f = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if "dgnn::" in r[0]]
csv.writer(open(os.path.join(root, "profiles", f"{tag}_kernel_stats_{name}.csv"), "w")).writerows(keep)
for r in keep[1:]:
    print(f"{r[0][:72]:72s} calls={r[1]:>4s} avg_us={float(r[3]) / 1e3:10.1f}")


def per_kernel(sub):
    """{counter: {kernel: [values per dispatch]}} of one --pmc pass."""
    fs = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    acc = {}
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if "dgnn::" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return acc


out = [["kernel", "counter", "dispatches", "avg_counter_value_KB", "avg_bytes_corrected", "note"]]
for sub, counter, corr in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
    for k, v in per_kernel(sub).get(counter, {}).items():
        avg = sum(v) / len(v)
        note = ("x2: gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM); "
                "calibrated on k_bwd_rowstat's known 5.0 GB read") if corr == 2 else "exact for 16 B/lane stores"
        out.append([k, counter, len(v), f"{avg:.1f}", f"{avg * 1024 * corr:.4e}", note])
csv.writer(open(os.path.join(root, "profiles", f"{tag}_pmc_hbm_bytes_{name}.csv"), "w")).writerows(out)
for r in out[1:]:
    if float(r[4]) > 5e7:
        print(f"{r[1]:10s} {r[0][:60]:60s} {float(r[4]) / 1e9:8.3f} GB")

tcc = per_kernel("tcc")
if tcc:
    hit, miss = tcc.get("TCC_HIT_sum", {}), tcc.get("TCC_MISS_sum", {})
    rows = [["kernel", "dispatches", "avg_TCC_HIT_sum", "avg_TCC_MISS_sum", "l2_hit_rate"]]
    for k in hit:
        h, m = sum(hit[k]) / len(hit[k]), sum(miss.get(k, [0])) / max(len(miss.get(k, [0])), 1)
        rows.append([k, len(hit[k]), f"{h:.0f}", f"{m:.0f}", f"{h / max(h + m, 1):.4f}"])
        if h + m > 1e6:
            print(f"L2 hit {h / max(h + m, 1):6.3f}  {k[:70]}")
    csv.writer(open(os.path.join(root, "profiles", f"{tag}_pmc_l2_hit_{name}.csv"), "w")).writerows(rows)

sq = per_kernel("sq")
if sq:
    names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_WAVES"]
    rows = [["kernel", "dispatches"] + [f"avg_{c}" for c in names]]
    kernels = sorted({k for c in sq.values() for k in c})
    for k in kernels:
        vals = [sum(sq.get(c, {}).get(k, [0])) / max(len(sq.get(c, {}).get(k, [0])), 1) for c in names]
        rows.append([k, len(sq.get(names[0], {}).get(k, []))] + [f"{v:.0f}" for v in vals])
        if vals[0] > 1e7:
            print(f"SQ wave-instructions  VALU {vals[0]:.3e}  SALU {vals[1]:.3e}  LDS {vals[2]:.3e}  SMEM {vals[3]:.3e}  "
                  f"VMEM_RD {vals[4]:.3e}  waves {vals[5]:.0f}  {k[:60]}")
    csv.writer(open(os.path.join(root, "profiles", f"{tag}_pmc_sq_{name}.csv"), "w")).writerows(rows)
