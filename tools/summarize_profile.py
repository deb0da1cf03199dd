#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (tools/profile.sh) into the committed summaries:
profiles/<tag>_kernel_stats_<workload>.csv and profiles/<tag>_pmc_hbm_bytes_<workload>.csv."""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
workload = sys.argv[2] if len(sys.argv) > 2 else "products"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)

f = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if "dgnn::" in r[0]]
csv.writer(open(os.path.join(root, "profiles", f"{tag}_kernel_stats_{workload}.csv"), "w")).writerows(keep)
for r in keep[1:]:
    print(f"{r[0][:64]:64s} calls={r[1]:>3s} avg_us={float(r[3]) / 1e3:10.1f}")

out = [["kernel", "counter", "dispatches", "avg_counter_value_KB", "avg_bytes_corrected", "note"]]
for name, counter, corr in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
    f = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if "dgnn::" in r["Kernel_Name"]:
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        avg = sum(v) / len(v)
        note = ("x2: gfx950 FETCH_SIZE counts 128-B requests as 64 B (MI355X_MICROARCH.md, HBM); "
                "calibrated on k_bwd_rowstat's known 5.0 GB read") if corr == 2 else "exact for 16 B/lane stores"
        out.append([k, counter, len(v), f"{avg:.1f}", f"{avg * 1024 * corr:.4e}", note])
csv.writer(open(os.path.join(root, "profiles", f"{tag}_pmc_hbm_bytes_{workload}.csv"), "w")).writerows(out)
for r in out[1:]:
    if "gather" in r[0] or "rowstat" in r[0] or "row_sum" in r[0]:
        print(r[1], r[0][:50], r[4])
