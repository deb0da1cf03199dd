"""gpurun_out/tolerance_report_<tag>.json (written by tests/conftest.py at the end of a pytest run)
-> a markdown table: per labelled check, how many elements sit outside the plain rel 1e-5 / abs 1e-6
band and pass only through the summand-mass term.   python tools/tolerance_table.py gpu > profiles/..."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "gpu"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(root, "gpurun_out", f"tolerance_report_{tag}.json")))
tot = sum(v["elements"] for v in d.values())
out = sum(v["outside_plain_band"] for v in d.values())
print(f"# Tolerance tally ({tag} run): {len(d)} labelled checks, {tot} elements, {out} outside the plain band\n")
print("Every check is `|got - want| <= 1e-5 |want| + 1e-6 + 4 ulp(float32) * mass`; `outside` counts elements that")
print("need the mass term, `worst` is the largest error in units of the plain band's tolerance.\n")
print("`literal f32`: the oracle's literal float32 restatement of the reference's arithmetic on the SAME inputs, same")
print("yardstick (worst / plain tol; rms of err / ulp(mass) over all elements, ours | literal) -- tests assert")
print("ours <= max(1, 2 x literal) on the worst and <= 1.25 x literal on the rms.\n")
print("| check | elements | outside plain band | worst / plain tol | literal f32 worst | rms err in ulp(mass): ours, literal |")
print("|---|---:|---:|---:|---:|---:|")
for k, v in sorted(d.items(), key=lambda kv: (-kv[1]["outside_plain_band"], kv[0])):
    if v["outside_plain_band"] == 0 and "literal_f32_worst_err_over_plain_tol" not in v:
        continue
    lit = v.get("literal_f32_worst_err_over_plain_tol")
    rms = (f"{v['rms_err_in_ulp_of_mass']:.3f}, {v['literal_f32_rms_err_in_ulp_of_mass']:.3f}"
           if "rms_err_in_ulp_of_mass" in v else "")
    print(f"| {k} | {v['elements']} | {v['outside_plain_band']} | {v['worst_err_over_plain_tol']:.2f} | "
          f"{'' if lit is None else format(lit, '.2f')} | {rms} |")
inside = [k for k, v in d.items() if v["outside_plain_band"] == 0]
worst_in = max((d[k]["worst_err_over_plain_tol"] for k in inside), default=0.0)
print(f"\n{len(inside)} further checks have no element outside the plain band (worst {worst_in:.2f} of it): "
      + "; ".join(sorted(inside)))
nonstd = {k: (v["rtol"], v["atol"]) for k, v in d.items() if (v["rtol"], v["atol"]) != (1e-5, 1e-6)}
if nonstd:
    print("\nChecks with a band other than rel 1e-5 / abs 1e-6: " + "; ".join(f"{k} {v}" for k, v in sorted(nonstd.items())))
