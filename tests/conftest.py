import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, so a bare
    `pytest tests/` works on the CPU container too."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Which parity checks lean on the mass term of util.assert_close: per label, the number of
    elements compared, how many lie outside the plain rel/abs band (and so pass only through the
    4-ulp-of-mass term), and the worst error in units of the plain tolerance."""
    import json
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from util import TOLERANCE_TALLY
    except Exception:  # pragma: no cover
        return
    if not TOLERANCE_TALLY:
        return
    rows = {k: {"elements": v[0], "outside_plain_band": v[1], "worst_err_over_plain_tol": round(v[2], 3),
                "rtol": v[3], "atol": v[4]} for k, v in sorted(TOLERANCE_TALLY.items())}
    try:
        from util import LITERAL_TALLY
        for k, v in LITERAL_TALLY.items():      # the reference's own float32 arithmetic on the same inputs
            rows[k]["literal_f32_worst_err_over_plain_tol"] = round(v, 3)
        from util import LITERAL_RMS
        for k, (a2, l2, cnt) in LITERAL_RMS.items():
            if cnt:
                rows[k]["rms_err_in_ulp_of_mass"] = round((a2 / cnt) ** 0.5, 4)
                rows[k]["literal_f32_rms_err_in_ulp_of_mass"] = round((l2 / cnt) ** 0.5, 4)
    except Exception:  # pragma: no cover
        pass
    leaning = {k: v for k, v in rows.items() if v["outside_plain_band"]}
    terminalreporter.write_line(f"tolerance tally: {len(rows)} labelled checks, {len(leaning)} with elements "
                                f"outside the plain band (passing through the mass term)")
    for k, v in leaning.items():
        terminalreporter.write_line(f"  {k}: {v['outside_plain_band']} / {v['elements']} outside rel {v['rtol']:g} "
                                    f"abs {v['atol']:g}, worst {v['worst_err_over_plain_tol']}x"
                                    + (f" (literal float32 on the same inputs: {v['literal_f32_worst_err_over_plain_tol']}x)"
                                       if "literal_f32_worst_err_over_plain_tol" in v else ""))
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        tag = "gpu" if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "") else "cpu"
        with open(os.path.join(out, f"tolerance_report_{tag}.json"), "w") as fh:
            json.dump(rows, fh, indent=1)
    except OSError:  # pragma: no cover
        pass
