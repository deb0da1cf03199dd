#!/usr/bin/env python3
"""Where the worst GmmConvFuse gradient elements sit: HIP and the literal float32 restatement against
float64, per width, with the column's degree and summand mass (diagnostic for tests/util.py's tally)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from util import parity_graphs  # noqa: E402
from dgnn_amd.operators import GmmConvFuse  # noqa: E402

g = parity_graphs()["powerlaw256"]
dev = torch.device("cuda:0")
G = {k: torch.from_numpy(g[k]).to(dev) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")}
k, d = 3, 2
for f in (16, 41, 7):
    rng = np.random.default_rng(4)
    node_feat = rng.standard_normal((g["n"], k, f)).astype(np.float32)
    pseudo = rng.random((g["nnz"], d), dtype=np.float32)
    mu = (0.1 * rng.standard_normal((k, d))).astype(np.float32)
    inv_sigma = (1 + 0.1 * rng.standard_normal((k, d))).astype(np.float32)
    go = rng.standard_normal((g["n"], k, f)).astype(np.float32)
    x, ps, m_, sg = (torch.from_numpy(t).to(dev).requires_grad_() for t in (node_feat, pseudo, mu, inv_sigma))
    out = GmmConvFuse(G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"], G["permute"], x, ps, m_, sg)
    out.backward(torch.from_numpy(go).to(dev))
    psc = pseudo[g["permute"]]
    want = oracle.gmmconv_f64(g["col_ptr"], g["row_ind"], go, psc, mu, inv_sigma)
    mass = oracle.gmmconv_f64(g["col_ptr"], g["row_ind"], np.abs(go), psc, mu, inv_sigma)
    lit = oracle.gmmconv(g["col_ptr"], g["row_ind"], go, psc, mu, inv_sigma)
    plain = 1e-6 + 1e-5 * np.abs(want)
    deg = np.diff(g["col_ptr"])
    for name, got in (("hip", x.grad.cpu().numpy()), ("literal", lit)):
        r = np.abs(got - want) / plain
        i = np.unravel_index(np.argmax(r), r.shape)
        ulp_mass = np.abs(got - want) / (1.1920929e-07 * mass + 1e-30)
        print(f"f={f} {name}: worst {r.max():.3f}x plain at {i} value {want[i]:.3e} err {abs(got[i] - want[i]):.2e} mass {mass[i]:.2f} "
              f"column degree {deg[i[0]]}; err/ulp(mass): worst {ulp_mass.max():.2f} mean {ulp_mass.mean():.3f} rms {np.sqrt((ulp_mass**2).mean()):.3f}")
