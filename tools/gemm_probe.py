#!/usr/bin/env python3
"""Probe: the weight gradient of the GAT layer's projection, dW = feat^T @ grad ([in, N] x [N, H*F] with
N = 2.4 M): one GEMM whose reduction runs over N, against the same product split over row blocks with a
batched GEMM (split-K by hand) and summed."""
import time
import torch
dev = "cuda:0"
n = 2449029
for k_in in (100, 256, 602):
    x = torch.randn(n, k_in, device=dev); g = torch.randn(n, 256, device=dev)

    def t(fn, it=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - s) / it * 1e3

    def splitk(b):
        m = n // b * b
        p = torch.bmm(x[:m].view(b, n // b, k_in).transpose(1, 2), g[:m].view(b, n // b, 256)).sum(0)
        return p if m == n else p.addmm_(x[m:].t(), g[m:])
    ref = x.t() @ g
    print(k_in, "x.T@g", round(t(lambda: x.t() @ g), 3), "(g.T@x).T", round(t(lambda: (g.t() @ x).t()), 3),
          {b: (round(t(lambda: splitk(b)), 3), float((splitk(b) - ref).abs().max() / ref.abs().max())) for b in (16, 64, 256, 1024)})
    del x, g
