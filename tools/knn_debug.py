"""Debug view of the filtered k-NN workspace: flagged blocks, the certificate's two sides."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgnn_amd import _lib  # noqa: E402

L = _lib.lib() if hasattr(_lib, "lib") else None
batch, points, k, dim = (int(v) for v in (sys.argv[1:] or ["32", "2048", "20", "64"]))
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn((batch, points, dim), device=dev)
lib = ctypes.CDLL(os.path.join(ROOT, "dgnn_amd", "csrc", "libdgnn_hip.so"))
lib.dgnn_knn_graph_workspace_bytes.restype = ctypes.c_size_t
nb = lib.dgnn_knn_graph_workspace_bytes(batch, points, dim, k)
print("workspace bytes", nb)
ws = torch.zeros(nb, dtype=torch.uint8, device=dev)
out = torch.empty(batch * points * k, dtype=torch.int32, device=dev)
rc = lib.dgnn_knn_graph_ws(batch, points, dim, k, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                           ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(nb), None)
torch.cuda.synchronize()
print("rc", rc)
al = lambda v: (v + 255) // 256 * 256
kl = 24 if k <= 20 else 32 if k <= 28 else 48 if k <= 44 else 64
pp = (points + 31) // 32 * 32
dimp = 32 if dim <= 32 else 64 if dim <= 64 else 128
nq, npad = batch * points, batch * pp
o = 0
offs = {}
for name, size in (("xh", npad * dimp * 2), ("xl", npad * dimp * 2), ("nyk", npad * 4), ("cand", nq * kl * 4),
                   ("key", nq * 4), ("bad", batch * 4), ("flags", batch * ((points + 63) // 64) * 4), ("qflags", nq * 4), ("rflags", batch * ((points + 63) // 64) * 4)):
    offs[name] = (o, size)
    o = al(o + size)
assert o == nb, (o, nb)
view = lambda name, dt: ws[offs[name][0]:offs[name][0] + offs[name][1]].view(dt)
flags = view("flags", torch.int32)
print("blocks for the exhaustive kernel", int((flags != 0).sum()), "for the redo pass", int((view("rflags", torch.int32) != 0).sum()), "of", flags.numel(), "flagged queries",
      int((view("qflags", torch.int32) != 0).sum()), "of", nq, "bad clouds", int(view("bad", torch.int32).sum()))
key = view("key", torch.float32)
nyk = view("nyk", torch.float32).view(batch, pp)[:, :points].reshape(-1)
cand = view("cand", torch.int32).view(kl, nq)
print('src_ind[:20]', out[:20].tolist())
