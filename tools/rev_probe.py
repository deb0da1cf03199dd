import torch, time, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dgnn_amd.native import format_conversion as fc, knn, fused_edgeconv as fe
dev = "cuda:0"
B, P, k = 32, 2048, 20
x = torch.randn(B, P, 3, device=dev)
src = knn.knn_graph(x, k)
M = B * P
rp = torch.arange(0, M * k + 1, k, dtype=torch.int32, device=dev)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / it * 1e3
print("csr2csc of the k-NN lists ms", round(t(lambda: fc.csr2csc_int(rp, src, n_cols=M)), 4))
for F in (64, 128, 256):
    hs, hd = torch.randn(M, F, device=dev), torch.randn(M, F, device=dev)
    out, idx_f, idx_i = fe.edgeconv_forward(k, src, hs, hd, return_int_idx=True)
    go = torch.randn(M, F, device=dev)
    print(F, "atomic backward ms", round(t(lambda: fe.edgeconv_backward(go, idx_i)), 4))
    print(F, "gathered backward ms", round(t(lambda: fe.edgeconv_backward(go, idx_i, src_ind=src)), 4))
