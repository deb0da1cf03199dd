"""Worker for the multi-process tests of dgnn_amd.dist (spawned by test_dist.py).

mode "cpu": gloo backend, CPU tensors, compute through a CPU stand-in built on the oracle
            (test infrastructure: exercises the partition, block extraction, inverse
            permutation and collectives without a GPU).
mode "cpuneg": the same with a stand-in that also produces the forward's negative-score pair, so the
            runner takes the path on which a row's owner computes grad_attn_row alone (no return
            exchange; dgnn_amd/dist.py).
mode "cpunegmix": the pair is missing on rank 1 only: every rank must fall back together.
mode "gpu": gloo backend, every rank on cuda:0, compute through the real HIP kernels.
mode "nccl": rank r on cuda:r with the nccl (= RCCL) backend: the multi-GPU path as it runs in
            production; needs a GPU per rank (tests skip themselves on a one-GPU box).
mode "cpuheads" / "gpuheads" / "ncclheads": the head-parallel runner (dgnn_amd.dist.HeadParallelGAT through
            make_gat_runner): every rank runs the whole graph on its heads, nothing is exchanged; gloo on the CPU
            (oracle stand-in), gloo with every rank on cuda:0, RCCL with a GPU per rank.
mode "nccl1": ONE rank on cuda:0 with the nccl (= RCCL) backend and DGNN_DIST_FORCE_COLLECTIVES=1:
            every collective of the step is really issued through RCCL (RCCL wants one GPU per
            rank, so one rank is what a one-GPU box can rehearse).
Each rank writes its local results to <out_dir>/rank<r>.npz; the parent assembles and checks.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleOps:
    """CPU stand-in for the kernel set PartitionedGAT needs (rectangular blocks)."""

    def __init__(self):
        import oracle
        self.o = oracle

    @staticmethod
    def _t(a, dtype=None):
        return torch.from_numpy(np.ascontiguousarray(a if dtype is None else a.astype(dtype)))

    @staticmethod
    def _emit(results, out):
        """Honour the `out=` convention of the native module: write into the given tensors."""
        if out is not None:
            results = tuple(r if o is None else o.copy_(r) for r, o in zip(results, out))
        return results

    def gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat, out=None):
        res = self.o.gat_forward(attn_row.numpy(), attn_col.numpy(), row_ptr.numpy(),
                                 col_ind.numpy(), slope, in_feat.numpy(), 0.0)
        return self._emit(tuple(self._t(r) for r in res), out)

    def row_dot(self, grad, out):
        return (grad.double() * out.double()).sum(-1).float()

    def transpose(self, ptr, ind, n_cols):
        cp, ri, pm = self.o.csr2csc(ptr.numpy(), ind.numpy(), n_cols)
        return self._t(cp), self._t(ri), self._t(pm)

    def gat_backward_packed(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat, in_feat,
                            attn_col, grad, out=None):
        """Stand-in for the packed entry point: odd row counts take the `None` (unsupported) way so
        that the runner's unpacking fallback is exercised too."""
        if row_stat.size(0) % 2 == 1:
            return None
        st = row_stat.permute(2, 0, 1).contiguous()
        return self.gat_backward(slope, row_ptr, col_ind, col_ptr, row_ind, permute, st[1], st[2], in_feat,
                                 st[0], attn_col, grad, st[3], out=out)

    def gat_backward(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                     in_feat, attn_row, attn_col, grad, row_dot, out=None):
        """Same dataflow as the HIP backward: one pass over the CSC, d written to CSR order through
        `permute`, row sums over the block's own row_ptr."""
        cp, ri, pm = col_ptr.numpy().astype(np.int64), row_ind.numpy().astype(np.int64), permute.numpy().astype(np.int64)
        rp = row_ptr.numpy().astype(np.int64)
        n, m = cp.size - 1, rp.size - 1
        col = np.repeat(np.arange(n), np.diff(cp))
        ar, ac = attn_row.double().numpy(), attn_col.double().numpy()
        em, es, S = edge_max.double().numpy(), edge_sum.double().numpy(), row_dot.double().numpy()
        x, g = in_feat.double().numpy(), grad.double().numpy()
        e = ar[ri] + ac[col]
        neg = e <= 0
        e = np.where(e > 0, e, e * slope)
        a = np.exp(e - em[ri]) / es[ri]
        dots = np.einsum("ehf,ehf->eh", g[ri], x[col])
        d = a * (dots - S[ri]) * np.where(e < 0, slope, 1.0)
        gf = np.zeros_like(x)
        np.add.at(gf, col, a[:, :, None] * g[ri])
        gc = np.zeros_like(ac)
        np.add.at(gc, col, d)
        dE = np.zeros_like(d)
        dE[pm] = d                                   # CSC order -> the block's CSR order
        assert np.array_equal(np.sort(pm), np.arange(pm.size)), "permute must be a permutation"
        rows_csr = np.repeat(np.arange(m), np.diff(rp))
        assert np.array_equal(rows_csr[pm], ri), "permute / row_ptr of the backward block disagree"
        gr = np.zeros((m, ar.shape[1]))
        np.add.at(gr, rows_csr, dE)
        return self._emit((self._t(gf, np.float32), self._t(gr, np.float32), self._t(gc, np.float32)), out)


class OracleOpsNeg(OracleOps):
    """The stand-in with the negative-score pair of include/dgnn_hip.h (dgnn_gat_forward): out_neg =
    the aggregate over the edges whose pre-activation score is negative, a_neg = the sum of their
    softmax weights; the packed backward can be asked to leave grad_attn_row out."""
    supports_neg = True

    def gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat, out=None, neg=False):
        base = super().gat_forward(attn_row, attn_col, row_ptr, col_ind, slope, in_feat)
        if not neg:
            return self._emit(base, out)
        rp, ci = row_ptr.numpy().astype(np.int64), col_ind.numpy().astype(np.int64)
        row = np.repeat(np.arange(rp.size - 1), np.diff(rp))
        e0 = attn_row.double().numpy()[row] + attn_col.double().numpy()[ci]
        e = np.where(e0 > 0, e0, e0 * slope)
        a = np.exp(e - base[1].double().numpy()[row]) / base[2].double().numpy()[row]
        an_e = np.where(e < 0, a, 0.0)
        a_neg = np.zeros((rp.size - 1, e.shape[1]))
        np.add.at(a_neg, row, an_e)
        out_neg = np.zeros((rp.size - 1,) + tuple(in_feat.shape[1:]))
        np.add.at(out_neg, row, an_e[:, :, None] * in_feat.double().numpy()[ci])
        return self._emit(base + (self._t(out_neg, np.float32), self._t(a_neg, np.float32)), out)

    def gat_backward_packed(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat, in_feat,
                            attn_col, grad, out=None, need_grad_attn_row=True):
        if need_grad_attn_row:
            return super().gat_backward_packed(slope, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat,
                                               in_feat, attn_col, grad, out=out)
        st = row_stat.permute(2, 0, 1).contiguous()
        gf, _, gc = self.gat_backward(slope, row_ptr, col_ind, col_ptr, row_ind, permute, st[1], st[2], in_feat,
                                      st[0], attn_col, grad, st[3])
        return self._emit((gf, None, gc), out)


class OracleOpsNegOnSomeRanks(OracleOpsNeg):
    """The pair exists on every rank but `missing_on`: what misaligned chunk cuts or an empty chunk
    used to cause on the GPU.  The runner has to notice and take the returning path on
    EVERY rank; ranks that disagree pair an all-to-all with nothing and hang."""

    def __init__(self, rank, missing_on=1):
        super().__init__()
        self.missing = rank == missing_on

    def gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat, out=None, neg=False):
        if neg and self.missing:
            base = OracleOps.gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat)
            return self._emit(base, None if out is None else out[:3]) + (None, None)
        return super().gat_forward(attn_row, attn_col, row_ptr, col_ind, slope, in_feat, out=out, neg=neg)


def run(rank, world, mode, port, out_dir, graph_spec, h, f):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if mode == "nccl1":      # one rank, RCCL backend, collectives forced: rehearses the calls on one GPU
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    elif mode in ("nccl", "ncclheads"):
        from datetime import timedelta
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"),
                                timeout=timedelta(seconds=120))
    else:
        from datetime import timedelta
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    from util import gat_inputs, rmat_graph
    from dgnn_amd.dist import PartitionedGAT, make_gat_runner
    g = rmat_graph(*graph_spec)
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed=21)
    device = torch.device("cpu" if mode in ("cpu", "cpuneg", "cpunegmix", "cpuheads") else
                          f"cuda:{rank}" if mode in ("nccl", "ncclheads") else "cuda:0")
    G = {k: torch.from_numpy(g[k]).to(device) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind")}
    if mode.endswith("heads"):
        # default strategy: the heads divide by the world size, so the factory must pick the head split itself
        runner = make_gat_runner(G, world, rank, device, h, feat=f, ops=OracleOpsNeg() if mode == "cpuheads" else None)
        assert runner.mode == "heads" and (runner.hi - runner.lo) * world == h * runner.q
        t = lambda a: torch.from_numpy(a).to(device)
        f_loc = f if runner.f_lo is None else runner.f_hi - runner.f_lo
        x_loc = runner.input_buffer("forward", (runner.hi - runner.lo, f_loc))
        x_loc.copy_(runner.local(t(in_feat)))
        out = runner.forward(x_loc, runner.local_scores(t(attn_row)), runner.local_scores(t(attn_col)), 0.2)
        gf, gr, gc = runner.backward(runner.local(t(grad)))
        out2 = runner.forward(x_loc, runner.local_scores(t(attn_row)), runner.local_scores(t(attn_col)), 0.2)
        gf2, gr2, gc2 = runner.backward(runner.local(t(grad)))
        for u, v in ((out, out2), (gf, gf2), (gr, gr2), (gc, gc2)):
            assert torch.equal(u, v), "the second step must repeat the first bit for bit"
        if device.type == "cuda":
            torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=runner.lo, hi=runner.hi,
                 f_lo=0 if runner.f_lo is None else runner.f_lo, f_hi=f if runner.f_hi is None else runner.f_hi,
                 out=out.cpu().numpy(), gf=gf.cpu().numpy(), gr=gr.cpu().numpy(), gc=gc.cpu().numpy(),
                 nnz_fwd=runner.nnz_fwd, nnz_bwd=runner.nnz_bwd)
        dist.barrier()
        dist.destroy_process_group()
        return
    ops = (OracleOps() if mode == "cpu" else OracleOpsNeg() if mode == "cpuneg" else
           OracleOpsNegOnSomeRanks(rank) if mode == "cpunegmix" else None)
    runner = PartitionedGAT(G, world, rank, device, ops=ops)
    lo, hi = runner.lo, runner.hi
    t = lambda a: torch.from_numpy(a).to(device)
    # the layer input is produced straight into the runner's gather table (no copy of local rows);
    # the output gradient goes the ordinary way (copied in)
    x_loc = runner.input_buffer("forward", (h, f))
    x_loc.copy_(t(in_feat[lo:hi]))
    out = runner.forward(x_loc, t(attn_row[lo:hi]), t(attn_col[lo:hi]), 0.2)
    gf, gr, gc = runner.backward(t(grad[lo:hi]))
    if mode == "cpuneg" and runner.mode == "halo":
        assert runner._neg is not None, "the stand-in offers the pair: the runner must take the owner-local path"
    if mode == "cpunegmix" and runner.mode == "halo" and world > 1:
        assert runner._neg is None, "one rank lacks the pair: NO rank may take the owner-local path"
    if mode in ("gpu", "nccl") and runner.mode == "halo" and f >= 4 and h <= 8:
        assert runner._neg is not None, "float4 rows: the HIP forward must produce the pair on every rank"
    if mode in ("cpuneg", "cpunegmix", "gpu", "nccl"):      # a second step takes the remembered decision
        out2 = runner.forward(x_loc, t(attn_row[lo:hi]), t(attn_col[lo:hi]), 0.2)
        gf2, gr2, gc2 = runner.backward(t(grad[lo:hi]))
        for u, v in ((out, out2), (gf, gf2), (gr, gr2), (gc, gc2)):
            assert torch.equal(u, v), "the second step must repeat the first bit for bit"
    if device.type == "cuda":
        torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, out=out.cpu().numpy(),
             gf=gf.cpu().numpy(), gr=gr.cpu().numpy(), gc=gc.cpu().numpy(),
             nnz_fwd=runner.nnz_fwd, nnz_bwd=runner.nnz_bwd)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rank, world, mode, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
    spec = tuple(int(v) for v in sys.argv[6].split(","))
    run(rank, world, mode, port, out_dir, spec, int(sys.argv[7]), int(sys.argv[8]))
