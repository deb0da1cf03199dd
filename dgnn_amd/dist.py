"""Row-partitioned full-graph fused GAT over the GPUs of one node (one process per GPU,
``torch.distributed`` with backend ``nccl`` = RCCL over xGMI).

The reference is single-GPU only; this is new design:

* nodes are cut into ``world`` CONTIGUOUS ranges of (almost) equal edge count (split on the CSR
  ``row_ptr`` prefix, not on node count: R-MAT ids are skewed, equal node ranges would give rank
  0 almost half of the edges);
* rank p owns the destination rows AND the source columns of its node range:
    forward   works on  A[lo:hi, :]  (its CSR rows, global column ids): it needs the source
              features / source score terms of every node -> all-gather of the boundary-node
              rows (``in_feat`` [N,H,F] and ``attn_col`` [N,H]);
    backward  works on  A[:, lo:hi]  (its CSC columns, global row ids), one gather pass as on a
              single GPU: it needs ``grad_out`` rows and the packed per-row statistics
              (attn_row, edge_max, edge_sum, row_dot) of every destination -> two all-gathers;
              grad_feat / grad_attn_col come out local, grad_attn_row comes out as a partial
              [N,H] per rank -> one all-reduce, sliced to the local range.
* every rank runs exactly the single-GPU kernels (C ABI of include/dgnn_hip.h) on its block:
  the forward on an (n_loc x N) rectangle, the backward on an (N x n_loc) rectangle.

All-gathers are over UNEVEN row blocks.  With RCCL, ``dist.all_gather`` on views of the full
buffer moves each block once to every peer over its own xGMI link (direct, not a ring of
7 hops of the whole tensor); with gloo (CPU tests) a broadcast per rank is used.
"""
import os

import torch
import torch.distributed as dist


def partition_bounds(row_ptr, world):
    """Node range bounds [world+1] with ~equal nnz per range (on the host)."""
    rp = row_ptr.detach().to("cpu", torch.int64)
    n, nnz = rp.numel() - 1, int(rp[-1])
    targets = torch.tensor([nnz * p // world for p in range(1, world)], dtype=torch.int64)
    cuts = torch.searchsorted(rp, targets, right=False).clamp_(0, n)
    bounds = [0] + [int(c) for c in cuts] + [n]
    for i in range(1, len(bounds)):           # keep the ranges monotone
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


class _NativeOps:
    """The product path: hand-written HIP kernels through the C ABI (no fallback)."""

    def __init__(self):
        from .native import fused_gatconv, format_conversion
        self._gat = fused_gatconv
        self._fmt = format_conversion

    def gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat):
        out, emax, esum, _ = self._gat.gat_forward(attn_row, attn_col, row_ptr, col_ind, slope, in_feat, 0.0)
        return out, emax, esum

    def row_dot(self, grad, out):
        return self._gat.gat_row_dot(grad, out)

    def gat_backward(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                     in_feat, attn_row, attn_col, grad, row_dot):
        return self._gat.gat_backward(slope, 0.0, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max,
                                      edge_sum, None, in_feat, attn_row, attn_col, grad, row_dot=row_dot)

    def transpose(self, ptr, ind, n_cols):
        """(ptr, ind) = CSR of an (r x n_cols) matrix -> (ptr_T [n_cols+1], ind_T, perm)."""
        return self._fmt.csr2csc_int(ptr, ind, n_cols=n_cols)


class PartitionedGAT:
    """Fused GAT forward+backward on rank ``rank``'s share of a row-partitioned graph.

    ``graph`` holds the FULL index arrays (row_ptr, col_ind, col_ptr, row_ind as int32 tensors on
    ``device``); only this rank's blocks are kept afterwards.  ``ops`` is the compute backend
    (default: the HIP kernels); tests inject a CPU stand-in to exercise the partition and the
    collectives without a GPU.
    """

    def __init__(self, graph, world, rank, device, ops=None, group=None):
        self.world, self.rank, self.device, self.group = world, rank, device, group
        self.ops = ops if ops is not None else _NativeOps()
        row_ptr, col_ind = graph["row_ptr"], graph["col_ind"]
        col_ptr, row_ind = graph["col_ptr"], graph["row_ind"]
        self.n = row_ptr.numel() - 1
        self.bounds = partition_bounds(row_ptr, world)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_loc = lo, hi, hi - lo
        # forward block  A[lo:hi, :]  (CSR, global column ids)
        e0, e1 = int(row_ptr[lo]), int(row_ptr[hi])
        self.f_row_ptr = (row_ptr[lo:hi + 1] - row_ptr[lo]).to(torch.int32).contiguous()
        self.f_col_ind = col_ind[e0:e1].contiguous()
        # backward block  A[:, lo:hi]  (CSC, global row ids) and its own CSR ordering
        c0, c1 = int(col_ptr[lo]), int(col_ptr[hi])
        self.b_col_ptr = (col_ptr[lo:hi + 1] - col_ptr[lo]).to(torch.int32).contiguous()
        self.b_row_ind = row_ind[c0:c1].contiguous()
        b_row_ptr, b_col_ind, perm2 = self.ops.transpose(self.b_col_ptr, self.b_row_ind, self.n)
        self.b_row_ptr, self.b_col_ind = b_row_ptr, b_col_ind
        # perm2[j] = CSC position of the block's CSR entry j; the kernels want the inverse
        inv = torch.empty_like(perm2)
        inv[perm2.long()] = torch.arange(perm2.numel(), dtype=perm2.dtype, device=perm2.device)
        self.b_permute = inv
        self.nnz_fwd, self.nnz_bwd = e1 - e0, c1 - c0
        self._ctx = None

    # ------------------------------------------------------------------ collectives
    def _all_gather_rows(self, local, full):
        """full[bounds[q]:bounds[q+1]] <- rank q's ``local`` rows, for every q."""
        if self.world == 1:
            full.copy_(local)
            return full
        views = [full[self.bounds[q]:self.bounds[q + 1]] for q in range(self.world)]
        if dist.get_backend(self.group) == "nccl" and os.environ.get("DGNN_DIST_GATHER", "allgather") != "broadcast":
            dist.all_gather(views, local.contiguous(), group=self.group)   # uneven blocks: grouped sends
        else:
            for q in range(self.world):
                if q == self.rank:
                    views[q].copy_(local)
                dist.broadcast(views[q], src=q, group=self.group)
        return full

    def _full(self, local):
        return torch.empty((self.n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, in_feat_loc, attn_row_loc, attn_col_loc, slope):
        """Local rows of GATConvFuse(...): [n_loc,H,F]."""
        in_full = self._all_gather_rows(in_feat_loc, self._full(in_feat_loc))
        ac_full = self._all_gather_rows(attn_col_loc, self._full(attn_col_loc))
        out, emax, esum = self.ops.gat_forward(attn_row_loc.contiguous(), ac_full, self.f_row_ptr,
                                               self.f_col_ind, slope, in_full)
        self._ctx = (in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope)
        return out

    def backward(self, grad_out_loc):
        """-> (grad_feat_loc [n_loc,H,F], grad_attn_row_loc [n_loc,H], grad_attn_col_loc [n_loc,H])."""
        in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope = self._ctx
        grad_out_loc = grad_out_loc.contiguous()
        grad_full = self._all_gather_rows(grad_out_loc, self._full(grad_out_loc))
        S_loc = self.ops.row_dot(grad_out_loc, out)
        stat_loc = torch.stack((attn_row_loc, emax, esum, S_loc), dim=1)          # [n_loc,4,H]
        stat_full = self._all_gather_rows(stat_loc, self._full(stat_loc))
        stat = stat_full.permute(1, 0, 2).contiguous()                            # [4,N,H]
        gf, gr_partial, gc = self.ops.gat_backward(
            slope, self.b_row_ptr, self.b_col_ind, self.b_col_ptr, self.b_row_ind, self.b_permute,
            stat[1], stat[2], in_feat_loc.contiguous(), stat[0], attn_col_loc.contiguous(), grad_full,
            stat[3])
        if self.world > 1:
            dist.all_reduce(gr_partial, group=self.group)
        return gf, gr_partial[self.lo:self.hi].contiguous(), gc

    def step(self, in_feat, attn_row, attn_col, grad_out, slope):
        """One forward+backward given FULL tensors (every rank slices its own range): the unit
        bench.py times.  Returns the local gradients."""
        lo, hi = self.lo, self.hi
        self.forward(in_feat[lo:hi], attn_row[lo:hi], attn_col[lo:hi], slope)
        return self.backward(grad_out[lo:hi])
