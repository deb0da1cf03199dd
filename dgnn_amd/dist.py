"""Row-partitioned full-graph fused GAT over the GPUs of one node (one process per GPU,
``torch.distributed`` with backend ``nccl`` = RCCL over xGMI).

The reference is single-GPU only; this is new design:

* nodes are cut into ``world`` CONTIGUOUS ranges of (almost) equal edge count (split on the CSR
  ``row_ptr`` prefix, not on node count: R-MAT ids are skewed, equal node ranges would give rank
  0 almost half of the edges);
* rank p owns the destination rows AND the source columns of its node range:
    forward   works on  A[lo:hi, :]  (its CSR rows): it needs the features / score terms of the
              BOUNDARY nodes = the distinct sources its rows reference;
    backward  works on  A[:, lo:hi]  (its CSC columns), one gather pass as on a single GPU: it
              needs ``grad_out`` rows and the packed per-row statistics (attn_row, edge_max,
              edge_sum, row_dot) of the distinct destinations its columns reference;
              grad_feat / grad_attn_col come out local, grad_attn_row comes out as partial sums
              for those destinations and travels back to their owners.
* every rank runs exactly the single-GPU kernels (C ABI of include/dgnn_hip.h) on its block:
  the forward on an (n_loc x n_halo) rectangle, the backward on an (n_halo x n_loc) rectangle.

Exchange (``mode``):
  "halo" (default)  only the rows a rank actually references move: index lists are exchanged
                    once at set-up, every step is one ``all_to_all_single`` with split sizes per
                    tensor (direct peer-to-peer over each pair's own xGMI link).  The received
                    rows arrive ordered by owner rank = globally sorted by node id, so the
                    receive buffer IS the compact feature table the kernels gather from (column
                    ids are relabelled to it once at set-up): no scatter pass.  On the
                    products-shaped R-MAT graph at 8 ranks a rank needs ~35 % of the remote rows.
  "allgather"       every rank receives every row (``dist.all_gather`` on uneven row blocks).
Set with ``DGNN_DIST_MODE``.  gloo (CPU tests) takes the same code path; CUDA tensors under a
non-NCCL backend are staged through the host.
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


class _Halo:
    """Who needs which of my rows, and which rows I need, for one direction of the exchange."""

    def __init__(self, owner, needed_ids):
        """needed_ids: sorted unique GLOBAL node ids (int64) this rank references."""
        self.n_needed = needed_ids.numel()
        b = torch.tensor(owner.bounds, dtype=torch.int64, device=needed_ids.device)
        cut = torch.searchsorted(needed_ids, b)                      # [world+1]
        self.recv_counts = [int(c) for c in (cut[1:] - cut[:-1]).cpu()]
        send_counts = owner._a2a_counts(self.recv_counts)
        self.send_counts = send_counts
        wanted = torch.empty(sum(send_counts), dtype=torch.int64, device=needed_ids.device)
        owner._a2a(wanted, needed_ids, send_counts, self.recv_counts)   # ids the peers want from me
        self.send_idx = (wanted - owner.lo).contiguous()                # local row indices, by peer
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < owner.n_loc)


class PartitionedGAT:
    """Fused GAT forward+backward on rank ``rank``'s share of a row-partitioned graph.

    ``graph`` holds the FULL index arrays (row_ptr, col_ind, col_ptr, row_ind as int32 tensors on
    ``device``); only this rank's blocks are kept afterwards.  ``ops`` is the compute backend
    (default: the HIP kernels); tests inject a CPU stand-in to exercise the partition and the
    collectives without a GPU.
    """

    def __init__(self, graph, world, rank, device, ops=None, group=None, mode=None):
        self.world, self.rank, self.device, self.group = world, rank, device, group
        self.ops = ops if ops is not None else _NativeOps()
        self.mode = mode or os.environ.get("DGNN_DIST_MODE", "halo")
        # one rank needs no collectives; DGNN_DIST_FORCE_COLLECTIVES=1 issues them anyway (a
        # single-GPU rehearsal of the RCCL calls: tests/test_dist.py)
        self._solo = world == 1 and os.environ.get("DGNN_DIST_FORCE_COLLECTIVES") != "1"
        if self.mode not in ("halo", "allgather"):
            raise ValueError(f"unknown exchange mode {self.mode!r}")
        row_ptr, col_ind = graph["row_ptr"], graph["col_ind"]
        col_ptr, row_ind = graph["col_ptr"], graph["row_ind"]
        self.n = row_ptr.numel() - 1
        self.bounds = partition_bounds(row_ptr, world)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_loc = lo, hi, hi - lo
        # forward block  A[lo:hi, :]  (CSR)
        e0, e1 = int(row_ptr[lo]), int(row_ptr[hi])
        self.f_row_ptr = (row_ptr[lo:hi + 1] - row_ptr[lo]).to(torch.int32).contiguous()
        f_col = col_ind[e0:e1].contiguous()
        # backward block  A[:, lo:hi]  (CSC)
        c0, c1 = int(col_ptr[lo]), int(col_ptr[hi])
        self.b_col_ptr = (col_ptr[lo:hi + 1] - col_ptr[lo]).to(torch.int32).contiguous()
        b_row = row_ind[c0:c1].contiguous()
        self.nnz_fwd, self.nnz_bwd = e1 - e0, c1 - c0
        if self.mode == "halo":
            uf = torch.unique(f_col.long())                           # sorted distinct sources
            self.f_col_ind = torch.searchsorted(uf, f_col.long()).to(torch.int32).contiguous()
            self.f_halo = _Halo(self, uf)
            ub = torch.unique(b_row.long())                           # sorted distinct destinations
            self.b_row_ind = torch.searchsorted(ub, b_row.long()).to(torch.int32).contiguous()
            self.b_halo = _Halo(self, ub)
            n_rows_b = ub.numel()
        else:
            self.f_col_ind, self.b_row_ind, n_rows_b = f_col, b_row, self.n
        # the backward block's own CSR ordering (for the per-row sum of the edge gradients)
        b_row_ptr, b_col_ind, perm2 = self.ops.transpose(self.b_col_ptr, self.b_row_ind, n_rows_b)
        self.b_row_ptr, self.b_col_ind = b_row_ptr, b_col_ind
        # perm2[j] = CSC position of the block's CSR entry j; the kernels want the inverse
        inv = torch.empty_like(perm2)
        inv[perm2.long()] = torch.arange(perm2.numel(), dtype=perm2.dtype, device=perm2.device)
        self.b_permute = inv
        self._ctx = None

    # ------------------------------------------------------------------ collectives
    def _nccl(self):
        return (not self._solo) and dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def _a2a_counts(self, counts):
        if self._solo:
            return list(counts)
        dev = self.device if self._nccl() else torch.device("cpu")
        inp = torch.tensor(counts, dtype=torch.int64, device=dev)
        out = torch.empty_like(inp)
        dist.all_to_all_single(out, inp, group=self.group)
        return [int(c) for c in out.cpu()]

    def _a2a(self, out, inp, out_splits, in_splits):
        """out (rows grouped by source rank) <- inp (rows grouped by destination rank)."""
        if self._solo:
            out.copy_(inp)
            return out
        if self._nccl() or not inp.is_cuda:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)
        else:                                   # e.g. gloo with device tensors: stage through the host
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.contiguous().cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        return out

    def _fetch(self, halo, local):
        """Rows of every peer that this rank references (plus its own), in node-id order."""
        send = local.index_select(0, halo.send_idx)
        table = torch.empty((halo.n_needed,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        return self._a2a(table, send, halo.recv_counts, halo.send_counts)

    def _return_rows(self, halo, partial):
        """Inverse of _fetch for additive results: partial [n_needed,...] rows go back to their
        owners and are summed there in a fixed (peer) order."""
        back = torch.empty((halo.send_idx.numel(),) + tuple(partial.shape[1:]), dtype=partial.dtype,
                           device=partial.device)
        self._a2a(back, partial, halo.send_counts, halo.recv_counts)
        acc = torch.zeros((self.n_loc,) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
        off = 0
        for cnt in halo.send_counts:              # ids are distinct inside one peer's list
            if cnt:
                idx = halo.send_idx[off:off + cnt]
                acc[idx] += back[off:off + cnt]
            off += cnt
        return acc

    def _all_gather_rows(self, local):
        """full[bounds[q]:bounds[q+1]] <- rank q's ``local`` rows, for every q."""
        full = torch.empty((self.n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if self._solo:
            full.copy_(local)
            return full
        views = [full[self.bounds[q]:self.bounds[q + 1]] for q in range(self.world)]
        if self._nccl() and os.environ.get("DGNN_DIST_GATHER", "allgather") != "broadcast":
            dist.all_gather(views, local.contiguous(), group=self.group)   # uneven blocks: grouped sends
        else:
            for q in range(self.world):
                if q == self.rank:
                    views[q].copy_(local)
                dist.broadcast(views[q], src=q, group=self.group)
        return full

    def _rows_for_forward(self, local):
        return self._fetch(self.f_halo, local) if self.mode == "halo" else self._all_gather_rows(local)

    def _rows_for_backward(self, local):
        return self._fetch(self.b_halo, local) if self.mode == "halo" else self._all_gather_rows(local)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, in_feat_loc, attn_row_loc, attn_col_loc, slope):
        """Local rows of GATConvFuse(...): [n_loc,H,F]."""
        in_tab = self._rows_for_forward(in_feat_loc.contiguous())
        ac_tab = self._rows_for_forward(attn_col_loc.contiguous())
        out, emax, esum = self.ops.gat_forward(attn_row_loc.contiguous(), ac_tab, self.f_row_ptr,
                                               self.f_col_ind, slope, in_tab)
        self._ctx = (in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope)
        return out

    def backward(self, grad_out_loc):
        """-> (grad_feat_loc [n_loc,H,F], grad_attn_row_loc [n_loc,H], grad_attn_col_loc [n_loc,H])."""
        in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope = self._ctx
        grad_out_loc = grad_out_loc.contiguous()
        grad_tab = self._rows_for_backward(grad_out_loc)
        S_loc = self.ops.row_dot(grad_out_loc, out)
        stat_loc = torch.stack((attn_row_loc, emax, esum, S_loc), dim=1).contiguous()   # [n_loc,4,H]
        stat = self._rows_for_backward(stat_loc).permute(1, 0, 2).contiguous()          # [4,rows,H]
        gf, gr_partial, gc = self.ops.gat_backward(
            slope, self.b_row_ptr, self.b_col_ind, self.b_col_ptr, self.b_row_ind, self.b_permute,
            stat[1], stat[2], in_feat_loc.contiguous(), stat[0], attn_col_loc.contiguous(), grad_tab,
            stat[3])
        if self.mode == "halo":
            gr = self._return_rows(self.b_halo, gr_partial)
        else:
            if not self._solo:
                dist.all_reduce(gr_partial, group=self.group)
            gr = gr_partial[self.lo:self.hi].contiguous()
        return gf, gr, gc

    def step(self, in_feat, attn_row, attn_col, grad_out, slope):
        """One forward+backward given FULL tensors (every rank slices its own range): the unit
        bench.py times.  Returns the local gradients."""
        lo, hi = self.lo, self.hi
        self.forward(in_feat[lo:hi], attn_row[lo:hi], attn_col[lo:hi], slope)
        return self.backward(grad_out[lo:hi])


# ----------------------------------------------------------------------------------------------
# EdgeConv / DGCNN: data parallel over point clouds (BASELINE.json configs[2]; SURVEY.md 8(e)
# "replicas only").  The k-NN graph never crosses clouds (script/train/train_edgeconv.py:49), so
# the fused operators need no exchange at all: every rank runs the single-GPU kernels on its own
# clouds and only the WEIGHT gradients are summed, once per step.
def cloud_range(batch, world, rank):
    """[lo, hi) of the clouds rank ``rank`` owns: contiguous, sizes differ by at most one."""
    return batch * rank // world, batch * (rank + 1) // world


class CloudDataParallel:
    """Replicas of ``module`` (one per rank), each stepping on its own slice of the batch.

    * parameters and buffers are broadcast from rank 0 once, as ONE flat message per dtype;
    * ``local(x)`` slices a [batch, ...] tensor to this rank's clouds;
    * the loss of a rank is the mean over ITS clouds; ``reduce_gradients()`` turns the local
      gradients into the gradient of the mean over the WHOLE batch: every gradient is packed
      into one flat bucket per ``bucket_bytes`` (default 64 MiB: DGCNN's 1.8 M parameters
      are a single 7 MB message, far below what a per-link-bound xGMI ring needs to amortise its
      latency), scaled by n_local / batch, summed with one all-reduce (RCCL) and unpacked.
      Uneven splits (batch % world != 0) are therefore exact, unlike an unweighted average.
    No overlap with the backward: the whole reduction is ~0.1 ms next to a ~10 ms step.
    Batch statistics of BatchNorm layers stay per rank (as torch's DistributedDataParallel)."""

    def __init__(self, module, batch, world, rank, group=None, bucket_bytes=64 << 20):
        self.module, self.batch, self.world, self.rank, self.group = module, batch, world, rank, group
        self.lo, self.hi = cloud_range(batch, world, rank)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.params = [p for p in module.parameters() if p.requires_grad]
        if world > 1:
            with torch.no_grad():
                self._broadcast(list(module.parameters()) + list(module.buffers()))

    # -- helpers --------------------------------------------------------------------------
    def _staged(self, t):
        """gloo has no device path for every collective: stage CUDA tensors through the host."""
        return t.is_cuda and dist.get_backend(self.group) != "nccl"

    def _broadcast(self, tensors):
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for ts in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            buf = flat.cpu() if self._staged(flat) else flat
            dist.broadcast(buf, src=0, group=self.group)
            flat = buf.to(flat.device)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    def local(self, x):
        return x[self.lo:self.hi]

    @property
    def n_local(self):
        return self.hi - self.lo

    def __call__(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def reduce_gradients(self):
        """grad <- sum over ranks of (n_local / batch) * grad, bucket by bucket."""
        if self.world == 1:
            return
        scale = self.n_local / self.batch
        bucket, size = [], 0
        for p in self.params + [None]:
            if p is not None:
                if p.grad is None:                      # unused on this rank: contributes zeros
                    p.grad = torch.zeros_like(p)
                bucket.append(p)
                size += p.numel()
            if bucket and (p is None or size >= self.bucket_elems):
                flat = torch.cat([q.grad.reshape(-1).to(bucket[0].grad.dtype) for q in bucket]).mul_(scale)
                buf = flat.cpu() if self._staged(flat) else flat
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                flat = buf.to(flat.device)
                off = 0
                for q in bucket:
                    q.grad.copy_(flat[off:off + q.numel()].view_as(q.grad))
                    off += q.numel()
                bucket, size = [], 0
