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
  "auto" (default)  "halo" unless every rank references >= 90 % of the remote rows (then
                    "allgather"); chunk count and shape from the wire / compute ratio of the
                    partition (PartitionedGAT._plan: 4 geometric chunks when compute-bound, 8 equal
                    chunks when wire-bound, e.g. two ranks over one link).
  "halo"            only the rows a rank actually references move, and own rows never travel:
                    index lists are exchanged once at set-up; the block is processed in S chunks
                    (``DGNN_DIST_CHUNKS``, default 4) of consecutive rows (forward) / columns
                    (backward), smallest first; chunk s fetches only the remote rows no earlier
                    chunk fetched, with one ``all_to_all_single`` per tensor (direct peer-to-peer
                    over each pair's own xGMI link) issued asynchronously on RCCL's stream, so
                    the exchange of chunk s+1 overlaps the kernels of chunk s.  Received rows
                    land directly behind the local rows in the table the kernels gather from
                    (column ids are relabelled to table positions once at set-up): no scatter
                    pass.  On the products-shaped R-MAT graph at 8 ranks a rank needs ~35 % of
                    the remote rows; with 4 chunks the first (unhidden) exchange carries ~38 %
                    of them, with 5 chunks ~24 % (hubs are needed by every chunk).
  "allgather"       every rank receives every row (``dist.all_gather`` on uneven row blocks).
Set with ``DGNN_DIST_MODE``.  gloo (CPU tests) takes the same code path; CUDA tensors under a
non-NCCL backend are staged through the host.
"""
import contextlib
import os

import torch
import torch.distributed as dist


def partition_bounds(row_ptr, world, node_weight=None):
    """Node range bounds [world+1] with ~equal cost per range (on the host), a node costing its
    edges + ``node_weight``: the gathers are paid per edge, the dense passes (output rows, row
    statistics, gradient rows) per node, so a range of many short rows is not as cheap as its
    edge count says.  Default: DGNN_DIST_NODE_WEIGHT or 2 (products shape, 8 ranks timed alone:
    slowest rank 6.33 ms with 0, 6.05 with 2 or 4, 6.5 with 8)."""
    if node_weight is None:
        node_weight = float(os.environ.get("DGNN_DIST_NODE_WEIGHT", "2"))
    rp = row_ptr.detach().to("cpu", torch.int64)
    n = rp.numel() - 1
    if node_weight:
        rp = rp + (torch.arange(n + 1, dtype=torch.float64) * node_weight).to(torch.int64)
    nnz = int(rp[-1])
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

    supports_neg = True     # the forward can return the negative-score pair (include/dgnn_hip.h)

    def gat_forward(self, attn_row, attn_col, row_ptr, col_ind, slope, in_feat, out=None, neg=False):
        """-> (out, edge_max, edge_sum) or, with neg, (+ out_neg, a_neg; None where unsupported);
        out = (out, edge_max, edge_sum[, None, out_neg, a_neg]) views to write into."""
        if not neg:
            out, emax, esum, _ = self._gat.gat_forward(attn_row, attn_col, row_ptr, col_ind, slope, in_feat, 0.0,
                                                       out=out, saved=False)
            return out, emax, esum
        res = self._gat.gat_forward(attn_row, attn_col, row_ptr, col_ind, slope, in_feat, 0.0, out=out, neg=True)
        return res[0], res[1], res[2], res[4], res[5]

    def row_dot(self, grad, out):
        return self._gat.gat_row_dot(grad, out)

    def row_stats(self, slope, attn_row, edge_max, edge_sum, grad, out, out_neg=None, a_neg=None):
        """(packed row records [m,h,4], grad_attn_row or None) in one pass, or None (unsupported shape)."""
        return self._gat.gat_row_stats(slope, attn_row, edge_max, edge_sum, grad, out, out_neg, a_neg)

    def gat_backward(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                     in_feat, attn_row, attn_col, grad, row_dot, out=None):
        return self._gat.gat_backward(slope, 0.0, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max,
                                      edge_sum, None, in_feat, attn_row, attn_col, grad, row_dot=row_dot, out=out)

    def gat_backward_packed(self, slope, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat, in_feat,
                            attn_col, grad, out=None, need_grad_attn_row=True):
        """row_stat [rows,H,4] = {attn_row, edge_max, edge_sum, row_dot}; None where unsupported."""
        return self._gat.gat_backward_packed(slope, 0.0, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat,
                                             None, in_feat, attn_col, grad, out=out,
                                             need_grad_attn_row=need_grad_attn_row)

    def pack_rows(self, src, idx):
        """src[idx] (int32 idx) by the pack kernel: one float4 per lane, four loads in flight."""
        return self._gat.pack_rows(src, idx)

    def transpose(self, ptr, ind, n_cols):
        """(ptr, ind) = CSR of an (r x n_cols) matrix -> (ptr_T [n_cols+1], ind_T, perm)."""
        return self._fmt.csr2csc_int(ptr, ind, n_cols=n_cols)


def chunk_fractions(chunks, shape="geometric"):
    """Shares of a block's edges given to its chunks.
    "geometric" (compute-bound pipelines): 1/2^(S-1), 1/2^(S-1), 1/2^(S-2), ..., 1/2.  The FIRST
    exchange cannot be hidden behind compute, so the first chunk is the smallest; on a power-law
    graph it still pulls in the hubs (~1/4 of the halo for 1/16 of the edges).
    "equal" (wire-bound pipelines, e.g. two ranks over ONE xGMI link): the wire never idles, what
    stays exposed is the compute of the LAST chunk behind the last message, so no chunk is large."""
    if chunks <= 1:
        return [1.0]
    if shape == "equal":
        return [1.0 / chunks] * chunks
    return [0.5 ** (chunks - 1)] + [0.5 ** (chunks - i) for i in range(1, chunks)]


# The two rates the exchange plan and bench.py's "expected" curve are priced with.
# LINK: what one xGMI link delivers to an all-to-all message in ONE direction: 153.6 GB/s per link is the
# two-way figure, 76.8 GB/s each way; 64 GB/s (5/6 of it) is assumed for RCCL's point-to-point sends -- an
# ASSUMPTION until the first multi-GPU record exists.  GATHER: the rate at which one rank's kernels consume
# gathered rows, measured on one GPU (products shape: 62 M edges x 1 KiB x 2 passes in 21.9 ms).
# WIRE_BOUND_ABOVE: wire / compute ratio from which the equal-chunk pipeline is planned; between the products
# shape on 8 ranks (1.26: 4 geometric chunks model best) and on 4 ranks (1.62: 8 equal chunks model best),
# profiles/r3k_dist_plan_model.jsonl.
LINK_GBS = float(os.environ.get("DGNN_DIST_LINK_GBS", "64"))
GATHER_GBS = float(os.environ.get("DGNN_DIST_GATHER_GBS", "5800"))
WIRE_BOUND_ABOVE = float(os.environ.get("DGNN_DIST_WIRE_BOUND_ABOVE", "1.4"))


class _NoWait:
    """Stand-in for a collective's work handle where the exchange ran synchronously on the pack
    stream (one rank, rehearsal copies): the chunk's event carries the ordering."""

    def wait(self):
        return None


class _Halo:
    """Who needs which of my rows, and which rows I need, for ONE chunk of one direction."""

    def __init__(self, owner, needed_ids):
        """needed_ids: sorted unique GLOBAL ids (int64) of REMOTE nodes this chunk adds."""
        self.n_needed = needed_ids.numel()
        b = torch.tensor(owner.bounds, dtype=torch.int64, device=needed_ids.device)
        cut = torch.searchsorted(needed_ids, b)                      # [world+1]
        self.recv_counts = [int(c) for c in (cut[1:] - cut[:-1]).cpu()]
        if owner._fake_peers:
            # timing rehearsal of ONE rank's share without peers (bench.py --simulate-rank): pretend
            # every peer wants from me as many rows as I want from it; WHICH rows does not matter
            self.send_counts = list(self.recv_counts)
            self.send_idx = (torch.arange(self.n_needed, dtype=torch.int64, device=needed_ids.device) * 7919
                             % max(owner.n_loc, 1)).contiguous()
            self.send_idx32 = self.send_idx.to(torch.int32)
            return
        self.send_counts = owner._a2a_counts(self.recv_counts)
        wanted = torch.empty(sum(self.send_counts), dtype=torch.int64, device=needed_ids.device)
        owner._a2a(wanted, needed_ids, self.send_counts, self.recv_counts)   # ids the peers want from me
        self.send_idx = (wanted - owner.lo).contiguous()                # local row indices, by peer
        assert self.send_idx.numel() == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < owner.n_loc)
        self.send_idx32 = self.send_idx.to(torch.int32)      # what the pack kernel reads


class _Side:
    """One direction of the exchange for a rank's block, cut into chunks.

    The block is a compressed matrix whose OUTER index is local (forward: the rank's CSR rows;
    backward: its CSC columns) and whose INNER index is global (sources / destinations).  Outer
    entries are independent in the kernels, so the block is processed as S chunks of consecutive
    outer entries, smallest first (``chunk_fractions``).  Chunk s needs the remote inner nodes no
    earlier chunk has fetched; they are appended to a table laid out as
        [ local rows (n_loc) | new rows of chunk 0 | new rows of chunk 1 | ... ]
    (inner ids are relabelled to table positions once), so chunk s reads only the prefix that has
    arrived and the exchange of chunk s+1 runs while chunk s computes.  Own rows never travel.
    """

    def __init__(self, owner, ptr, ind, chunks, shape="geometric"):
        dev = ind.device
        n, lo, hi, n_loc = owner.n, owner.lo, owner.hi, owner.n_loc
        ptr = ptr.to(torch.int64)
        ind = ind.to(torch.int64)
        nnz = int(ptr[-1]) if ptr.numel() else 0
        acc, cuts = 0.0, [0]
        for fr in chunk_fractions(chunks, shape)[:-1]:
            acc += fr
            cuts.append(int(torch.searchsorted(ptr, torch.tensor([int(acc * nnz)], device=dev))[0]))
        cuts.append(n_loc)
        cuts = [min(max(c, 0), n_loc) for c in cuts]
        for i in range(1, len(cuts)):
            cuts[i] = max(cuts[i], cuts[i - 1])
        self.cuts = cuts
        pos = torch.full((n,), -1, dtype=torch.int64, device=dev)     # global id -> table position
        pos[lo:hi] = torch.arange(n_loc, device=dev)
        self.halos, self.offs = [], []
        off = n_loc
        for s in range(len(cuts) - 1):
            c = ind[int(ptr[cuts[s]]):int(ptr[cuts[s + 1]])]
            u = torch.unique(c[(c < lo) | (c >= hi)])                    # sorted distinct remote ids
            new = u[pos[u] < 0]
            pos[new] = off + torch.arange(new.numel(), device=dev)
            self.halos.append(_Halo(owner, new))
            self.offs.append(off)
            off += new.numel()
        self.n_table = off
        rel = pos[ind]
        assert rel.numel() == 0 or int(rel.min()) >= 0
        self.blocks = []                                                  # (outer range, ptr_s, ind_s, visible table rows)
        for s in range(len(cuts) - 1):
            a, b = cuts[s], cuts[s + 1]
            e0, e1 = int(ptr[a]), int(ptr[b])
            self.blocks.append((a, b, (ptr[a:b + 1] - ptr[a]).to(torch.int32).contiguous(),
                                rel[e0:e1].to(torch.int32).contiguous(), self.offs[s] + self.halos[s].n_needed))
        # return path (additive results of remote table rows go back to their owners in ONE
        # all-to-all): table rows regrouped by owner, chunks ascending inside an owner
        world = owner.world
        take, give = [], []
        self.ret_send_counts, self.ret_recv_counts = [0] * world, [0] * world
        for q in range(world):
            for s, h in enumerate(self.halos):
                r0 = sum(h.recv_counts[:q])
                take.append(torch.arange(self.offs[s] + r0, self.offs[s] + r0 + h.recv_counts[q], device=dev))
                s0 = sum(h.send_counts[:q])
                give.append(h.send_idx[s0:s0 + h.send_counts[q]])
                self.ret_send_counts[q] += h.recv_counts[q]
                self.ret_recv_counts[q] += h.send_counts[q]
        self.ret_take = torch.cat(take) if take else torch.empty(0, dtype=torch.int64, device=dev)
        self.ret_give = torch.cat(give) if give else torch.empty(0, dtype=torch.int64, device=dev)


class PartitionedGAT:
    """Fused GAT forward+backward on rank ``rank``'s share of a row-partitioned graph.

    ``graph`` holds the FULL index arrays (row_ptr, col_ind, col_ptr, row_ind as int32 tensors on
    ``device``); only this rank's blocks are kept afterwards.  ``ops`` is the compute backend
    (default: the HIP kernels); tests inject a CPU stand-in to exercise the partition and the
    collectives without a GPU.  ``chunks`` (halo mode; default DGNN_DIST_CHUNKS or 4 when
    world > 1) pipelines the exchange against the compute, see ``_Side``.
    """

    def __init__(self, graph, world, rank, device, ops=None, group=None, mode=None, chunks=None,
                 fake_peers=False):
        self.world, self.rank, self.device, self.group = world, rank, device, group
        self.ops = ops if ops is not None else _NativeOps()
        self.mode = mode or os.environ.get("DGNN_DIST_MODE", "auto")
        # one rank needs no collectives; DGNN_DIST_FORCE_COLLECTIVES=1 issues them anyway (a
        # single-GPU rehearsal of the RCCL calls: tests/test_dist.py)
        self._solo = world == 1 and os.environ.get("DGNN_DIST_FORCE_COLLECTIVES") != "1"
        # fake_peers: this process plays rank `rank` of `world` ALONE, every exchange replaced by a
        # local copy of the same size -- the results are meaningless, the time is what one rank
        # spends on everything except the wire (bench.py --simulate-rank)
        self._fake_peers = bool(fake_peers)
        if self._fake_peers:
            self._solo = True
        if self.mode not in ("auto", "halo", "allgather"):
            raise ValueError(f"unknown exchange mode {self.mode!r}")
        if chunks is None and "DGNN_DIST_CHUNKS" in os.environ:
            chunks = int(os.environ["DGNN_DIST_CHUNKS"])
        row_ptr, col_ind = graph["row_ptr"], graph["col_ind"]
        col_ptr, row_ind = graph["col_ptr"], graph["row_ind"]
        self.n = row_ptr.numel() - 1
        self.bounds = partition_bounds(row_ptr, world)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.lo, self.hi, self.n_loc = lo, hi, hi - lo
        # forward block  A[lo:hi, :]  (CSR)
        e0, e1 = int(row_ptr[lo]), int(row_ptr[hi])
        f_ptr = row_ptr[lo:hi + 1] - row_ptr[lo]
        f_col = col_ind[e0:e1].contiguous()
        # backward block  A[:, lo:hi]  (CSC)
        c0, c1 = int(col_ptr[lo]), int(col_ptr[hi])
        b_ptr = col_ptr[lo:hi + 1] - col_ptr[lo]
        b_row = row_ind[c0:c1].contiguous()
        self.nnz_fwd, self.nnz_bwd = e1 - e0, c1 - c0
        self.plan = self._plan(f_col, chunks)
        self.mode, self.chunks, shape = self.plan["mode"], self.plan["chunks"], self.plan["chunk_shape"]
        if self.mode == "halo":
            self.f_side = _Side(self, f_ptr, f_col, self.chunks, shape)
            self.b_side = _Side(self, b_ptr, b_row, self.chunks, shape)
            b_blocks = self.b_side.blocks
        else:
            self.f_row_ptr = f_ptr.to(torch.int32).contiguous()
            self.f_col_ind = f_col
            b_blocks = [(0, self.n_loc, b_ptr.to(torch.int32).contiguous(), b_row, self.n)]
        # every backward block's own CSR ordering (for the per-row sum of the edge gradients)
        self.b_blocks = []
        for a, b, cptr, rind, n_rows in b_blocks:
            rptr, cind, perm2 = self.ops.transpose(cptr, rind, n_rows)
            # perm2[j] = CSC position of the block's CSR entry j; the kernels want the inverse
            inv = torch.empty_like(perm2)
            inv[perm2.long()] = torch.arange(perm2.numel(), dtype=perm2.dtype, device=perm2.device)
            self.b_blocks.append((a, b, cptr, rind, n_rows, rptr, cind, inv))
        # dimensions of the largest chunk (bench.py prices the kernels of the LAST launch)
        self.last_fwd = ((self.f_side.blocks[-1][1] - self.f_side.blocks[-1][0], self.f_side.blocks[-1][3].numel())
                         if self.mode == "halo" else (self.n_loc, self.nnz_fwd))
        self.last_bwd = (self.b_blocks[-1][1] - self.b_blocks[-1][0], self.b_blocks[-1][3].numel())
        self._ctx = None
        self._tables = {}
        self._pair_agreed = {}      # (heads, features) -> every rank's forward produced the negative-score pair

    # ------------------------------------------------------------------ exchange plan
    def _plan(self, f_col, chunks):
        """Which exchange this partition gets.  Two numbers of the rank's forward block decide, both
        independent of the row width: ``halo_fraction`` = distinct remote rows referenced / remote
        rows that exist, and ``wire_over_compute`` = time the halo spends on the wire (its rows
        spread over world-1 links at LINK_GBS) / time the rank's kernels spend gathering the
        block's edges (GATHER_GBS).  Every rank must issue the same collectives, so the ranks take
        the MAX of the ratio and the MIN of the fraction over the group (one all-reduce at set-up).
          * DGNN_DIST_MODE=auto (default): "allgather" when every rank references >=
            DGNN_DIST_ALLGATHER_ABOVE (0.9) of the remote rows -- the halo lists, packs and
            relabelled tables then buy nothing over two plain all-gathers; "halo" otherwise.
          * chunks (when neither the argument nor DGNN_DIST_CHUNKS fixes them): ratio <=
            WIRE_BOUND_ABOVE (products on 8 ranks: 1.26) -> 4 geometric chunks, the small first
            chunk bounds the exposed exchange; above it (products on 4 ranks: 1.62; on 2 ranks,
            56 % of the other half over ONE link: 2.0) -> 8 equal chunks, the exchange is the
            critical path from the first byte and what can still be saved is the compute behind
            the last message (1/8 of the block instead of 1/2).  The model's own comparison of
            the candidates: profiles/r3k_dist_plan_model.jsonl.
        DGNN_DIST_CHUNK_SHAPE=geometric|equal overrides the shape."""
        remote = f_col[(f_col < self.lo) | (f_col >= self.hi)]
        halo_rows = int(torch.unique(remote).numel()) if remote.numel() else 0
        n_remote = max(self.n - self.n_loc, 1)
        fraction = halo_rows / n_remote
        wire = halo_rows / max(self.world - 1, 1) / LINK_GBS
        compute = max(self.nnz_fwd, 1) / GATHER_GBS
        ratio = wire / compute
        if not self._solo:
            dev = self.device if self._nccl() else torch.device("cpu")
            t = torch.tensor([ratio, -fraction], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            ratio, fraction = float(t[0]), -float(t[1])
        mode = self.mode
        if mode == "auto":
            above = float(os.environ.get("DGNN_DIST_ALLGATHER_ABOVE", "0.9"))
            mode = "allgather" if (self.world > 1 and fraction >= above) else "halo"
        wire_bound = ratio > WIRE_BOUND_ABOVE
        shape = os.environ.get("DGNN_DIST_CHUNK_SHAPE") or ("equal" if wire_bound else "geometric")
        if shape not in ("geometric", "equal"):
            raise ValueError(f"unknown DGNN_DIST_CHUNK_SHAPE {shape!r}")
        if chunks is None:
            chunks = 1 if self.world == 1 else 8 if wire_bound else 4
        chunks = max(1, int(chunks)) if mode == "halo" else 1
        return {"mode": mode, "chunks": chunks, "chunk_shape": shape, "halo_rows": halo_rows,
                "halo_fraction_min_over_ranks": round(fraction, 4), "wire_over_compute_max_over_ranks": round(ratio, 3),
                "link_GBs_assumed": LINK_GBS, "gather_GBs": GATHER_GBS}

    def expected_ms(self, h, f, chunk_overhead_ms=0.1, latency_ms=0.03):
        """Model of one forward+backward step of THIS rank over real links (bench.py's "expected"
        key; a prediction to judge the first multi-GPU record against, not a measurement).
        Per pass the exchanges of the chunks queue on the communicator's stream in chunk order
        (chunk s: its largest per-peer message / LINK_GBS + a launch latency; messages to different
        peers use different links), chunk s computes once its rows arrived and chunk s-1 is done
        (its edges x row bytes / GATHER_GBS + a fixed per-chunk cost: directory pass, merges of cut
        rows, launches).  The backward's records add 16 bytes per (row, head)."""
        row = 4.0 * h * f
        total = 0.0
        detail = {}
        for name, extra in (("forward", 4.0 * h), ("backward", 16.0 * h)):
            if self.mode == "halo":
                side = self.f_side if name == "forward" else self.b_side
                msgs = [max(hl.recv_counts + hl.send_counts + [0]) * (row + extra) for hl in side.halos]
                edges = [blk[3].numel() for blk in side.blocks]
            else:
                biggest = max(self.bounds[q + 1] - self.bounds[q] for q in range(self.world))
                msgs = [biggest * (row + extra)]          # every peer's whole block, one link each
                edges = [self.nnz_fwd if name == "forward" else self.nnz_bwd]
            t_wire = t_comp = wire_total = 0.0
            for m, e in zip(msgs, edges):
                w = (m / (LINK_GBS * 1e6) + latency_ms) if (self.world > 1 and m > 0) else 0.0
                t_wire += w
                wire_total += w
                t_comp = max(t_comp, t_wire) + e * row / (GATHER_GBS * 1e6) + chunk_overhead_ms
            total += t_comp
            detail[name] = {"ms": round(t_comp, 3), "wire_ms": round(wire_total, 3),
                            "compute_ms": round(sum(edges) * row / (GATHER_GBS * 1e6) + chunk_overhead_ms * len(edges), 3)}
        return {"ms_per_step": round(total, 3), **detail,
                "model": (f"per pass: chunk exchanges queue in order at {LINK_GBS:g} GB/s per link (assumed; largest "
                          f"per-peer message of a chunk) + {latency_ms * 1e3:g} us each, a chunk computes when its rows are "
                          f"in and the previous chunk is done, at {GATHER_GBS:g} GB/s of gathered rows (measured on one "
                          f"GPU) + {chunk_overhead_ms:g} ms per chunk; this rank's share, ranks are cost-balanced"),
                "plan": self.plan}

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

    def _a2a(self, out, inp, out_splits, in_splits, async_op=False):
        """out (rows grouped by source rank) <- inp (rows grouped by destination rank).  With
        ``async_op`` (RCCL only) the exchange runs on the communicator's stream and the returned
        handle's wait() orders the CURRENT stream after it; the handle keeps ``inp`` alive."""
        if self._solo:
            out.copy_(inp)
            return None
        if self._nccl() or not inp.is_cuda:
            inp = inp.contiguous()
            work = dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group,
                                          async_op=async_op and self._nccl())
            return (work, inp) if work is not None else None
        # e.g. gloo with device tensors: stage through the host
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.contiguous().cpu(), out_splits, in_splits, group=self.group)
        out.copy_(o)
        return None

    def _agree_on_pair(self, key, mine):
        """The backward issues different collectives with and without the forward's negative-score
        pair (no return exchange with it), so one rank falling back alone would pair an all-to-all
        with nothing and hang the group: the pair is used only if EVERY rank has it (one MIN
        all-reduce at the first forward of a shape, remembered afterwards)."""
        if not self._solo:
            dev = self.device if self._nccl() else torch.device("cpu")
            flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            mine = bool(int(flag.item()))
        self._pair_agreed[key] = mine
        return mine

    @staticmethod
    def _wait(handles):
        """Order the CURRENT stream after one chunk's exchange: the collectives' own handles (RCCL
        stream) and the event recorded on the pack stream behind them (local copies / staged paths)."""
        hs, ev = handles if isinstance(handles, tuple) else (handles, None)
        for h in hs:
            if h is not None:
                h[0].wait()
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _chunk_streams(self, dev):
        """Two compute streams taken in turn by the chunks of a pass (GPU, halo mode): a chunk's
        tail -- the last waves of its gather, the merges of rows cut by group boundaries (a chain
        of small launches), the next chunk's directory pass -- runs under the next chunk's gather
        instead of in front of it.  Chunks write disjoint rows, so the order between them is free.
        DGNN_DIST_TWO_STREAMS=0 keeps every chunk on the caller's stream."""
        if dev.type != "cuda" or self.chunks < 2 or os.environ.get("DGNN_DIST_TWO_STREAMS", "1") == "0":
            return None
        if getattr(self, "_cstreams", None) is None:
            self._cstreams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        cur = torch.cuda.current_stream(dev)
        for st in self._cstreams:
            st.wait_stream(cur)
        return self._cstreams

    @staticmethod
    def _join_streams(streams, dev):
        if streams is not None:
            cur = torch.cuda.current_stream(dev)
            for st in streams:
                cur.wait_stream(st)

    def _pack(self, local, h):
        """The rows of ``local`` the peers want for this chunk, as one contiguous message."""
        pack = getattr(self.ops, "pack_rows", None)       # injected test back ends have none
        if pack is not None:
            return pack(local, h.send_idx32)
        return local.index_select(0, h.send_idx)

    def _start_fetch(self, side, tables, locals_, only=None, skip=()):
        """Start every chunk's exchange, in chunk order: for each chunk the rows the peers need of
        each tensor in ``locals_`` are packed and sent, and the rows this rank needs land directly
        in their place in ``tables`` (no scatter pass).  -> per-chunk (handle list, event).
        ``only`` restricts the call to those chunks (the others get empty entries); ``skip`` lists
        (chunk, tensor position) pairs an earlier call has started already.  Every rank makes the
        same calls in the same order, so the collectives still pair up.

        On the GPU the packs run on a SIDE stream (ordered after the current stream once, so the
        inputs are complete): only chunk 0's pack and exchange precede chunk 0's kernels; the packs
        of the later chunks (1.2 ms per step at 8 ranks on the products shape, as much as one gather
        pass) overlap the compute stream instead of sitting in front of it.  DGNN_DIST_PACK_STREAM=0
        issues them on the compute stream as before (A/B runs)."""
        first = next((l for l in locals_ if l is not None), None)
        on_gpu = first is not None and first.is_cuda and os.environ.get("DGNN_DIST_PACK_STREAM", "1") != "0"
        if not on_gpu:
            handles = []
            for s, h in enumerate(side.halos):
                hs = []
                for i, (table, local) in enumerate(zip(tables, locals_)):
                    if (only is not None and s not in only) or (s, i) in skip:
                        continue
                    send = self._pack(local, h)
                    hs.append(self._a2a(table[side.offs[s]:side.offs[s] + h.n_needed], send, h.recv_counts,
                                        h.send_counts, async_op=True))
                handles.append((hs, None))
            return handles
        dev = first.device
        if getattr(self, "_pack_stream", None) is None:
            self._pack_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        self._pack_stream.wait_stream(cur)            # inputs (and last step's readers of the tables) are done
        handles = []
        with torch.cuda.stream(self._pack_stream):
            for s, h in enumerate(side.halos):
                hs = []
                for i, (table, local) in enumerate(zip(tables, locals_)):
                    if (only is not None and s not in only) or (s, i) in skip:
                        continue
                    send = self._pack(local, h)
                    hs.append(self._a2a(table[side.offs[s]:side.offs[s] + h.n_needed], send, h.recv_counts,
                                        h.send_counts, async_op=True))
                    if hs[-1] is None:
                        hs[-1] = (_NoWait(), send)     # keep the message alive until the chunk is consumed
                ev = torch.cuda.Event()
                ev.record(self._pack_stream)
                handles.append((hs, ev))
        return handles

    def _return_rows(self, side, partial):
        """Inverse of the fetches for additive results: the remote rows of ``partial``
        [n_table, ...] go back to their owners (one all-to-all) and are added there in a fixed
        (peer, chunk) order; -> this rank's [n_loc, ...] rows."""
        acc = partial[:self.n_loc].clone()
        give = side.ret_give
        back = torch.empty((give.numel(),) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
        self._a2a(back, partial.index_select(0, side.ret_take), side.ret_recv_counts, side.ret_send_counts)
        off = 0
        for cnt in side.ret_recv_counts:          # ids are distinct inside one peer's list
            if cnt:
                acc[give[off:off + cnt]] += back[off:off + cnt]
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

    def _table(self, side, local, name):
        """The gather table of ``side`` for tensor ``name``: [n_table, ...] with the local rows in
        front.  Tables are kept between steps; when ``local`` IS the table's local part (the
        caller filled ``input_buffer``) no copy is made."""
        key = (id(side), name, tuple(local.shape[1:]), local.dtype)
        t = self._tables.get(key)
        if t is None:
            t = torch.empty((side.n_table,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            self._tables[key] = t
        if local.data_ptr() != t.data_ptr():
            t[:self.n_loc].copy_(local)
        return t

    def input_buffer(self, which, tail_shape, dtype=torch.float32):
        """[n_loc, *tail_shape] view of the local part of a gather table (halo mode): produce the
        layer input (``which`` = "forward": in_feat) or the output gradient ("backward": grad_out)
        directly into it and pass it to forward() / backward(): the step then starts without
        copying the local rows.  The table is overwritten by the next step of this runner (one
        runner per layer)."""
        if self.mode != "halo":
            return torch.empty((self.n_loc,) + tuple(tail_shape), dtype=dtype, device=self.device)
        side, name = (self.f_side, "in_feat") if which == "forward" else (self.b_side, "grad")
        key = (id(side), name, tuple(tail_shape), dtype)
        if key not in self._tables:
            self._tables[key] = torch.empty((side.n_table,) + tuple(tail_shape), dtype=dtype, device=self.device)
        return self._tables[key][:self.n_loc]

    # ------------------------------------------------------------------ forward / backward
    def forward(self, in_feat_loc, attn_row_loc, attn_col_loc, slope):
        """Local rows of GATConvFuse(...): [n_loc,H,F]."""
        in_feat_loc, attn_row_loc = in_feat_loc.contiguous(), attn_row_loc.contiguous()
        attn_col_loc = attn_col_loc.contiguous()
        if self.mode == "halo":
            side = self.f_side
            in_tab, ac_tab = self._table(side, in_feat_loc, "in_feat"), self._table(side, attn_col_loc, "attn_col")
            handles = self._start_fetch(side, (ac_tab, in_tab), (attn_col_loc, in_feat_loc))
            h = attn_row_loc.size(1)
            out = torch.empty((self.n_loc,) + tuple(in_feat_loc.shape[1:]), dtype=in_feat_loc.dtype, device=in_feat_loc.device)
            emax = torch.empty((self.n_loc, h), dtype=in_feat_loc.dtype, device=in_feat_loc.device)
            esum = torch.empty_like(emax)
            # the negative-score pair: grad_attn_row then needs nothing from other ranks (backward).
            # Whether the pair is used decides WHICH collectives the backward issues (with it no
            # partial rows travel back), so it must be the same on every rank: the first forward of
            # a shape agrees on it over the group (_agree_on_pair) and later steps reuse the answer.
            key = (h, in_feat_loc.size(-1))
            agreed = self._pair_agreed.get(key)
            want_neg = getattr(self.ops, "supports_neg", False) and in_feat_loc.size(-1) >= 4 and agreed is not False
            out_neg = torch.empty_like(out) if want_neg else None
            a_neg = torch.empty_like(emax) if want_neg else None
            streams = self._chunk_streams(in_feat_loc.device)
            for s, (a, b, ptr, ind, n_vis) in enumerate(side.blocks):   # every chunk writes its own rows
                with (torch.cuda.stream(streams[s % 2]) if streams is not None else contextlib.nullcontext()):
                    self._wait(handles[s])
                    if b == a:                         # an empty chunk writes nothing (and decides nothing)
                        continue
                    # the kernels move the [rows,h] tensors as 16-byte vectors: a chunk whose first row
                    # is not 16-byte aligned inside them (h = 1, 2, 3, 5 ... and an odd cut) gets
                    # aligned temporaries, copied into place (a few KB) -- the kernel choice, and with
                    # it the pair, must not depend on where a rank's cuts happen to fall
                    direct = (a * h) % 4 == 0
                    small = (emax[a:b], esum[a:b], a_neg[a:b] if want_neg else None) if direct else \
                        tuple(torch.empty((b - a, h), dtype=emax.dtype, device=emax.device) if t is not None else None
                              for t in (emax, esum, a_neg if want_neg else None))
                    if want_neg:
                        res = self.ops.gat_forward(attn_row_loc[a:b], ac_tab[:n_vis], ptr, ind, slope, in_tab[:n_vis],
                                                   out=(out[a:b], small[0], small[1], out_neg[a:b], small[2]), neg=True)
                        if res[3] is None:             # shape outside the pair's kernels: the edge-gradient path
                            if agreed:
                                raise RuntimeError("dgnn_amd.dist: the ranks agreed on the negative-score pair for this "
                                                   "shape, but a chunk's forward did not produce it")
                            want_neg, out_neg, a_neg = False, None, None
                    else:
                        self.ops.gat_forward(attn_row_loc[a:b], ac_tab[:n_vis], ptr, ind, slope, in_tab[:n_vis],
                                             out=(out[a:b], small[0], small[1]))
                    if not direct:
                        emax[a:b].copy_(small[0])
                        esum[a:b].copy_(small[1])
                        if want_neg:
                            a_neg[a:b].copy_(small[2])
            self._join_streams(streams, in_feat_loc.device)
            if agreed is None:
                want_neg = self._agree_on_pair(key, want_neg)
            self._neg = (out_neg, a_neg) if want_neg else None
        else:
            in_tab, ac_tab = self._all_gather_rows(in_feat_loc), self._all_gather_rows(attn_col_loc)
            out, emax, esum = self.ops.gat_forward(attn_row_loc, ac_tab, self.f_row_ptr, self.f_col_ind, slope, in_tab)
            self._neg = None
        self._ctx = (in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope)
        return out

    def backward(self, grad_out_loc):
        """-> (grad_feat_loc [n_loc,H,F], grad_attn_row_loc [n_loc,H], grad_attn_col_loc [n_loc,H])."""
        in_feat_loc, attn_row_loc, attn_col_loc, out, emax, esum, slope = self._ctx
        grad_out_loc = grad_out_loc.contiguous()
        neg_pair = getattr(self, "_neg", None) if self.mode == "halo" else None
        # the 16-byte records the column-gather kernel reads per (edge, head): they travel packed.
        # One kernel makes them (and, from the forward's negative-score pair, this rank's
        # grad_attn_row) from a single read of grad / out / out_neg; back ends without it take the
        # row dots one by one.
        # The first chunk's exchange is the exposed one, and its big message (grad_out rows) does not
        # depend on the records: it is started before the row pass and travels under it.
        early = None
        if self.mode == "halo" and os.environ.get("DGNN_DIST_EARLY_GRAD", "1") != "0":
            grad_tab = self._table(self.b_side, grad_out_loc, "grad")
            early = self._start_fetch(self.b_side, (None, grad_tab), (None, grad_out_loc), only=(0,), skip=((0, 0),))
        fused = None
        if hasattr(self.ops, "row_stats"):
            fused = self.ops.row_stats(slope, attn_row_loc, emax, esum, grad_out_loc, out,
                                       *(neg_pair if neg_pair is not None else (None, None)))
        if fused is not None:
            stat_loc, fused_gr = fused
        else:
            S_loc = self.ops.row_dot(grad_out_loc, out)
            stat_loc = torch.stack((attn_row_loc, emax, esum, S_loc), dim=2).contiguous()   # [n_loc,H,4]
            fused_gr = None
        if self.mode == "halo":
            side = self.b_side
            if early is None:
                grad_tab = self._table(side, grad_out_loc, "grad")
            stat_tab = self._table(side, stat_loc, "stat")
            handles = self._start_fetch(side, (stat_tab, grad_tab), (stat_loc, grad_out_loc),
                                        skip=((0, 1),) if early is not None else ())
            if early is not None:             # chunk 0 waits for both parts; the later event covers the pack stream
                handles[0] = (early[0][0] + handles[0][0], handles[0][1])
        else:
            grad_tab, stat_tab = self._all_gather_rows(grad_out_loc), self._all_gather_rows(stat_loc)
            handles = [[]]
        # With the forward's negative-score pair a row's owner computes grad_attn_row alone:
        #   (1 - slope) (S * a_neg - <grad, out_neg>)      (include/dgnn_hip.h, dgnn_gat_forward)
        # so the column chunks produce no edge gradients and no partial rows travel back.
        local_gr = None
        if neg_pair is not None:
            if fused_gr is not None:
                local_gr = fused_gr
            else:
                out_neg, a_neg = neg_pair
                S_of = stat_loc[:, :, 3] if fused is not None else S_loc
                local_gr = (1.0 - slope) * (S_of * a_neg - self.ops.row_dot(grad_out_loc, out_neg))
        gr_tab = None if local_gr is not None else \
            torch.zeros((stat_tab.size(0), stat_tab.size(1)), dtype=stat_tab.dtype, device=stat_tab.device)
        gf = torch.empty_like(in_feat_loc)
        gc = torch.empty_like(attn_col_loc)
        # (the chunks share gr_tab when partial grad_attn_row rows are summed: one stream then)
        streams = self._chunk_streams(gf.device) if (local_gr is not None and self.mode == "halo") else None
        for s, (a, b, cptr, rind, n_vis, rptr, cind, perm) in enumerate(self.b_blocks):
            # every chunk writes its own columns of grad_feat / grad_attn_col; its grad_attn_row
            # is a partial sum over the table rows it can see
            outs = (gf[a:b], None, gc[a:b])
            if b == a:                      # an empty chunk: no columns to write, no partial rows
                self._wait(handles[s])
                continue
            if local_gr is not None:
                with (torch.cuda.stream(streams[s % 2]) if streams is not None else contextlib.nullcontext()):
                    self._wait(handles[s])
                    res = self.ops.gat_backward_packed(slope, rptr, cind, cptr, rind, perm, stat_tab[:n_vis],
                                                       in_feat_loc[a:b], attn_col_loc[a:b], grad_tab[:n_vis], out=outs,
                                                       need_grad_attn_row=False)
                if res is not None:
                    continue
                # falling back HERE would issue a return exchange the other ranks do not: fail loudly
                # (cannot happen for shapes the pair exists for: float4 rows, 16-byte records)
                raise RuntimeError("dgnn_amd.dist: the packed column pass refused a shape the forward produced the "
                                   "negative-score pair for")
            else:
                self._wait(handles[s])
            res = self.ops.gat_backward_packed(slope, rptr, cind, cptr, rind, perm, stat_tab[:n_vis],
                                               in_feat_loc[a:b], attn_col_loc[a:b], grad_tab[:n_vis], out=outs)
            if res is None:             # shape outside the packed kernels: unpack to four [rows,H] arrays
                stat = stat_tab[:n_vis].permute(2, 0, 1).contiguous()
                res = self.ops.gat_backward(slope, rptr, cind, cptr, rind, perm, stat[1], stat[2], in_feat_loc[a:b],
                                            stat[0], attn_col_loc[a:b], grad_tab[:n_vis], stat[3], out=outs)
            gr_tab[:n_vis] += res[1]
        self._join_streams(streams, gf.device)
        if local_gr is not None:
            gr = local_gr
        elif self.mode == "halo":
            gr = self._return_rows(self.b_side, gr_tab)
        else:
            if not self._solo:
                dist.all_reduce(gr_tab, group=self.group)
            gr = gr_tab[self.lo:self.hi].contiguous()
        return gf, gr, gc

    def step(self, in_feat, attn_row, attn_col, grad_out, slope):
        """One forward+backward given FULL tensors (every rank slices its own range): the unit
        bench.py times.  Returns the local gradients."""
        lo, hi = self.lo, self.hi
        self.forward(in_feat[lo:hi], attn_row[lo:hi], attn_col[lo:hi], slope)
        return self.backward(grad_out[lo:hi])


# ----------------------------------------------------------------------------------------------
# Head-parallel GAT: the other way to put the operator on several GPUs.  Heads never interact inside
# the fused operator -- softmax and aggregation run per (row, head), fused_gatconv_kernel.cu:32-35,
# and so do all three gradients -- so rank p can run the WHOLE graph on heads [p H/P, (p+1) H/P) of
# in_feat / attn_row / attn_col / grad with NO exchange at all: no halo, no index lists, perfect
# balance whatever the graph looks like.  With more ranks than heads (P = q H) a head's FEATURES are
# cut into q slices as well: out and grad_feat are independent per feature given the scores, and the
# two score gradients are linear in <grad, in_feat> and <grad, out>, so every rank of a head computes
# them from its partial dot products and ONE all-reduce of [N, 2] floats per head group (10 MB at the
# products shape, against 0.75 GB of halo rows per pass) sums them.  The price is paid outside the
# operator (the layer's projection becomes column-parallel: every rank needs the layer input, i.e.
# tensor parallelism) and in row width: a rank's rows are P times narrower than the single-GPU rows,
# so the per-rank kernels are the narrow-row ones.
#
# Which one: on the products shape the row partition moves 35-56 % of the remote rows per pass and
# is predicted to LOSE at two ranks (one xGMI link), while one rank's step on
# 2 x 64 / 1 x 64 / 1 x 32 slices takes 11.3 / 6.25 / 4.1 ms against 21.85 ms on one GPU -- the last
# equal to what a rank of the 8-way row partition computes BEFORE its exchange.  make_gat_runner()
# therefore picks the slice split whenever heads and features divide and the row partition
# otherwise; DGNN_DIST_STRATEGY=rows|heads overrides.  The choice depends on (heads, features, world)
# alone, so every rank makes it without talking to the others.
def slice_layout(heads, world, feat=None):
    """(heads per rank, feature slices per head) of the head-parallel split, or None when (heads, feat) do not
    divide over ``world`` ranks: whole heads when heads % world == 0, else world = q * heads with q slices
    of feat / q >= 4 floats per head."""
    if heads % world == 0:
        return heads // world, 1
    if world % heads == 0 and feat is not None:
        q = world // heads
        if feat % q == 0 and feat // q >= 4:
            return 1, q
    return None


class HeadParallelGAT:
    """Fused GAT forward+backward on rank ``rank``'s heads (and, with more ranks than heads, its feature
    slice of one head) of the whole graph.  Same calling convention as PartitionedGAT, with ``local()`` /
    ``local_scores()`` cutting full [N,H,F] / [N,H] tensors to this rank's part; ``lo`` / ``hi`` are the
    head range, ``f_lo`` / ``f_hi`` the feature range (None = all)."""

    mode = "heads"

    def __init__(self, graph, world, rank, device, heads, feat=None, ops=None, group=None, fake_peers=False):
        lay = slice_layout(heads, world, feat)
        if lay is None:
            raise ValueError(f"head-parallel GAT: heads={heads}, feat={feat} do not divide over {world} ranks")
        self.world, self.rank, self.device, self.heads = world, rank, device, heads
        self.h_local, self.q = lay
        if self.q == 1:
            self.lo, self.hi = rank * self.h_local, (rank + 1) * self.h_local
            self.f_lo = self.f_hi = None
        else:
            self.lo, self.hi = rank // self.q, rank // self.q + 1
            fq = feat // self.q
            self.f_lo, self.f_hi = (rank % self.q) * fq, (rank % self.q + 1) * fq
        # the ranks that share my head sum their partial score gradients; every rank creates every group,
        # in the same order (torch.distributed's rule)
        self._group = None
        if self.q > 1 and not fake_peers and dist.is_initialized() and dist.get_world_size(group) > 1:
            for hh in range(heads):
                grp = dist.new_group([hh * self.q + j for j in range(self.q)])
                if hh == self.lo:
                    self._group = grp
        # one rank's worth of the row-partitioned runner on the whole graph: the same kernels in the same
        # order as the single-GPU operator (pair forward, one row pass, one column-gather pass), its
        # persistent input buffers, and no collective (fake_peers: nothing is ever exchanged)
        self._inner = PartitionedGAT(graph, 1, 0, device, ops=ops, mode="halo", chunks=1, fake_peers=True)
        self.n, self.n_loc = self._inner.n, self._inner.n
        self.nnz_fwd, self.nnz_bwd = self._inner.nnz_fwd, self._inner.nnz_bwd
        self.chunks = 1
        self.last_fwd, self.last_bwd = self._inner.last_fwd, self._inner.last_bwd
        self.plan = {"mode": "heads", "heads_per_rank": self.h_local, "feature_slices_per_head": self.q,
                     "exchange": "none" if self.q == 1 else
                     f"one all-reduce of [N, 2] floats over the {self.q} ranks of a head per step",
                     "gather_GBs": GATHER_GBS, "link_GBs_assumed": LINK_GBS}

    def local(self, t):
        """This rank's heads (and feature slice) of a full [N, H, F] tensor."""
        t = t[:, self.lo:self.hi]
        return (t if self.f_lo is None else t[:, :, self.f_lo:self.f_hi]).contiguous()

    def local_scores(self, t):
        """This rank's heads of a full [N, H] tensor (attn_row / attn_col: whole on every rank of a head)."""
        return t[:, self.lo:self.hi].contiguous()

    def input_buffer(self, which, tail_shape, dtype=torch.float32):
        return self._inner.input_buffer(which, tail_shape, dtype)

    def forward(self, in_feat_loc, attn_row_loc, attn_col_loc, slope):
        """[N, heads per rank, feature slice] of GATConvFuse(...)."""
        return self._inner.forward(in_feat_loc, attn_row_loc, attn_col_loc, slope)

    def backward(self, grad_out_loc):
        """-> (grad_feat of this rank's slice, grad_attn_row [N,h], grad_attn_col [N,h] of its heads; with
        feature slices the last two are summed over the head's ranks and identical on all of them)."""
        gf, gr, gc = self._inner.backward(grad_out_loc)
        if self._group is not None:
            both = torch.cat((gr, gc), dim=1)
            if both.is_cuda and dist.get_backend(self._group) != "nccl":
                host = both.cpu()                       # e.g. gloo ranks sharing one card (tests)
                dist.all_reduce(host, group=self._group)
                both.copy_(host)
            else:
                dist.all_reduce(both, group=self._group)
            gr, gc = both[:, :gr.size(1)].contiguous(), both[:, gr.size(1):].contiguous()
        return gf, gr, gc

    def step(self, in_feat, attn_row, attn_col, grad_out, slope):
        self.forward(self.local(in_feat), self.local_scores(attn_row), self.local_scores(attn_col), slope)
        return self.backward(self.local(grad_out))

    def expected_ms(self, h, f, chunk_overhead_ms=0.1, latency_ms=0.03):
        """No halo: the rank's kernels on rows of h x f floats (its slice) at GATHER_GBS -- optimistic for
        narrow slices, whose kernels run below that rate; bench.py --simulate-rank R/N
        measures the real figure on one GPU, since no row is exchanged -- plus, with feature slices, the
        ring all-reduce of [N, 2h] floats over the head's q ranks at LINK_GBS."""
        e = self._inner.expected_ms(h, f, chunk_overhead_ms, latency_ms)
        ar_ms = 0.0
        if self.q > 1:
            ar_ms = 2.0 * (self.q - 1) / self.q * (8.0 * h * self.n) / (LINK_GBS * 1e6) + 2 * (self.q - 1) * latency_ms
        e["ms_per_step"] = round(e["ms_per_step"] + ar_ms, 3)
        e["all_reduce_ms"] = round(ar_ms, 3)
        e["plan"] = self.plan
        e["model"] = (f"head-parallel: no rows exchanged; the whole graph on a slice of {h} head(s) x {f} features per rank at "
                      f"{GATHER_GBS:g} GB/s of gathered rows + {chunk_overhead_ms:g} ms per pass"
                      + (f" + one ring all-reduce of [N,{2 * h}] floats over {self.q} ranks at {LINK_GBS:g} GB/s (assumed)"
                         if self.q > 1 else ""))
        return e


def make_gat_runner(graph, world, rank, device, heads, feat=None, ops=None, group=None, strategy=None, **kw):
    """The multi-GPU runner for a fused GAT of ``heads`` x ``feat``: HeadParallelGAT when heads (and, beyond
    one head per rank, features) divide over the ranks -- no rows exchanged --, PartitionedGAT (row partition,
    halo exchange) otherwise.  ``strategy`` / DGNN_DIST_STRATEGY: "auto" (default), "heads", "rows"."""
    strategy = strategy or os.environ.get("DGNN_DIST_STRATEGY", "auto")
    if strategy not in ("auto", "heads", "rows"):
        raise ValueError(f"unknown strategy {strategy!r}")
    if strategy == "heads" or (strategy == "auto" and world > 1 and slice_layout(heads, world, feat) is not None):
        return HeadParallelGAT(graph, world, rank, device, heads, feat=feat, ops=ops, group=group,
                               fake_peers=bool(kw.get("fake_peers", False)))
    return PartitionedGAT(graph, world, rank, device, ops=ops, group=group, **kw)


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
