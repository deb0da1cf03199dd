"""Graph index preparation on the device: the five index arrays every fused operator takes.

The reference scripts build them on the host with DGL + scipy, once per graph
(dgNN/script/train/train_gatconv.py:73-94; GMM variant train_gmmconv.py:76-89):
  * remove, then add self loops (:75-76);
  * ``col, row = g.edges(order='srcdst')`` (:78): ROW = destination, COL = source;
  * ``csr_matrix((arange(E), (row, col)))`` (:82): CSR by destination, sources ascending inside a
    row, duplicate edges merged;
  * ``.tocsc()`` of it (:87-90) and ``permute`` = the CSR position of every CSC entry (:92-94);
  * everything int32 on the device (:109-112).
Here the same arrays come out of one sort of (dst, src) keys plus the device transpose
(csrc/convert.hip), bit-identical to the scipy recipe (tests/test_gpu_parity.py).
"""
import torch

from .native import format_conversion


def prepare_graph(src, dst, n, self_loops="replace"):
    """Edges src -> dst (integer tensors on the GPU, any order, duplicates allowed) of a graph
    with ``n`` nodes -> (row_ptr, col_ind, col_ptr, row_ind, permute), all int32.

    self_loops: "replace" = drop the given ones and add one per node, as the reference scripts
    do (train_gatconv.py:75-76); "keep" = leave the edge list as it is."""
    if src.shape != dst.shape or src.dim() != 1:
        raise ValueError("src and dst must be 1-d tensors of the same length")
    if self_loops not in ("replace", "keep"):
        raise ValueError(f"self_loops must be 'replace' or 'keep', not {self_loops!r}")
    src, dst = src.to(torch.int64), dst.to(torch.int64)
    if src.numel() and (min(int(src.min()), int(dst.min())) < 0 or max(int(src.max()), int(dst.max())) >= n):
        raise ValueError("node ids must lie in [0, n)")
    if self_loops == "replace":
        keep = src != dst
        loops = torch.arange(n, dtype=torch.int64, device=src.device)
        src, dst = torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])
    key = torch.unique(dst * n + src)                       # sorted: destinations, then sources
    rows = (key // n).to(torch.int32)
    col_ind = (key % n).to(torch.int32)
    row_ptr = format_conversion.coo2csr(rows, n)
    col_ptr, row_ind, permute = format_conversion.csr2csc_int(row_ptr, col_ind)
    return row_ptr, col_ind, col_ptr, row_ind, permute


def relabel_nodes(src, dst, n, how="scramble", seed=0):
    """A node relabelling for the memory system, applied ONCE at load time like the index
    preparation itself: -> (new_src, new_dst, new_of_old) with new_of_old[old id] = new id (int64,
    on the edges' device).  Permute node features / labels once with ``x_new[new_of_old] = x_old``
    (``permute_rows``), run every layer in the new labelling, and read results back through
    ``unpermute_rows``; the operators' results are the same up to this permutation.

    Why: a node's feature row lives at ``id * row_bytes``.  Generators such as R-MAT give the hubs
    the ids with few set bits, so the rows gathered most often sit at power-of-two strides and
    alias onto a few HBM channels: on the products-shaped R-MAT graph the forward gather runs
    11.5 ms with the generator's ids and 9.6 ms after ANY relabelling that breaks the pattern,
    with identical bytes fetched and identical L2 hit rate (profiles/r2b_*relabel*).  Graph500
    prescribes the same vertex scrambling for the same reason.  No counterpart in the reference
    (its scripts use the dataset's ids as they come).

    how: "scramble" = a seeded random permutation; "degree" = descending (in + out) degree,
    ties by old id (hubs contiguous at the front)."""
    if how not in ("scramble", "degree"):
        raise ValueError(f"how must be 'scramble' or 'degree', not {how!r}")
    src, dst = src.to(torch.int64), dst.to(torch.int64)
    dev = src.device
    if how == "degree":
        deg = torch.bincount(dst, minlength=n) + torch.bincount(src, minlength=n)
        order = torch.argsort(deg, descending=True, stable=True)           # order[new] = old
    else:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        order = torch.randperm(n, device=dev, generator=gen)
    new_of_old = torch.empty_like(order)
    new_of_old[order] = torch.arange(n, device=dev)
    return new_of_old[src], new_of_old[dst], new_of_old


def permute_rows(x_old, new_of_old):
    """Node-indexed tensor in the old labelling -> the new one (x_new[new_of_old[i]] = x_old[i])."""
    out = torch.empty_like(x_old)
    out[new_of_old] = x_old
    return out


def unpermute_rows(x_new, new_of_old):
    """Inverse of permute_rows: x_old[i] = x_new[new_of_old[i]]."""
    return x_new[new_of_old]
