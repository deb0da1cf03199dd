"""Multi-process tests of the row-partitioned GAT (dgnn_amd/dist.py): the partitioned result
must equal the single-graph result (oracle, float64) within the same tolerance as one GPU.

CPU: world_size 2 and 3 over gloo with an oracle-backed compute stand-in (partition logic,
     block extraction, inverse permutation, collectives).
GPU: world_size 2 over gloo, both ranks on cuda:0, the real HIP kernels on rectangular blocks
     (RCCL needs one GPU per rank, so the nccl path itself is exercised by bench.py --gpus N).
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle
from util import assert_close, gat_inputs, gat_masses, rmat_graph

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, mode, spec, h, f, exchange="halo", chunks=None, extra_env=None):
    port = free_port()
    env = dict(os.environ, DGNN_DIST_MODE=exchange, **(extra_env or {}))
    env.pop("DGNN_DIST_CHUNKS", None)
    if chunks is not None:
        env["DGNN_DIST_CHUNKS"] = str(chunks)
    with tempfile.TemporaryDirectory() as out_dir:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world),
                                   mode, str(port), out_dir, ",".join(map(str, spec)), str(h), str(f)], env=env)
                 for r in range(world)]
        try:
            for p in procs:
                assert p.wait(timeout=300) == 0, "a rank failed"
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        parts = [dict(np.load(os.path.join(out_dir, f"rank{r}.npz"))) for r in range(world)]
    return parts


def check(parts, spec, h, f):
    g = rmat_graph(*spec)
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed=21)
    o64, _, _ = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    gf64, gr64, gc64 = oracle.gat_backward_f64(0.2, 0.0, g["row_ptr"], g["col_ind"], None, in_feat,
                                               attn_row, attn_col, grad)
    m_out, m_gf, m_gr, m_gc = gat_masses(g, attn_row, attn_col, in_feat, grad, 0.2)
    # ranges tile [0, n) and the edge split is balanced
    los, his = [int(p["lo"]) for p in parts], [int(p["hi"]) for p in parts]
    assert los[0] == 0 and his[-1] == g["n"] and all(his[i] == los[i + 1] for i in range(len(parts) - 1))
    assert sum(int(p["nnz_fwd"]) for p in parts) == g["nnz"] == sum(int(p["nnz_bwd"]) for p in parts)
    # balanced by cost = edges + 2 per node (partition_bounds' default node weight)
    w = 2
    fair = (g["nnz"] + w * g["n"]) / len(parts)
    max_row = int(np.diff(g["row_ptr"]).max())
    assert max(int(p["nnz_fwd"]) + w * (int(p["hi"]) - int(p["lo"])) for p in parts) <= fair + max_row + w
    out = np.concatenate([p["out"] for p in parts])
    gf = np.concatenate([p["gf"] for p in parts])
    gr = np.concatenate([p["gr"] for p in parts])
    gc = np.concatenate([p["gc"] for p in parts])
    assert_close(out, o64, "partitioned out_feat", mass=m_out)
    assert_close(gf, gf64, "partitioned grad_feat", mass=m_gf)
    assert_close(gr, gr64, "partitioned grad_attn_row", mass=m_gr)
    assert_close(gc, gc64, "partitioned grad_attn_col", mass=m_gc)


@pytest.mark.parametrize("world,exchange", [(2, "halo"), (3, "halo"), (4, "halo"), (2, "allgather"), (3, "allgather")])
def test_partitioned_gat_cpu_gloo(world, exchange):
    spec, h, f = (10, 700, 6000, 5), 2, 8
    check(run_world(world, "cpu", spec, h, f, exchange), spec, h, f)


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 2), (2, 6), (8, 4)])
def test_partitioned_gat_cpu_gloo_chunked_pipeline(world, chunks):
    """The chunked exchange (default 4 chunks above) with 1, 2 and 6 chunks: same result."""
    spec, h, f = (10, 700, 6000, 5), 2, 8
    check(run_world(world, "cpu", spec, h, f, "halo", chunks), spec, h, f)


@pytest.mark.parametrize("world,chunks", [(2, None), (3, 2), (4, 1)])
def test_partitioned_gat_cpu_gloo_owner_local_grad_attn_row(world, chunks):
    """With the forward's negative-score pair a row's owner computes grad_attn_row from its own
    tensors: the column chunks return no partial rows and no return exchange runs
    (dgnn_amd/dist.py: backward).  Same results as the returning path."""
    spec, h, f = (10, 700, 6000, 5), 2, 8
    check(run_world(world, "cpuneg", spec, h, f, "halo", chunks), spec, h, f)


@pytest.mark.parametrize("world,h", [(2, 2), (3, 1), (4, 3)])
def test_partitioned_gat_cpu_gloo_ranks_agree_on_the_pair(world, h):
    """a rank whose forward lacks the negative-score pair used to fall back alone and
    issue a return exchange nobody else did.  With the pair missing on rank 1 only, every rank
    must take the returning path (no hang, same results); odd head counts put the chunk cuts at
    unaligned offsets of the [rows,h] tensors."""
    spec, f = (10, 700, 6000, 5), 8
    check(run_world(world, "cpunegmix", spec, h, f, "halo", 4), spec, h, f)
    check(run_world(world, "cpuneg", spec, h, f, "halo", 4), spec, h, f)


@pytest.mark.parametrize("world,extra", [(2, {}), (3, {}), (2, {"DGNN_DIST_ALLGATHER_ABOVE": "0.2"}),
                                         (3, {"DGNN_DIST_LINK_GBS": "100000"})])
def test_partitioned_gat_cpu_gloo_auto_plan(world, extra):
    """DGNN_DIST_MODE=auto (the default): every rank derives the same plan from the group-wide wire /
    compute ratio and halo fraction (PartitionedGAT._plan) -- wire-bound: 8 equal chunks; a fast
    assumed link: 4 geometric chunks; a low all-gather threshold: the all-gather exchange.  Results
    do not depend on the plan; a rank choosing alone would hang the group (120 s timeout)."""
    spec, h, f = (10, 700, 6000, 5), 2, 8
    check(run_world(world, "cpuneg", spec, h, f, "auto", None, extra), spec, h, f)


def test_exchange_plan_and_expected_step_model():
    import torch
    import dist_worker
    from dgnn_amd.dist import PartitionedGAT, chunk_fractions
    assert chunk_fractions(8, "equal") == [0.125] * 8 and chunk_fractions(1, "equal") == [1.0]
    g = rmat_graph(10, 700, 6000, 5)
    G = {k: torch.from_numpy(g[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind")}
    plans = {}
    for world in (1, 2, 8):
        r = PartitionedGAT(G, world, 0, torch.device("cpu"), ops=dist_worker.OracleOps(), fake_peers=True)
        plans[world] = r.plan
        e = r.expected_ms(4, 64)
        assert e["ms_per_step"] > 0 and abs(e["forward"]["ms"] + e["backward"]["ms"] - e["ms_per_step"]) < 2e-3
        assert e["forward"]["ms"] >= max(e["forward"]["compute_ms"] - 1e-3, 0) and e["plan"] is r.plan
        if world == 1:
            assert e["forward"]["wire_ms"] == 0 and r.plan["chunks"] == 1
    # this small graph's halo is large against its edges: wire-bound -> equal chunks
    assert plans[2]["mode"] == "halo" and plans[2]["chunk_shape"] == "equal" and plans[2]["chunks"] == 8
    assert 0 < plans[8]["halo_fraction_min_over_ranks"] <= 1
    # explicit arguments win over the plan
    r = PartitionedGAT(G, 2, 0, torch.device("cpu"), ops=dist_worker.OracleOps(), fake_peers=True, mode="allgather")
    assert r.mode == "allgather" and r.chunks == 1 and r.expected_ms(4, 64)["ms_per_step"] > 0
    r = PartitionedGAT(G, 2, 0, torch.device("cpu"), ops=dist_worker.OracleOps(), fake_peers=True, chunks=3)
    assert r.chunks == 3 and len(r.f_side.blocks) == 3


def check_heads(parts, spec, h, f):
    """Head-parallel ranks: every rank holds ALL rows of its head range; the ranges tile [0, h)."""
    g = rmat_graph(*spec)
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed=21)
    o64, _, _ = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    gf64, gr64, gc64 = oracle.gat_backward_f64(0.2, 0.0, g["row_ptr"], g["col_ind"], None, in_feat,
                                               attn_row, attn_col, grad)
    m_out, m_gf, m_gr, m_gc = gat_masses(g, attn_row, attn_col, in_feat, grad, 0.2)
    assert all(int(p["nnz_fwd"]) == g["nnz"] == int(p["nnz_bwd"]) for p in parts)      # the whole graph on every rank
    # the slices tile [0, h) x [0, f) exactly once
    cover = np.zeros((h, f), np.int32)
    out, gf = np.zeros_like(o64, dtype=np.float32), np.zeros_like(gf64, dtype=np.float32)
    gr, gc = np.zeros_like(gr64, dtype=np.float32), np.zeros_like(gc64, dtype=np.float32)
    for p in parts:
        lo, hi, f_lo, f_hi = int(p["lo"]), int(p["hi"]), int(p["f_lo"]), int(p["f_hi"])
        cover[lo:hi, f_lo:f_hi] += 1
        out[:, lo:hi, f_lo:f_hi], gf[:, lo:hi, f_lo:f_hi] = p["out"], p["gf"]
        if f_lo == 0:
            gr[:, lo:hi], gc[:, lo:hi] = p["gr"], p["gc"]
        else:       # the score gradients are summed over a head's ranks: identical on all of them
            assert np.array_equal(gr[:, lo:hi], p["gr"]) and np.array_equal(gc[:, lo:hi], p["gc"])
    assert (cover == 1).all()
    assert_close(out, o64, "head-parallel out_feat", mass=m_out)
    assert_close(gf, gf64, "head-parallel grad_feat", mass=m_gf)
    assert_close(gr, gr64, "head-parallel grad_attn_row", mass=m_gr)
    assert_close(gc, gc64, "head-parallel grad_attn_col", mass=m_gc)


@pytest.mark.parametrize("world,h,f", [(2, 4, 8), (4, 4, 8), (2, 2, 8), (3, 6, 8), (8, 4, 16), (4, 2, 8), (2, 1, 12)])
def test_head_parallel_gat_cpu_gloo(world, h, f):
    """dgnn_amd.dist.make_gat_runner picks the head split when heads % world == 0 -- every rank runs the whole graph
    on its heads with no exchange -- and heads x feature slices when world is a multiple of the heads (8 ranks on
    4 heads: two feature halves per head, the score gradients summed by one all-reduce inside each head's pair).
    The slices assemble to the single-graph result."""
    spec = (10, 700, 6000, 5)
    check_heads(run_world(world, "cpuheads", spec, h, f, "auto"), spec, h, f)


def test_runner_factory_picks_heads_or_rows():
    import torch
    import dist_worker
    from dgnn_amd.dist import HeadParallelGAT, PartitionedGAT, make_gat_runner
    g = rmat_graph(10, 700, 6000, 5)
    G = {k: torch.from_numpy(g[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind")}
    mk = lambda world, heads, **kw: make_gat_runner(G, world, 0, torch.device("cpu"), heads, ops=dist_worker.OracleOps(),
                                                    fake_peers=True, **kw)
    assert isinstance(mk(2, 4), HeadParallelGAT) and isinstance(mk(4, 4), HeadParallelGAT)
    assert isinstance(mk(8, 4), PartitionedGAT) and isinstance(mk(3, 4), PartitionedGAT)      # heads do not divide
    assert isinstance(mk(8, 4, feat=64), HeadParallelGAT) and mk(8, 4, feat=64).q == 2         # ... but 4 x 2 slices do
    assert isinstance(mk(8, 4, feat=4), PartitionedGAT) and isinstance(mk(8, 4, feat=63), PartitionedGAT)
    r8 = make_gat_runner(G, 8, 5, torch.device("cpu"), 4, feat=64, ops=dist_worker.OracleOps(), fake_peers=True)
    assert (r8.lo, r8.hi, r8.f_lo, r8.f_hi) == (2, 3, 32, 64) and r8.expected_ms(1, 32)["all_reduce_ms"] > 0
    assert isinstance(mk(1, 4), PartitionedGAT)
    assert isinstance(mk(2, 4, strategy="rows"), PartitionedGAT)
    with pytest.raises(ValueError):
        mk(3, 4, strategy="heads")
    r = mk(4, 8)
    assert (r.lo, r.hi, r.h_local) == (0, 2, 2) and r.plan["exchange"] == "none"
    e = r.expected_ms(r.h_local, 16)
    assert e["forward"]["wire_ms"] == 0 and e["ms_per_step"] > 0 and e["plan"]["mode"] == "heads"


def test_chunk_fractions_and_incremental_tables():
    import torch
    from dgnn_amd.dist import PartitionedGAT, chunk_fractions
    for c in (1, 2, 4, 7):
        fr = chunk_fractions(c)
        assert len(fr) == c and abs(sum(fr) - 1.0) < 1e-12 and fr == sorted(fr)
    # one rank, no collectives: every chunk's table prefix is just the local rows
    g = rmat_graph(10, 700, 6000, 5)
    G = {k: torch.from_numpy(g[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind")}
    import dist_worker
    r = PartitionedGAT(G, 1, 0, torch.device("cpu"), ops=dist_worker.OracleOps(), chunks=3)
    assert [b[1] - b[0] for b in r.f_side.blocks] and r.f_side.blocks[-1][1] == g["n"]
    assert all(b[4] == g["n"] for b in r.f_side.blocks) and r.f_side.n_table == g["n"]
    assert sum(b[3].numel() for b in r.f_side.blocks) == g["nnz"] == sum(b[3].numel() for b in r.b_blocks)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_partitioned_gat_gpu_two_ranks(exchange):
    spec, h, f = (12, 4000, 60000, 7), 4, 64
    check(run_world(2, "gpu", spec, h, f, exchange), spec, h, f)


@pytest.mark.gpu
@pytest.mark.parametrize("h,f", [(1, 64), (2, 16), (3, 8), (8, 8)])
def test_partitioned_gat_gpu_two_ranks_unaligned_chunk_cuts(h, f):
    """Heads 1 / 2 / 3 with four chunks: a chunk's first row sits at a*h*4 bytes of the [rows,h] tensors,
    16-byte aligned or not depending on where the rank's cuts fall.  Both ranks must
    produce the pair (asserted in the worker), agree on it and finish."""
    spec = (12, 4001, 60000, 7)
    check(run_world(2, "gpu", spec, h, f, "halo", 4), spec, h, f)


@pytest.mark.gpu
@pytest.mark.parametrize("world,h,f", [(2, 4, 64), (4, 4, 64), (2, 8, 8), (2, 2, 41), (8, 4, 64), (4, 1, 48)])
def test_head_parallel_gat_gpu_ranks_on_one_card(world, h, f):
    """The head-parallel runner on the real kernels: gloo ranks sharing cuda:0 (no collective touches the data)."""
    spec = (12, 4001, 60000, 7)
    check_heads(run_world(world, "gpuheads", spec, h, f, "auto"), spec, h, f)


@pytest.mark.gpu
def test_head_parallel_gat_nccl_ranks():
    """One GPU per rank under RCCL (runs wherever two or more GPUs are visible)."""
    import torch
    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one GPU per rank)")
    spec, h, f = (12, 4001, 60000, 7), 4, 64
    world = 8 if n_gpu >= 8 else 4 if n_gpu >= 4 else 2      # 8: feature halves, all-reduce inside each head's pair
    check_heads(run_world(world, "ncclheads", spec, h, f, "auto"), spec, h, f)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange,h,f", [("halo", 4, 64), ("allgather", 4, 64), ("halo", 1, 16), ("halo", 2, 32)])
def test_partitioned_gat_nccl_ranks(exchange, h, f):
    """Real RCCL, one GPU per rank: halo exchange and all-gather against the single-graph oracle.
    Runs wherever at least two GPUs are visible (the driver's multi-GPU node), skipped on a one-GPU box."""
    import torch
    n_gpu = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if n_gpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one GPU per rank)")
    spec = (12, 4001, 60000, 7)
    check(run_world(min(n_gpu, 4), "nccl", spec, h, f, exchange, 4), spec, h, f)


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_partitioned_gat_one_rank_through_rccl(exchange, monkeypatch):
    """Every collective of the step issued through RCCL itself (one rank, collectives forced)."""
    monkeypatch.setenv("DGNN_DIST_FORCE_COLLECTIVES", "1")
    spec, h, f = (12, 4000, 60000, 7), 4, 64
    check(run_world(1, "nccl1", spec, h, f, exchange), spec, h, f)


def test_partition_bounds_balance_edges_not_nodes():
    import torch
    from dgnn_amd.dist import partition_bounds
    g = rmat_graph(12, 4000, 60000, 7)
    b = partition_bounds(torch.from_numpy(g["row_ptr"]), 4, node_weight=0)
    assert b[0] == 0 and b[-1] == g["n"] and all(b[i] <= b[i + 1] for i in range(4))
    per = [int(g["row_ptr"][b[i + 1]] - g["row_ptr"][b[i]]) for i in range(4)]
    assert max(per) - min(per) <= 2 * int(np.diff(g["row_ptr"]).max())
    nodes = [b[i + 1] - b[i] for i in range(4)]
    assert nodes[0] < nodes[-1], "R-MAT ids are skewed: the first range must hold fewer nodes"
    # a node also costs `node_weight` edges (dense per-node passes): equal edges + w * nodes
    w = 3
    bw = partition_bounds(torch.from_numpy(g["row_ptr"]), 4, node_weight=w)
    assert bw[0] == 0 and bw[-1] == g["n"] and all(bw[i] <= bw[i + 1] for i in range(4))
    cost = [int(g["row_ptr"][bw[i + 1]] - g["row_ptr"][bw[i]]) + w * (bw[i + 1] - bw[i]) for i in range(4)]
    assert max(cost) - min(cost) <= 2 * (int(np.diff(g["row_ptr"]).max()) + w)
    assert bw[-1] - bw[-2] < nodes[-1], "the range of many short rows shrinks"
    assert partition_bounds(torch.from_numpy(g["row_ptr"]), 1) == [0, g["n"]]
