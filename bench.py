#!/usr/bin/env python3
"""Headline benchmark: fused GATConv forward+backward on an ogbn-products-shaped R-MAT graph.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload products|reddit|cora]

A "step" is one GATConvFuse forward + backward (the autograd operator, allocations included)
over the whole graph with inputs already resident in HBM.  Prints ONE JSON line (rank 0) with
BASELINE.json's metric (edges/s), the HBM roofline of the dominant kernel and a CPU baseline
(the oracle's float32 restatement of the reference kernels, timed on the host cores).

N > 1, one rank per GPU: the same graph is row-partitioned across ranks; each step exchanges only
the boundary rows a rank references (all-to-all over RCCL, or all-gathers with
DGNN_DIST_MODE=allgather; dgnn_amd/dist.py); the work is fixed, so scaling is "strong".  The ranks
come either from torch.distributed.run (RANK / WORLD_SIZE in the environment; --gpus must agree)
or, when `python bench.py --gpus N` is started plainly, from this process itself: it starts N
fresh child ranks BEFORE touching the GPU (it never initialises HIP), relays rank 0's JSON line
and exits non-zero if any rank does.  --dry replaces the workload by one gloo all-reduce on the
CPU (a check of the launcher alone, tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("DGNN_TIMING", "1")     # event-time the dominant kernels (include/dgnn_hip.h)

WORKLOADS = {
    # name: (nodes, unique directed edges incl. self loops, rmat scale)   SURVEY.md section 8(d)
    "products": (2_449_029, 62_000_000, 22),
    "reddit": (232_965, 114_848_857, 18),
    "cora": (2_708, 13_264, 12),
}
H, F = 4, 64
SLOPE = 0.2
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# measured single-GPU steps (scrambled labels, profiles/r3h_bench_line_products.json, r3i_bench_line_reddit.json):
# the denominator of the "expected" multi-GPU curve
SINGLE_GPU_MS = {"products": 21.85, "reddit": 29.4}
# one rank's head-parallel step timed ALONE on one GPU (bench.py --simulate-rank R/N; nothing is exchanged in that
# split but one small all-reduce at 8 ranks, so this is a rank's step time): profiles/r3p_dist_head_parallel_ranks.jsonl
RANK_ALONE_MS = {"products": {2: 11.33, 4: 6.25, 8: 4.14}}
# collectives of a bench run take milliseconds: a rank that waits this long has lost a peer
TORCH_THREADS = torch.get_num_threads()     # (read at start-up: loading the OpenMP oracle later changes what torch reports)
PG_TIMEOUT = __import__("datetime").timedelta(seconds=int(os.environ.get("DGNN_PG_TIMEOUT", "120")))


def rmat_edges(n, target, scale, device, seed, abcd=(0.57, 0.19, 0.19, 0.05)):
    """Unique directed R-MAT edges (ids folded by mod n) + every self loop, as sorted int64 keys
    dst*n+src.  Exactly `target` keys are returned."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    a, b, c, _ = abcd
    want_rand = target - n
    keys = torch.empty(0, dtype=torch.int64, device=device)
    while keys.numel() < want_rand:
        batch = int((want_rand - keys.numel()) * 1.35) + 1024
        dst = torch.zeros(batch, dtype=torch.int64, device=device)
        src = torch.zeros(batch, dtype=torch.int64, device=device)
        for _ in range(scale):
            u = torch.rand(batch, device=device, generator=gen)
            dbit = (u >= a + b)
            sbit = ((u >= a) & (u < a + b)) | (u >= a + b + c)
            dst = dst * 2 + dbit
            src = src * 2 + sbit
        dst %= n
        src %= n
        k = dst * n + src
        k = k[dst != src]
        keys = torch.unique(torch.cat([keys, k]))
        del dst, src, k, u, dbit, sbit
    if keys.numel() > want_rand:
        perm = torch.randperm(keys.numel(), device=device, generator=gen)[:want_rand]
        keys = keys[perm]
    loops = torch.arange(n, dtype=torch.int64, device=device) * (n + 1)
    keys = torch.sort(torch.cat([keys, loops])).values
    return keys


def graph_from_keys(keys, n, device):
    """Sorted unique int64 keys dst*n+src -> CSR by destination / CSC / permute in the reference's
    convention (script/train/train_gatconv.py:78-94), built on the device."""
    import dgnn_amd  # noqa: F401
    from dgnn_amd.native import format_conversion
    dst = keys // n
    col_ind = (keys % n).to(torch.int32)
    counts = torch.bincount(dst, minlength=n)
    row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(counts, 0)
    row_ptr = row_ptr.to(torch.int32)
    del dst, counts
    col_ptr, row_ind, permute = format_conversion.csr2csc_int(row_ptr, col_ind)
    torch.cuda.synchronize()
    return dict(n=n, nnz=int(col_ind.numel()), row_ptr=row_ptr, col_ind=col_ind, col_ptr=col_ptr,
                row_ind=row_ind, permute=permute)


def build_graph(name, device, seed=0, labels="natural"):
    """labels: "natural" = the R-MAT generator's ids (hubs at the ids with few set bits);
    "scrambled" = the same graph after dgnn_amd.graph.relabel_nodes(how="scramble"), the load-time
    vertex scrambling Graph500 prescribes for R-MAT (real datasets carry no such id structure)."""
    n, target, scale = WORKLOADS[name]
    keys = rmat_edges(n, target, scale, device, seed)
    if labels == "scrambled":
        from dgnn_amd.graph import relabel_nodes
        src, dst, _ = relabel_nodes(keys % n, keys // n, n, how="scramble", seed=seed)
        keys = torch.sort(dst * n + src).values
        del src, dst
    elif labels != "natural":
        raise ValueError(labels)
    return graph_from_keys(keys, n, device)


def algorithmic_bytes(n, nnz, h, f, neg=True):
    """SURVEY.md section 8(d) gather model (attn_drop = 0).  Returns per-phase byte counts.
    neg: the training path of round 2 -- the forward also writes the negative-score pair (out_neg
    [n,h,f], a_neg [n,h]) and the backward's gather pass neither reads permute nor writes the
    [nnz,h] edge gradients (grad_attn_row comes out of the per-row pass)."""
    hf = h * f
    fwd = nnz * (4 * hf + 4 * h + 4) + n * (4 * hf + 16 * h + 4)          # gather + out + stats
    # backward, one gather pass: grad rows + packed row stats + row_ind (+ permute + dE write),
    # per node in_feat row + grad_feat write + pointers/scalars
    bwd_csc = nnz * (4 * hf + 16 * h + 8 + 4 * h) + n * (8 * hf + 8 * h + 4)
    if neg:
        fwd += n * (4 * hf + 4 * h)
        bwd_csc -= nnz * (4 + 4 * h)
    total_model = nnz * (12 * hf + 28 * h + 12) + n * (12 * hf + 36 * h + 8)  # BASELINE.md formula
    return dict(fwd=fwd, bwd_csc=bwd_csc, survey_total=total_model)


def measured_traffic(workload, kernel_name):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    of this workload (profiles/*pmc_hbm_bytes_<workload>.csv, the last by tag name: FETCH_SIZE and WRITE_SIZE collected
    in separate runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM
    prescribes for gfx950).  Counters cannot be read from inside the process, so the figure is
    only as fresh as that profile: it is returned with the file it came from, and it is None when
    no profile of this workload is committed or when the profile does not hold the exact template
    instantiation just timed (`kernel_name`, from dgnn_debug_last_kernel_name) -- i.e. when the
    kernel has changed since.  -> (bytes or None, source file or None)"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_hbm_bytes_{workload}.csv")))   # r1_a < ... < r2d: newest tag last
    if not files or not kernel_name:
        return None, None
    src = files[-1]
    parts = {}
    for row in csv.DictReader(open(src)):
        if f"dgnn::{kernel_name}(" in row["kernel"]:
            parts[row["counter"]] = float(row["avg_bytes_corrected"])
    if set(parts) != {"FETCH_SIZE", "WRITE_SIZE"}:
        return None, os.path.relpath(src, ROOT) + f" (no {kernel_name}: stale)"
    return parts["FETCH_SIZE"] + parts["WRITE_SIZE"], os.path.relpath(src, ROOT)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_signature_function():
    """The reference's own autograd wrapper (dgNN/operators/fused_gatconv.py:12-34) restated on our
    drop-in module: gat_forward with the reference's seven positional arguments, the SAME eleven saved
    tensors (the fourth result included, handed back untouched), gat_backward with its fourteen.  What
    INTEGRATION.md option A gives a user who keeps the reference's Python unmodified."""
    from dgnn_amd.native import fused_gatconv as fused_gat

    class FusedGATFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope, in_feat, attn_drop):
            out_feat, edge_max, edge_sum, edge_mask = fused_gat.gat_forward(attn_row, attn_col, row_ptr, col_ind,
                                                                            negative_slope, in_feat, attn_drop)
            ctx.save_for_backward(row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat,
                                  attn_row, attn_col)
            ctx.negative_slope, ctx.attn_drop = negative_slope, attn_drop
            return out_feat

        @staticmethod
        def backward(ctx, grad_out):
            (row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat, attn_row,
             attn_col) = ctx.saved_tensors
            grad_out = grad_out.contiguous()
            grad_feat, grad_attn_row, grad_attn_col = fused_gat.gat_backward(
                ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                edge_mask, in_feat, attn_row, attn_col, grad_out)
            return grad_attn_row, grad_attn_col, None, None, None, None, None, None, grad_feat, None

    return FusedGATFunction.apply


def cpu_baseline(name, seed=0):
    """The oracle (float32 restatement of the reference kernels, OpenMP over rows) timed on the
    host cores on a bounded sample: an R-MAT graph of the same shape family at 1/8 scale (tens
    of core-seconds).  kind = "port" (the reference has no CPU path)."""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import gat_inputs, rmat_graph
    n, target, scale = WORKLOADS[name]
    div = 8 if name != "cora" else 1          # ~2 s wall on 16 cores = ~30 core-seconds of CPU work
    ns, es, sc = max(n // div, 64), max(target // div, 256), max(scale - 3, 8) if div > 1 else scale
    g = rmat_graph(sc, ns, max(es - ns, 1), seed)
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], H, F, seed)
    cores = min(oracle.max_threads(), 16)
    oracle.set_threads(cores)
    t0 = time.perf_counter()
    out, emax, esum = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], SLOPE, in_feat, 0.0)
    oracle.gat_backward(SLOPE, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"],
                        emax, esum, None, in_feat, attn_row, attn_col, grad)
    dt = time.perf_counter() - t0
    oracle.set_threads(1)
    return {"value": g["nnz"] / dt, "unit": "edges/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "cpu_count": os.cpu_count(), "torch_threads": TORCH_THREADS,
            "sample": f"oracle f32 fwd+bwd on a 1/{div} scale R-MAT of the same shape "
                      f"(n={g['n']}, nnz={g['nnz']}, {H}x{F}), {dt:.2f} s wall"}


# ---------------------------------------------------------------------------------------------
# The other BASELINE.json configurations (configs[1]: fused GMMConv on a PubMed-shaped graph, also on
# the Reddit shape of test/gmm_result.csv:10; configs[2]: fused EdgeConv on a ModelNet40-shaped batch),
# same JSON schema.  Algorithmic bytes: SURVEY.md section 8(d) "EdgeConv / GMM algorithmic bytes".
GMM_K, GMM_D, GMM_F = 3, 2, 64            # test/gmm_result.csv:6 (hidden 64; the script's default is 16)
DGCNN = dict(batch=32, points=2048, k=20, dims=((3, 64), (64, 64), (64, 128), (128, 256)))   # train_edgeconv.py:244


def _events():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _time_steps(step_fn, steps, warmup, sync):
    for _ in range(warmup):
        step_fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    return time.perf_counter() - t0


def _host_info():
    return {"cpu": cpu_model(), "torch_threads": TORCH_THREADS, "cpu_count": os.cpu_count()}


def bench_gmm(args, device):
    """A step = GmmConvFuse forward + backward (autograd operator) over the whole graph."""
    import dgnn_amd  # noqa: F401
    from dgnn_amd import _lib
    from dgnn_amd.graph import prepare_graph
    from dgnn_amd.operators import GmmConvFuse
    K, D, F = GMM_K, GMM_D, GMM_F
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    if args.workload == "gmm-pubmed":
        n, e = 19_717, 88_648                  # directed edges, + one self loop per node
        src = torch.randint(0, n, (e,), device=device, generator=gen)
        dst = torch.randint(0, n, (e,), device=device, generator=gen)
        row_ptr, col_ind, col_ptr, row_ind, permute = prepare_graph(src, dst, n, "replace")
        shape = f"PubMed-shaped random graph (test/gmm_result.csv:6), n={n}"
    else:
        g = build_graph("reddit", device, seed=0, labels=args.labels)
        n = g["n"]
        row_ptr, col_ind, col_ptr, row_ind, permute = (g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute"))
        shape = f"Reddit-shaped R-MAT (test/gmm_result.csv:10), n={n}, node labels {args.labels}"
    nnz = int(col_ind.numel())
    # pseudo coordinates of the reference's MoNet script (train_gmmconv.py:120-122): 1/sqrt(deg) of both ends
    deg = (row_ptr[1:] - row_ptr[:-1]).float().clamp_(min=1)
    row_of = torch.repeat_interleave(torch.arange(n, device=device), (row_ptr[1:] - row_ptr[:-1]).long())
    pseudo = torch.stack([deg[col_ind.long()].rsqrt(), deg[row_of].rsqrt()], 1).contiguous().requires_grad_()
    del row_of
    x = torch.randn((n, K, F), device=device, generator=gen).requires_grad_()
    mu = (0.1 * torch.randn((K, D), device=device, generator=gen)).requires_grad_()     # gmmconv_layer.py:48-49
    sg = torch.ones((K, D), device=device).requires_grad_()
    go = torch.randn((n, K, F), device=device, generator=gen)

    def step_fn():
        x.grad = pseudo.grad = mu.grad = sg.grad = None
        GmmConvFuse(row_ptr, col_ind, col_ptr, row_ind, permute, x, pseudo, mu, sg).backward(go)

    torch.cuda.reset_peak_memory_stats(device)
    steps = args.steps if args.workload != "gmm-pubmed" else max(args.steps, 200)     # 50-us steps: time many
    dt = _time_steps(step_fn, steps, args.warmup, torch.cuda.synchronize)
    ms = dt / steps * 1e3
    peak = round(torch.cuda.max_memory_allocated(device) / 2**20)
    kf, kb = _lib.last_kernel_ms(0) or 0.0, _lib.last_kernel_ms(1) or 0.0
    b_fwd = nnz * (4 + 4 * D + 4 * K * F) + n * (4 * K * F + 4)
    # the one-pass backward: grad row gather + row_ind + permute + pseudo through permute + grad_gauss scatter per
    # edge; the column's own node_feat row, grad_feat row and pointer per node
    b_bwd = nnz * (4 * K * F + 8 + 4 * D + 4 * K) + n * (8 * K * F + 4)
    b_survey = nnz * (2 * (4 + 4 * D + 4 * K * F) + 8 * K + 12 * D + 4) + n * 12 * K * F     # the reference's three kernels
    name, nbytes, kms = (_lib.last_kernel_name(1), b_bwd, kb) if kb >= kf else (_lib.last_kernel_name(0), b_fwd, kf)
    roof = None
    if kms > 0:
        ach = nbytes / (kms * 1e-3) / 1e9
        inst = (name or "").split("dgnn::")[-1].split("(")[0]        # "k_bwd_col_gather<4, 2, 4, 1, true, 0>"
        traffic, traffic_src = measured_traffic(args.workload, inst)
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "traffic_from_profile": True,
                "kernel": name, "kernel_instance": name, "algorithmic_bytes": nbytes, "kernel_ms": round(kms, 4),
                "fwd_kernel_ms": round(kf, 4), "bwd_kernel_ms": round(kb, 4),
                "survey_reference_3kernel_backward_bytes": b_survey}
        if n * K * F * 4 <= 256 << 20:
            roof["note"] = ("feature table %.0f MB fits L2 / Infinity Cache: frac is against the HBM peak although the rows "
                            "come from cache" % (n * K * F * 4 / 1e6)) + \
                           ("; kernels of ~20 us: the step is launch-bound, not bandwidth-bound" if nnz < 1_000_000 else "")
    line = {"metric": "GMMConv fwd+bwd edges/s", "value": nnz / (ms * 1e-3), "unit": "edges/s", "n_gpus": 1, "steps": steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "peak_mem_MB": peak, "host": _host_info(),
            "config": {"workload": f"fused GmmConvFuse fwd+bwd, {shape}, nnz={nnz}, kernels={K}, pseudo_dim={D}, feat={F}",
                       "parallelism": "single GPU"}}
    if roof:
        line["roofline"] = roof
    if nnz < 1_000_000:
        # launch-bound sizes: the same step captured once in a HIP graph and replayed (the library neither
        # allocates nor synchronises) -- what is left when the host issues nothing per step.
        # An extra key: `value` stays the eager figure.
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    step_fn()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize()
            hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hip_graph):
                step_fn()
            dt_g = _time_steps(hip_graph.replay, steps, args.warmup, torch.cuda.synchronize)
            line["hip_graph_replay"] = {"ms_per_step": round(dt_g / steps * 1e3, 5), "value": nnz / (dt_g / steps),
                                        "note": "the step captured in one HIP graph (torch.cuda.CUDAGraph) and replayed"}
        except Exception as exc:  # pragma: no cover - reported, never fatal for the line
            line["hip_graph_replay"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_gmm(n, nnz, row_ptr, col_ind, col_ptr, row_ind, permute, pseudo)
    print(json.dumps(line), flush=True)


def cpu_baseline_gmm(n, nnz, row_ptr, col_ind, col_ptr, row_ind, permute, pseudo):
    """Oracle port of the reference's GMM step on the host: gmmconv forward, mhsddmm + gmmconv on the CSC +
    GaussianBp backward (dgNN/operators/fused_gmmconv.py:14-27).  PubMed shape: the whole graph; Reddit
    shape: the first rows holding ~1/64 of the edges (a bounded sample)."""
    import oracle
    K, D, F = GMM_K, GMM_D, GMM_F
    rp = row_ptr.cpu().numpy()
    rows = n if nnz <= 2_000_000 else int(np.searchsorted(rp, nnz // 64))
    e = int(rp[rows])
    sub_rp, sub_ci = rp[:rows + 1].copy(), col_ind[:e].cpu().numpy()
    ps = pseudo.detach()[:e].cpu().numpy()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, K, F)).astype(np.float32)
    go = rng.standard_normal((rows, K, F)).astype(np.float32)
    mu = (0.1 * rng.standard_normal((K, D))).astype(np.float32)
    sg = np.ones((K, D), np.float32)
    cores = min(oracle.max_threads(), 16)
    oracle.set_threads(cores)
    cp, ri, pm = oracle.csr2csc(sub_rp, sub_ci, n)
    t0 = time.perf_counter()
    oracle.gmmconv(sub_rp, sub_ci, x, ps, mu, sg)
    gg = oracle.mhsddmm(sub_rp, sub_ci, go, x)
    oracle.gmmconv(cp, ri, go, oracle.mhtranspose(pm, ps), mu, sg)
    oracle.gaussian_bp(ps, mu, sg, gg)
    dt = time.perf_counter() - t0
    oracle.set_threads(1)
    return {"value": e / dt, "unit": "edges/s", "cores": cores, "kind": "port", "cpu": cpu_model(), "cpu_count": os.cpu_count(),
            "torch_threads": TORCH_THREADS,
            "sample": f"oracle f32 GMM fwd + three-call bwd on {rows} rows / {e} edges of the same graph, {dt:.2f} s wall"}


def bench_dgcnn(args, device, world, rank, json_fd=None):
    """A step = the four EdgeConv layers of DGCNN (train_edgeconv.py:43-70) as the fused path sees them: per
    layer the k-NN graph in the layer's input space, then EdgeConvFuse forward + backward at the layer's
    width.  Data parallel over clouds (replicas only): every rank steps on its own 32
    clouds, no data-path collective; value = edges of all ranks / slowest rank."""
    import dgnn_amd  # noqa: F401
    from dgnn_amd import _lib
    from dgnn_amd.native.knn import knn_graph
    from dgnn_amd.operators import EdgeConvFuse
    B, P, k = DGCNN["batch"], DGCNN["points"], DGCNN["k"]
    m = B * P
    gen = torch.Generator(device=device)
    gen.manual_seed(rank)
    layers = []
    for din, f in DGCNN["dims"]:
        xin = torch.randn((B, P, din), device=device, generator=gen)
        if din > 3:                       # feature spaces sit behind an activation: positive, off-centre
            xin = torch.nn.functional.leaky_relu(xin + 1.0, 0.2)
        layers.append(dict(x=xin, hs=torch.randn((m, f), device=device, generator=gen).requires_grad_(),
                           hd=torch.randn((m, f), device=device, generator=gen).requires_grad_(),
                           go=torch.randn((m, f), device=device, generator=gen), f=f, src=None))

    def step_fn(with_knn=True):
        for L in layers:
            if with_knn or L["src"] is None:
                L["src"] = knn_graph(L["x"], k)
            L["hs"].grad = L["hd"].grad = None
            EdgeConvFuse(k, L["src"], L["hs"], L["hd"]).backward(L["go"])

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats(device)
    steps = max(args.steps, 50)
    dt = _time_steps(step_fn, steps, args.warmup, sync)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / steps * 1e3
    peak = round(torch.cuda.max_memory_allocated(device) / 2**20)
    ms_noknn = _time_steps(lambda: step_fn(False), steps, 2, torch.cuda.synchronize) / steps * 1e3
    # dominant EdgeConv kernel: the forward at the widest layer, event-timed by the library
    per = []
    for L in layers:
        EdgeConvFuse(k, L["src"], L["hs"], L["hd"]).backward(L["go"])
        torch.cuda.synchronize()
        per.append((L["f"], _lib.last_kernel_ms(0) or 0.0, _lib.last_kernel_ms(1) or 0.0, _lib.last_kernel_name(0)))
    f_dom, kf, kb, kname = max(per, key=lambda t: t[1])
    nbytes = m * (4 * k + 4 * k * f_dom + 4 * f_dom + 8 * f_dom)
    edges = 4 * m * k * world
    if rank != 0:
        return
    line = {"metric": "EdgeConv fwd+bwd edges/s", "value": edges / (ms * 1e-3), "unit": "edges/s", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "peak_mem_MB": peak, "host": _host_info(),
            "config": {"workload": f"DGCNN EdgeConv stack on a ModelNet40-shaped batch: {B} clouds x {P} points per GPU, k={k}, "
                                   f"widths {[d[1] for d in DGCNN['dims']]}, k-NN graph rebuilt per layer in dims "
                                   f"{[d[0] for d in DGCNN['dims']]} (INCLUDED in the step), fused EdgeConvFuse fwd+bwd",
                       "parallelism": "single GPU" if world == 1 else f"data parallel over clouds x{world} (replicas, no data-path collective)"},
            "knn_excluded": {"ms_per_step": round(ms_noknn, 4), "value": 4 * m * k / (ms_noknn * 1e-3),
                             "note": "rank 0 alone, graphs reused: the four EdgeConvFuse fwd+bwd only"},
            "per_layer_kernel_ms": [{"feat": f, "fwd": round(a, 4), "bwd_gather": round(b, 4)} for f, a, b, _ in per]}
    if kf > 0:
        ach = nbytes / (kf * 1e-3) / 1e9
        line["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel": f"{kname} at feat={f_dom}",
                            "kernel_instance": kname, "algorithmic_bytes": nbytes, "kernel_ms": round(kf, 4),
                            "note": "a point's neighbours lie in its own cloud (2 MB of rows at 256 features): the gather is "
                                    "served by L2, so frac against the HBM peak exceeds 1 by design"}
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_dgcnn()
    if json_fd is None:
        print(json.dumps(line), flush=True)
    else:                                  # fd 1 points at stderr while RCCL may print its banner (see main)
        os.write(json_fd, (json.dumps(line) + "\n").encode())


def cpu_baseline_dgcnn():
    """Oracle port on the host: k-NN graph + EdgeConv forward / backward of the four layers on a bounded
    sample of 2 clouds."""
    import oracle
    B, P, k = 2, DGCNN["points"], DGCNN["k"]
    rng = np.random.default_rng(0)
    cores = min(oracle.max_threads(), 16)
    oracle.set_threads(cores)
    t0 = time.perf_counter()
    for din, f in DGCNN["dims"]:
        x = rng.standard_normal((B, P, din)).astype(np.float32)
        src = oracle.knn_graph(x, k)
        hs, hd = (rng.standard_normal((B * P, f)).astype(np.float32) for _ in range(2))
        out, idx = oracle.edgeconv_forward(k, src, hs, hd)
        oracle.edgeconv_backward(out, idx.astype(np.int32))
    dt = time.perf_counter() - t0
    oracle.set_threads(1)
    return {"value": 4 * B * P * k / dt, "unit": "edges/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "cpu_count": os.cpu_count(), "torch_threads": TORCH_THREADS,
            "sample": f"oracle k-NN + EdgeConv fwd+bwd of the four layers on {B} clouds x {P} points, {dt:.2f} s wall "
                      "(array generation included)"}


def self_launch(n_ranks, argv):
    """Start `n_ranks` child ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1) from a parent that has not touched the GPU, wait for all of them,
    pass rank 0's stdout through.  No exec: the children are ordinary subprocesses."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # Watchdog: every child is polled; the first rank that exits non-zero (bad device index, OOM,
    # import error) takes the others down within seconds -- they would otherwise sit in RCCL's
    # bootstrap or in a collective until the process-group timeout -- and an overall limit
    # (DGNN_BENCH_LAUNCH_TIMEOUT seconds, default 900) bounds the whole run.
    limit = float(os.environ.get("DGNN_BENCH_LAUNCH_TIMEOUT", "900"))
    procs = []
    out0 = tempfile.TemporaryFile()           # rank 0's stdout (the JSON line): a file, so nothing blocks on a pipe
    try:
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                       LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        t0 = time.monotonic()
        why = None
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes):
                why = "a rank failed"
                break
            if time.monotonic() - t0 > limit:
                why = f"no result after {limit:.0f} s"
                break
            time.sleep(0.2)
    finally:
        # the exact children started above, by pid: terminate, then kill what ignores it
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in live) and time.monotonic() - t1 < 10.0:
            time.sleep(0.1)
        for p in live:
            if p.poll() is None:
                p.kill()
        codes = [p.wait() for p in procs]
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    out0.close()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if why is not None or bad:
        print(f"bench.py: {why or 'ranks failed'}; (rank, exit code) of the ranks that did not exit 0: {bad}",
              file=sys.stderr)
        return 1
    return 0


def dry_run(args):
    """The launcher alone: every rank joins a gloo group on the CPU, one all-reduce counts the
    ranks, rank 0 prints the line.  Touches no GPU."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen, backend = 1, None
    if os.environ.get("DGNN_BENCH_DRY_FAIL_RANK") == str(rank) and "RANK" in os.environ:
        sys.exit(3)        # launcher test: a rank that dies before the rendezvous (tests/test_bench_launcher.py)
    if "RANK" in os.environ:
        dist.init_process_group("gloo", timeout=PG_TIMEOUT)
        t = torch.ones(1)
        dist.all_reduce(t)
        seen, backend = int(t.item()), dist.get_backend()
        assert seen == dist.get_world_size() == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "GATConv fwd+bwd edges/s", "value": None, "unit": "edges/s", "dry": True,
                          "n_gpus": args.gpus, "ranks_seen": seen, "backend": backend}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="products", choices=list(WORKLOADS) + ["gmm-pubmed", "gmm-reddit", "dgcnn"],
                    help="products (default, the headline), reddit, cora: fused GAT; gmm-pubmed / gmm-reddit: fused "
                         "GMMConv (BASELINE configs[1]); dgcnn: fused EdgeConv stack with k-NN (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--labels", default="scrambled", choices=["scrambled", "natural"],
                    help="node ids of the R-MAT graph: scrambled = after the load-time relabelling of "
                         "dgnn_amd.graph.relabel_nodes (Graph500-style; the default), natural = the generator's ids; "
                         "the single-GPU line reports the other labelling too")
    ap.add_argument("--attn-drop", type=float, default=0.0,
                    help="secondary run (test/gat_result.sh:10 uses 0.5); the headline is 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the row-partitioned path even with one rank (sanity check of dgnn_amd.dist)")
    ap.add_argument("--simulate-rank", default=None, metavar="R/N",
                    help="time rank R of an N-rank partition ALONE on this GPU, exchanges replaced by local "
                         "copies of the same size (everything but the wire; results are not meaningful)")
    ap.add_argument("--strategy", default=None, choices=["auto", "heads", "rows"],
                    help="multi-GPU: heads = every rank runs the whole graph on heads/N heads, no exchange (needs heads %% N "
                         "== 0); rows = row partition with halo exchange; auto (default, or DGNN_DIST_STRATEGY): heads when possible")
    ap.add_argument("--graph", action="store_true",
                    help="with --simulate-rank / --force-dist: capture the partitioned step in a HIP graph and replay it")
    ap.add_argument("--no-other-labels", action="store_true",
                    help="skip the extra steps on the other node labelling (profiling runs: one labelling per trace)")
    ap.add_argument("--path", default="operator", choices=["operator", "reference-signature"],
                    help="operator: dgnn_amd.operators.GATConvFuse (the default); reference-signature: the reference's "
                         "own wrapper restated on the drop-in module (gat_forward / gat_backward with the reference's "
                         "positional arguments only).  The single-GPU line reports the other path too")
    ap.add_argument("--dry", action="store_true", help="launcher check: gloo all-reduce on the CPU, no GPU work")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1 and not args.simulate_rank:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))        # the parent never initialises the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and ("RANK" in os.environ and world > 1 or args.gpus > 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry:
        return dry_run(args)
    # a process group also for ONE rank when DGNN_DIST_FORCE_COLLECTIVES=1 (rehearsal of the whole
    # multi-GPU path, RCCL calls included, on a one-GPU box; launch through torch.distributed.run)
    use_pg = world > 1 or (os.environ.get("DGNN_DIST_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ)
    json_fd = None
    if use_pg:
        import torch.distributed as dist
        # RCCL writes a version banner to stdout (fd 1, from C) when its first communicator comes up:
        # keep stdout for the ONE JSON line by pointing fd 1 at stderr for everything else
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=PG_TIMEOUT)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    if args.workload in ("gmm-pubmed", "gmm-reddit"):
        if world > 1:
            print("bench.py: the GMMConv workloads are single-GPU configurations (BASELINE.json configs[1])", file=sys.stderr)
            sys.exit(2)
        return bench_gmm(args, device)
    if args.workload == "dgcnn":
        bench_dgcnn(args, device, world, rank, json_fd)
        if use_pg:
            torch.distributed.destroy_process_group()
        return

    import dgnn_amd  # noqa: F401
    from dgnn_amd.operators import GATConvFuse

    if use_pg:
        # one graph for everybody: rank 0 builds it, the index arrays are broadcast over RCCL
        import torch.distributed as dist
        g = build_graph(args.workload, device, seed=0, labels=args.labels) if rank == 0 else None
        meta = torch.tensor([g["n"], g["nnz"]] if rank == 0 else [0, 0], dtype=torch.int64, device=device)
        dist.broadcast(meta, src=0)
        n, nnz = int(meta[0]), int(meta[1])
        if rank != 0:
            g = dict(n=n, nnz=nnz,
                     row_ptr=torch.empty(n + 1, dtype=torch.int32, device=device),
                     col_ind=torch.empty(nnz, dtype=torch.int32, device=device),
                     col_ptr=torch.empty(n + 1, dtype=torch.int32, device=device),
                     row_ind=torch.empty(nnz, dtype=torch.int32, device=device),
                     permute=torch.empty(nnz, dtype=torch.int32, device=device))
        for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute"):
            dist.broadcast(g[k], src=0)
    else:
        g = build_graph(args.workload, device, seed=0, labels=args.labels)
    n, nnz = g["n"], g["nnz"]
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    # value distribution of the reference's synthetic test (script/test/test_gat_tb.py:100-102)
    in_feat = (torch.rand((n, H, F), device=device, generator=gen) - 0.4)
    attn_row = (torch.rand((n, H), device=device, generator=gen) - 0.6)
    attn_col = (torch.rand((n, H), device=device, generator=gen) - 0.4)
    grad_out = (torch.rand((n, H, F), device=device, generator=gen) - 0.5)

    if use_pg or args.force_dist or args.simulate_rank:
        # head split (no exchange) when the heads divide by the ranks, row partition + halo exchange otherwise
        # (dgnn_amd.dist.make_gat_runner; --strategy / DGNN_DIST_STRATEGY overrides)
        from dgnn_amd.dist import make_gat_runner
        if args.simulate_rank:
            sim_r, sim_n = (int(v) for v in args.simulate_rank.split("/"))
            runner = make_gat_runner(g, sim_n, sim_r, device, H, feat=F, strategy=args.strategy,
                                     fake_peers=True)    # chunks: DGNN_DIST_CHUNKS or the plan's
        else:
            runner = make_gat_runner(g, world, rank, device, H, feat=F, strategy=args.strategy)
        lo, hi = runner.lo, runner.hi
        # the rank's inputs live where the operator gathers from (as the GEMM in front of the
        # operator / autograd behind it would produce them): no copy of local rows inside a step
        if runner.mode == "heads":          # [lo, hi) is this rank's head range (and a feature slice beyond one head per rank)
            H_loc = hi - lo
            F_loc = F if runner.f_lo is None else runner.f_hi - runner.f_lo
            x_loc = runner.input_buffer("forward", (H_loc, F_loc))
            x_loc.copy_(runner.local(in_feat))
            g_loc = runner.input_buffer("backward", (H_loc, F_loc))
            g_loc.copy_(runner.local(grad_out))
            ar_loc, ac_loc = runner.local_scores(attn_row), runner.local_scores(attn_col)
        else:
            H_loc, F_loc = H, F
            x_loc = runner.input_buffer("forward", (H, F))
            x_loc.copy_(in_feat[lo:hi])
            g_loc = runner.input_buffer("backward", (H, F))
            g_loc.copy_(grad_out[lo:hi])
            ar_loc, ac_loc = attn_row[lo:hi].contiguous(), attn_col[lo:hi].contiguous()
        del in_feat, grad_out

        def step_fn():
            runner.forward(x_loc, ar_loc, ac_loc, SLOPE)
            return runner.backward(g_loc)

        if args.graph:
            # the whole partitioned step as ONE HIP graph (shapes and buffers are fixed between steps): what is left
            # when the host issues nothing per step.  Rehearsal runs only (--simulate-rank / --force-dist): RCCL
            # collectives under capture are untested on this pool.
            if use_pg:
                raise SystemExit("--graph: rehearsal runs only (--simulate-rank / --force-dist)")
            eager_step = step_fn
            for _ in range(3):
                eager_step()
            torch.cuda.synchronize()
            hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hip_graph):
                eager_step()

            def step_fn():
                hip_graph.replay()
    else:
        x = in_feat.requires_grad_()
        ar = attn_row.requires_grad_()
        ac = attn_col.requires_grad_()

        paths = {"operator": GATConvFuse, "reference-signature": reference_signature_function()}
        fuse = paths[args.path]

        def step_fn():
            x.grad = ar.grad = ac.grad = None
            out = fuse(ar, ac, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"],
                       g["permute"], SLOPE, x, args.attn_drop)
            out.backward(grad_out)

    def sync():
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    torch.cuda.reset_peak_memory_stats(device)     # graph construction is not part of a step
    for _ in range(args.warmup):
        step_fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn()
    enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3     # host time to issue a step (before the GPU is waited for)
    sync()
    dt = time.perf_counter() - t0
    # peak device memory over the warm-up and timed steps, inputs and graph resident (the reference
    # publishes peak memory per run: test/gat_result.csv)
    peak_mem_mb = round(torch.cuda.max_memory_allocated(device) / 2**20)
    if use_pg:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # Dominant-kernel timing, live: with DGNN_TIMING=1 the library brackets its forward
    # row-gather kernel and its backward column-gather kernel with HIP events on the stream it
    # launches on (dgnn_debug_last_kernel_ms); the whole forward / backward calls are also timed
    # with events on the same (torch current) stream.
    roof = None
    if not use_pg and not args.force_dist and not args.simulate_rank:
        from dgnn_amd import _lib
        from dgnn_amd.native import fused_gatconv
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = kf = kb = 0.0
        reps = max(3, min(args.steps, 10))
        xd, ard, acd = in_feat.detach(), attn_row.detach(), attn_col.detach()
        for _ in range(reps):
            ev[0].record()
            out, emax, esum, emask, out_neg, a_neg = fused_gatconv.gat_forward(
                ard, acd, g["row_ptr"], g["col_ind"], SLOPE, xd, args.attn_drop, neg=True)     # as the operator calls it
            ev[1].record()
            fused_gatconv.gat_backward(SLOPE, args.attn_drop, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"],
                                       g["permute"], emax, esum, emask, xd, ard, acd, grad_out, out_feat=out,
                                       out_neg=out_neg, a_neg=a_neg)
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
            kf += _lib.last_kernel_ms(0) or 0.0
            kb += _lib.last_kernel_ms(1) or 0.0
        tf, tb, kf, kb = tf / reps, tb / reps, kf / reps, kb / reps
        ab = algorithmic_bytes(n, nnz, H, F)
        if kf > 0 and kb > 0:
            dom = ("k_bwd_col_gather (backward, one CSC gather pass)", ab["bwd_csc"], kb, _lib.last_kernel_name(1)) if kb >= kf else \
                  ("k_fast_row_gather (forward, CSR gather pass)", ab["fwd"], kf, _lib.last_kernel_name(0))
        else:   # generic kernels in use: fall back to the call-level times
            dom = ("gat_backward call", ab["bwd_csc"], tb, None) if tb >= tf else ("gat_forward call", ab["fwd"], tf, None)
        achieved = dom[1] / (dom[2] * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(args.workload, dom[3])
        # device-copy bandwidth of this very GPU, same run: a 2 GiB
        # device-to-device copy, read + written bytes over the event-timed duration
        src = torch.empty(1 << 31, dtype=torch.uint8, device=device)
        dst = torch.empty_like(src)
        dst.copy_(src)
        ev[0].record()
        for _ in range(5):
            dst.copy_(src)
        ev[1].record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
        del src, dst
        roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_from_profile": True,      # PMC bytes of the committed rocprofv3 run named above, not of this run
                "kernel_instance": dom[3],
                "device_copy_GBps": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4),
                "kernel": dom[0], "algorithmic_bytes": dom[1], "kernel_ms": round(dom[2], 3),
                "fwd_kernel_ms": round(kf, 3), "bwd_kernel_ms": round(kb, 3),
                "fwd_call_ms": round(tf, 3), "bwd_call_ms": round(tb, 3),
                "fwd_kernel_GBps": round(ab["fwd"] / (kf * 1e-3) / 1e9, 1) if kf > 0 else None,
                "bwd_kernel_GBps": round(ab["bwd_csc"] / (kb * 1e-3) / 1e9, 1) if kb > 0 else None,
                "survey_3gather_model_GBps": round(ab["survey_total"] / (ms_per_step * 1e-3) / 1e9, 1)}
        table_bytes = n * H * F * 4
        if table_bytes <= 256 << 20:
            # the gathered table fits the 256 MiB Infinity Cache: the bytes counted are served by it,
            # not by HBM, so the HBM peak is not the bound (MI355X_MICROARCH.md, gather table:
            # uniformly random rows of a 151 MB table 7.4-7.9 TB/s, of a 38 MB table 8.6 TB/s)
            roof["note"] = (f"feature table {table_bytes / 1e6:.0f} MB is Infinity-Cache resident: frac is against the HBM "
                            "peak although the rows come from the cache (guide: 7.4-8.6 TB/s for random rows of "
                            "cache-resident tables); frac > 1 is expected here")

    if roof is None and (use_pg or args.force_dist or args.simulate_rank):
        # partitioned run: the same two kernels on this rank's blocks (forward on n_loc rows x
        # nnz_fwd edges, backward on n_loc columns x nnz_bwd edges); rank 0's figures are reported
        from dgnn_amd import _lib
        step_fn()
        torch.cuda.synchronize()
        kf, kb = _lib.last_kernel_ms(0) or 0.0, _lib.last_kernel_ms(1) or 0.0
        if kf > 0 and kb > 0:
            # the kernels timed are those of the LAST (largest) chunk of rank 0's block
            (rows_f, nnz_f), (cols_b, nnz_b) = runner.last_fwd, runner.last_bwd
            bf = algorithmic_bytes(rows_f, nnz_f, H_loc, F_loc)["fwd"]
            bb = algorithmic_bytes(cols_b, nnz_b, H_loc, F_loc)["bwd_csc"]
            name, nbytes, ms = ("k_bwd_col_gather", bb, kb) if kb >= kf else ("k_fast_row_gather", bf, kf)
            achieved = nbytes / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": (f"{name} on rank 0's slice ({H_loc} of {H} heads x {F_loc} of {F} features) of the whole graph"
                               if runner.mode == "heads" else
                               f"{name} on the last of {runner.chunks} chunk(s) of rank 0's block ({rows_f} rows x "
                               f"{nnz_f} edges forward, {cols_b} columns x {nnz_b} edges backward; the block: "
                               f"{runner.n_loc} nodes, {runner.nnz_fwd} / {runner.nnz_bwd} edges)"),
                    "algorithmic_bytes": nbytes, "kernel_ms": round(ms, 3),
                    "fwd_kernel_ms": round(kf, 3), "bwd_kernel_ms": round(kb, 3)}
    if rank == 0:
        line = {
            "metric": "GATConv fwd+bwd edges/s", "value": nnz / (ms_per_step * 1e-3), "unit": "edges/s",
            "n_gpus": world, "ranks_seen": torch.distributed.get_world_size() if use_pg else 1,
            "backend": (f"{torch.distributed.get_backend()} (RCCL {'.'.join(str(v) for v in torch.cuda.nccl.version())})"
                        if use_pg else None),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "peak_mem_MB": peak_mem_mb, "host_enqueue_ms_per_step": round(enqueue_ms, 3),
            "host": {"cpu": cpu_model(), "torch_threads": TORCH_THREADS, "cpu_count": os.cpu_count()},
            "config": {"workload": f"{args.workload}-shaped R-MAT, n={n}, nnz={nnz}, heads={H}, feat={F}, "
                                   f"attn_drop={args.attn_drop:g}, fused GATConvFuse fwd+bwd, node labels {args.labels}"
                                   + (" (load-time relabelling, dgnn_amd.graph.relabel_nodes)" if args.labels == "scrambled" else ""),
                       "parallelism": "single GPU" if world == 1 else
                       (f"head-parallel x{world}: every rank runs the whole graph on {H_loc} of the {H} heads"
                        + (f" x {F_loc} of the {F} features" if F_loc != F else "") + f"; exchange: {runner.plan['exchange']}"
                        if runner.mode == "heads" else
                        f"row-partition x{world}, RCCL {runner.mode} exchange of boundary rows"
                        + (f", {runner.chunks} pipelined chunks" if runner.mode == "halo" else ""))},
        }
        if roof is not None:
            line["roofline"] = roof
        if use_pg or args.force_dist or args.simulate_rank:
            # a PREDICTION of this step over real links (dgnn_amd.dist.PartitionedGAT.expected_ms): the curve a
            # multi-GPU record is to be judged against; speedup against the measured single-GPU step
            exp = runner.expected_ms(H_loc, F_loc)
            exp["single_gpu_ms_per_step"] = SINGLE_GPU_MS.get(args.workload)
            n_ranks = sim_n if args.simulate_rank else world
            alone = RANK_ALONE_MS.get(args.workload, {}).get(n_ranks) if (runner.mode == "heads" and (H, F) == (4, 64)) else None
            if alone:        # measured beats modelled: the narrow slices run below the wide rows' gather rate
                exp["model_ms_per_step"] = exp["ms_per_step"]
                exp["rank_alone_ms_per_step_measured"] = alone
                exp["ms_per_step"] = round(alone + exp.get("all_reduce_ms", 0.0), 3)
            if exp["single_gpu_ms_per_step"]:
                exp["speedup_over_one_gpu"] = round(exp["single_gpu_ms_per_step"] / exp["ms_per_step"], 2)
            line["expected"] = exp
        if args.simulate_rank:
            if runner.mode == "heads":
                line["simulated"] = (f"rank {sim_r} of {sim_n} alone on its slice ({H_loc} of {H} heads x {F_loc} of {F} features; "
                                     f"head-parallel: no rows are exchanged, so this IS a rank's step time"
                                     + ("" if runner.q == 1 else " up to one small all-reduce")
                                     + "): value = its share of the work / time, NOT a multi-GPU result")
                line["value"] = nnz * (H_loc * F_loc) / (H * F) / (ms_per_step * 1e-3)
            else:
                line["simulated"] = (f"rank {sim_r} of {sim_n} alone, exchanges as local copies: "
                                     f"value = this rank's {runner.nnz_fwd} edges / time, NOT a multi-GPU result")
                line["value"] = runner.nnz_fwd / (ms_per_step * 1e-3)
        if world == 1 and not use_pg and not args.force_dist and not args.simulate_rank:
            # the same steps through the other calling convention (INTEGRATION.md option A / the operator)
            line["config"]["path"] = args.path
            other_path = "reference-signature" if args.path == "operator" else "operator"
            fuse = paths[other_path]
            for _ in range(2):
                step_fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k_path = max(3, min(args.steps, 10))
            for _ in range(k_path):
                step_fn()
            torch.cuda.synchronize()
            ms_path = (time.perf_counter() - t0) / k_path * 1e3
            line["other_path"] = {"path": other_path, "ms_per_step": round(ms_path, 3), "steps": k_path,
                                  "value": nnz / (ms_path * 1e-3), "ratio_to_timed_path": round(ms_path / ms_per_step, 4)}
            fuse = paths[args.path]
            x.grad = ar.grad = ac.grad = None
        if (world == 1 and not use_pg and not args.force_dist and not args.simulate_rank and args.workload != "cora"
                and not args.no_other_labels):
            # the same step on the other labelling of the same graph (continuity with round 1,
            # which ran on the generator's ids)
            other = "natural" if args.labels == "scrambled" else "scrambled"
            del g
            torch.cuda.empty_cache()
            g = build_graph(args.workload, device, seed=0, labels=other)
            for _ in range(2):
                step_fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k_other = max(3, min(args.steps, 10))
            for _ in range(k_other):
                step_fn()
            torch.cuda.synchronize()
            ms_other = (time.perf_counter() - t0) / k_other * 1e3
            line["other_labels"] = {"labels": other, "ms_per_step": round(ms_other, 3), "steps": k_other,
                                    "value": nnz / (ms_other * 1e-3)}
        if not args.no_cpu_baseline and world == 1 and not args.simulate_rank:      # a reported baseline of the N = 1 run only
            line["cpu_baseline"] = cpu_baseline(args.workload)
        if json_fd is None:
            print(json.dumps(line), flush=True)
        else:
            os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
