"""Shared pieces of the example models: synthetic datasets of the reference's benchmark shapes
(no dataset can be downloaded here), device-side graph preparation, and the reference scripts'
measurement protocol (10 warm-up epochs, n timed epochs bracketed by device synchronisation, one
CSV line: script/train/train_gatconv.py:125-172)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dgnn_amd.graph import prepare_graph  # noqa: E402,F401

# name: (nodes, directed edges before self loops, input features, classes)
CITATION_SHAPES = {
    "cora": (2708, 10556, 1433, 7),
    "citeseer": (3327, 9104, 3703, 6),
    "pubmed": (19717, 88648, 500, 3),
    "reddit": (232965, 114615892, 602, 41),
    "products": (2449029, 62000000, 100, 47),      # R-MAT of bench.py's headline shape
}


def synthetic_citation(name, device, seed=0):
    """Random graph + features + labels of a citation dataset's shape, self loops removed then
    added as the reference scripts do (train_gatconv.py:75-76)."""
    n, e, n_feats, n_classes = CITATION_SHAPES[name]
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    if name == "products":
        import bench                               # skewed R-MAT graph, self loops included
        # node ids relabelled at load time (dgnn_amd.graph.relabel_nodes), as bench.py does
        g = bench.build_graph("products", device, seed, labels="scrambled")
        graph = tuple(g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute"))
    else:
        src = torch.randint(0, n, (e,), device=device, generator=gen)
        dst = torch.randint(0, n, (e,), device=device, generator=gen)
        graph = prepare_graph(src, dst, n)      # self loops removed then added (train_gatconv.py:75-76)
    features = torch.randn((n, n_feats), device=device, generator=gen)
    labels = torch.randint(0, n_classes, (n,), device=device, generator=gen)
    train_mask = torch.rand(n, device=device, generator=gen) < 0.3
    return dict(n=n, graph=graph, features=features, labels=labels, n_feats=n_feats,
                n_classes=n_classes, train_mask=train_mask, test_mask=~train_mask)


def time_epochs(fn, n_epochs, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start = time.time()
    for _ in range(n_epochs):
        fn()
    torch.cuda.synchronize()
    return (time.time() - start) / n_epochs


def capture_step(step_fn, warmup=3):
    """One call of ``step_fn`` as a HIP graph (torch.cuda.CUDAGraph): a small-graph epoch is a few
    dozen launches of microsecond kernels, so replaying it as ONE graph launch removes the host
    from the loop (Cora-sized GAT training epoch: 0.83 ms eager -> 0.53 ms replayed).  The inputs
    must be static tensors and the optimizer capturable; the library itself neither allocates nor
    synchronises, so its launches capture as they are.  Keep random-number ops (dropout) out of the
    captured step: capturing them aborted the process on this ROCm build.  -> (replay, value step_fn returned during
    capture: static tensors that every replay overwrites)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step_fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        result = step_fn()
    torch.cuda.synchronize()
    return graph.replay, result


def max_memory_mb(device):
    return int(torch.cuda.max_memory_allocated(device) // (1 << 20))


def accuracy(logits, labels):
    return (logits.argmax(1) == labels).float().mean().item()


def append_csv(path, line):
    print(line)
    if path:
        with open(path, "a") as f:
            print(line, file=f)


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() (mean over rows) as log_softmax + a gather + a mean: torch's
    nll_loss kernel reduces all rows in ONE workgroup (1.2 ms forward + 0.9 ms backward for the
    196 k training rows of the products-shaped run); the gather and the mean are ordinary
    parallel kernels.  Same value up to the order of the float sum."""
    if logits.shape[0] < 8192:       # launch-bound sizes: the two-kernel library form is shorter
        return torch.nn.functional.cross_entropy(logits, labels)
    logp = torch.log_softmax(logits, dim=1)
    return -logp.gather(1, labels.unsqueeze(1)).mean()
