"""Data parallel over point clouds for the EdgeConv / DGCNN path (BASELINE.json configs[2],
dgnn_amd.dist.CloudDataParallel): after the weighted gradient all-reduce every rank holds the
gradient of the mean loss over the WHOLE batch, with rank 0's parameters.

CPU: world_size 2 and 3 over gloo (5 clouds: uneven split), pure-torch float64 layer stack.
GPU: world_size 2 over gloo, both ranks on cuda:0, dgnn_amd.layers.EdgeConv / KNNGraph.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
from torch import nn

import dp_worker
from test_dist import free_port

HERE = os.path.dirname(os.path.abspath(__file__))
BATCH, POINTS, K = 5, 40, 6


def run_world(world, mode):
    port = free_port()
    with tempfile.TemporaryDirectory() as out_dir:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), str(world), mode,
                                   str(port), out_dir, str(BATCH), str(POINTS), str(K)]) for r in range(world)]
        try:
            for p in procs:
                assert p.wait(timeout=300) == 0, "a rank failed"
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        return [dict(np.load(os.path.join(out_dir, f"rank{r}.npz"))) for r in range(world)]


def full_batch(mode):
    """Single process, whole batch, rank 0's parameters."""
    model = dp_worker.build(mode, K, seed=100)
    x, y = dp_worker.data(mode, BATCH, POINTS)
    nn.functional.cross_entropy(model(x), y).backward()
    return dp_worker.grads_of(model), {n: p.detach().cpu().double().numpy() for n, p in model.named_parameters()}


def check(parts, mode, rtol, atol):
    grads, params = full_batch(mode)
    los, his = [int(p["lo"]) for p in parts], [int(p["hi"]) for p in parts]
    assert los[0] == 0 and his[-1] == BATCH and all(his[i] == los[i + 1] for i in range(len(parts) - 1))
    assert max(h - l for l, h in zip(los, his)) - min(h - l for l, h in zip(los, his)) <= 1
    for part in parts:
        for n, ref in params.items():
            assert np.array_equal(part["p:" + n], ref), f"parameter {n} was not broadcast from rank 0"
        for n, ref in grads.items():
            np.testing.assert_allclose(part["g:" + n], ref, rtol=rtol, atol=atol, err_msg=n)
    assert np.abs(grads["conv.0.theta.weight"]).max() > 0 and np.all(parts[0]["g:unused"] == 0)


@pytest.mark.parametrize("world", [2, 3])
def test_cloud_data_parallel_cpu_gloo(world):
    check(run_world(world, "cpu"), "cpu", rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_cloud_data_parallel_gpu_two_ranks():
    # float32 on the GPU: per-cloud results are identical (exact k-NN ids and arg-max), only the
    # summation order of the weight gradients differs
    check(run_world(2, "gpu"), "gpu", rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_cloud_data_parallel_nccl_ranks():
    """Real RCCL, one GPU per rank: runs wherever at least two GPUs are visible (the driver's multi-GPU
    node), skipped on a one-GPU box."""
    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one GPU per rank)")
    check(run_world(min(n_gpu, 4), "nccl"), "gpu", rtol=1e-4, atol=1e-6)


def test_cloud_range_covers_the_batch():
    from dgnn_amd.dist import cloud_range
    for batch in (1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            r = [cloud_range(batch, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == batch and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
