"""Multi-process (gloo, world_size 2, CPU) test of the view-sharded step: the
all-reduced gradient equals the sum over views of single-process gradients,
and the densification statistics reduce with sum / sum / max."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    g = torch.Generator().manual_seed(0)
    P = 257
    params = [torch.randn(P, 3, generator=g), torch.randn(P, 16, 3, generator=g), torch.randn(P, 1, generator=g)]
    views = [torch.randn(P, generator=g) for _ in range(5)]   # 5 views over 2 ranks: ragged split
    return params, views


def _loss(params, v):
    # any differentiable per-view function of all parameters
    a, b, c = params
    return ((a.sum(1) * v).tanh().sum() + (b.sum((1, 2)) * v * v).sin().sum() + (c[:, 0] * v).pow(2).sum())


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gsdp
    params, views = _make_problem()
    params = [p.clone().requires_grad_(True) for p in params]
    gsdp.view_sharded_step(params, views, lambda v: _loss(params, v))
    stats = [torch.full((4,), float(rank + 1)), torch.full((4,), 2.0 * (rank + 1)), torch.tensor([1.0, 5.0, 2.0, 0.0]) * (rank + 1)]
    gsdp.reduce_densify_stats(*stats)
    torch.save(dict(grads=[p.grad for p in params], stats=stats), os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_shard_views_partitions_the_batch():
    import gsdp
    for n in (0, 1, 5, 8, 13):
        for w in (1, 2, 4, 8):
            parts = [gsdp.shard_views(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_view_sharded_step_world2(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    params, views = _make_problem()
    params = [p.clone().requires_grad_(True) for p in params]
    sum(_loss(params, v) for v in views).backward()
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True)
        for g, p in zip(got["grads"], params):
            assert torch.allclose(g, p.grad, rtol=1e-5, atol=1e-6)
        assert torch.equal(got["stats"][0], torch.full((4,), 3.0))
        assert torch.equal(got["stats"][1], torch.full((4,), 6.0))
        assert torch.equal(got["stats"][2], torch.tensor([2.0, 10.0, 4.0, 0.0]))


def test_single_process_is_a_no_op_reduce():
    import gsdp
    params, views = _make_problem()
    params = [p.clone().requires_grad_(True) for p in params]
    gsdp.view_sharded_step(params, views, lambda v: _loss(params, v))
    ref = [p.clone().detach().requires_grad_(True) for p in params]
    sum(_loss(ref, v) for v in views).backward()
    for a, b in zip(params, ref):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
