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


# ---------------------------------------------------------------- ParamGradSink -> GradReducer, chunked
def _sink_problem():
    g = torch.Generator().manual_seed(3)
    P = 1000   # four blocks of 256 rows, the last one partial
    shapes = dict(xyz=(P, 3), dc=(P, 1, 3), rest=(P, 15, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4), reps=(P, 4))
    params = {k: torch.randn(*s, generator=g) for k, s in shapes.items()}
    views = [torch.randn(P, generator=g) for _ in range(5)]
    return params, views


def _view_grad(p, w):
    """The stand-in for one view's per-Gaussian backward: row i of every gradient is w[i] * sin(parameter row)."""
    return w.view(-1, *([1] * (p.dim() - 1))) * torch.sin(p)


def _make_fake_sink(params):
    from diff_gaussian_rasterization import ParamGradSink

    class FakeSink(ParamGradSink):
        """The kernels replaced by CPU arithmetic with the same contract: rows [first, first + count) of every
        gradient are overwritten or added to, rows outside are not touched."""
        launches = []

        def _geometry(self, cfg, views, first, count, accumulate):
            self.launches.append((first, count, len(views), accumulate))
            with torch.no_grad():
                for k, g in self.grads.items():
                    rows = sum(_view_grad(self.params[k], v["w"])[first:first + count] for v in views)
                    if accumulate:
                        g[first:first + count] += rows
                    else:
                        g[first:first + count] = rows

    leaves = [params[k].clone().requires_grad_(True) for k in ("xyz", "dc", "rest", "opacity", "scaling", "rotation", "reps")]
    sink = FakeSink(*leaves, defer=True)
    return sink, leaves


def _sink_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gsdp
    params, views = _sink_problem()
    sink, leaves = _make_fake_sink(params)
    assert sink.defer
    out = {}
    for name, nviews, chunks in (("five_views_3_chunks", 5, 3), ("one_view_8_chunks", 1, 8), ("five_views_unchunked", 5, 1)):
        for p in leaves:
            p.grad = None              # what optimizer.zero_grad(set_to_none=True) does between steps
        sink.begin_step()
        sink.flat.fill_(float("nan"))  # stale contents must not survive
        for vi in gsdp.shard_views(nviews, rank, world):
            sink.pending.append(dict(cfg=(1.0, 3, False), w=views[vi], means2D=torch.zeros(3, requires_grad=True),
                                     dL_dmeans2D=torch.ones(3)))
        red = gsdp.GradReducer()
        seen = []
        grads = sink.tensors()

        def on_chunk(first, count):
            seen.append((first, count))
            red.start_rows(grads, first, count)

        sink.flush(chunks=chunks, on_chunk=on_chunk)
        red.wait()
        assert all(p.grad is sink.grads[k] for p, k in zip(leaves, sink.NAMES)), "gradients are not the sink's buffers"
        assert all(f % 256 == 0 for f, _ in seen) and sum(c for _, c in seen) == 1000
        out[name] = dict(grads=[p.grad.clone() for p in leaves], chunks=seen)
    torch.save(out, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_param_grad_sink_chunked_reduce_world2(tmp_path):
    world = 2
    mp.start_processes(_sink_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    params, views = _sink_problem()
    order = ("xyz", "dc", "rest", "opacity", "scaling", "rotation", "reps")
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"s{r}.pt"), weights_only=True)
        for name, nviews in (("five_views_3_chunks", 5), ("one_view_8_chunks", 1), ("five_views_unchunked", 5)):
            for g, k in zip(got[name]["grads"], order):
                ref = sum(_view_grad(params[k], views[v]) for v in range(nviews))
                assert torch.allclose(g, ref, rtol=1e-5, atol=1e-6), (name, k, r)
        assert len(got["five_views_3_chunks"]["chunks"]) == 3
        assert len(got["one_view_8_chunks"]["chunks"]) == 4     # 1000 rows are four blocks of 256
        # chunked and unchunked reduce give the same sums (same per-row addition order on both ranks)
        for a, b in zip(got["five_views_3_chunks"]["grads"], got["five_views_unchunked"]["grads"]):
            assert torch.equal(a, b)


def test_param_grad_sink_flat_buffer_and_reattach():
    params, _ = _sink_problem()
    sink, leaves = _make_fake_sink(params)
    # every gradient is a contiguous view into the one flat buffer, in NAMES order
    base = sink.flat.data_ptr()
    last = -1
    for k in sink.NAMES:
        g = sink.grads[k]
        assert g.is_contiguous() and g.shape == sink.params[k].shape
        off = g.data_ptr() - base
        assert off > last and off % 256 == 0 and off + g.numel() * 4 <= sink.flat.numel() * 4
        last = off
    leaves[0].grad = None
    sink.begin_step()
    assert leaves[0].grad is sink.grads["means3D"]
    sink.params["means3D"] = torch.zeros(7, 3)
    with pytest.raises(RuntimeError):
        sink.begin_step()


# ---------------------------------------------------------------- sparse exchange of the touched rows
def _sparse_views(P, nviews, density):
    """Per-view row weights that are zero outside a random subset of the rows (the Gaussians a view reached)."""
    g = torch.Generator().manual_seed(11)
    out = []
    for _ in range(nviews):
        w = torch.randn(P, generator=g)
        w[torch.rand(P, generator=g) >= density] = 0.0
        out.append(w)
    return out


def _sparse_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gsdp
    params, _ = _sink_problem()
    out = {}
    for name, density, mode in (("sparse_auto", 0.03, "auto"), ("dense_auto", 0.6, "auto"), ("dense_forced_sparse", 0.6, "always"),
                                ("never", 0.03, "never"), ("nothing_touched", 0.0, "auto"), ("sparse_wide_mask", 0.03, "auto")):
        views = _sparse_views(1000, 5, density)
        sink, leaves = _make_fake_sink(params)
        sink.begin_step()
        sink.flat.fill_(float("nan"))
        # the bookkeeping of keep_zero_rows (the stand-in kernels do not keep it: what shows afterwards is what
        # the exchange reported as written by other ranks)
        sink.row_blocks = torch.zeros(((1000 + 63) // 64,), dtype=torch.uint8)
        for vi in gsdp.shard_views(5, rank, world):
            sink.pending.append(dict(cfg=(1.0, 3, False), w=views[vi], means2D=torch.zeros(3, requires_grad=True),
                                     dL_dmeans2D=torch.ones(3)))
            sink.hit_counts.append(views[vi].ne(0).to(torch.int32))   # what the forward's gauss_count says
        red = gsdp.GradReducer(sparse=mode)
        if name == "sparse_wide_mask":
            red.mask_as_bytes = False     # the row masks as int32 with SUM, for backends without 8-bit MAX
        took = gsdp.reduce_flush(sink, red, chunks=3)
        out[name] = dict(grads=[p.grad.clone() for p in leaves], took=took, union=red.last_union,
                         launches=len(sink.launches), dirty_blocks=sink.row_blocks.clone())
        type(sink).launches.clear()
    # three dense steps in a row: the fourth one does not probe (no mask exchange), the 16th plan does
    red = gsdp.GradReducer(sparse="auto", probe_every=4)
    dense = torch.ones(1000, dtype=torch.uint8)
    calls = []

    def mask():
        calls.append(1)
        return dense.clone()

    for _ in range(8):
        assert red.plan_rows(mask) is None
    out["probes"] = len(calls)
    torch.save(out, os.path.join(out_dir, f"p{rank}.pt"))
    dist.destroy_process_group()


def test_sparse_row_exchange_world2(tmp_path):
    world = 2
    mp.start_processes(_sparse_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    params, _ = _sink_problem()
    order = ("xyz", "dc", "rest", "opacity", "scaling", "rotation", "reps")
    expect = dict(sparse_auto=("sparse", 0.03), dense_auto=("dense", 0.6), dense_forced_sparse=("sparse", 0.6),
                  never=("dense", 0.03), nothing_touched=("sparse", 0.0), sparse_wide_mask=("sparse", 0.03))
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"p{r}.pt"), weights_only=True)
        for name, (took, density) in expect.items():
            views = _sparse_views(1000, 5, density)
            assert got[name]["took"] == took, (name, got[name]["took"])
            for g, k in zip(got[name]["grads"], order):
                ref = sum(_view_grad(params[k], v) for v in views)
                assert torch.allclose(g, ref, rtol=1e-5, atol=1e-6), (name, k, r)
            touched = torch.stack([v.ne(0) for v in views]).any(0)
            if name != "never":
                assert tuple(got[name]["union"]) == (int(touched.sum()), 1000)
            # rows that other ranks made non-zero are reported to the sink: the union's blocks of 64 after a
            # sparse exchange, every block after a dense one
            pad = torch.zeros((16 * 64,), dtype=torch.bool)
            pad[:1000] = touched
            want = pad.view(16, 64).any(1) if took == "sparse" else torch.ones((16,), dtype=torch.bool)
            assert torch.equal(got[name]["dirty_blocks"].bool(), want), name
        # the sparse path flushes in one launch, the dense one in three ranges
        assert got["sparse_auto"]["launches"] == 1 and got["dense_auto"]["launches"] == 3
        # 8 plans with probe_every 4: plans 1-3 probe (streak builds), then only plans 4 and 8
        assert got["probes"] == 5
