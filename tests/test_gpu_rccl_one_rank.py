"""The gradient exchange through RCCL itself, as far as one GPU allows: a process group of ONE rank on
the "nccl" backend (RCCL on ROCm).  The sums are trivial, the calls are not: initialisation bound to
the device, the coalesced row-range all-reduce issued while later ranges are still being computed,
the 8-bit MAX all-reduce of the row masks, the gathered sparse exchange, the asynchronous work
objects the compute stream waits on.  The two-rank arithmetic is covered on CPU tensors by
tests/test_gsdp_gloo.py; RCCL refuses two ranks on one device."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import synth
from gpu_util import rel_err
from test_gpu_params_mode import _arc_settings, _raw_scene

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


@pytest.fixture(scope="module")
def one_rank_group():
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _one_rank_reducer(**kw):
    import gsdp

    class OneRank(gsdp.GradReducer):   # a group of one is a no-op for GradReducer unless told otherwise
        def _active(self):
            return dist.is_initialized()
    return OneRank(**kw)


@pytest.mark.parametrize("mode,chunks,nstreams", [("never", 3, 0), ("never", 6, 3), ("always", 1, 2), ("auto", 3, 3)])
def test_reduce_flush_through_rccl(one_rank_group, mode, chunks, nstreams):
    import gsdp
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    nviews, W, H = 5, 96, 64
    sc = synth.tiny_scene(P=1500, width=W, height=H, seed=47)
    raw = _raw_scene(sc, seed=9)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names], defer=True)
    sink_b = ParamGradSink(*[B[n] for n in names], defer=True, keep_zero_rows=True)
    settings = _arc_settings(nviews, W, H)
    rng = np.random.default_rng(19)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(nviews)]

    def loss_of(P_, sink):
        def f(i, rs):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(rs).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            return (o[0] * ws[i][0]).sum() + (o[2] * ws[i][1]).sum() + (o[5] * ws[i][2]).sum()
        return f

    gsdp.run_views(sink_a, settings, loss_of(A, sink_a))
    red = _one_rank_reducer(sparse=mode)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for _ in range(3):   # the exchange writes into the persistent buffers: later steps must start clean
        gsdp.run_views(sink_b, settings, loss_of(B, sink_b), streams, reducer=red, chunks=chunks)
    torch.cuda.synchronize()
    if mode == "never":
        assert red.calls > 0 and red.last_union is None
    else:
        assert red.last_union is not None and 0 < red.last_union[0] <= red.last_union[1] == 1500
    for n in names:
        ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))


def test_timing_reduction_and_barrier(one_rank_group):
    """What bench.py does around its timed region with more than one rank."""
    dist.barrier()
    tmax = torch.tensor([1.25], device="cuda:0", dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    assert float(tmax.item()) == 1.25
    buf = torch.arange(1 << 20, device="cuda:0", dtype=torch.float32)
    _one_rank_reducer().start([buf]).wait()
    torch.cuda.synchronize()
    assert float(buf[-1].item()) == float((1 << 20) - 1)
