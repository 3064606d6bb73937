"""Reads the debug counters of an instrumented build (GSPLAT_AMD_LIB=tools/lib_count.so):
per-wave instance visits / threshold passes of the two blend kernels and the spread of
workgroup durations, on the bench workload.  Development tool, not part of the product."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import bench  # noqa: E402
import synth  # noqa: E402
import _gsplat_lib as G  # noqa: E402
from diff_gaussian_rasterization import ParamGradSink  # noqa: E402

P, W, H = 6_000_000, 1920, 1080
dev = torch.device("cuda:0")
sc = synth.street_scene(P, W, H, seed=2)
params = bench.build_params(sc, dev)
rs = bench.make_settings(synth.street_camera(W, H, 0, 8), dev, sc["sh_degree"])
gt = torch.rand((3, H, W), device=dev)
sink = ParamGradSink(*params.values(), defer=True)
L = G.lib()
buf = (C.c_ulonglong * 16)()


def read(fn, reset=1):
    getattr(L, fn)(buf, reset)
    return list(buf)


for it in range(3):
    read("gsr_dbg_fwd"), read("gsr_dbg_bwd")
    sink.begin_step()
    out = bench.render_params(params, rs, sink)
    bench.train_loss(out, gt).backward()
    sink.flush()
    torch.cuda.synchronize()
f, b = read("gsr_dbg_fwd"), read("gsr_dbg_bwd")
print("fwd: wave-instances visited %d, hit %d, pass tau %d, blend %d, rounds %d" % (f[0], f[1], f[2], f[3], f[8]))
print("fwd: workgroups %d, list entries total %d; duration sum %.1f us, mean %.2f us, max %.2f us, kernel span %.1f us"
      % (f[9], f[10], f[4] / 100, f[4] / 100 / max(1, f[9]), f[5] / 100, (f[7] - f[6]) / 100))
print("fwd: mean concurrency %.1f workgroups" % (f[4] / max(1, f[7] - f[6])))
print("bwd: wave-instances visited %d, pass tau %d, valid %d" % (b[0], b[2], b[3]))
print("bwd: workgroups %d, walked entries total %d; duration sum %.1f us, mean %.2f us, max %.2f us, kernel span %.1f us"
      % (b[9], b[10], b[4] / 100, b[4] / 100 / max(1, b[9]), b[5] / 100, (b[7] - b[6]) / 100))
print("bwd: mean concurrency %.1f workgroups" % (b[4] / max(1, b[7] - b[6])))
