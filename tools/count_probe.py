"""Per-workgroup start / end timestamps of the two blend kernels from an instrumented build
(GSPLAT_AMD_LIB=tools/lib_wgt.so: a copy of csrc/ in which both kernels store wall_clock64() at entry and
exit per workgroup into a __device__ array gs_wgt_f / gs_wgt_b and export it through gsr_dbg_wgt_f / _b,
built with tools/build_variant_dir.sh): how many
workgroups run concurrently over the kernel's life, how long the tail is, how the duration
relates to the entries a tile walks.  Development tool, not part of the product."""
import ctypes as C
import os
import sys

import numpy as np
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
NT = ((W + 15) // 16) * ((H + 15) // 16)

for it in range(3):
    sink.begin_step()
    out = bench.render_params(params, rs, sink)
    bench.train_loss(out, gt).backward()
    sink.flush()
    torch.cuda.synchronize()

for tag in ("f", "b"):
    buf = (C.c_ulonglong * (4 * NT))()
    rc = getattr(L, "gsr_dbg_wgt_" + tag)(buf, NT)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NT, 4).astype(np.int64)
    t0, t1, n = a[:, 0], a[:, 1], a[:, 2]
    base = t0.min()
    t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0   # microseconds (100 MHz counter)
    dur = t1 - t0
    span = t1.max()
    print(f"== {tag}: rc {rc}, workgroups {NT}, span {span:.1f} us, duration mean {dur.mean():.1f} median {np.median(dur):.1f} "
          f"p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us; "
          f"sum {dur.sum():.0f} us -> mean concurrency {dur.sum() / span:.0f}")
    # concurrency timeline in 20 slices
    edges = np.linspace(0, span, 21)
    conc = [(np.minimum(t1, e1) - np.maximum(t0, e0)).clip(min=0).sum() / (e1 - e0) for e0, e1 in zip(edges[:-1], edges[1:])]
    print("   concurrency per 5% slice:", " ".join(f"{c:.0f}" for c in conc))
    print(f"   last start at {t0.max():.1f} us; entries walked: mean {n.mean():.0f} max {n.max()}; "
          f"corr(duration, entries) {np.corrcoef(dur, n)[0, 1]:.2f}; us per entry (sum/sum) {dur.sum() / max(1, n.sum()):.3f}")
    order = np.argsort(-dur)[:5]
    print("   longest:", [(int(i), round(float(dur[i]), 1), int(n[i]), round(float(t0[i]), 1)) for i in order])
