"""Host-side enqueue time per view against the GPU time of the same work (is the host ahead of the GPU?)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import bench, synth
from diff_gaussian_rasterization import ParamGradSink
dev = torch.device("cuda:0")
P, W, H = 6_000_000, 1920, 1080
sc = synth.street_scene(P, W, H, seed=2)
params = bench.build_params(sc, dev)
rs = bench.make_settings(sc["camera"], dev, sc["sh_degree"])
gt = torch.rand((3, H, W), device=dev)
sink = ParamGradSink(*params.values(), defer=True)
def view(timing=None):
    t0 = time.perf_counter()
    out = bench.render_params(params, rs, sink)
    t1 = time.perf_counter()
    loss = bench.train_loss(out, gt)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    if timing is not None:
        timing.append((t1 - t0, t2 - t1, t3 - t2))
for _ in range(3):
    sink.begin_step(); view(); sink.flush()
torch.cuda.synchronize()
tm = []
t0 = time.perf_counter()
for k in range(5):
    sink.begin_step()
    for v in range(8):
        view(tm)
    sink.flush()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 40
a = np.array(tm) * 1e3
print("per view wall %.3f ms; host: forward call (incl. waiting at the sync) %.3f, loss enqueue %.3f, backward enqueue %.3f" %
      (tot * 1e3, a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean()))
# the same with the GPU idle when the host starts (sync before every view): pure host cost of the forward call
tm2 = []
for k in range(10):
    torch.cuda.synchronize(); sink.begin_step(); view(tm2); sink.flush()
b = np.array(tm2) * 1e3
print("GPU idle at start: forward call %.3f (GPU forward part before sync ~0.55), loss %.3f, backward %.3f" %
      (b[:, 0].mean(), b[:, 1].mean(), b[:, 2].mean()))
