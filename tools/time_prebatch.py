"""Times gsr_preprocess_batch alone (batched forward preprocess + scans + batched depth order) on the bench
scene: python tools/time_prebatch.py [gaussians].  GSPLAT_AMD_LIB selects the library."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import synth  # noqa: E402
import _gsplat_lib as G  # noqa: E402
from diff_gaussian_rasterization import _C  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
W, H, V = 1920, 1080, 8
dev = torch.device("cuda:0")
sc = synth.street_scene(P, W, H, seed=2)
p = bench.build_params(sc, dev)
views = [bench.make_settings(synth.street_camera(W, H, v, V), dev, sc["sh_degree"]) for v in range(V)]
tm = bench.KernelTimer(1)
tm.arm(0)
ms = {"preprocess_batch": [], "depth_order_batch": []}
for it in range(6):
    with torch.no_grad():
        _C.preprocess_batch(p["xyz"], p["opacity"], p["scaling"], p["rotation"], 1.0, p["features_dc"],
                            p["features_rest"], 3, p["reps"], True, views)
    torch.cuda.synchronize()
    if it:
        ev = tm.events[0]
        ms["preprocess_batch"].append(ev[16].elapsed_time(ev[17]))
        ms["depth_order_batch"].append(ev[19].elapsed_time(ev[20]))
print(os.environ.get("GSPLAT_AMD_LIB", "in-tree"), {k: round(float(np.mean(v)), 4) for k, v in ms.items()})
