"""Workload counters of a synthetic scene on the GPU: visible Gaussians, instances R, list entries the
blend kernels walk, (pixel, Gaussian) pairs tested.  usage: scene_stats.py P W H [logit_mean scale_factor]..."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from _gsplat_lib import GsrLayout, lib
from gpu_util import run_forward_raw, scene_to_torch


def stats(sc, W, H):
    P = sc["means3D"].shape[0]
    rs, ten = scene_to_torch(sc, antialiasing=True, debug=False)
    o = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    R = o["num_rendered"]
    lay = GsrLayout()
    assert lib().gsr_layout(P, W, H, R, ctypes.byref(lay)) == 0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    n = o["img"][lay.n_contrib:lay.n_contrib + W * H * 4].view(torch.int32).reshape(H, W)
    n = torch.nn.functional.pad(n, (0, gx * 16 - W, 0, gy * 16 - H))
    per_tile = n.reshape(gy, 16, gx, 16).amax(dim=(1, 3)).reshape(-1).float()
    walked = int(per_tile.sum(dtype=torch.float64))
    q = torch.quantile(per_tile, torch.tensor([0.5, 0.9, 0.99, 1.0], device=per_tile.device)).tolist()
    print("   walked per tile: median %.0f, p90 %.0f, p99 %.0f, max %.0f; list length max %d" %
          (q[0], q[1], q[2], q[3], int((o["img"][lay.ranges:lay.ranges + gx * gy * 8].view(torch.int32).reshape(-1, 2)[:, 1] -
                                         o["img"][lay.ranges:lay.ranges + gx * gy * 8].view(torch.int32).reshape(-1, 2)[:, 0]).max())))
    # per quadrant (8x8) share of instances whose splat reaches it with alpha >= 1/255 is not observable here
    vis = int((o["radii"] > 0).sum())
    pairs = int(o["gauss_count"].sum(dtype=torch.int64))
    return dict(visible=vis, R=R, R_per_visible=round(R / max(vis, 1), 2), walked=walked, walked_over_R=round(walked / max(R, 1), 3),
                pairs_tested=pairs, mean_T=round(float(o["accum_alpha"].mean()), 3))


if __name__ == "__main__":
    P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    print("opaque", stats(synth.street_scene(P, W, H, seed=2), W, H), flush=True)
    rest = sys.argv[4:]
    for i in range(0, len(rest), 2):
        m, f = float(rest[i]), float(rest[i + 1])
        print("translucent", m, f, stats(synth.street_scene(P, W, H, seed=2, translucent=True, opacity_logit_mean=m,
                                                          scale_factor=f), W, H), flush=True)
