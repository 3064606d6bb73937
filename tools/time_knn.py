"""Times simple_knn._C.distCUDA2 on the points of the street scene (HIP events, mean of 3 calls after one warm-up).
usage: time_knn.py [POINTS]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import synth
from simple_knn._C import distCUDA2

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
pts = torch.tensor(synth.street_scene(P, 1920, 1080, seed=2)["means3D"], dtype=torch.float32, device="cuda")
distCUDA2(pts)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    d = distCUDA2(pts)
e1.record()
torch.cuda.synchronize()
print("distCUDA2 %d points: %.2f ms; mean dist2 %.6g" % (P, e0.elapsed_time(e1) / 3, float(d.mean())))
