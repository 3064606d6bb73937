"""Times fused_ssim.l1_ssim_loss on one 3 x H x W image plus l1_loss on a 4 x H x W feature image, forward +
backward (HIP events, mean of 200 runs).
usage: time_loss.py [W H]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import fused_ssim

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
g = torch.Generator(device="cuda").manual_seed(0)
img = torch.rand((3, H, W), device="cuda", generator=g).requires_grad_(True)
gt = torch.rand((3, H, W), device="cuda", generator=g)
rep = (10 * torch.rand((4, H, W), device="cuda", generator=g)).requires_grad_(True)


def once():
    img.grad = None
    rep.grad = None
    (fused_ssim.l1_ssim_loss(img, gt) + 0.01 * fused_ssim.l1_loss(rep, 10.0)).backward()


for _ in range(20):
    once()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(200):
    once()
e1.record()
torch.cuda.synchronize()
print("l1_ssim_loss + l1_loss fwd+bwd %dx%d: %.4f ms" % (W, H, e0.elapsed_time(e1) / 200))
