"""How many (wave, instance) visits the blend kernels would make under different work mappings, counted
with PyTorch on the lists and records a forward pass of the library leaves behind (a sample of tiles).

  quadrant : a wave owns an 8x8 quadrant and visits every instance the tile walks (what the kernels do);
             `quad_hit` of those visits find a pixel that passes the alpha test.
  rows     : a row of 16 lanes owns a 4x4 block and only visits the instances that can reach the block
             (exactly, or by the bounding box of the alpha >= 1/255 ellipse); the four rows of a wave step
             together, so a wave makes max-over-its-rows visits per round of 128 instances.

usage: blend_stats.py P W H [translucent] [ntiles]"""
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

BATCH = 128


def main():
    P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    translucent = len(sys.argv) > 4 and sys.argv[4] == "translucent"
    nsample = int(sys.argv[5]) if len(sys.argv) > 5 else 300
    sc = synth.street_scene(P, W, H, seed=2, translucent=translucent)
    rs, ten = scene_to_torch(sc, antialiasing=True, debug=False)
    o = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    R = o["num_rendered"]
    lay = GsrLayout()
    assert lib().gsr_layout(P, W, H, R, ctypes.byref(lay)) == 0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    dev = o["img"].device
    ncon = o["img"][lay.n_contrib:lay.n_contrib + W * H * 4].view(torch.int32).reshape(H, W)
    ranges = o["img"][lay.ranges:lay.ranges + gx * gy * 8].view(torch.int32).reshape(-1, 2)
    rec = o["geom"][lay.rec:lay.rec + P * 64].view(torch.float32).reshape(P, 16)
    v = (lay.vals0, lay.vals1)[lay.sorted_buffer]
    plist = o["binning"][v:v + R * 4].view(torch.int32)

    rng = np.random.default_rng(0)
    tiles = rng.choice(gx * gy, size=min(nsample, gx * gy), replace=False)
    tot = dict(walked=0, quad_visits=0, quad_hit=0, rows_exact=0, rows_box=0, rows_exact_sum=0, rows_box_sum=0,
               fwd_walked=0, fwd_quad_hit=0, fwd_rows_exact=0, pairs_hit=0)
    hist = torch.zeros(16, dtype=torch.int64)
    py, px = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
    blk = ((py // 4) * 4 + (px // 4)).reshape(-1)                    # 4x4 block of every pixel of the tile
    quad = ((py // 8) * 2 + (px // 8)).reshape(-1)
    blk_quad = torch.tensor([(b // 4 // 2) * 2 + (b % 4) // 2 for b in range(16)], device=dev)
    for t in tiles:
        tx, ty = int(t % gx), int(t // gx)
        x0, y0 = tx * 16, ty * 16
        n = torch.zeros((16, 16), dtype=torch.int64, device=dev)
        hh, ww = min(16, H - y0), min(16, W - x0)
        n[:hh, :ww] = ncon[y0:y0 + hh, x0:x0 + ww]
        n = n.reshape(-1)
        Lw = int(n.max())
        if Lw == 0:
            continue
        r0 = int(ranges[t, 0])
        ids = plist[r0:r0 + Lw].long()
        rc = rec[ids]
        x, y, a, b, c, op = rc[:, 0:1], rc[:, 1:2], rc[:, 2:3], rc[:, 3:4], rc[:, 4:5], rc[:, 5:6]
        tau = torch.clamp(-torch.log(255.0 * op) - 0.01, min=-87.0)
        X = (x0 + px.reshape(1, -1)).float()
        Y = (y0 + py.reshape(1, -1)).float()
        dx, dy = x - X, y - Y
        power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy        # [Lw, 256]
        reach = (power <= 0) & (power >= tau)
        alpha = torch.clamp(op * torch.exp(power), max=0.99)
        j = torch.arange(Lw, device=dev).reshape(-1, 1)
        hit = reach & (alpha >= 1.0 / 255.0) & (j < n.reshape(1, -1))   # what the backward blends
        tot["pairs_hit"] += int(hit.sum())
        tot["walked"] += Lw
        tot["quad_visits"] += 4 * Lw
        qh = torch.zeros((Lw, 4), dtype=torch.bool, device=dev)
        for q in range(4):
            qh[:, q] = hit[:, quad == q].any(1)
        tot["quad_hit"] += int(qh.sum())
        for q in range(4):   # how many of a quadrant's 64 pixels a blending visit blends
            cq = hit[:, quad == q].sum(1)
            cq = cq[cq > 0]
            hist += torch.bincount(torch.clamp((cq - 1) // 4, max=15), minlength=16).cpu()
        # rows: per block, the instances that can reach it and that one of its pixels still needs
        blk_last = torch.zeros(16, dtype=torch.int64, device=dev)
        need_exact = torch.zeros((Lw, 16), dtype=torch.bool, device=dev)
        for bb in range(16):
            blk_last[bb] = n[blk == bb].max()
            need_exact[:, bb] = reach[:, blk == bb].any(1)
        live_blk = j < blk_last.reshape(1, -1)
        need_exact &= live_blk
        # bounding box of the ellipse power >= tau
        L = torch.clamp(-2.0 * tau, min=0.0)
        det = a * c - b * b
        ok = (det > 0) & (a > 0) & (c > 0)
        hx = torch.where(ok, torch.sqrt(L * c / det.clamp(min=1e-30)), torch.full_like(a, 1e9)) * 1.001 + 0.01
        hy = torch.where(ok, torch.sqrt(L * a / det.clamp(min=1e-30)), torch.full_like(a, 1e9)) * 1.001 + 0.01
        bxs = torch.arange(4, device=dev).float().reshape(1, -1)
        colok = (x - hx <= x0 + 4 * bxs + 3) & (x + hx >= x0 + 4 * bxs)      # [Lw, 4]
        rowok = (y - hy <= y0 + 4 * bxs + 3) & (y + hy >= y0 + 4 * bxs)
        need_box = (rowok.unsqueeze(2) & colok.unsqueeze(1)).reshape(Lw, 16) & live_blk & (tau <= 0)
        assert not (need_exact & ~need_box).any(), "the box test dropped a block a pixel reaches"
        # walking order is back to front; rounds of BATCH instances starting from the last walked one
        order = torch.arange(Lw - 1, -1, -1, device=dev)
        for name, need in (("rows_exact", need_exact), ("rows_box", need_box)):
            nd = need[order]
            pad = (-Lw) % BATCH
            nd = torch.nn.functional.pad(nd, (0, 0, 0, pad)).reshape(-1, BATCH, 16).sum(1)   # [rounds, 16]
            per_wave = torch.stack([nd[:, blk_quad == q].max(1).values for q in range(4)], 1)
            tot[name] += int(per_wave.sum())
            tot[name + "_sum"] += int(nd.sum())
    s = tot
    print(("translucent" if translucent else "street"), "tiles sampled", len(tiles))
    print("  walked instances                         ", s["walked"])
    print("  blended (pixel, instance) pairs          ", s["pairs_hit"], " per walked instance %.1f" % (s["pairs_hit"] / s["walked"]))
    print("  quadrant visits (now)                    ", s["quad_visits"], " of which find a pixel: %.3f" % (s["quad_hit"] / s["quad_visits"]))
    print("  row scheme, exact reach: wave visits     ", s["rows_exact"], " (block visits %d, %.2f blocks per instance)" %
          (s["rows_exact_sum"], s["rows_exact_sum"] / s["walked"]))
    print("  row scheme, box reach:   wave visits     ", s["rows_box"], " (block visits %d, %.2f blocks per instance)" %
          (s["rows_box_sum"], s["rows_box_sum"] / s["walked"]))
    print("  pixels blended per blending quadrant visit, share of visits by 1-4, 5-8, ..., 61-64:",
          " ".join("%.3f" % (float(h) / max(1, int(hist.sum()))) for h in hist))
    full, skip = 160.0, 20.0
    now = s["quad_hit"] * full + (s["quad_visits"] - s["quad_hit"]) * skip
    print("  instruction estimate (160 per blending visit, 20 per skipped one): now %.3g, rows exact %.3g (%.2fx), rows box %.3g (%.2fx)" %
          (now, s["rows_exact"] * full, now / max(1.0, s["rows_exact"] * full), s["rows_box"] * full,
           now / max(1.0, s["rows_box"] * full)))


if __name__ == "__main__":
    main()
