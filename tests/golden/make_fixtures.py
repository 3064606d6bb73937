"""Generates the golden vectors under tests/golden/ by importing the parts of
the reference that run on a CPU in the build container:

    utils/sh_utils.eval_sh              (SH -> RGB, same constants as auxiliary.h:21-38)
    utils/graphics_utils.getWorld2View2 / getProjectionMatrix / focal2fov
    utils/loss_utils.ssim               (conv2d SSIM, value and autograd gradient;
                                         the pin submodules/fused-ssim/tests/test.py:82-91 uses)
    utils/loss_utils.l1_loss + ssim     (the training loss as train.py:222-230, 248-251 composes it,
                                         value and autograd gradient through the clamp)

Run once here:  python tests/golden/make_fixtures.py   (needs /root/reference).
Only the resulting .npz files (inputs + expected outputs) are committed; the
reference itself never travels.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    from utils.graphics_utils import focal2fov, getProjectionMatrix, getWorld2View2
    from utils.loss_utils import ssim
    from utils.sh_utils import eval_sh

    rng = np.random.default_rng(20240611)

    # --- SH evaluation -------------------------------------------------
    N = 257
    sh = rng.standard_normal((N, 16, 3)).astype(np.float32)       # rasterizer layout [P, M, 3]
    pos = (rng.standard_normal((N, 3)) * 2).astype(np.float32)
    campos = np.array([0.3, -0.2, 1.5], np.float32)
    d = torch.tensor(pos) - torch.tensor(campos)
    d = d / d.norm(dim=1, keepdim=True)
    sh_t = torch.tensor(sh).transpose(1, 2)                        # eval_sh wants [..., C, coeffs]
    rgb = {f"rgb_deg{deg}": torch.clamp_min(eval_sh(deg, sh_t, d) + 0.5, 0.0).numpy() for deg in range(4)}
    np.savez(os.path.join(OUT, "sh_eval.npz"), sh=sh, pos=pos, campos=campos, **rgb)

    # --- camera matrices ----------------------------------------------
    cams = {}
    for i in range(4):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        r, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        t = rng.standard_normal(3) * 3
        fovx = float(rng.uniform(0.4, 1.4))
        W, H = int(rng.integers(200, 2000)), int(rng.integers(200, 1200))
        fovy = focal2fov(W / (2 * np.tan(fovx / 2)), H)
        view = torch.tensor(getWorld2View2(R, t, np.array([0.1, -0.2, 0.3]), 1.5)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = view.inverse()[3, :3]
        cams.update({f"R{i}": R, f"t{i}": t, f"fov{i}": np.array([fovx, fovy, W, H]), f"view{i}": view.numpy(),
                     f"proj{i}": proj.numpy(), f"full{i}": full.numpy(), f"center{i}": center.numpy()})
    np.savez(os.path.join(OUT, "camera_matrices.npz"), **cams)

    # --- SSIM value and gradient --------------------------------------
    out = {}
    for i, shape in enumerate([(1, 3, 48, 64), (2, 1, 33, 29), (1, 3, 12, 80)]):
        a = rng.random(shape).astype(np.float32)
        b = np.clip(a + 0.15 * rng.standard_normal(shape), 0, 1).astype(np.float32)
        ta = torch.tensor(a, requires_grad=True)
        val = ssim(ta, torch.tensor(b))
        val.backward()
        out.update({f"img1_{i}": a, f"img2_{i}": b, f"ssim_{i}": np.float32(val.item()), f"grad_{i}": ta.grad.numpy()})
    np.savez(os.path.join(OUT, "ssim_ref.npz"), **out)
    # --- the photometric training loss as train.py:222-230 composes it, and the feature-map L1 of :248-251
    from utils.loss_utils import l1_loss
    out = {}
    lam = 0.2
    for i, shape in enumerate([(3, 40, 72), (3, 17, 33)]):
        raw = (rng.random(shape) * 1.4 - 0.2).astype(np.float32)        # an unclamped render: some values outside [0, 1]
        gt = rng.random(shape).astype(np.float32)
        t = torch.tensor(raw, requires_grad=True)
        image = t.clamp(0, 1)                                            # gaussian_renderer/__init__.py:179
        Ll1 = l1_loss(image, torch.tensor(gt))
        ssim_value = ssim(image, torch.tensor(gt))
        loss = (1.0 - lam) * Ll1 + lam * (1.0 - ssim_value)
        loss.backward()
        rep = (rng.standard_normal((4,) + shape[1:]) * 3 + 8).astype(np.float32)
        tr = torch.tensor(rep, requires_grad=True)
        Ll1rep = l1_loss(tr, torch.full_like(tr, 10.0))
        Ll1rep.backward()
        out.update({f"raw_{i}": raw, f"gt_{i}": gt, f"loss_{i}": np.float32(loss.item()), f"l1_{i}": np.float32(Ll1.item()),
                    f"ssim_{i}": np.float32(ssim_value.item()), f"grad_{i}": t.grad.numpy(), f"rep_{i}": rep,
                    f"l1rep_{i}": np.float32(Ll1rep.item()), f"gradrep_{i}": tr.grad.numpy()})
    out["lambda_dssim"] = np.float32(lam)
    np.savez(os.path.join(OUT, "train_loss_ref.npz"), **out)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
