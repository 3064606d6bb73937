"""End to end: a short optimisation with the operator packages, the raw-parameter
path, the deferred multi-view backward, fused-ssim, the fused Adam step and the
densification statistics, on a small scene with a known answer (the target
images are renders of the unperturbed parameters).  The loss has to fall."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_short_optimisation_converges():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, ParamGradSink
    from fused_ssim import fused_ssim
    from gsplat_optim import FusedGaussianAdam, add_densification_stats
    dev = "cuda:0"
    W, H, P = 128, 96, 1500
    sc = synth.tiny_scene(P=P, width=W, height=H, seed=77, scale=0.08)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    op = np.clip(sc["opacities"].reshape(-1, 1), 1e-3, 1 - 1e-3)
    truth = dict(xyz=t(sc["means3D"]), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]),
                 opacity=t(np.log(op / (1 - op))), scaling=t(np.log(sc["scales"])), rotation=t(sc["rotations"]),
                 reps=t(sc["reps"]))
    cams = []
    for v in range(3):
        ang = 0.3 * (v - 1)
        cam = synth.make_camera(W, H, 0.9, (0.1 + 2.5 * np.sin(ang), -0.05, -2.5 * np.cos(ang)), (0.0, 0.0, 1.0))
        cams.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
            campos=t(cam["campos"]), prefiltered=False, debug=False, antialiasing=True))

    def render(p, rs, sink=None, m2=None):
        m2 = torch.zeros_like(p["xyz"], requires_grad=True) if m2 is None else m2
        return GaussianRasterizer(rs).forward_params(p["xyz"], m2, p["f_dc"], p["f_rest"], p["opacity"], p["scaling"],
                                                     p["rotation"], p["reps"], grad_sink=sink), m2

    with torch.no_grad():
        targets = [render(truth, rs)[0][0].clamp(0, 1) for rs in cams]
    g = torch.Generator(device=dev).manual_seed(3)
    noise = lambda x, s: x + s * torch.randn(x.shape, device=dev, generator=g)
    params = dict(xyz=noise(truth["xyz"], 0.02), f_dc=noise(truth["f_dc"], 0.3), f_rest=noise(truth["f_rest"], 0.05),
                  opacity=noise(truth["opacity"], 0.3), scaling=noise(truth["scaling"], 0.1),
                  rotation=noise(truth["rotation"], 0.05), reps=truth["reps"].clone())
    params = {k: torch.nn.Parameter(v.contiguous()) for k, v in params.items()}
    lrs = dict(xyz=2e-3, f_dc=1e-2, f_rest=5e-4, opacity=2e-2, scaling=5e-3, rotation=2e-3, reps=1e-3)
    opt = FusedGaussianAdam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in params], lr=0.0, eps=1e-15)
    sink = ParamGradSink(*[params[k] for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "reps")],
                         defer=True)
    acc, den, mx = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros((P,), device=dev)

    def batch_loss(backward):
        total, keep = 0.0, []
        for rs, gt in zip(cams, targets):
            out, m2 = render(params, rs, sink if backward else None)
            img = out[0].clamp(0, 1)
            loss = 0.8 * (img - gt).abs().mean() + 0.2 * (1 - fused_ssim(img[None], gt[None]))
            if backward:
                loss.backward()
                keep.append((out[1], m2))
            total += float(loss.detach())
        return total / len(cams), keep

    with torch.no_grad():
        first, _ = batch_loss(False)
    for it in range(60):
        sink.begin_step()
        _, keep = batch_loss(True)
        sink.flush()
        for radii, m2 in keep:
            add_densification_stats(radii, m2.grad, acc, den, mx)
        opt.step()
    with torch.no_grad():
        last, _ = batch_loss(False)
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    assert float(den.max()) == 180.0 and float(acc.sum()) > 0 and float(mx.max()) > 0
    for p in params.values():
        assert torch.isfinite(p).all()
