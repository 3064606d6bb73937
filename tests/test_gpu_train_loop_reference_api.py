"""A train.py-shaped loop on the GPU (train.py:188-368 restated): one random camera per iteration,
render through the operator, L1 + fused-ssim + feature loss, backward, densification statistics from
`viewspace_points.grad`, densify / prune every N iterations (the model's size changes: every buffer of
the next forward has another size, the optimizer state is cut and extended row by row), opacity reset,
Adam step, zero_grad(set_to_none=True).

Two variants of the same loop:
  reference  the unchanged reference signature, exactly what gaussian_renderer/__init__.py:86-169 calls:
             activations and torch.cat in PyTorch, `GaussianRasterizer(settings)(means3D=..., shs=..., ...)`,
             gradients through autograd into the leaves;
  params     `forward_params` on the raw parameters with a ParamGradSink, rebuilt after every resize and
             after the opacity reset (both replace parameter objects).
The model starts as a noisy part of the scene that made the target images plus transparent junk, so that
densification both adds rows (clone, split) and removes them (prune).  Asserted: the loss falls, every
tensor stays finite, no sort reported an error, the size went up and went down, the Adam moments follow
the rows."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "reps")


def _settings(W, H, n, dev):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    out = []
    for v in range(n):
        ang = 0.25 * (v - (n - 1) / 2)
        cam = synth.make_camera(W, H, 0.9, (0.1 + 2.5 * np.sin(ang), -0.05, -2.5 * np.cos(ang)), (0.0, 0.0, 1.0))
        out.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
            campos=t(cam["campos"]), prefiltered=False, debug=False, antialiasing=True))
    return out


def _render_reference(p, rs):
    """gaussian_renderer/__init__.py:49-54, 86-169: the placeholder for the screen-space gradient, the model's
    activations (scene/gaussian_model.py:470-502) and the rasterizer call with the reference's keywords."""
    from diff_gaussian_rasterization import GaussianRasterizer
    screenspace_points = torch.zeros_like(p["xyz"], requires_grad=True) + 0
    screenspace_points.retain_grad()
    out = GaussianRasterizer(raster_settings=rs)(
        means3D=p["xyz"], means2D=screenspace_points, shs=torch.cat((p["f_dc"], p["f_rest"]), dim=1),
        colors_precomp=None, opacities=torch.sigmoid(p["opacity"]), scales=torch.exp(p["scaling"]),
        rotations=torch.nn.functional.normalize(p["rotation"]), cov3D_precomp=None, reps=p["reps"])
    return out, screenspace_points


def _render_params(p, rs, sink):
    from diff_gaussian_rasterization import GaussianRasterizer
    m2 = torch.zeros_like(p["xyz"], requires_grad=True)
    out = GaussianRasterizer(rs).forward_params(p["xyz"], m2, p["f_dc"], p["f_rest"], p["opacity"], p["scaling"],
                                                p["rotation"], p["reps"], grad_sink=sink)
    return out, m2


def _reset_opacity(opt, params):
    """GaussianModel.reset_opacity + replace_tensor_to_optimizer (scene/gaussian_model.py:433-458)."""
    with torch.no_grad():
        new = torch.min(torch.sigmoid(params["opacity"]), torch.full_like(params["opacity"], 0.01))
        new = torch.log(new / (1 - new))      # inverse_sigmoid, utils/general_utils.py:18-19
    for group in opt.param_groups:
        if group["name"] == "opacity":
            st = opt.state.get(group["params"][0], None)
            if st is not None:
                st["exp_avg"] = torch.zeros_like(new)
                st["exp_avg_sq"] = torch.zeros_like(new)
                del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(new.requires_grad_(True))
            if st is not None:
                opt.state[group["params"][0]] = st
            params["opacity"] = group["params"][0]


@pytest.mark.parametrize("api", ["reference", "params"])
def test_training_loop_with_densification(api):
    from _gsplat_lib import sort_errors
    from diff_gaussian_rasterization import ParamGradSink
    from fused_ssim import fused_ssim
    from gsplat_optim import FusedGaussianAdam, add_densification_stats, densify_and_prune
    dev = "cuda:0"
    W, H, P_truth = 160, 112, 2400
    sort_errors.poll(block=True)
    sc = synth.tiny_scene(P=P_truth, width=W, height=H, seed=77, scale=0.07)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    op = np.clip(sc["opacities"].reshape(-1, 1), 1e-3, 1 - 1e-3)
    truth = dict(xyz=t(sc["means3D"]), f_dc=t(sc["shs"][:, :1]), f_rest=t(sc["shs"][:, 1:]),
                 opacity=t(np.log(op / (1 - op))), scaling=t(np.log(sc["scales"])), rotation=t(sc["rotations"]),
                 reps=t(sc["reps"]))
    cams = _settings(W, H, 5, dev)
    with torch.no_grad():
        targets = [_render_params(truth, rs, None)[0][0].clamp(0, 1) for rs in cams]

    # the start: every third Gaussian of the scene, perturbed, a little too large (so that the view-space
    # gradients ask for splits), plus junk far below the prune threshold (sigmoid(-7) < 0.005)
    g = torch.Generator(device=dev).manual_seed(3)
    noise = lambda x, s: x + s * torch.randn(x.shape, device=dev, generator=g)
    keep = torch.arange(0, P_truth, 3, device=dev)
    start = {k: v[keep].clone() for k, v in truth.items()}
    start.update(xyz=noise(start["xyz"], 0.03), f_dc=noise(start["f_dc"], 0.3), opacity=noise(start["opacity"], 0.3),
                 scaling=start["scaling"] + 0.25)
    n_junk = 600
    junk = {k: v[1:3 * n_junk:3].clone() for k, v in truth.items()}
    junk["opacity"] = torch.full_like(junk["opacity"], -7.0)
    params = {k: torch.nn.Parameter(torch.cat((start[k], junk[k])).contiguous()) for k in NAMES}
    # training_setup (scene/gaussian_model.py:97-113): one group per tensor, the reference's names
    lrs = dict(xyz=1.6e-3, f_dc=2.5e-3, f_rest=2.5e-3 / 20, opacity=2.5e-2, scaling=5e-3, rotation=1e-3, reps=2.5e-3)
    opt = FusedGaussianAdam([{"params": [params[k]], "lr": lrs[k], "name": k} for k in NAMES], lr=0.0, eps=1e-15)
    P0 = params["xyz"].shape[0]
    stats = dict(xyz_gradient_accum=torch.zeros((P0, 1), device=dev), denom=torch.zeros((P0, 1), device=dev),
                 max_radii2D=torch.zeros((P0,), device=dev))
    sink = ParamGradSink(*[params[k] for k in NAMES]) if api == "params" else None

    iterations, densify_from, densify_every, densify_until, reset_every = 200, 30, 25, 160, 90
    lambda_dssim, lambda_rep, grad_threshold, extent = 0.2, 0.01, 0.0008, 3.0
    order = np.random.default_rng(5)
    losses, sizes, resized = [], [P0], []
    for it in range(1, iterations + 1):
        v = int(order.integers(len(cams)))                      # train.py:172-178: one random camera
        rs, gt = cams[v], targets[v]
        if sink is not None:
            sink.begin_step()
            out, viewspace = _render_params(params, rs, sink)
        else:
            out, viewspace = _render_reference(params, rs)
        image, radii, rep = out[0].clamp(0, 1), out[1], out[5]   # gaussian_renderer/__init__.py:179
        Ll1 = torch.abs(image - gt).mean()                       # train.py:222
        loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
        loss = loss + lambda_rep * torch.abs(rep - torch.full_like(rep, 10.0)).mean()     # train.py:248-251
        loss.backward()
        losses.append(float(loss.detach()))
        with torch.no_grad():
            assert viewspace.grad is not None and viewspace.grad.shape == params["xyz"].shape
            if it < densify_until:
                add_densification_stats(radii, viewspace.grad, stats["xyz_gradient_accum"], stats["denom"],
                                        stats["max_radii2D"])                               # train.py:298-312
                if it > densify_from and it % densify_every == 0:
                    size_threshold = 20 if it > reset_every else None                      # train.py:320
                    before = params["xyz"].shape[0]
                    new, stats, _ = densify_and_prune(opt, stats, grad_threshold, 0.005, extent, size_threshold, radii)
                    for k in NAMES:
                        params[k] = new[k]
                    after = params["xyz"].shape[0]
                    sizes.append(after)
                    resized.append(after - before)
                    for k in NAMES:    # the Adam moments followed their rows
                        st = opt.state[params[k]]
                        assert st["exp_avg"].shape == params[k].shape and st["exp_avg_sq"].shape == params[k].shape
                    if sink is not None:
                        sink = ParamGradSink(*[params[k] for k in NAMES])   # the old one belongs to the replaced tensors
                if it % reset_every == 0:
                    _reset_opacity(opt, params)
                    if sink is not None:
                        sink = ParamGradSink(*[params[k] for k in NAMES])
            # (a parameter replaced above has no gradient from this iteration: Adam skips it in the reference
            # variant, as in the reference; a fresh sink hands it zeros)
            opt.step()                                                                       # train.py:342
            opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    sort_errors.poll(block=True)                                     # raises if any forward's sorts reported an error
    print(f"{api}: sizes {sizes}, loss {np.mean(losses[:10]):.4f} -> {np.mean(losses[-10:]):.4f}")
    assert all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < 0.75 * np.mean(losses[:10]), (np.mean(losses[:10]), np.mean(losses[-10:]))
    assert max(resized) > 0 and min(resized) < 0, sizes              # the model grew and shrank
    assert len(set(sizes)) >= 3
    for k in NAMES:
        assert torch.isfinite(params[k]).all(), k
        assert params[k].shape[0] == sizes[-1]
