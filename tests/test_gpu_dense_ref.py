"""The HIP kernels against tests/dense_ref.py directly (not through the CPU checker): a float64
PyTorch restatement from the textbook formulas, differentiated by autograd, that shares no code
with the kernels or with oracle/ (the checker includes the kernels' gs_math.h, so an error in that
header would be invisible to a kernel-vs-checker comparison).  Scenes include antialiasing and
Gaussians outside the 1.3 x frustum whose projection is clamped.  Integer outputs must be equal;
images 1e-5 absolute; gradients element-wise |got - ref| <= 1e-4 |ref| + 1e-6 max|ref|, where
dense_ref(ref_quirks=True) models the two places in which the reference's hand-written backward is
not the derivative of its forward (backward.cu:185-186, 234-249, 313-315)."""
import numpy as np
import pytest
import torch

import synth
from dense_ref import dense_render
from gpu_util import rel_err, scene_to_torch

pytestmark = pytest.mark.gpu

CASES = [("seed7", dict(seed=7), False), ("seed7_aa", dict(seed=7), True), ("seed8_aa", dict(seed=8), True),
         ("wide_aa", dict(P=96, width=56, height=40, seed=21, spread=1.7), True),
         ("wide", dict(P=96, width=56, height=40, seed=22, spread=1.7, scale=0.3), False),
         # long lists of nearly transparent splats with entries of alpha 0.9 .. 0.99 among them: the
         # backward recovers T by dozens of divisions in a row, some by 1 - alpha = 0.01 .. 0.1
         ("deep_aa", "deep", True)]


@pytest.mark.parametrize("name,kw,aa", CASES, ids=[c[0] for c in CASES])
def test_hip_forward_and_backward_match_dense_autograd(name, kw, aa):
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synth.deep_scene() if kw == "deep" else synth.tiny_scene(**kw)
    P = sc["means3D"].shape[0]
    bg = (0.3, 0.6, 0.1)
    cam = sc["camera"]
    H, W = cam["image_height"], cam["image_width"]
    # how many Gaussians have a clamped projection: the case the comparison is meant to include
    pv = np.concatenate([sc["means3D"], np.ones((P, 1), np.float32)], 1) @ cam["viewmatrix"]
    outside = (np.abs(pv[:, 0] / pv[:, 2]) > 1.3 * cam["tanfovx"]) | (np.abs(pv[:, 1] / pv[:, 2]) > 1.3 * cam["tanfovy"])

    rs, ten = scene_to_torch(sc, bg=bg, antialiasing=aa)
    leaves = {k: v.clone().requires_grad_(True) for k, v in ten.items()}
    m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, invd, gsum, gcnt, oreps = GaussianRasterizer(rs)(
        means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], shs=leaves["shs"],
        scales=leaves["scales"], rotations=leaves["rotations"], reps=leaves["reps"])

    names = ("means3D", "scales", "rotations", "opacities", "shs", "reps")
    inp = {k: torch.tensor(np.asarray(sc[k], np.float64), requires_grad=True) for k in names}
    inp["means2D"] = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    ref = dense_render(inp, cam, bg, sc["sh_degree"], antialiasing=aa, ref_quirks=True)

    if kw == "deep":
        assert np.median(ref["n_contrib"].numpy()) >= 24 and int(ref["n_contrib"].max()) >= 60
    else:
        assert ((radii.cpu().numpy() > 0) & outside).sum() > 0, "no rendered Gaussian has a clamped projection"
    assert np.array_equal(radii.cpu().numpy(), ref["radii"].numpy())
    assert np.array_equal(gcnt.cpu().numpy(), ref["gauss_count"].numpy())
    assert np.abs(color.detach().cpu().numpy() - ref["color"].detach().numpy()).max() <= 1e-5
    assert np.abs(invd.detach().cpu().numpy() - ref["invdepth"].detach().numpy()).max() <= 1e-5
    assert np.abs(oreps.detach().cpu().numpy() - ref["reps"].detach().numpy()).max() <= 1e-5
    assert rel_err(gsum.detach().cpu().numpy(), ref["gauss_sum"].detach().numpy()) <= 1e-4

    rng = np.random.default_rng(3)
    gc, gd, gr = rng.standard_normal((3, H, W)), rng.standard_normal((1, H, W)), rng.standard_normal((4, H, W))
    t32 = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda:0")
    ((color * t32(gc)).sum() + (invd * t32(gd)).sum() + (oreps * t32(gr)).sum()).backward()
    ((ref["color"] * torch.tensor(gc)).sum() + (ref["invdepth"] * torch.tensor(gd)).sum() +
     (ref["reps"] * torch.tensor(gr)).sum()).backward()
    torch.cuda.synchronize()
    got = dict(leaves, means2D=m2)
    for k in names + ("means2D",):
        g = got[k].grad.cpu().numpy()
        r = inp[k].grad.numpy().reshape(g.shape)
        assert np.isfinite(g).all(), k
        assert rel_err(g, r) <= 1e-4, (k, rel_err(g, r))
