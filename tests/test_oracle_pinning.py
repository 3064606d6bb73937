"""Pins the CPU checker (oracle/) before anything is compared against it:
  * against outputs of the reference's own Python that imports on a CPU
    (fixtures written by tests/golden/make_fixtures.py: eval_sh, camera
    matrices, conv2d SSIM value + gradient);
  * against an independent float64 autograd restatement (tests/dense_ref.py);
  * against brute force / numpy for knn, sort and exp;
  * against its own committed outputs (tests/golden/raster_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

import synth
from gpu_util import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_matches_reference_eval_sh(oracle):
    fx = np.load(os.path.join(GOLD, "sh_eval.npz"))
    for deg in range(4):
        rgb, clamped = oracle.sh_eval(deg, fx["pos"], fx["campos"], fx["sh"])
        ref = fx[f"rgb_deg{deg}"]
        assert np.abs(rgb - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
        assert np.array_equal(clamped.astype(bool) & (ref > 1e-5), np.zeros_like(ref, bool))
        assert (clamped.sum() > 0) and np.all(rgb[clamped.astype(bool)] == 0.0)


def test_camera_matrices_match_reference_graphics_utils():
    fx = np.load(os.path.join(GOLD, "camera_matrices.npz"))
    for i in range(4):
        R, t = fx[f"R{i}"], fx[f"t{i}"]
        fovx, fovy, W, H = fx[f"fov{i}"]
        view = synth.world2view(R, t, (0.1, -0.2, 0.3), 1.5).transpose()
        proj = synth.projection_matrix(0.01, 100.0, fovx, fovy).transpose()
        assert np.allclose(view, fx[f"view{i}"], atol=1e-6)
        assert np.allclose(proj, fx[f"proj{i}"], atol=1e-6)
        assert np.allclose((view @ proj).astype(np.float32), fx[f"full{i}"], atol=1e-5)
        assert np.allclose(np.linalg.inv(view)[3, :3], fx[f"center{i}"], atol=1e-5)
        assert abs(synth.focal2fov(W / (2 * np.tan(fovx / 2)), H) - fovy) < 1e-12


def test_ssim_matches_reference_conv2d_ssim(oracle):
    """Value and gradient, the check submodules/fused-ssim/tests/test.py:82-91 makes."""
    fx = np.load(os.path.join(GOLD, "ssim_ref.npz"))
    for i in range(3):
        a, b = fx[f"img1_{i}"], fx[f"img2_{i}"]
        m, d1, d2, d3 = oracle.ssim_fwd(a, b)
        assert abs(float(m.mean()) - float(fx[f"ssim_{i}"])) <= 2e-6
        dl = np.full(a.shape, 1.0 / a.size, np.float32)
        g = oracle.ssim_bwd(a, b, dl, d1, d2, d3)
        ref = fx[f"grad_{i}"]
        assert np.abs(g - ref).max() <= 1e-4 * np.abs(ref).max()
        m2, e1, e2, e3 = oracle.ssim_fwd(a, b, train=False)
        assert e1 is None and np.array_equal(m, m2)


def test_expf_is_faithful(oracle):
    x = np.concatenate([np.linspace(-87, 0, 200001), -np.logspace(-8, 1.9, 5000)]).astype(np.float32)
    y = oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    rel = np.abs(y.astype(np.float64) - ref) / ref
    assert rel.max() < 1.5e-7          # ~1 ulp
    assert y[x == 0][0] == 1.0
    assert np.all(np.diff(y[:200001]) >= 0)  # monotone on the grid
    assert oracle.expf(np.array([-200.0], np.float32))[0] == oracle.expf(np.array([-87.0], np.float32))[0]


def test_sort_pairs_is_a_stable_lsd_sort(oracle):
    rng = np.random.default_rng(0)
    for n, end_bit in ((0, 44), (1, 44), (1000, 13), (50_000, 44), (50_000, 64)):
        keys = rng.integers(0, 2 ** 62, size=n, dtype=np.int64).astype(np.uint64) % np.uint64(977 if end_bit < 64 else 2 ** 62)
        keys |= np.uint64(0xABC) << np.uint64(52) if end_bit < 52 else np.uint64(0)
        vals = np.arange(n, dtype=np.uint32)
        ko, vo = oracle.sort_pairs(keys, vals, end_bit)
        mask = np.uint64((1 << end_bit) - 1) if end_bit < 64 else np.uint64(2 ** 64 - 1)
        order = np.argsort(keys & mask, kind="stable")
        assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])


def test_knn_matches_brute_force(oracle):
    rng = np.random.default_rng(5)
    pts = (rng.standard_normal((2000, 3)) * np.array([1, 2, 0.5]) + np.array([3, 0, -1])).astype(np.float32)
    got = oracle.dist2(pts)
    d = pts[:, None, :] - pts[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    np.fill_diagonal(d2, np.inf)
    part = np.sort(d2, axis=1)[:, :3].astype(np.float32)
    bf = ((part[:, 0] + part[:, 1]) + part[:, 2]) / np.float32(3.0)
    assert np.allclose(got, bf, rtol=1e-6, atol=0)
    few = oracle.dist2(pts[:3])
    assert np.all(few > 1e37)          # fewer than 3 neighbours: FLT_MAX terms stay in the mean, as in the reference


def _leaves(sc, P, extra=()):
    names = ("means3D", "scales", "rotations", "opacities", "shs", "reps") + tuple(extra)
    inp = {k: torch.tensor(np.asarray(sc[k], np.float64), requires_grad=True) for k in names}
    inp["means2D"] = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    return inp


@pytest.mark.parametrize("aa", [False, True])
@pytest.mark.parametrize("seed", [7, 8])
def test_forward_and_backward_match_dense_autograd(oracle, aa, seed):
    from dense_ref import dense_render
    sc = synth.tiny_scene(seed=seed)
    bg = (0.3, 0.6, 0.1)
    f = oracle.forward_from_scene(sc, bg=bg, antialiasing=aa)
    inp = _leaves(sc, f.P)
    out = dense_render(inp, sc["camera"], bg, sc["sh_degree"], antialiasing=aa, ref_quirks=True)
    assert np.array_equal(out["radii"].numpy(), f.radii)
    assert np.array_equal(out["tiles_touched"].numpy(), f.tiles_touched)
    assert np.array_equal(out["n_contrib"].numpy().reshape(-1), f.n_contrib)
    assert np.array_equal(out["gauss_count"].numpy(), f.gauss_count)
    assert np.abs(out["color"].detach().numpy() - f.out_color).max() < 1e-5
    assert np.abs(out["invdepth"].detach().numpy() - f.invdepth).max() < 1e-5
    assert np.abs(out["reps"].detach().numpy() - f.out_reps).max() < 1e-5
    assert np.abs(out["gauss_sum"].numpy() - f.gauss_sum).max() <= 1e-4 * np.abs(f.gauss_sum).max()
    rng = np.random.default_rng(3)
    gc, gd, gr = rng.standard_normal((3, f.H, f.W)), rng.standard_normal((1, f.H, f.W)), rng.standard_normal((4, f.H, f.W))
    loss = (out["color"] * torch.tensor(gc)).sum() + (out["invdepth"] * torch.tensor(gd)).sum() + \
           (out["reps"] * torch.tensor(gr)).sum()
    loss.backward()
    g = f.backward(gc.astype(np.float32), gd.astype(np.float32), gr.astype(np.float32))
    for a, b in (("means3D", "dL_dmean3D"), ("means2D", "dL_dmean2D"), ("scales", "dL_dscale"),
                 ("rotations", "dL_drot"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("reps", "dL_dreps")):
        ref = inp[a].grad.numpy()
        assert rel_err(g[b].reshape(ref.shape), ref) <= 1e-4, (a, rel_err(g[b].reshape(ref.shape), ref))


def test_deep_lists_with_high_alpha_entries_match_dense_autograd(oracle):
    """Dozens of blended entries per pixel, among them splats with alpha in 0.9 .. 0.99 (the list depth and
    the alpha range in which recovering T by division in the backward, backward.cu:615, is least forgiving)."""
    from dense_ref import dense_render
    sc = synth.deep_scene()
    bg = (0.3, 0.6, 0.1)
    f = oracle.forward_from_scene(sc, bg=bg, antialiasing=True)
    nb = _blended_per_pixel(f)
    assert np.median(nb) >= 24 and nb.max() >= 60
    inp = _leaves(sc, f.P)
    out = dense_render(inp, sc["camera"], bg, sc["sh_degree"], antialiasing=True, ref_quirks=True)
    assert np.array_equal(out["radii"].numpy(), f.radii)
    assert np.array_equal(out["n_contrib"].numpy().reshape(-1), f.n_contrib)
    assert np.array_equal(out["gauss_count"].numpy(), f.gauss_count)
    assert np.abs(out["color"].detach().numpy() - f.out_color).max() < 1e-5
    rng = np.random.default_rng(3)
    gc, gd, gr = rng.standard_normal((3, f.H, f.W)), rng.standard_normal((1, f.H, f.W)), rng.standard_normal((4, f.H, f.W))
    ((out["color"] * torch.tensor(gc)).sum() + (out["invdepth"] * torch.tensor(gd)).sum() +
     (out["reps"] * torch.tensor(gr)).sum()).backward()
    g = f.backward(gc.astype(np.float32), gd.astype(np.float32), gr.astype(np.float32))
    for a, b in (("means3D", "dL_dmean3D"), ("means2D", "dL_dmean2D"), ("scales", "dL_dscale"),
                 ("rotations", "dL_drot"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("reps", "dL_dreps")):
        ref = inp[a].grad.numpy()
        assert rel_err(g[b].reshape(ref.shape), ref) <= 1e-4, (a, rel_err(g[b].reshape(ref.shape), ref))


def _blended_per_pixel(f):
    """How many list entries each pixel walks up to its last contributor (n_contrib is a 1-based position)."""
    return f.n_contrib.astype(np.int64)


def test_backward_is_the_true_gradient_where_the_reference_is_exact(oracle):
    """Without antialiasing, and for Gaussians inside the 1.3 x frustum, the
    reference backward is the exact derivative: compare with plain autograd."""
    from dense_ref import dense_render
    sc = synth.tiny_scene(seed=7)
    bg = (0.0, 0.0, 0.0)
    f = oracle.forward_from_scene(sc, bg=bg)
    inp = _leaves(sc, f.P)
    out = dense_render(inp, sc["camera"], bg, sc["sh_degree"], ref_quirks=False)
    rng = np.random.default_rng(4)
    gc = rng.standard_normal((3, f.H, f.W))
    (out["color"] * torch.tensor(gc)).sum().backward()
    g = f.backward(gc.astype(np.float32), None, None)
    cam = sc["camera"]
    pv = np.concatenate([sc["means3D"], np.ones((f.P, 1), np.float32)], 1) @ cam["viewmatrix"]
    ok = (np.abs(pv[:, 0] / pv[:, 2]) < 1.3 * cam["tanfovx"]) & (np.abs(pv[:, 1] / pv[:, 2]) < 1.3 * cam["tanfovy"])
    ref = inp["means3D"].grad.numpy()
    assert ok.sum() > 10 and (~ok & (f.radii > 0)).sum() > 0
    assert np.abs(ref[ok] - g["dL_dmean3D"][ok]).max() <= 1e-4 * np.abs(ref).max()
    for a, b in (("scales", "dL_dscale"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh")):
        r = inp[a].grad.numpy()
        assert rel_err(g[b].reshape(r.shape), r) <= 1e-4, (a, rel_err(g[b].reshape(r.shape), r))


def test_precomputed_inputs_and_no_invdepth(oracle):
    from dense_ref import dense_render
    sc = synth.tiny_scene(seed=12)
    f0 = oracle.forward_from_scene(sc)
    P = f0.P
    cov = f0.cov3D.copy()
    cov[f0.radii == 0] = np.array([1e-3, 0, 0, 1e-3, 0, 1e-3], np.float32)
    cols = np.random.default_rng(2).random((P, 3)).astype(np.float32)
    f = oracle.forward_from_scene(sc, colors_precomp=cols, use_cov_precomp=cov, want_invdepth=False)
    assert f.invdepth is None
    inp = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True)
           for k, v in (("means3D", sc["means3D"]), ("opacities", sc["opacities"]), ("reps", sc["reps"]),
                        ("colors_precomp", cols), ("cov3D_precomp", cov))}
    inp["means2D"] = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    out = dense_render(inp, sc["camera"], (0, 0, 0), 0, ref_quirks=True)
    assert np.abs(out["color"].detach().numpy() - f.out_color).max() < 1e-5
    gc = np.random.default_rng(3).standard_normal((3, f.H, f.W))
    (out["color"] * torch.tensor(gc)).sum().backward()
    g = f.backward(gc.astype(np.float32), None, None)
    assert g["dL_dinvdepths"] is None
    for a, b in (("colors_precomp", "dL_dcolor"), ("cov3D_precomp", "dL_dcov3D"), ("means3D", "dL_dmean3D")):
        r = inp[a].grad.numpy()
        assert rel_err(g[b], r) <= 1e-4, (a, rel_err(g[b], r))
    assert np.all(g["dL_dsh"].size == 0) and np.all(g["dL_dscale"] == 0)


def test_fork_statistics_invariant(oracle):
    """The commented-out check of the fork author
    (diff_gaussian_rasterization/__init__.py:130-147): gauss_count counts every
    tested pair with power <= 0, so it is >= the number of pixels whose last
    contributor is the Gaussian, and gauss_sum <= gauss_count (T <= 1)."""
    sc = synth.tiny_scene(P=200, width=64, height=48, seed=5)
    f = oracle.forward_from_scene(sc)
    last = f.last_contr.reshape(-1)
    per_g = np.bincount(last[last >= 0], minlength=f.P)
    assert np.all(f.gauss_count >= per_g)
    assert np.all(f.gauss_sum <= f.gauss_count + 1e-3)
    assert np.all((f.gauss_count == 0) == (f.gauss_sum == 0))
    assert np.all(f.last_contr[f.n_contrib.reshape(f.last_contr.shape) == 0] == -1)


def test_edge_cases(oracle):
    sc = synth.tiny_scene(P=50, seed=3)
    z = dict(sc)
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "reps"):
        z[k] = sc[k][:0]
    f = oracle.forward_from_scene(z, bg=(0.5, 0.25, 0.125))      # P == 0 short-circuit
    assert f.R == 0 and np.all(f.out_color == 0)
    behind = dict(sc)
    behind["means3D"] = sc["means3D"].copy()
    behind["means3D"][:, 2] = -50.0                               # everything behind the near plane
    f = oracle.forward_from_scene(behind, bg=(0.5, 0.25, 0.125))
    assert f.R == 0 and np.all(f.radii == 0) and np.all(f.ranges == 0)
    assert np.allclose(f.out_color[0], 0.5) and np.allclose(f.out_color[2], 0.125)
    assert np.all(f.last_contr == -1) and np.all(f.final_T == 1.0)
    vis = oracle.mark_visible(sc["means3D"], sc["camera"]["viewmatrix"], sc["camera"]["projmatrix"])
    pv = np.concatenate([sc["means3D"], np.ones((50, 1), np.float32)], 1) @ sc["camera"]["viewmatrix"]
    assert np.array_equal(vis, pv[:, 2] > 0.2)


def test_oracle_reproduces_committed_golden(oracle):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_oracle_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fx = np.load(os.path.join(GOLD, "raster_tiny.npz"))
    for name, c in mk.CASES.items():
        sc, f, _, g = mk.run_case(c)
        for k in ("radii", "keys", "point_list", "ranges", "n_contrib", "gauss_count", "last_contr"):
            assert np.array_equal(getattr(f, k), fx[f"{name}_{k}"]), k
        assert np.abs(f.out_color - fx[f"{name}_color"]).max() <= 1e-6
        for k, v in g.items():
            assert np.abs(v - fx[f"{name}_{k}"]).max() <= 1e-5 * max(1.0, np.abs(v).max()), k


def test_training_loss_matches_reference_composition(oracle):
    """train.py:222-230: (1 - l) * l1_loss(c, gt) + l * (1 - ssim(c, gt)) on c = clamp(render, 0, 1), value and
    gradient from the reference's own l1_loss / ssim through autograd (fixture), against the same loss composed
    from the checker's ssim and numpy: what smoke() and bench.py's CPU baseline differentiate through."""
    fx = np.load(os.path.join(GOLD, "train_loss_ref.npz"))
    lam = float(fx["lambda_dssim"])
    for i in range(2):
        raw, gt = fx[f"raw_{i}"], fx[f"gt_{i}"]
        c = np.clip(raw, 0, 1)[None]
        m, d1, d2, d3 = oracle.ssim_fwd(c, gt[None])
        l1 = np.abs(c - gt[None]).mean(dtype=np.float64)
        loss = (1 - lam) * l1 + lam * (1 - m.mean(dtype=np.float64))
        assert abs(loss - float(fx[f"loss_{i}"])) <= 2e-6
        assert abs(l1 - float(fx[f"l1_{i}"])) <= 1e-6 and abs(m.mean(dtype=np.float64) - float(fx[f"ssim_{i}"])) <= 2e-6
        g = oracle.ssim_bwd(c, gt[None], np.full(c.shape, -lam / c.size, np.float32), d1, d2, d3)[0]
        g = g + (1 - lam) * np.sign(c[0] - gt) / c.size
        g = np.where((raw > 0) & (raw < 1), g, 0.0)          # the clamp passes no gradient outside (0, 1)
        assert rel_err(g, fx[f"grad_{i}"]) <= 1e-4
