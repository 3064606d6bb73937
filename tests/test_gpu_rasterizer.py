"""GPU parity: the HIP path (through the C ABI) against the CPU checker on
the same seeded inputs.  Bars: integer outputs (radii, tile counts, sort keys,
point list, tile ranges, n_contrib, gauss_count, last_contr_gauss) bit-exact;
colour / inverse depth / reps / transmittance within 1e-5 absolute; gauss_sum
and every gradient within 1e-4 relative (summation order of the atomics)."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import worst, read_internals, rel_err, run_forward_raw, scene_to_torch

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-5
REL_TOL = 1e-4


def _scenes():
    return [
        ("tiny", lambda: synth.tiny_scene(seed=7), False, (0.3, 0.6, 0.1)),
        ("tiny_aa", lambda: synth.tiny_scene(seed=8), True, (1.0, 1.0, 1.0)),
        ("tiny_ragged", lambda: synth.tiny_scene(P=300, width=75, height=45, seed=9, scale=0.1), True, (0, 0, 0)),
        ("ball20k", lambda: synth.ball_scene(20000, width=400, height=400), False, (0, 0, 0)),
        ("street50k", lambda: synth.street_scene(50000, 640, 360, seed=3), True, (0.1, 0.2, 0.3)),
    ]


def _check_forward(oracle, sc, aa, bg, **variant):
    rs, ten = scene_to_torch(sc, bg=bg, antialiasing=aa)
    P, W, H = ten["means3D"].shape[0], rs.image_width, rs.image_height
    kw_o, kw_g = {}, {}
    if variant.get("precomp"):
        f0 = oracle.forward_from_scene(sc, bg=bg, antialiasing=aa)
        cols = np.clip(np.random.default_rng(5).random((P, 3)), 0, 1).astype(np.float32)
        kw_o = dict(colors_precomp=cols, use_cov_precomp=f0.cov3D.copy())
        # Gaussians the first pass culled never had their covariance computed: give them one
        bad = f0.radii == 0
        kw_o["use_cov_precomp"][bad] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32)
        kw_g = dict(colors_precomp=torch.tensor(cols, device="cuda:0"),
                    cov3D_precomp=torch.tensor(kw_o["use_cov_precomp"], device="cuda:0"))
    with_reps = variant.get("with_reps", True)
    f = oracle.forward_from_scene(sc, bg=bg, antialiasing=aa, with_reps=with_reps, **kw_o)
    out = run_forward_raw(rs, ten, with_reps=with_reps, **kw_g)
    torch.cuda.synchronize()
    it = read_internals(out, P, W, H, packed_pairs=variant.get("packed_pairs"))
    vis = f.radii > 0
    # bit-exact integers
    assert out["num_rendered"] == f.R
    assert np.array_equal(out["radii"].cpu().numpy(), f.radii)
    assert np.array_equal(it["bin"][:, 3], f.tiles_touched)
    assert np.array_equal(it["bin"][vis, 2], f.depths[vis].view(np.uint32))
    assert np.array_equal(it["keys"], f.keys)
    assert np.array_equal(it["point_list"], f.point_list)
    assert np.array_equal(it["ranges"], f.ranges)
    assert np.array_equal(it["n_contrib"], f.n_contrib)
    assert np.array_equal(out["gauss_count"].cpu().numpy(), f.gauss_count)
    assert np.array_equal(out["last_contr_gauss"].cpu().numpy(), f.last_contr)
    # per-Gaussian floats come from the same arithmetic: exact
    rec = it["rec"]
    assert np.array_equal(rec[vis, 0:2], f.means2D[vis])
    assert np.array_equal(rec[vis][:, [2, 3, 4, 5]], f.conic_opacity[vis])
    if not variant.get("precomp"):
        assert np.array_equal(rec[vis, 6:9], f.rgb[vis])
        assert np.array_equal(it["cov3D"][vis], f.cov3D[vis])
        cl = it["clamped"][vis]
        assert np.array_equal(np.stack([cl & 1, (cl >> 1) & 1, (cl >> 2) & 1], 1), f.clamped[vis])
    assert np.array_equal(out["out_depths"].cpu().numpy()[vis], f.out_depths[vis])
    # per-pixel floats: same sequential arithmetic per pixel
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    assert np.abs(out["invdepth"].cpu().numpy() - f.invdepth).max() <= ABS_TOL
    assert np.abs(out["accum_alpha"].cpu().numpy() - f.accum_alpha).max() <= ABS_TOL
    assert np.abs(it["final_T"] - f.final_T).max() <= ABS_TOL
    if with_reps:
        # reps are N(0, 1), not colours in [0, 1]: 1e-5 of the largest feature value a pixel can blend
        rep_scale = max(1.0, float(np.abs(ten["reps"].cpu().numpy()[vis]).max())) if vis.any() else 1.0
        assert np.abs(out["out_reps"].cpu().numpy() - f.out_reps).max() <= ABS_TOL * rep_scale
    else:
        assert out["out_reps"].shape[0] == 0
    assert rel_err(out["gauss_sum"].cpu().numpy(), f.gauss_sum) <= REL_TOL
    return f, out, rs, ten, kw_g


@pytest.mark.parametrize("name,make,aa,bg", _scenes(), ids=[s[0] for s in _scenes()])
def test_forward_parity_with_instances_sorted_as_pairs(oracle, name, make, aa, bg):
    """Between the key emission and the pass that orders each segment of tiles an instance normally travels
    as one word (low tile digit above the Gaussian index); models with more Gaussians than fit below the
    digit take (tile id, index) pairs.  That path forced on every scene: the same lists and ranges, and the
    sorted tile ids it stores equal the tile part of the rebuilt keys."""
    from _gsplat_lib import lib
    try:
        lib().gsr_test_packed_instances(0)
        _check_forward(oracle, make(), aa, bg, packed_pairs=False)
    finally:
        lib().gsr_test_packed_instances(1)


@pytest.mark.parametrize("name,make,aa,bg", _scenes(), ids=[s[0] for s in _scenes()])
@pytest.mark.parametrize("reach", [-1, 0, 1], ids=["auto", "plain_kernel", "reach_masks"])
def test_forward_parity(oracle, name, make, aa, bg, reach):
    """reach: both forward blend kernels on every scene (the one that leaves out, per quadrant, the list entries
    that cannot reach it is otherwise only chosen for views of small splats); integer outputs, hit counts
    included, are bit-exact with either."""
    from _gsplat_lib import lib
    try:
        lib().gsr_test_forward_reach_masks(reach)
        _check_forward(oracle, make(), aa, bg)
    finally:
        lib().gsr_test_forward_reach_masks(-1)


def test_forward_parity_precomputed_and_no_reps(oracle):
    sc = synth.tiny_scene(P=200, width=64, height=48, seed=11)
    _check_forward(oracle, sc, True, (0.2, 0.2, 0.2), precomp=True)
    _check_forward(oracle, sc, False, (0.2, 0.2, 0.2), with_reps=False)


def _grads_gpu(rs, ten, gc, gd, gr, kw_g, out):
    from diff_gaussian_rasterization import _C
    e = torch.Tensor([])
    cp, cv = kw_g.get("colors_precomp"), kw_g.get("cov3D_precomp")
    res = _C.rasterize_gaussians_backward(
        rs.bg, ten["means3D"], out["radii"], e if cp is None else cp, ten["opacities"],
        ten["scales"] if cv is None else e, ten["rotations"] if cv is None else e, rs.scale_modifier,
        e if cv is None else cv, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, gc,
        e if gd is None else gd, gr, ten["shs"] if cp is None else e, rs.sh_degree, rs.campos, out["geom"],
        out["num_rendered"], out["binning"], out["img"], rs.antialiasing, rs.debug, ten["reps"])
    names = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot",
             "dL_dreps")
    return dict(zip(names, [r.cpu().numpy() for r in res]))


@pytest.mark.parametrize("name,make,aa,bg", _scenes(), ids=[s[0] for s in _scenes()])
@pytest.mark.parametrize("with_depth", [True, False], ids=["invdepth", "noinvdepth"])
def test_backward_parity(oracle, name, make, aa, bg, with_depth):
    sc = make()
    f, out, rs, ten, kw_g = _check_forward(oracle, sc, aa, bg)
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(3)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32) if with_depth else None
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, gd, gr)
    t = lambda a: None if a is None else torch.tensor(a, device="cuda:0")
    got = _grads_gpu(rs, ten, t(gc), t(gd), t(gr), kw_g, out)
    torch.cuda.synchronize()
    for k in got:
        r = ref[k].reshape(got[k].shape)
        assert np.isfinite(got[k]).all(), k
        assert rel_err(got[k], r) <= REL_TOL, (k, rel_err(got[k], r))


@pytest.mark.parametrize("dim", [8, 16])
@pytest.mark.parametrize("name,make,aa,bg", [
    ("tiny_aa", lambda d: synth.tiny_scene(seed=8, dim=d), True, (1.0, 1.0, 1.0)),
    ("ragged", lambda d: synth.tiny_scene(P=300, width=75, height=45, seed=9, scale=0.1, dim=d), False, (0.3, 0.6, 0.1)),
    ("street50k", lambda d: synth.street_scene(50000, 640, 360, seed=3, dim=d), True, (0.1, 0.2, 0.3)),
    ("deep", lambda d: synth.deep_scene(dim=d), True, (0.0, 0.0, 0.0))], ids=["tiny_aa", "ragged", "street50k", "deep"])
def test_wide_feature_rows(oracle, dim, name, make, aa, bg):
    """8 and 16 feature ("reps") channels.  The reference compiles one width in (NUM_DIM, config.h:16; its CLI
    default --dim is 16, arguments/__init__.py:98); here the width is read off reps.shape[1].  Forward with both
    blend kernels, backward with and without the inverse-depth gradient, everything against the checker."""
    from _gsplat_lib import lib
    sc = make(dim)
    assert sc["reps"].shape[1] == dim
    for reach in (0, 1):
        try:
            lib().gsr_test_forward_reach_masks(reach)
            f, out, rs, ten, kw_g = _check_forward(oracle, sc, aa, bg)
        finally:
            lib().gsr_test_forward_reach_masks(-1)
    assert out["out_reps"].shape[0] == dim
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(3)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((dim, H, W)).astype(np.float32)
    t = lambda a: None if a is None else torch.tensor(a, device="cuda:0")
    for with_depth in (True, False):
        ref = f.backward(gc, gd if with_depth else None, gr)
        got = _grads_gpu(rs, ten, t(gc), t(gd) if with_depth else None, t(gr), kw_g, out)
        torch.cuda.synchronize()
        assert got["dL_dreps"].shape == (ten["means3D"].shape[0], dim)
        for k in got:
            r = ref[k].reshape(got[k].shape)
            assert np.isfinite(got[k]).all(), k
            assert rel_err(got[k], r) <= REL_TOL, (k, rel_err(got[k], r))


def test_backward_parity_precomputed(oracle):
    sc = synth.tiny_scene(P=200, width=64, height=48, seed=11)
    f, out, rs, ten, kw_g = _check_forward(oracle, sc, True, (0.2, 0.2, 0.2), precomp=True)
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(4)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, gd, gr)
    t = lambda a: torch.tensor(a, device="cuda:0")
    got = _grads_gpu(rs, ten, t(gc), t(gd), t(gr), kw_g, out)
    for k in ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dreps"):
        assert rel_err(got[k], ref[k].reshape(got[k].shape)) <= REL_TOL, (k, worst(got[k], ref[k]), got[k].shape)


def test_autograd_module_end_to_end(oracle):
    """The nn.Module / autograd.Function layer: 6 outputs, gradients on the leaves."""
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synth.tiny_scene(P=150, width=48, height=32, seed=13)
    rs, ten = scene_to_torch(sc, bg=(0.5, 0.5, 0.5), antialiasing=True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in ten.items()}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    color, radii, invd, gsum, gcnt, oreps = GaussianRasterizer(rs)(
        means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
        scales=leaves["scales"], rotations=leaves["rotations"], reps=leaves["reps"])
    assert color.shape == (3, 32, 48) and invd.shape == (1, 32, 48) and oreps.shape == (4, 32, 48)
    assert radii.dtype == torch.int32 and gcnt.dtype == torch.int32 and gsum.shape == radii.shape
    rng = np.random.default_rng(1)
    gc, gd, gr = (rng.standard_normal(s).astype(np.float32) for s in ((3, 32, 48), (1, 32, 48), (4, 32, 48)))
    loss = (color * torch.tensor(gc, device="cuda:0")).sum() + (invd * torch.tensor(gd, device="cuda:0")).sum() + \
           (oreps * torch.tensor(gr, device="cuda:0")).sum()
    loss.backward()
    f = oracle.forward_from_scene(sc, bg=(0.5, 0.5, 0.5), antialiasing=True)
    ref = f.backward(gc, gd, gr)
    pairs = (("means3D", "dL_dmean3D"), ("scales", "dL_dscale"), ("rotations", "dL_drot"), ("opacities", "dL_dopacity"),
             ("shs", "dL_dsh"), ("reps", "dL_dreps"))
    for a, b in pairs:
        assert rel_err(leaves[a].grad.cpu().numpy(), ref[b].reshape(leaves[a].shape)) <= REL_TOL, a
    assert rel_err(means2D.grad.cpu().numpy(), ref["dL_dmean2D"]) <= REL_TOL


def test_mark_visible_and_api_errors(oracle):
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synth.tiny_scene(P=500, seed=21)
    rs, ten = scene_to_torch(sc)
    vis = GaussianRasterizer(rs).markVisible(ten["means3D"])
    ref = oracle.mark_visible(sc["means3D"], sc["camera"]["viewmatrix"], sc["camera"]["projmatrix"])
    assert vis.dtype == torch.bool and np.array_equal(vis.cpu().numpy(), ref)
    r = GaussianRasterizer(rs)
    m2 = torch.zeros_like(ten["means3D"])
    with pytest.raises(Exception):
        r(means3D=ten["means3D"], means2D=m2, opacities=ten["opacities"], scales=ten["scales"],
          rotations=ten["rotations"])  # neither SHs nor colours
    with pytest.raises(Exception):
        r(means3D=ten["means3D"], means2D=m2, opacities=ten["opacities"], shs=ten["shs"], scales=ten["scales"])
    with pytest.raises(RuntimeError):
        r(means3D=ten["means3D"][:, :2], means2D=m2, opacities=ten["opacities"], shs=ten["shs"],
          scales=ten["scales"], rotations=ten["rotations"], reps=ten["reps"])
    with pytest.raises(RuntimeError):
        r(means3D=ten["means3D"], means2D=m2, opacities=ten["opacities"], shs=ten["shs"], scales=ten["scales"],
          rotations=ten["rotations"], reps=ten["reps"][:, :3])


def test_empty_inputs():
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synth.tiny_scene(P=10, seed=2)
    rs, ten = scene_to_torch(sc, bg=(0.25, 0.5, 0.75))
    z = {k: v[:0].contiguous() for k, v in ten.items()}
    color, radii, invd, gsum, gcnt, oreps = GaussianRasterizer(rs)(
        means3D=z["means3D"], means2D=torch.zeros_like(z["means3D"]), opacities=z["opacities"], shs=z["shs"],
        scales=z["scales"], rotations=z["rotations"], reps=None)
    assert radii.numel() == 0 and gsum.numel() == 0
    c = color.cpu().numpy()
    assert np.allclose(c[0], 0.25) and np.allclose(c[1], 0.5) and np.allclose(c[2], 0.75)
    assert float(invd.abs().max()) == 0.0


def test_extension_and_ctypes_host_layers_agree(oracle):
    """The compiled host extension and the ctypes host layer drive the same C ABI: identical forward
    outputs bit for bit, gradients equal up to the order of the float atomics."""
    from diff_gaussian_rasterization import _C
    assert _C.host_layer() == "torch extension"
    sc = synth.tiny_scene(P=300, width=75, height=45, seed=9, scale=0.1)
    rs, ten = scene_to_torch(sc, bg=(0.2, 0.1, 0.3), antialiasing=True)
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(8)
    t = lambda a: torch.tensor(a.astype(np.float32), device="cuda:0")
    gc, gd, gr = t(rng.standard_normal((3, H, W))), t(rng.standard_normal((1, H, W))), t(rng.standard_normal((4, H, W)))
    a = run_forward_raw(rs, ten)
    ga = _grads_gpu(rs, ten, gc, gd, gr, {}, a)
    ext = _C._hip_ext
    try:
        _C._hip_ext = None
        assert _C.host_layer() == "ctypes"
        b = run_forward_raw(rs, ten)
        gb = _grads_gpu(rs, ten, gc, gd, gr, {}, b)
    finally:
        _C._hip_ext = ext
    torch.cuda.synchronize()
    assert a["num_rendered"] == b["num_rendered"]
    for k in ("color", "radii", "invdepth", "accum_alpha", "gauss_count", "last_contr_gauss", "out_reps"):
        assert torch.equal(a[k], b[k]), k
    for k in ga:
        assert rel_err(gb[k], ga[k]) <= REL_TOL, k
