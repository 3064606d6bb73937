"""GPU parity of the raw-parameter mode (activations and the SH concatenation
folded into the kernels) and of gradient accumulation.

The CPU checker applies the same activation functions (shared header) and then
runs the reference algorithm on the activated values, so the integer outputs
stay bit-exact; raw-parameter gradients are the checker's gradients pushed
through its activation backward (1e-4 relative).  A second test compares the
autograd layer with the reference formulation (torch.sigmoid / exp / normalize
/ cat in front of the ordinary operator), where only the last bits of the
activations differ."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import read_internals, rel_err, scene_to_torch

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-5
REL_TOL = 1e-4


def _raw_scene(sc, seed=0):
    """Raw parameters whose activations reproduce the scene (up to rounding)."""
    rng = np.random.default_rng(seed)
    op = np.clip(sc["opacities"].reshape(-1), 1e-4, 1 - 1e-4)
    raw = dict(
        opacity=np.log(op / (1 - op)).astype(np.float32).reshape(-1, 1),
        scaling=np.log(sc["scales"]).astype(np.float32),
        rotation=(sc["rotations"] * rng.uniform(0.3, 3.0, (sc["rotations"].shape[0], 1))).astype(np.float32),
        features_dc=np.ascontiguousarray(sc["shs"][:, :1]),
        features_rest=np.ascontiguousarray(sc["shs"][:, 1:]),
    )
    return raw


def _oracle_forward(oracle, sc, raw, aa, bg):
    o, s, r = oracle.activate(raw["opacity"], raw["scaling"], raw["rotation"])
    sc2 = dict(sc)
    sc2["opacities"], sc2["scales"], sc2["rotations"] = o, s, r
    return oracle.forward_from_scene(sc2, bg=bg, antialiasing=aa)


def _gpu_forward(rs, sc, raw, split=True):
    from diff_gaussian_rasterization import _C
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    ten = dict(xyz=t(sc["means3D"]), dc=t(raw["features_dc"]), rest=t(raw["features_rest"]), opacity=t(raw["opacity"]),
               scaling=t(raw["scaling"]), rotation=t(raw["rotation"]), reps=t(sc["reps"]))
    e = torch.Tensor([])
    sh = ten["dc"] if split else torch.cat((ten["dc"], ten["rest"]), dim=1)
    out = _C.rasterize_gaussians(
        rs.bg, ten["xyz"], e, ten["opacity"], ten["scaling"], ten["rotation"], rs.scale_modifier, e, rs.viewmatrix,
        rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, False,
        rs.antialiasing, rs.debug, ten["reps"], sh_rest=ten["rest"] if split else None, raw_params=True)
    names = ("num_rendered", "color", "radii", "geom", "binning", "img", "invdepth", "accum_alpha", "gauss_sum",
             "gauss_count", "last_contr_gauss", "out_depths", "out_reps")
    return dict(zip(names, out)), ten, sh


def _gpu_backward(rs, ten, sh, out, gc, gd, gr, split, into=None, accumulate=False):
    from diff_gaussian_rasterization import _C
    e = torch.Tensor([])
    return _C.rasterize_gaussians_backward(
        rs.bg, ten["xyz"], out["radii"], e, ten["opacity"], ten["scaling"], ten["rotation"], rs.scale_modifier, e,
        rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, gc, gd, gr, sh, rs.sh_degree, rs.campos, out["geom"],
        out["num_rendered"], out["binning"], out["img"], rs.antialiasing, rs.debug, ten["reps"],
        sh_rest=ten["rest"] if split else None, raw_params=True, into=into, accumulate=accumulate)


SCENES = [
    ("tiny_aa", lambda: synth.tiny_scene(seed=8), True, (1.0, 1.0, 1.0)),
    ("ragged", lambda: synth.tiny_scene(P=300, width=75, height=45, seed=9, scale=0.1), True, (0, 0, 0)),
    ("street50k", lambda: synth.street_scene(50000, 640, 360, seed=3), True, (0.1, 0.2, 0.3)),
]


@pytest.mark.parametrize("name,make,aa,bg", SCENES, ids=[s[0] for s in SCENES])
@pytest.mark.parametrize("split", [True, False], ids=["split_sh", "cat_sh"])
def test_raw_params_forward_backward(oracle, name, make, aa, bg, split):
    sc = make()
    raw = _raw_scene(sc)
    rs, _ = scene_to_torch(sc, bg=bg, antialiasing=aa)
    P, W, H = sc["means3D"].shape[0], rs.image_width, rs.image_height
    f = _oracle_forward(oracle, sc, raw, aa, bg)
    out, ten, sh = _gpu_forward(rs, sc, raw, split)
    torch.cuda.synchronize()
    it = read_internals(out, P, W, H)
    assert out["num_rendered"] == f.R
    assert np.array_equal(out["radii"].cpu().numpy(), f.radii)
    assert np.array_equal(it["keys"], f.keys)
    assert np.array_equal(it["point_list"], f.point_list)
    assert np.array_equal(it["n_contrib"], f.n_contrib)
    assert np.array_equal(out["gauss_count"].cpu().numpy(), f.gauss_count)
    vis = f.radii > 0
    assert np.array_equal(it["rec"][vis, 6:9], f.rgb[vis])
    assert np.array_equal(it["rec"][vis][:, [2, 3, 4, 5]], f.conic_opacity[vis])
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    assert np.abs(out["out_reps"].cpu().numpy() - f.out_reps).max() <= ABS_TOL * 10

    rng = np.random.default_rng(3)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, gd, gr)
    r_op, r_sc, r_rot = oracle.activate_bwd(raw["opacity"], raw["scaling"], raw["rotation"], ref["dL_dopacity"],
                                            ref["dL_dscale"], ref["dL_drot"])
    t = lambda a: torch.tensor(a, device="cuda:0")
    res = _gpu_backward(rs, ten, sh, out, t(gc), t(gd), t(gr), split)
    torch.cuda.synchronize()
    g = [r.cpu().numpy() for r in res]
    dsh = ref["dL_dsh"].reshape(P, -1, 3)
    assert rel_err(g[0], ref["dL_dmean2D"]) <= REL_TOL
    assert rel_err(g[2].reshape(-1), r_op) <= REL_TOL
    assert rel_err(g[3], ref["dL_dmean3D"]) <= REL_TOL
    assert rel_err(g[6], r_sc) <= REL_TOL
    assert rel_err(g[7], r_rot) <= REL_TOL
    assert rel_err(g[8], ref["dL_dreps"]) <= REL_TOL
    if split:
        assert g[5].shape == (P, 1, 3) and g[9].shape == (P, dsh.shape[1] - 1, 3)
        assert rel_err(g[5], dsh[:, :1]) <= REL_TOL
        assert rel_err(g[9], dsh[:, 1:]) <= REL_TOL
    else:
        assert len(g) == 9 and rel_err(g[5], dsh) <= REL_TOL
    for a in g:
        assert np.isfinite(a).all()

    # accumulation: a second backward with other pixel gradients added into the first one's results
    gc2 = rng.standard_normal((3, H, W)).astype(np.float32)
    gr2 = rng.standard_normal((4, H, W)).astype(np.float32)
    res2 = _gpu_backward(rs, ten, sh, out, t(gc2), None, t(gr2), split)
    into = dict(means3D=res[3].clone(), opacity=res[2].clone(), sh=res[5].clone(), scales=res[6].clone(),
                rotations=res[7].clone(), reps=res[8].clone())
    if split:
        into["sh_rest"] = res[9].clone()
    res3 = _gpu_backward(rs, ten, sh, out, t(gc2), None, t(gr2), split, into=into, accumulate=True)
    torch.cuda.synchronize()
    pairs = [(3, "means3D"), (2, "opacity"), (5, "sh"), (6, "scales"), (7, "rotations"), (8, "reps")]
    if split:
        pairs.append((9, "sh_rest"))
    for i, key in pairs:
        assert res3[i].data_ptr() == into[key].data_ptr()          # written in place
        want = (res[i].double() + res2[i].double()).cpu().numpy()
        assert rel_err(into[key].cpu().numpy(), want) <= 1e-6, key
    # the means2D gradient is per view, never accumulated (equal up to the order of the atomics)
    assert rel_err(res3[0].cpu().numpy(), res2[0].cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("degree,M", [(1, 4), (0, 1), (2, 9)])
@pytest.mark.parametrize("split", [True, False], ids=["split_sh", "cat_sh"])
def test_other_sh_sizes(oracle, degree, M, split):
    if split and M == 1:
        pytest.skip("a degree-0 model has no second SH tensor")
    sc = synth.tiny_scene(P=257, width=64, height=48, seed=31)
    sc["shs"] = np.ascontiguousarray(sc["shs"][:, :M])
    sc["sh_degree"] = degree
    raw = _raw_scene(sc)
    rs, _ = scene_to_torch(sc, bg=(0.2, 0.3, 0.4), antialiasing=False)
    P, W, H = 257, 64, 48
    f = _oracle_forward(oracle, sc, raw, False, (0.2, 0.3, 0.4))
    out, ten, sh = _gpu_forward(rs, sc, raw, split)
    assert out["num_rendered"] == f.R
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    rng = np.random.default_rng(5)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, None, gr)
    t = lambda a: torch.tensor(a, device="cuda:0")
    res = _gpu_backward(rs, ten, sh, out, t(gc), None, t(gr), split)
    dsh = ref["dL_dsh"].reshape(P, M, 3)
    got = torch.cat((res[5], res[9]), dim=1) if split else res[5]
    assert rel_err(got.cpu().numpy(), dsh) <= REL_TOL
    assert rel_err(res[3].cpu().numpy(), ref["dL_dmean3D"]) <= REL_TOL


@pytest.mark.parametrize("use_sink", [False, True], ids=["autograd", "sink"])
def test_forward_params_matches_reference_formulation(use_sink):
    """forward_params == activations + cat in torch, then the ordinary operator."""
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    sc = synth.tiny_scene(P=400, width=80, height=56, seed=17)
    raw = _raw_scene(sc, seed=2)
    rs, _ = scene_to_torch(sc, bg=(0.1, 0.1, 0.1), antialiasing=True)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(1)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(2)]
    r = GaussianRasterizer(rs)
    sink = ParamGradSink(*[B[n] for n in names]) if use_sink else None

    def loss(o, w):
        return (o[0] * w[0]).sum() + (o[2] * w[1]).sum() + (o[5] * w[2]).sum()

    m2a = torch.zeros_like(A["xyz"], requires_grad=True)
    m2b = torch.zeros_like(B["xyz"], requires_grad=True)
    for view in range(2):   # two backward passes: gradients accumulate
        oa = r(means3D=A["xyz"], means2D=m2a, opacities=torch.sigmoid(A["opacity"]),
               shs=torch.cat((A["features_dc"], A["features_rest"]), dim=1), scales=torch.exp(A["scaling"]),
               rotations=torch.nn.functional.normalize(A["rotation"]), reps=A["reps"])
        loss(oa, ws[view]).backward()
        if sink is not None and view == 0:
            sink.begin_step()
        ob = r.forward_params(B["xyz"], m2b, B["features_dc"], B["features_rest"], B["opacity"], B["scaling"],
                              B["rotation"], B["reps"], grad_sink=sink)
        loss(ob, ws[view]).backward()
        assert torch.equal(oa[1], ob[1])                       # radii
        assert float((oa[0] - ob[0]).detach().abs().max()) <= 1e-5
    for n in names:
        assert B[n].grad is not None, n
        assert rel_err(B[n].grad.cpu().numpy(), A[n].grad.cpu().numpy()) <= REL_TOL, n
    assert rel_err(m2b.grad.cpu().numpy(), m2a.grad.cpu().numpy()) <= REL_TOL
