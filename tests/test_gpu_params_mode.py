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
        # two launches of the backward blend order their float atomics differently: the recomputed second
        # term differs from res2 by that noise; an accumulation that overwrote or doubled would be off by O(1)
        assert rel_err(into[key].cpu().numpy(), want) <= 2e-5, key
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


@pytest.mark.parametrize("nviews,dim", [(1, 4), (3, 4), (10, 4), (3, 8), (10, 16)])
def test_deferred_batch_backward_matches_per_view_backward(oracle, nviews, dim):
    """ParamGradSink(defer=True): blend backward per view, one per-Gaussian backward for the batch.
    Compared with the per-view accumulate path and, for the sum, with the CPU checker.  dim: feature
    channels (4 as the reference compiles them, config.h:16; 8 and 16 take the wide instantiations)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, ParamGradSink
    sc = synth.tiny_scene(P=500, width=96, height=64, seed=23, dim=dim)
    raw = _raw_scene(sc, seed=4)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names])
    sink_b = ParamGradSink(*[B[n] for n in names], defer=True)
    assert sink_b.defer
    cams, settings = [], []
    for v in range(nviews):
        ang = 0.25 * (v - nviews / 2)
        cam = synth.make_camera(96, 64, 0.9, (0.1 + 2.5 * np.sin(ang), -0.05 + 0.1 * v, -2.5 * np.cos(ang)), (0.0, 0.0, 1.0))
        cams.append(cam)
        settings.append(GaussianRasterizationSettings(
            image_height=64, image_width=96, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t((0.1, 0.2, 0.3)),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
            campos=t(cam["campos"]), prefiltered=False, debug=True, antialiasing=True))
    rng = np.random.default_rng(9)
    ws = [[rng.standard_normal(s).astype(np.float32) for s in ((3, 64, 96), (1, 64, 96), (dim, 64, 96))]
          for _ in range(nviews)]

    def run(P_, sink, step_scale):
        sink.begin_step()
        m2s = []
        for v in range(nviews):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(settings[v]).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            w = ws[v]
            ((o[0] * t(w[0])).sum() + (o[2] * t(w[1])).sum() + (o[5] * t(w[2])).sum()).mul(step_scale).backward()
            m2s.append(m2)
        sink.flush() if sink.defer else None      # also fills the means2D.grad of the deferred views
        return [m.grad.clone() for m in m2s]

    for step_scale in (1.0, -0.5):      # two steps: the second must overwrite, not add to, the first
        ma = run(A, sink_a, step_scale)
        mb = run(B, sink_b, step_scale)
        for n in names:
            ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
            assert np.isfinite(gb).all(), n
            assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))
        for x, y in zip(ma, mb):
            assert rel_err(y.cpu().numpy(), x.cpu().numpy()) <= REL_TOL
    # the sum over views against the CPU checker (last step: scale -0.5)
    o_, s_, r_ = oracle.activate(raw["opacity"], raw["scaling"], raw["rotation"])
    tot = None
    for v in range(nviews):
        sc2 = dict(sc)
        sc2["opacities"], sc2["scales"], sc2["rotations"], sc2["camera"] = o_, s_, r_, cams[v]
        f = oracle.forward_from_scene(sc2, bg=(0.1, 0.2, 0.3), antialiasing=True)
        g = f.backward(*[(-0.5 * w).astype(np.float32) for w in ws[v]])
        g_op, g_sc, g_rot = oracle.activate_bwd(raw["opacity"], raw["scaling"], raw["rotation"], g["dL_dopacity"],
                                                g["dL_dscale"], g["dL_drot"])
        cur = dict(xyz=g["dL_dmean3D"], sh=g["dL_dsh"].reshape(500, 16, 3), opacity=g_op, scaling=g_sc, rotation=g_rot,
                   reps=g["dL_dreps"])
        tot = cur if tot is None else {k: tot[k] + cur[k] for k in cur}
    got_sh = np.concatenate([B["features_dc"].grad.cpu().numpy(), B["features_rest"].grad.cpu().numpy()], axis=1)
    assert rel_err(got_sh, tot["sh"]) <= REL_TOL
    for n in ("xyz", "opacity", "scaling", "rotation", "reps"):
        assert rel_err(B[n].grad.cpu().numpy().reshape(tot[n].shape), tot[n]) <= REL_TOL, n


def _arc_settings(nviews, W, H, aa=True, debug=True):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    out = []
    for v in range(nviews):
        ang = 0.25 * (v - nviews / 2)
        cam = synth.make_camera(W, H, 0.9, (0.1 + 2.5 * np.sin(ang), -0.05 + 0.1 * v, -2.5 * np.cos(ang)), (0.0, 0.0, 1.0))
        out.append(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t((0.1, 0.2, 0.3)),
            scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3,
            campos=t(cam["campos"]), prefiltered=False, debug=debug, antialiasing=aa))
    return out


@pytest.mark.parametrize("nviews,P", [(1, 300), (3, 500), (11, 777)])
@pytest.mark.parametrize("depth_keys", ["narrow", "forced_wide", "far_gaussian"])
def test_batched_forward_preprocess_is_bit_identical(nviews, P, depth_keys):
    """gsr_preprocess_batch + stage 1/2 with `preprocessed` set leaves every output and every
    scratch array that later stages read exactly as the one-view preprocess does.  The batch call orders the
    Gaussians by 27-bit keys in three passes when every depth allows it (narrow), by all 32 depth bits in four
    passes when told to (test hook) or when a Gaussian lies beyond 13107.2 (far_gaussian); the one-view path
    always takes the 32 bits: the sorted lists must agree in all three cases."""
    from _gsplat_lib import lib
    from diff_gaussian_rasterization import _C
    W, H = 96, 64
    sc = synth.tiny_scene(P=P, width=W, height=H, seed=31)
    if depth_keys == "far_gaussian":
        # a huge Gaussian far behind the scene, on camera 0's axis (the arc cameras look at (0, 0, 1))
        cam0 = np.asarray(_arc_settings(nviews, W, H)[0].campos.cpu(), dtype=np.float64)
        axis = np.array([0.0, 0.0, 1.0]) - cam0
        sc["means3D"][3] = cam0 + axis / np.linalg.norm(axis) * 20000.0
        sc["scales"][3] = 400.0
    raw = _raw_scene(sc, seed=5)
    settings = _arc_settings(nviews, W, H)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    ten = dict(xyz=t(sc["means3D"]), dc=t(raw["features_dc"]), rest=t(raw["features_rest"]), opacity=t(raw["opacity"]),
               scaling=t(raw["scaling"]), rotation=t(raw["rotation"]), reps=t(sc["reps"]))
    e = torch.Tensor([])
    names = ("num_rendered", "color", "radii", "geom", "binning", "img", "invdepth", "accum_alpha", "gauss_sum",
             "gauss_count", "last_contr_gauss", "out_depths", "out_reps")

    def fwd(rs, prepared):
        return dict(zip(names, _C.rasterize_gaussians(
            rs.bg, ten["xyz"], e, ten["opacity"], ten["scaling"], ten["rotation"], rs.scale_modifier, e, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, ten["dc"], rs.sh_degree, rs.campos,
            False, rs.antialiasing, rs.debug, ten["reps"], sh_rest=ten["rest"], raw_params=True, prepared=prepared)))

    lib().gsr_test_wide_depth_keys(1 if depth_keys == "forced_wide" else 0)
    try:
        prep = _C.preprocess_batch(ten["xyz"], ten["opacity"], ten["scaling"], ten["rotation"], 1.0, ten["dc"], ten["rest"],
                                   3, ten["reps"], True, settings, debug=True)
    finally:
        lib().gsr_test_wide_depth_keys(0)
    assert len(prep) == nviews
    bits = {pr["depth_key_bits"] for pr in prep if pr["depth_ordered"]}
    assert bits == ({27} if depth_keys == "narrow" else {32}), bits
    seen_any = 0
    for rs, pr in zip(settings, prep):
        a = fwd(rs, None)
        b = fwd(rs, pr)
        torch.cuda.synchronize()
        assert a["num_rendered"] == b["num_rendered"]
        assert pr["depth_ordered"] == (a["num_rendered"] > 0)   # the batch call also made the front-to-back order
        seen_any += a["num_rendered"]
        vis = a["radii"].cpu().numpy() > 0
        for k in ("color", "radii", "invdepth", "accum_alpha", "gauss_count", "last_contr_gauss", "out_reps"):
            assert torch.equal(a[k], b[k]), k
        assert rel_err(b["gauss_sum"].cpu().numpy(), a["gauss_sum"].cpu().numpy()) <= REL_TOL   # float atomics
        assert np.array_equal(a["out_depths"].cpu().numpy()[vis], b["out_depths"].cpu().numpy()[vis])
        ia, ib = read_internals(a, P, W, H), read_internals(b, P, W, H)
        assert np.array_equal(ia["rec"][vis][:, :15], ib["rec"][vis][:, :15])
        assert np.array_equal(ia["bin"], ib["bin"])
        assert np.array_equal(ia["clamped"][vis], ib["clamped"][vis])
        for k in ("keys", "point_list", "n_contrib", "ranges", "final_T"):
            assert np.array_equal(ia[k], ib[k]), k
    assert seen_any > 0
    # gsr_forward_stage1 behind the batch call (allowed: the sums are in place, it only fetches the totals
    # again): the same counts, and the view renders the same image from the same order
    # (a prepared view serves one forward: the hit counts accumulate in its geometry buffer)
    prep2 = _C.preprocess_batch(ten["xyz"], ten["opacity"], ten["scaling"], ten["rotation"], 1.0, ten["dc"], ten["rest"], 3,
                                ten["reps"], True, settings, debug=True)
    rs, pr = settings[0], dict(prep2[0])
    ref = fwd(rs, None)
    R, nvis = pr.pop("num_rendered"), pr.pop("num_visible")
    c = fwd(rs, pr)                      # without the counts the host layer asks stage 1 for them
    torch.cuda.synchronize()
    assert c["num_rendered"] == R == ref["num_rendered"]
    assert torch.equal(c["color"], ref["color"]) and torch.equal(c["gauss_count"], ref["gauss_count"])
    assert np.array_equal(read_internals(c, P, W, H)["point_list"], read_internals(ref, P, W, H)["point_list"])


def test_declined_batch_path_warns_once():
    """A model without 16 SH coefficients cannot take the batched kernels; the sink says so once."""
    import warnings
    import diff_gaussian_rasterization as dgr
    P = 64
    z = lambda *sh: torch.zeros(sh, device="cuda:0", requires_grad=True)
    dgr._warned.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sink = dgr.ParamGradSink(z(P, 3), z(P, 1, 3), z(P, 8, 3), z(P, 1), z(P, 3), z(P, 4), z(P, 4), defer=True)
        assert not sink.defer
        dgr.ParamGradSink(z(P, 3), z(P, 1, 3), z(P, 8, 3), z(P, 1), z(P, 3), z(P, 4), z(P, 4), defer=True)
        sink.prepare_views(_arc_settings(2, 32, 32))
        sink.prepare_views(_arc_settings(2, 32, 32))
    msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    assert len(msgs) == 2 and "16 coefficients" in msgs[0] and "prepare_views" in msgs[1]


@pytest.mark.parametrize("defer", [False, True], ids=["per_view_backward", "deferred_backward"])
def test_prepared_views_give_the_same_gradients(defer):
    """ParamGradSink.prepare_views in front of the views of a step: same images, same gradients
    (the backward of a prepared view recomputes the 3D covariance instead of reading it)."""
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    nviews, W, H = 4, 96, 64
    sc = synth.tiny_scene(P=600, width=W, height=H, seed=37)
    raw = _raw_scene(sc, seed=6)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names], defer=defer)
    sink_b = ParamGradSink(*[B[n] for n in names], defer=defer)
    settings = _arc_settings(nviews, W, H)
    rng = np.random.default_rng(11)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(nviews)]

    def run(P_, sink, prepare):
        sink.begin_step()
        if prepare:
            sink.prepare_views(settings)
            assert len(sink.prepared) == nviews
        imgs, m2s = [], []
        for v in range(nviews):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(settings[v]).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            ((o[0] * ws[v][0]).sum() + (o[2] * ws[v][1]).sum() + (o[5] * ws[v][2]).sum()).backward()
            imgs.append(o[0].detach().clone())
            m2s.append(m2)
        sink.flush()
        assert not sink.prepared
        return imgs, [m.grad.clone() for m in m2s]

    ia, ma = run(A, sink_a, False)
    ib, mb = run(B, sink_b, True)
    for x, y in zip(ia, ib):
        assert torch.equal(x, y)
    for x, y in zip(ma, mb):
        assert rel_err(y.cpu().numpy(), x.cpu().numpy()) <= REL_TOL
    for n in names:
        ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))


@pytest.mark.parametrize("nstreams", [0, 2, 3])
def test_run_views_on_side_streams(nstreams):
    """gsdp.run_views: prepared forward, deferred backward, views dealt to side streams; the same
    gradients as one view after the other through the plain per-view path."""
    import gsdp
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    nviews, W, H = 5, 96, 64
    sc = synth.tiny_scene(P=700, width=W, height=H, seed=41)
    raw = _raw_scene(sc, seed=7)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names])
    sink_b = ParamGradSink(*[B[n] for n in names], defer=True)
    settings = _arc_settings(nviews, W, H)
    rng = np.random.default_rng(13)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(nviews)]

    def loss_of(P_, sink):
        def f(i, rs):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(rs).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            return (o[0] * ws[i][0]).sum() + (o[2] * ws[i][1]).sum() + (o[5] * ws[i][2]).sum()
        return f

    sink_a.begin_step()
    la = []
    for i, rs in enumerate(settings):
        l = loss_of(A, sink_a)(i, rs)
        l.backward()
        la.append(l.detach())
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for _ in range(2):   # twice: the second step must start from clean buffers
        lb = gsdp.run_views(sink_b, settings, loss_of(B, sink_b), streams)
    torch.cuda.synchronize()
    for x, y in zip(la, lb):
        assert abs(float(x) - float(y)) <= 1e-4 * max(1.0, abs(float(x)))
    for n in names:
        ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))


@pytest.mark.parametrize("nviews,nstreams,defer,chunks", [(5, 3, False, 1), (11, 3, True, 1), (11, 0, True, 3),
                                                          (9, 2, True, 8)],
                         ids=["no_defer_with_streams", "more_than_8_views", "chunked_flush", "chunked_on_streams"])
def test_run_views_variants(nviews, nstreams, defer, chunks):
    """gsdp.run_views where it used to be fragile: a sink that does not defer (the side streams must
    not be used: every view's backward writes the shared buffers), more pending views than one
    batched launch takes, the flush run over ranges of Gaussians (what a multi-GPU step does, here
    with a recording stand-in for the reducer), and views whose SH degree differs inside one step."""
    import gsdp
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    W, H = 96, 64
    sc = synth.tiny_scene(P=700, width=W, height=H, seed=43)
    raw = _raw_scene(sc, seed=8)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names])
    sink_b = ParamGradSink(*[B[n] for n in names], defer=defer)
    settings = _arc_settings(nviews, W, H)
    settings[1] = settings[1]._replace(sh_degree=2)   # a second (scale modifier, degree) group inside the step
    rng = np.random.default_rng(17)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(nviews)]

    def loss_of(P_, sink):
        def f(i, rs):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(rs).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            return (o[0] * ws[i][0]).sum() + (o[2] * ws[i][1]).sum() + (o[5] * ws[i][2]).sum()
        return f

    sink_a.begin_step()
    for i, rs in enumerate(settings):
        loss_of(A, sink_a)(i, rs).backward()

    class Recorder:   # stands in for gsdp.GradReducer: the ranges it is handed, in order
        def __init__(self):
            self.rows = []

        def start_rows(self, tensors, first, count):
            assert all(x.size(0) == 700 for x in tensors)
            self.rows.append((first, count))

        def plan_rows(self, touched):   # always the dense exchange
            return None

        def wait(self):
            pass

    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    rec = Recorder() if chunks > 1 else None
    for _ in range(2):
        for p in B.values():
            p.grad = None   # optimizer.zero_grad(set_to_none=True): the sink re-attaches its buffers
        gsdp.run_views(sink_b, settings, loss_of(B, sink_b), streams, reducer=rec, chunks=chunks)
    torch.cuda.synchronize()
    if rec is not None:
        per_step = rec.rows[:len(rec.rows) // 2]
        assert per_step == [(0, 256), (256, 256), (512, 188)]   # 700 rows are three blocks of 256
    for n in names:
        assert B[n].grad is not None, n
        ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))


@pytest.mark.gpu
@pytest.mark.parametrize("defer", [True, False])
def test_untouched_rows_have_zero_gradients(defer):
    """What the sparse gradient exchange of gsdp.GradReducer rests on: a Gaussian that no live pixel of any
    view of the step reached (gauss_count == 0 in all of them, forward.cu:423-432) has all-zero gradient
    rows, and `ParamGradSink.touched_rows()` is exactly the OR of the views' gauss_count != 0."""
    import gsdp
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    W, H = 96, 64
    sc = synth.tiny_scene(P=3000, width=W, height=H, seed=47)
    raw = _raw_scene(sc, seed=9)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = (sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"])
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink = ParamGradSink(*[A[n] for n in names], defer=defer)
    settings = _arc_settings(3, W, H)
    rng = np.random.default_rng(19)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(3)]
    counts = []

    def f(i, rs):
        m2 = torch.zeros_like(A["xyz"], requires_grad=True)
        o = GaussianRasterizer(rs).forward_params(A["xyz"], m2, A["features_dc"], A["features_rest"], A["opacity"],
                                                  A["scaling"], A["rotation"], A["reps"], grad_sink=sink)
        counts.append(o[4])
        return (o[0] * ws[i][0]).sum() + (o[2] * ws[i][1]).sum() + (o[5] * ws[i][2]).sum()

    sink.flat.fill_(float("nan"))
    gsdp.run_views(sink, settings, f, [torch.cuda.Stream()])
    touched = sink.touched_rows()
    torch.cuda.synchronize()
    want = torch.stack([c != 0 for c in counts]).any(0)
    assert torch.equal(touched.bool(), want)
    n_touched = int(want.sum())
    assert 0 < n_touched < 3000, n_touched      # the scene has both kinds
    for k, g in sink.grads.items():
        g2 = g.view(g.size(0), -1)
        assert torch.isfinite(g2).all(), k
        assert not g2[~want].any(), (k, "a Gaussian no pixel reached has a non-zero gradient row")
    assert sink.grads["sh"].view(3000, -1)[want].abs().sum() > 0
    # the next step starts from an empty record
    sink.begin_step()
    assert not sink.touched_rows().any()


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [1, 3], ids=["one_launch", "chunked_flush"])
def test_keep_zero_rows_gives_the_same_gradients(chunks):
    """ParamGradSink(keep_zero_rows=True): blocks of 64 gradient rows that are zero and stay zero are not
    rewritten.  Over steps whose views reach different Gaussians the buffers must equal those of a plain
    sink: exact zeros wherever no view of THAT step reached (nothing stale from an earlier step), the same
    values elsewhere, the same per-view screen-space gradients; the block states say exactly which blocks
    hold a reached Gaussian; rows that someone else wrote are cleaned after `mark_rows_dirty`."""
    from diff_gaussian_rasterization import GaussianRasterizer, ParamGradSink
    W, H, P = 96, 64, 4000
    sc = synth.tiny_scene(P=P, width=W, height=H, seed=53)
    raw = _raw_scene(sc, seed=10)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "reps")
    vals = [sc["means3D"], raw["features_dc"], raw["features_rest"], raw["opacity"], raw["scaling"], raw["rotation"],
            sc["reps"]]
    settings = _arc_settings(11, W, H)
    rng = np.random.default_rng(23)
    ws = [[t(rng.standard_normal(s).astype(np.float32)) for s in ((3, H, W), (1, H, W), (4, H, W))] for _ in range(11)]

    def render(P_, sink, i, m2):
        o = GaussianRasterizer(settings[i]).forward_params(
            P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
            P_["reps"], grad_sink=sink)
        return o, (o[0] * ws[i][0]).sum() + (o[2] * ws[i][1]).sum() + (o[5] * ws[i][2]).sum()

    # put the Gaussians that some view reaches first, so that whole blocks of 64 are never reached
    with torch.no_grad():
        T0 = {n: t(v) for n, v in zip(names, vals)}
        reached = torch.zeros((P,), dtype=torch.bool, device="cuda:0")
        for i in range(11):
            reached |= render(T0, None, i, torch.zeros_like(T0["xyz"]))[0][4] != 0
    order = torch.argsort((~reached).to(torch.int8), stable=True).cpu().numpy()
    assert 64 < int(reached.sum()) < P - 256
    vals = [np.asarray(v)[order] for v in vals]
    A = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    B = {n: t(v).requires_grad_(True) for n, v in zip(names, vals)}
    sink_a = ParamGradSink(*[A[n] for n in names], defer=True)
    sink_b = ParamGradSink(*[B[n] for n in names], defer=True, keep_zero_rows=True)
    assert sink_b.row_blocks is not None and sink_a.row_blocks is None

    def step(P_, sink, views):
        sink.begin_step()
        m2s, counts = [], []
        for i in views:
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o, loss = render(P_, sink, i, m2)
            loss.backward()
            m2s.append(m2)
            counts.append(o[4])
        sink.flush(chunks=chunks)
        return m2s, torch.stack([c != 0 for c in counts])

    def check(views, label):
        ma, _ = step(A, sink_a, views)
        mb, hit = step(B, sink_b, views)
        torch.cuda.synchronize()
        want = hit.any(0)
        for k, g in sink_b.grads.items():
            g2, r2 = g.view(P, -1), sink_a.grads[k].view(P, -1)
            assert torch.isfinite(g2).all(), (label, k)
            assert not g2[~want].any(), (label, k, "stale values in rows no view of this step reached")
            assert rel_err(g2.cpu().numpy(), r2.cpu().numpy()) <= REL_TOL, (label, k)
        for v, (x, y) in enumerate(zip(ma, mb)):
            assert not y.grad[~hit[v]].any(), (label, v)
            assert rel_err(y.grad.cpu().numpy(), x.grad.cpu().numpy()) <= REL_TOL, (label, v)
        nb = (P + 63) // 64
        pad = torch.zeros((nb * 64,), dtype=torch.bool, device="cuda:0")
        pad[:P] = want
        assert torch.equal(sink_b.row_blocks.bool(), pad.view(nb, 64).any(1)), label
        for v, (_, blocks) in enumerate(sink_b.m2_pool[:len(views)]):
            pad[:P] = hit[v]
            assert torch.equal(blocks.bool(), pad.view(nb, 64).any(1)), (label, v)
        return want

    w1 = check([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10], "eleven views: two launches, the second one adds")
    w2 = check([5], "one view: most of the rows of the first step must be cleared")
    assert int((w1 & ~w2).sum()) > 0
    check([0, 10], "two other views")
    # somebody else writes into rows nobody reaches, and says so
    rows = torch.tensor([P - 1, P - 70, P - 200], device="cuda:0")
    for g in sink_b.grads.values():
        g.view(P, -1)[rows] = 7.0
    sink_b.mark_rows_dirty(rows)
    check([3, 4], "after foreign writes")
    assert len(sink_b.m2_pool) == 11
