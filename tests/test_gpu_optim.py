"""GPU parity of the fused Adam step: bit-exact against the CPU checker (same
arithmetic, shared header), close to torch.optim.Adam (which forms its bias
corrections in double precision), and a drop-in for it as far as the
reference's optimizer-state surgery is concerned."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu_update(p, g, m, v, vis, lr, b1, b2, eps, step):
    from _gsplat_lib import check, lib
    t = lambda a: torch.tensor(a, device="cuda:0")
    tp, tg, tm, tv = t(p), t(g), t(m), t(v)
    tvis = None if vis is None else torch.tensor(vis.astype(np.uint8), device="cuda:0")
    N = p.shape[0]
    M = p.size // max(N, 1)
    bc1 = 1.0 - b1 ** step if step > 0 else 1.0
    bc2s = float(np.sqrt(1.0 - b2 ** step)) if step > 0 else 1.0
    check(lib().gso_adam_update(tp.data_ptr(), tg.data_ptr(), tm.data_ptr(), tv.data_ptr(),
                                None if tvis is None else tvis.data_ptr(), N, M, lr, b1, b2, eps, bc1, bc2s,
                                torch.cuda.current_stream().cuda_stream), "adam")
    torch.cuda.synchronize()
    return tp.cpu().numpy(), tm.cpu().numpy(), tv.cpu().numpy()


@pytest.mark.parametrize("shape", [(1, 1), (1000, 3), (777, 4), (513, 15, 3), (4097, 1), (300, 1, 3)])
@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
def test_adam_matches_checker_bit_for_bit(oracle, shape, sparse):
    rng = np.random.default_rng(sum(shape))
    p = rng.standard_normal(shape).astype(np.float32)
    m = (rng.standard_normal(shape) * 0.1).astype(np.float32)
    v = (rng.standard_normal(shape) ** 2 * 0.01).astype(np.float32)
    vis = (rng.random(shape[0]) < 0.6) if sparse else None
    for step in (1, 2, 7):
        g = rng.standard_normal(shape).astype(np.float32)
        st = 0 if sparse else step
        rp, rm, rv = oracle.adam_update(p, g, m, v, vis, 1.6e-4, 0.9, 0.999, 1e-15, st)
        gp, gm, gv = _gpu_update(p, g, m, v, vis, 1.6e-4, 0.9, 0.999, 1e-15, st)
        assert np.array_equal(gp, rp) and np.array_equal(gm, rm) and np.array_equal(gv, rv)
        if sparse:
            assert np.array_equal(gp[~vis], p[~vis]) and np.array_equal(gm[~vis], m[~vis])
        p, m, v = gp, gm, gv


def test_optimizer_class_against_torch_adam():
    from gsplat_optim import FusedGaussianAdam
    rng = np.random.default_rng(3)
    shapes = {"xyz": (2000, 3), "f_dc": (2000, 1, 3), "f_rest": (2000, 15, 3), "opacity": (2000, 1), "scaling": (2000, 3),
              "rotation": (2000, 4), "reps": (2000, 4)}
    lrs = dict(xyz=1.6e-4, f_dc=2.5e-3, f_rest=1.25e-4, opacity=2.5e-2, scaling=5e-3, rotation=1e-3, reps=1e-2)
    init = {k: rng.standard_normal(s).astype(np.float32) for k, s in shapes.items()}

    def make(cls):
        ps = {k: torch.nn.Parameter(torch.tensor(v, device="cuda:0")) for k, v in init.items()}
        groups = [{"params": [ps[k]], "lr": lrs[k], "name": k} for k in shapes]
        return ps, cls(groups, lr=0.0, eps=1e-15)

    pa, oa = make(torch.optim.Adam)
    pb, ob = make(FusedGaussianAdam)
    for step in range(5):
        for k in shapes:
            g = torch.tensor(rng.standard_normal(shapes[k]).astype(np.float32), device="cuda:0")
            pa[k].grad = g.clone()
            pb[k].grad = g.clone()
        if step == 2:   # the scheduler changes the position learning rate every iteration (gaussian_model.py:186-195)
            for o in (oa, ob):
                for grp in o.param_groups:
                    if grp["name"] == "xyz":
                        grp["lr"] = 1.1e-4
        oa.step()
        ob.step()
    for k in shapes:
        a, b = pa[k].detach().cpu().numpy(), pb[k].detach().cpu().numpy()
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(a).max()), k
        sa, sb = oa.state[pa[k]], ob.state[pb[k]]
        assert set(sb.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sb["step"]) == 5.0
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    # the state surgery of densification: replace a tensor and keep (masked) moments, as _prune_optimizer does
    mask = torch.arange(2000, device="cuda:0") % 3 != 0
    for grp in ob.param_groups:
        old = grp["params"][0]
        st = ob.state.pop(old)
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
        new = torch.nn.Parameter(old[mask].detach().clone())
        grp["params"][0] = new
        ob.state[new] = st
        new.grad = torch.ones_like(new)
    ob.step()
    sd = ob.state_dict()
    assert len(sd["state"]) == len(shapes)
    # sparse step: only visible rows move, parameter groups that are not per-Gaussian are left alone
    n = int(mask.sum())
    vis = torch.rand(n, device="cuda:0") < 0.5
    before = {grp["name"]: grp["params"][0].detach().clone() for grp in ob.param_groups}
    ob.step(vis, n)
    for grp in ob.param_groups:
        p, b = grp["params"][0].detach(), before[grp["name"]]
        assert torch.equal(p[~vis], b[~vis])
        assert not torch.equal(p[vis], b[vis])


def test_densify_stats(oracle):
    from gsplat_optim import add_densification_stats
    rng = np.random.default_rng(11)
    P = 100_003
    radii = (rng.integers(0, 40, P) * (rng.random(P) < 0.7)).astype(np.int32)
    grad = rng.standard_normal((P, 3)).astype(np.float32)
    acc = rng.random((P, 1)).astype(np.float32)
    den = rng.integers(0, 5, (P, 1)).astype(np.float32)
    mx = (rng.random(P) * 30).astype(np.float32)
    t = lambda a: torch.tensor(a, device="cuda:0")
    tr, tg, ta, td, tm = t(radii), t(grad), t(acc), t(den), t(mx)
    add_densification_stats(tr, tg, ta, td, tm)
    ra, rd, rm = oracle.densify_stats(radii, grad, acc, den, mx)
    assert np.array_equal(td.cpu().numpy(), rd) and np.array_equal(tm.cpu().numpy(), rm)
    assert np.allclose(ta.cpu().numpy(), ra, rtol=1e-6, atol=0)
    # the reference's own formulation in torch
    vf = (tr > 0).nonzero()
    ma, mdn, mm = t(acc), t(den), t(mx)
    mm[vf] = torch.max(mm[vf], tr[vf].float())
    ma[vf] += torch.norm(tg[vf, :2], dim=-1, keepdim=True)
    mdn[vf] += 1
    assert torch.equal(mm, tm) and torch.equal(mdn, td)
    assert torch.allclose(ma, ta, rtol=1e-6, atol=0)


@pytest.mark.parametrize("N,frac", [(1, 1.0), (255, 0.5), (256, 0.0), (1000, 0.3), (70001, 0.9)])
def test_row_compactor_equals_boolean_indexing(N, frac):
    from gsplat_optim import RowCompactor
    g = torch.Generator(device="cuda").manual_seed(N)
    mask = torch.rand(N, device="cuda", generator=g) < frac
    comp = RowCompactor(mask)
    assert comp.kept == int(mask.sum())
    for t in (torch.randn(N, 3, device="cuda"), torch.randn(N, 15, 3, device="cuda"), torch.randn(N, device="cuda"),
              torch.randn(N, 1, device="cuda"), torch.arange(N, device="cuda", dtype=torch.int32),
              torch.randn(N, 2, device="cuda", dtype=torch.float64)):
        got = comp(t)
        assert got.dtype == t.dtype and torch.equal(got, t[mask])


def _adam_with_state(N, fused):
    from gsplat_optim import FusedGaussianAdam
    g = torch.Generator().manual_seed(5)
    shapes = dict(xyz=(N, 3), f_dc=(N, 1, 3), f_rest=(N, 15, 3), opacity=(N, 1), scaling=(N, 3), rotation=(N, 4), reps=(N, 4))
    groups = [{"params": [torch.nn.Parameter(torch.randn(*s, generator=g).cuda())], "lr": 1e-3, "name": n}
              for n, s in shapes.items()]
    opt = (FusedGaussianAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    for gr in opt.param_groups:
        gr["params"][0].grad = torch.randn(*gr["params"][0].shape, generator=g).cuda()
    opt.step()
    return opt


def test_prune_and_cat_match_the_reference_formulation():
    """prune_optimizer / cat_tensors_to_optimizer against scene/gaussian_model.py:325-346, 427-441 restated."""
    from gsplat_optim import cat_tensors_to_optimizer, prune_optimizer
    N = 3001
    a, b = _adam_with_state(N, True), _adam_with_state(N, True)
    mask = torch.rand(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) < 0.7
    got, comp = prune_optimizer(a, mask)
    for group in b.param_groups:   # the reference's loop
        p = group["params"][0]
        st = b.state.get(p)
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
        del b.state[p]
        group["params"][0] = torch.nn.Parameter(p[mask].requires_grad_(True))
        b.state[group["params"][0]] = st
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert got[ga["name"]] is pa and pa.requires_grad
        assert torch.equal(pa, pb)
        assert torch.equal(a.state[pa]["exp_avg"], b.state[pb]["exp_avg"])
        assert torch.equal(a.state[pa]["exp_avg_sq"], b.state[pb]["exp_avg_sq"])
    stats = torch.randn(N, 1, device="cuda")
    assert torch.equal(comp(stats), stats[mask])
    K = int(mask.sum())
    ext = {g["name"]: torch.randn((17,) + tuple(g["params"][0].shape[1:]), device="cuda") for g in a.param_groups}
    got = cat_tensors_to_optimizer(a, ext)
    for g in a.param_groups:
        p = g["params"][0]
        assert got[g["name"]] is p and p.shape[0] == K + 17
        assert torch.equal(p[K:], ext[g["name"]])
        assert torch.count_nonzero(a.state[p]["exp_avg"][K:]) == 0 and a.state[p]["exp_avg"].shape == p.shape
    for g in a.param_groups:   # the optimizer still steps after the surgery
        g["params"][0].grad = torch.randn_like(g["params"][0])
    a.step()


def _reference_densify_and_prune(model, opt, max_grad, min_opacity, extent, max_screen_size, radii, percent_dense, N,
                                 normals):
    """GaussianModel.densify_and_prune / densify_and_clone / densify_and_split / prune_points
    (scene/gaussian_model.py:216-425) restated on a dict of tensors with plain PyTorch, statement by statement;
    the split's torch.normal(mean=0, std=stds) is taken as normals * stds with the normals handed in."""
    import torch.nn as nn
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "reps")

    def param(n):
        return [g for g in opt.param_groups if g["name"] == n][0]["params"][0]

    def cat(d):
        for g in opt.param_groups:
            ext = d[g["name"]]
            p = g["params"][0]
            st = opt.state.get(p, None)
            newp = nn.Parameter(torch.cat((p.data, ext), dim=0).requires_grad_(True))
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del opt.state[p]
                opt.state[newp] = st
            g["params"][0] = newp

    def prune(mask):
        keep = ~mask
        for g in opt.param_groups:
            p = g["params"][0]
            st = opt.state.get(p, None)
            newp = nn.Parameter(p.data[keep].requires_grad_(True))
            if st is not None:
                st["exp_avg"] = st["exp_avg"][keep]
                st["exp_avg_sq"] = st["exp_avg_sq"][keep]
                del opt.state[p]
                opt.state[newp] = st
            g["params"][0] = newp
        for k in ("xyz_gradient_accum", "denom", "max_radii2D", "tmp_radii"):
            model[k] = model[k][keep]

    def postfix(d, new_radii):
        cat(d)
        n = param("xyz").shape[0]
        model["tmp_radii"] = torch.cat((model["tmp_radii"], new_radii))
        model["xyz_gradient_accum"] = torch.zeros((n, 1), device="cuda")
        model["denom"] = torch.zeros((n, 1), device="cuda")
        model["max_radii2D"] = torch.zeros((n,), device="cuda")

    grads = model["xyz_gradient_accum"] / model["denom"]
    grads[grads.isnan()] = 0.0
    model["tmp_radii"] = radii
    # clone
    sel = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(torch.exp(param("scaling").data), dim=1).values <= percent_dense * extent)
    postfix({n: param(n).data[sel] for n in names}, model["tmp_radii"][sel])
    # split
    n_init = param("xyz").shape[0]
    padded = torch.zeros((n_init,), device="cuda")
    padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.where(padded >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(torch.exp(param("scaling").data), dim=1).values > percent_dense * extent)
    stds = torch.exp(param("scaling").data)[sel].repeat(N, 1)
    samples = normals[: stds.shape[0]] * stds
    q = param("rotation").data[sel]
    qn = q / torch.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])[:, None]
    r, x, y, z = qn[:, 0], qn[:, 1], qn[:, 2], qn[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    rots = R.repeat(N, 1, 1)
    d = {n: param(n).data[sel].repeat(*([N] + [1] * (param(n).dim() - 1))) for n in names}
    d["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + param("xyz").data[sel].repeat(N, 1)
    d["scaling"] = torch.log(torch.exp(param("scaling").data)[sel].repeat(N, 1) / (0.8 * N))
    postfix(d, model["tmp_radii"][sel].repeat(N))
    prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), device="cuda", dtype=bool))))
    # prune
    mask = (torch.sigmoid(param("opacity").data) < min_opacity).squeeze()
    if max_screen_size:
        big_vs = model["max_radii2D"] > max_screen_size
        big_ws = torch.exp(param("scaling").data).max(dim=1).values > 0.1 * extent
        mask = torch.logical_or(torch.logical_or(mask, big_vs), big_ws)
    prune(mask)
    return int(sel.sum())


@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_and_prune_matches_the_reference_formulation(max_screen_size):
    """gsplat_optim.densify_and_prune (selection, split and prune-mask kernels + one row compaction per stage)
    against the reference's method restated above: the same rows in the same order in every parameter and Adam
    moment (bit for bit where rows are copied; the split children's positions and scales to 1e-6, a matrix
    product and a logarithm)."""
    import gsplat_optim
    g = torch.Generator(device="cuda").manual_seed(5)
    N, extent, pd = 5000, 4.0, 0.01
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    base = dict(xyz=rnd(N, 3), f_dc=rnd(N, 1, 3), f_rest=rnd(N, 15, 3), opacity=rnd(N, 1) * 3 - 2,
                scaling=rnd(N, 3) * 0.8 + np.log(pd * extent), rotation=rnd(N, 4), reps=rnd(N, 4))
    accum = torch.rand(N, 1, device="cuda", generator=g) * 0.001
    denom = torch.randint(0, 4, (N, 1), device="cuda", generator=g).float()      # zeros give NaN gradients
    radii = torch.randint(0, 40, (N,), device="cuda", generator=g).int()
    normals = rnd(2 * N, 3)

    def make():
        params = {k: torch.nn.Parameter(v.clone()) for k, v in base.items()}
        opt = torch.optim.Adam([{"params": [p], "lr": 1e-3, "name": k} for k, p in params.items()], eps=1e-15)
        for p in params.values():            # moments that can be told apart
            p.grad = torch.sin(p.data * 3)
        opt.step()
        return opt

    ref_opt, opt = make(), make()
    model = dict(xyz_gradient_accum=accum.clone(), denom=denom.clone(), max_radii2D=torch.rand(N, device="cuda") * 30)
    n_split = _reference_densify_and_prune(model, ref_opt, 0.0004, 0.05, extent, max_screen_size, radii.clone(), pd, 2,
                                           normals)
    stats = dict(xyz_gradient_accum=accum.clone(), denom=denom.clone(), max_radii2D=torch.rand(N, device="cuda") * 30)
    new_params, new_stats, new_radii = gsplat_optim.densify_and_prune(
        opt, stats, 0.0004, 0.05, extent, max_screen_size, radii.clone(), percent_dense=pd, n_split=2,
        normals=lambda n: normals[:n].contiguous())
    torch.cuda.synchronize()
    assert n_split > 100
    for gr, gg in zip(ref_opt.param_groups, opt.param_groups):
        name = gr["name"]
        pr, pg = gr["params"][0], gg["params"][0]
        assert pr.shape == pg.shape and pg.shape[0] != N, name
        assert new_params[name] is pg
        tol = dict(rtol=1e-6, atol=1e-6) if name in ("xyz", "scaling") else dict(rtol=0, atol=0)
        assert torch.allclose(pg.data, pr.data, **tol), name
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(opt.state[pg][k], ref_opt.state[pr][k]), (name, k)
    assert torch.equal(new_radii, model["tmp_radii"])
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        assert torch.equal(new_stats[k], model[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("P,narrays", [(1, 1), (3, 2), (1023, 8), (4096, 11), (100001, 0)])
def test_mark_nonzero_is_the_or_of_the_counts(P, narrays):
    """gso_mark_nonzero: the row mask of the Gaussians some view reached, from the views' hit counts (more
    arrays than one launch takes, lengths that are not a multiple of four, no array at all)."""
    import ctypes
    from _gsplat_lib import check, lib
    g = torch.Generator(device="cuda").manual_seed(P)
    cs = [(torch.rand(P, device="cuda", generator=g) < 0.05).int() * 7 for _ in range(narrays)]
    want = torch.stack(cs).ne(0).any(0) if cs else torch.zeros(P, dtype=torch.bool, device="cuda")
    t = torch.full(((P + 3) // 4 * 4,), 255, dtype=torch.uint8, device="cuda")[:P]
    ptrs = (ctypes.c_void_p * max(1, narrays))(*[c.data_ptr() for c in cs])
    check(lib().gso_mark_nonzero(P, narrays, ptrs, t.data_ptr(), torch.cuda.current_stream().cuda_stream),
          "gso_mark_nonzero")
    torch.cuda.synchronize()
    assert torch.equal(t, want.to(torch.uint8))
