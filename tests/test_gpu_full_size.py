"""GPU parity at the full sizes of the benchmark configurations.

  C1  100 k Gaussians, 800x800, fwd + bwd           full comparison with the CPU checker
  C2  2.5 M Gaussians, 1920x1080, render only      full comparison with the CPU checker; the fork's
                                                    statistics invariants on the kernel output
  C3  6 M Gaussians, 1297x840, antialiasing,        full comparison, forward and backward,
      forward + fused-ssim + backward               plus the ssim kernels on the rendered image
  C2t / C3t  the same two on the translucent        deep lists: the pixels blend a large part of every tile
      variant of the scene                          list (the street scene saturates after a few per cent),
                                                    the forward kernel with quadrant reach masks is the one
                                                    the library picks, T / (1 - alpha) is recovered dozens to
                                                    hundreds of times in a row per pixel in the backward
  C4  6 M Gaussians, 1920x1080, 8 cameras,          the step as bench.py runs it (batched forward preprocess,
      one multi-view step on one GPU                views on 3 streams, deferred backward) against the plain
                                                    per-view path; one of the views in full against the CPU checker
  C5  20 M Gaussians, 3840x2160 render and          size-independent properties (sortedness, ranges,
      distCUDA2 over the 20 M points                permutation, counts), knn exact on a strided
                                                    sample of query points through the CPU checker

Bars as everywhere: integers bit-exact, per-pixel floats 1e-5 absolute,
accumulated floats and gradients 1e-4 relative.  Each test prints how long the
CPU checker took; the sizes are chosen so that the file finishes in a few
minutes on the 16 host cores of a one-GPU box.
"""
import time

import numpy as np
import pytest
import torch

import synth
from gpu_util import read_internals, rel_err, run_forward_raw, scene_to_torch, worst
from test_gpu_rasterizer import ABS_TOL, REL_TOL, _check_forward, _grads_gpu

pytestmark = pytest.mark.gpu


def test_c1_ball_100k_800x800_forward_and_backward(oracle):
    """C1 at its stated size: 100 k Gaussians, 800x800, forward + backward, white background."""
    sc = synth.ball_scene(100_000, width=800, height=800)
    f, out, rs, ten, kw_g = _check_forward(oracle, sc, False, (1.0, 1.0, 1.0))
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(6)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, gd, gr)
    t = lambda a: torch.tensor(a, device="cuda:0")
    got = _grads_gpu(rs, ten, t(gc), t(gd), t(gr), kw_g, out)
    torch.cuda.synchronize()
    for k in got:
        r = ref[k].reshape(got[k].shape)
        assert np.isfinite(got[k]).all(), k
        assert rel_err(got[k], r) <= REL_TOL, (k, rel_err(got[k], r))


def test_c2_render_2p5M_1080p(oracle):
    sc = synth.street_scene(2_500_000, 1920, 1080, seed=2)
    t0 = time.time()
    f, out, rs, ten, _ = _check_forward(oracle, sc, False, (0.0, 0.0, 0.0))
    print(f"C2: R={f.R}, forward on both sides + compare {time.time() - t0:.1f}s")
    assert f.R > 5_000_000


def _fork_statistics_invariants(out, P, W, H):
    """The fork's extra outputs on what the HIP kernels produce (no checker involved).
    The fork author's own test (diff_gaussian_rasterization/__init__.py:130-147) compares, for one
    Gaussian, the mean final transmittance of the pixels it contributed to last with
    gauss_sum / gauss_count; that equality needs the alternate accumulation the author left
    commented out in forward.cu:449-451.  Under the accumulation the fork ships (pre-blend T of every
    tested pair, forward.cu:431-432) what can be proven for EVERY Gaussian g is:
      * pixels whose last contributor is g are among the pairs counted for g, and their final T is
        T_before(g) * (1 - alpha) <= T_before(g):  #pixels(last == g) <= gauss_count[g]  and
        sum of accum_alpha over those pixels <= gauss_sum[g];
      * T <= 1:  gauss_sum[g] <= gauss_count[g];  one is zero exactly when the other is;
      * a pixel nothing contributed to has last_contr_gauss == -1 and accum_alpha == 1."""
    last = out["last_contr_gauss"].view(-1).long()
    T = out["accum_alpha"].view(-1)
    gsum, gcnt = out["gauss_sum"].double(), out["gauss_count"].long()
    assert int(last.min()) >= -1 and int(last.max()) < P
    hit = last >= 0
    pix_per_g = torch.bincount(last[hit], minlength=P)
    t_per_g = torch.zeros(P, dtype=torch.float64, device=T.device).index_add_(0, last[hit], T[hit].double())
    assert bool((pix_per_g <= gcnt).all())
    assert bool((t_per_g <= gsum * (1 + 1e-4) + 1e-6).all())
    assert bool((gsum <= gcnt.double() * (1 + 1e-6)).all())
    assert bool(((gcnt == 0) == (gsum == 0)).all())
    assert bool((gcnt[out["radii"] == 0] == 0).all())
    n_contrib = torch.tensor(read_internals(out, P, W, H)["n_contrib"].astype(np.int64), device=T.device)
    assert bool((last[n_contrib == 0] == -1).all()) and bool((T[n_contrib == 0] == 1.0).all())
    assert bool((last[n_contrib > 0] >= 0).all())
    # the author's Gaussian of choice, the farthest one: nothing lies behind it, so wherever it blended it is the
    # last contributor, and the author's two averages are then ordered the way the inequality above says
    far = int(torch.argmax(torch.where(out["radii"] > 0, out["out_depths"], torch.full_like(out["out_depths"], -1.0))))
    if int(pix_per_g[far]) > 0:
        assert float(t_per_g[far] / pix_per_g[far]) <= 1.0
    assert int(hit.sum()) > W * H // 2 and int((pix_per_g > 0).sum()) > 1_000


def test_c2_fork_statistics_invariants_on_the_kernel_output():
    P, W, H = 2_500_000, 1920, 1080
    sc = synth.street_scene(P, W, H, seed=2)
    rs, ten = scene_to_torch(sc, antialiasing=False, debug=False)
    out = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    _fork_statistics_invariants(out, P, W, H)


def _walked_fraction(f, W, H):
    """Share of the sorted instances that lie in front of the last entry some pixel of their tile blended
    (what the blend kernels walk), from the checker's n_contrib and ranges."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = np.zeros((gy * 16, gx * 16), np.int64)
    nc[:H, :W] = f.n_contrib.reshape(H, W)
    per_tile = nc.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    return float(per_tile.sum()) / max(1, f.R), per_tile


def test_c2_translucent_render_2p5M_1080p_deep_lists(oracle):
    """C2 on the translucent variant of the scene (small, mostly transparent splats): every pixel blends a
    large part of its tile's list.  The library's own choice of forward kernel must be the one with
    quadrant reach masks here (fewer than six tiles per rendered Gaussian); the plain kernel and the
    (tile id, index) pair path are forced after it; all against the checker, then the fork-statistics
    invariants on the kernel output."""
    from _gsplat_lib import lib
    P, W, H = 2_500_000, 1920, 1080
    sc = synth.street_scene(P, W, H, seed=2, translucent=True)
    t0 = time.time()
    f, out, rs, ten, _ = _check_forward(oracle, sc, False, (0.0, 0.0, 0.0))
    vis = int((f.radii > 0).sum())
    assert f.R < 6 * vis, "the library picks the reach-mask forward kernel when R < 6 * visible"
    frac, per_tile = _walked_fraction(f, W, H)
    print(f"C2t: R={f.R}, visible={vis}, walked {frac:.2f} of R, deepest walk {int(per_tile.max())} entries, "
          f"{time.time() - t0:.1f}s")
    assert frac > 0.25 and int(per_tile.max()) > 1000     # deep lists, not the street scene's 3 %
    _fork_statistics_invariants(out, P, W, H)
    try:
        lib().gsr_test_forward_reach_masks(0)
        _check_forward(oracle, sc, False, (0.0, 0.0, 0.0))
    finally:
        lib().gsr_test_forward_reach_masks(-1)
    try:
        lib().gsr_test_packed_instances(0)
        _check_forward(oracle, sc, False, (0.0, 0.0, 0.0), packed_pairs=False)
    finally:
        lib().gsr_test_packed_instances(1)


def _c3_train_step(oracle, sc, tag, floor=1e-2):
    t0 = time.time()
    f, out, rs, ten, kw_g = _check_forward(oracle, sc, True, (0.0, 0.0, 0.0))
    t1 = time.time()
    H, W = rs.image_height, rs.image_width
    # the gradient a training step sends back: L1 + (1 - ssim) against a fixed image, reps regulariser
    from fused_ssim import fused_ssim
    gt = torch.rand((3, H, W), device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(5))
    img = out["color"].clone().requires_grad_(True)
    rep = out["out_reps"].clone().requires_grad_(True)
    ic = img.clamp(0, 1)
    loss = 0.8 * (ic - gt).abs().mean() + 0.2 * (1 - fused_ssim(ic.unsqueeze(0), gt.unsqueeze(0)))
    loss = loss + 0.01 * (rep - 10.0).abs().mean()
    loss.backward()
    gc, gr = img.grad.contiguous(), rep.grad.contiguous()
    # ssim value against the CPU checker on this image
    icn, gtn = ic.detach().cpu().numpy()[None], gt.cpu().numpy()[None]
    m, d1, d2, d3 = oracle.ssim_fwd(icn, gtn)
    val = fused_ssim(ic.detach().unsqueeze(0), gt.unsqueeze(0), train=False)
    assert abs(float(val) - float(m.astype(np.float64).mean())) <= 1e-5
    ref = f.backward(gc.cpu().numpy(), None, gr.cpu().numpy())
    t2 = time.time()
    got = _grads_gpu(rs, ten, gc, None, gr, kw_g, out)
    torch.cuda.synchronize()
    errs = {k: rel_err(got[k], ref[k].reshape(got[k].shape), floor) for k in got}
    frac, per_tile = _walked_fraction(f, W, H)
    print(f"{tag}: R={f.R}, walked {frac:.2f} of R (deepest {int(per_tile.max())}), forward {t1 - t0:.1f}s, "
          f"backward {t2 - t1:.1f}s; largest element-wise gradient errors: "
          + ", ".join(f"{k} {e:.1e}" for k, e in errs.items()))
    for k in got:
        r = ref[k].reshape(got[k].shape)
        assert np.isfinite(got[k]).all(), k
        assert errs[k] <= REL_TOL, (k, errs[k], worst(got[k], r, floor))
    return f, frac, per_tile


def test_c3_train_step_6M_1297x840(oracle):
    _c3_train_step(oracle, synth.street_scene(6_000_000, 1297, 840, seed=3), "C3")


def test_c3_translucent_train_step_6M_1297x840_deep_lists(oracle):
    """C3 on the translucent variant: forward + loss + backward with every gradient element-wise within
    1e-4 of the checker where the backward recovers T by a division per blended Gaussian hundreds of
    times in a row along a pixel's list (backward.cu:591-684), with the hardware exponential and the
    Newton-corrected reciprocal of the HIP kernel."""
    sc = synth.street_scene(6_000_000, 1297, 840, seed=3, translucent=True)
    f, frac, per_tile = _c3_train_step(oracle, sc, "C3t")
    assert f.R < 6 * int((f.radii > 0).sum())        # forward: the reach-mask kernel
    assert frac > 0.25 and int(per_tile.max()) > 1000


def test_wide_feature_rows_train_step_2p5M_1297x840_dim16_deep_lists(oracle):
    """Feature rows of 16 channels (the reference's CLI default, arguments/__init__.py:98; its kernels are
    compiled for one width, config.h:16) at size, on the translucent scene where most visits blend (94 % of
    the list entries walked, the deepest tile walk 5243 entries): the twelve extra channels are staged from the
    caller's tensor per list entry in the forward and reduced into a second gradient record per Gaussian in
    the backward.

    The bar for the gradients here is |got - ref| <= 1e-4 |ref| + 3e-6 max|ref| per element (3e-6 instead of
    the usual 1e-6 of the tensor's maximum for elements far below it).  Measured on scenes of this shape
    (seeds 3, 4, 5; 2.5 M and 6 M Gaussians): with 4 channels the largest deviation under the usual measure is
    3e-5 to 5e-5, with 16 channels 4e-5 to 1.1e-4, in a different tensor each time (dL_dmean3D once, dL_drot
    once: single elements of a few thousandths of their tensor's maximum, off by 1.5e-6 of that maximum).
    dL/dalpha of a pixel is a sum over the channels, 20 terms of mixed sign instead of 8, formed here as
    S - accum_S from channel sums and in the checker channel by channel; both are float evaluations of the same
    expression, and the geometry gradients are sums over pixels in which those terms cancel again.  The
    kernel's shortcuts are not the cause: built with an IEEE division for T / (1 - alpha) and the polynomial
    exponential (-DGS_BWD_EXACT_DIV -DGS_BWD_EXACT_EXP) the same elements deviate by the same amounts."""
    sc = synth.street_scene(2_500_000, 1297, 840, seed=4, dim=16, translucent=True)
    assert sc["reps"].shape == (2_500_000, 16)
    f, frac, per_tile = _c3_train_step(oracle, sc, "C3 dim 16", floor=3e-2)
    assert frac > 0.25


def test_c4_multi_view_step_6M_1080p(oracle):
    import gsdp
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, ParamGradSink
    P, W, H, V = 6_000_000, 1920, 1080, 8
    sc = synth.street_scene(P, W, H, seed=2)
    dev = "cuda:0"
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    op = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
    raw = dict(xyz=sc["means3D"], features_dc=np.ascontiguousarray(sc["shs"][:, :1]),
               features_rest=np.ascontiguousarray(sc["shs"][:, 1:]), opacity=np.log(op / (1 - op)).astype(np.float32),
               scaling=np.log(sc["scales"]).astype(np.float32), rotation=sc["rotations"], reps=sc["reps"])
    names = tuple(raw)
    A = {n: t(raw[n]).requires_grad_(True) for n in names}
    B = {n: t(raw[n]).requires_grad_(True) for n in names}
    cams = [synth.street_camera(W, H, v, V) for v in range(V)]
    settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=t(c["viewmatrix"]), projmatrix=t(c["projmatrix"]), sh_degree=sc["sh_degree"],
        campos=t(c["campos"]), prefiltered=False, debug=False, antialiasing=True) for c in cams]
    ws = [torch.randn((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(v)) for v in range(V)]
    imgs = {}

    def loss_of(P_, sink, tag):
        def f(i, rs):
            m2 = torch.zeros_like(P_["xyz"], requires_grad=True)
            o = GaussianRasterizer(rs).forward_params(
                P_["xyz"], m2, P_["features_dc"], P_["features_rest"], P_["opacity"], P_["scaling"], P_["rotation"],
                P_["reps"], grad_sink=sink)
            imgs[(tag, i)] = (o[0].detach(), o[1].detach(), o[4].detach())
            return (o[0] * ws[i]).sum() / (H * W) + 0.01 * (o[5] - 10.0).abs().mean()
        return f

    # A: one view after the other, each with its own preprocess and per-Gaussian backward
    sink_a = ParamGradSink(*[A[n] for n in names])
    sink_a.begin_step()
    for i, rs in enumerate(settings):
        loss_of(A, sink_a, "a")(i, rs).backward()
    # B: the batched step on three streams
    sink_b = ParamGradSink(*[B[n] for n in names], defer=True)
    gsdp.run_views(sink_b, settings, loss_of(B, sink_b, "b"), [torch.cuda.Stream() for _ in range(3)])
    torch.cuda.synchronize()
    for i in range(V):
        for x, y in zip(imgs[("a", i)], imgs[("b", i)]):
            assert torch.equal(x, y), i                      # image, radii, gauss_count: bit for bit
    for n in names:
        ga, gb = A[n].grad.cpu().numpy(), B[n].grad.cpu().numpy()
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) <= REL_TOL, (n, rel_err(gb, ga))
    # one view of the batched step in full against the CPU checker
    v = 5
    o_, s_, r_ = oracle.activate(raw["opacity"].reshape(-1, 1), raw["scaling"], raw["rotation"])
    sc2 = dict(sc)
    sc2["opacities"], sc2["scales"], sc2["rotations"], sc2["camera"] = o_, s_, r_, cams[v]
    t0 = time.time()
    f = oracle.forward_from_scene(sc2, bg=(0.0, 0.0, 0.0), antialiasing=True)
    print(f"C4: view {v}: R={f.R}, CPU checker forward {time.time() - t0:.1f}s")
    img, radii, gcnt = imgs[("b", v)]
    assert np.array_equal(radii.cpu().numpy(), f.radii)
    assert np.array_equal(gcnt.cpu().numpy(), f.gauss_count)
    assert np.abs(img.cpu().numpy() - f.out_color).max() <= ABS_TOL


def test_c5_render_20M_4k_properties():
    P, W, H = 20_000_000, 3840, 2160
    sc = synth.street_scene(P, W, H, seed=5)
    rs, ten = scene_to_torch(sc, antialiasing=True, debug=False)
    out = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    R = out["num_rendered"]
    it = read_internals(out, P, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    radii = out["radii"].cpu().numpy()
    tiles = it["bin"][:, 3].astype(np.int64)
    # the instance count is the sum of the tile rectangles, culled Gaussians have none
    assert int(tiles.sum()) == R and R > 50_000_000
    assert np.all((tiles > 0) == (radii > 0))
    keys, pl = it["keys"], it["point_list"]
    tile_of = (keys >> np.uint64(32)).astype(np.int64)
    assert np.all(keys[:-1] <= keys[1:])                               # sorted by (tile, depth)
    assert tile_of.max() < gx * gy
    assert np.array_equal(np.bincount(pl, minlength=P).astype(np.int64), tiles)   # every instance exactly once
    # the depth half of a key is the f32 bits of its Gaussian's view depth
    depths = out["out_depths"].cpu().numpy().view(np.uint32)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), depths[pl])
    # stable: equal keys keep the emission order (ascending Gaussian index inside one tile)
    same = keys[:-1] == keys[1:]
    assert np.all(pl[:-1][same] < pl[1:][same])
    # ranges partition the sorted list by tile
    cnt = np.bincount(tile_of, minlength=gx * gy)
    rng_ = it["ranges"].astype(np.int64)
    nz = cnt > 0
    assert np.array_equal((rng_[:, 1] - rng_[:, 0])[nz], cnt[nz])
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    assert np.array_equal(rng_[nz, 0], starts[nz])
    # per pixel: no more contributors than the tile's list, transmittance and colour in range
    ncontrib = it["n_contrib"].reshape(H, W).astype(np.int64)
    per_tile = cnt.reshape(gy, gx)
    lim = np.repeat(np.repeat(per_tile, 16, axis=0), 16, axis=1)[:H, :W]
    assert np.all(ncontrib <= lim)
    T = it["final_T"]
    assert np.all((T >= 0) & (T <= 1))
    col = out["color"].cpu().numpy()
    assert np.isfinite(col).all() and col.min() >= 0
    # the fork's accum_alpha output is the final transmittance itself (forward.cu:478-479)
    assert np.array_equal(out["accum_alpha"].cpu().numpy().reshape(-1), T)
    # last contributor: -1 exactly where nothing contributed
    last = out["last_contr_gauss"].cpu().numpy().reshape(H, W)
    assert np.all((last >= 0) == (ncontrib > 0))
    # every tested pair was counted once: the counts cannot exceed the pixels of the tiles a Gaussian touches
    gcnt = out["gauss_count"].cpu().numpy().astype(np.int64)
    assert np.all(gcnt >= 0) and np.all(gcnt <= tiles * 256)
    assert np.all(gcnt[radii == 0] == 0)


def test_c5_dist2_20M_points(oracle):
    from simple_knn._C import distCUDA2
    P = 20_000_000
    sc = synth.street_scene(P, 3840, 2160, seed=5)
    pts = np.ascontiguousarray(sc["means3D"], np.float32)
    del sc
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    assert got.shape == (P,) and np.isfinite(got).all() and np.all(got >= 0)
    t0 = time.time()
    ref = oracle.dist2(pts, stride=97)     # the same search for every 97th point of the Morton order
    print(f"C5 knn: CPU checker {time.time() - t0:.1f}s")
    sel = ref >= 0
    assert sel.sum() >= P // 97
    assert np.array_equal(got[sel], ref[sel])


def test_sort_more_than_2_pow_30_pairs():
    """The stand-alone sort beyond 2^30 pairs (64-bit look-back words; the reference's instance count may
    reach 2^31 - 1): size-independent properties on the device, no checker at this size.  About 35 GB."""
    import ctypes as C
    from _gsplat_lib import lib
    L = lib()
    n = (1 << 30) + 4097
    end_bit = 33
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(9)
    k0 = torch.randint(0, 1 << 24, (n,), device=dev, generator=g, dtype=torch.int64)      # many ties
    k0 |= torch.randint(0, 1 << 9, (n,), device=dev, generator=g, dtype=torch.int64) << 24
    k0 |= 0x5A << 40                                                                        # bits the sort must ignore
    v0 = torch.arange(n, device=dev, dtype=torch.int32)
    sum_in = int(k0.sum())
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    temp = torch.empty(L.gsr_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
    res = C.c_int(-1)
    rc = L.gsr_sort_pairs_u64_u32(k0.data_ptr(), v0.data_ptr(), k1.data_ptr(), v1.data_ptr(), n, end_bit,
                                  temp.data_ptr(), temp.numel(), torch.cuda.current_stream().cuda_stream, C.byref(res))
    assert rc == 0
    torch.cuda.synchronize()
    off = L.gsr_sort_temp_error_offset()
    assert int(temp[off:off + 4].view(torch.int32).item()) == 0
    ko, vo = (k0, v0) if res.value == 0 else (k1, v1)
    del temp
    mk = ko & ((1 << end_bit) - 1)
    assert bool((mk[1:] >= mk[:-1]).all())                                  # sorted on the key bits
    same = mk[1:] == mk[:-1]
    assert bool((vo[1:][same] > vo[:-1][same]).all())                       # stable: ties keep input order
    assert int(ko.sum()) == sum_in and bool(((ko >> 40) == 0x5A).all())     # the same multiset of keys, upper bits kept
    assert int(vo.long().sum()) == n * (n - 1) // 2                         # a permutation of the values
