"""GPU parity on the corner cases of the operator's argument space: scale
modifier, active SH degree below the allocated one, one Gaussian, a splat
covering the whole image, images smaller than a tile and not multiples of 16,
everything culled.  Same bars as the main parity tests."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import read_internals, rel_err, run_forward_raw, scene_to_torch
from test_gpu_rasterizer import ABS_TOL, REL_TOL, _grads_gpu

pytestmark = pytest.mark.gpu


def _compare(oracle, sc, rs, ten, aa, bg, scale_modifier=1.0, sh_degree=None, backward=True):
    P, W, H = ten["means3D"].shape[0], rs.image_width, rs.image_height
    f = oracle.forward_from_scene(sc, bg=bg, antialiasing=aa, scale_modifier=scale_modifier, sh_degree=sh_degree)
    out = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    it = read_internals(out, P, W, H)
    assert out["num_rendered"] == f.R
    assert np.array_equal(out["radii"].cpu().numpy(), f.radii)
    assert np.array_equal(it["keys"], f.keys)
    assert np.array_equal(it["point_list"], f.point_list)
    assert np.array_equal(it["ranges"], f.ranges)
    assert np.array_equal(it["n_contrib"], f.n_contrib)
    assert np.array_equal(out["gauss_count"].cpu().numpy(), f.gauss_count)
    assert np.array_equal(out["last_contr_gauss"].cpu().numpy(), f.last_contr)
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    assert np.abs(out["invdepth"].cpu().numpy() - f.invdepth).max() <= ABS_TOL
    assert np.abs(out["out_reps"].cpu().numpy() - f.out_reps).max() <= ABS_TOL * 10
    if not backward:
        return f, out
    rng = np.random.default_rng(P + W)
    gc = rng.standard_normal((3, H, W)).astype(np.float32)
    gd = rng.standard_normal((1, H, W)).astype(np.float32)
    gr = rng.standard_normal((4, H, W)).astype(np.float32)
    ref = f.backward(gc, gd, gr)
    t = lambda a: torch.tensor(a, device="cuda:0")
    got = _grads_gpu(rs, ten, t(gc), t(gd), t(gr), {}, out)
    for k in got:
        r = ref[k].reshape(got[k].shape)
        assert np.isfinite(got[k]).all(), k
        assert rel_err(got[k], r) <= REL_TOL, (k, rel_err(got[k], r))
    return f, out


@pytest.mark.parametrize("mod", [0.5, 1.7])
def test_scale_modifier(oracle, mod):
    sc = synth.tiny_scene(P=300, width=96, height=64, seed=41)
    rs, ten = scene_to_torch(sc, bg=(0.2, 0.4, 0.6), antialiasing=True, scale_modifier=mod)
    _compare(oracle, sc, rs, ten, True, (0.2, 0.4, 0.6), scale_modifier=mod)


@pytest.mark.parametrize("degree", [0, 1, 2])
def test_active_degree_below_allocated(oracle, degree):
    """The model starts at degree 0 with 16 coefficients allocated (train.py:146-147): the inactive
    coefficients must neither colour the image nor receive gradient."""
    sc = synth.tiny_scene(P=300, width=80, height=48, seed=43)
    sc["sh_degree"] = degree
    rs, ten = scene_to_torch(sc, antialiasing=False)
    f, out = _compare(oracle, sc, rs, ten, False, (0.0, 0.0, 0.0), sh_degree=degree)
    H, W = rs.image_height, rs.image_width
    gc = torch.ones((3, H, W), device="cuda:0")
    gr = torch.zeros((4, H, W), device="cuda:0")
    got = _grads_gpu(rs, ten, gc, None, gr, {}, out)
    n = (degree + 1) ** 2
    assert np.all(got["dL_dsh"][:, n:] == 0)


@pytest.mark.parametrize("W,H", [(7, 5), (16, 16), (17, 33), (250, 3)])
def test_small_and_ragged_images(oracle, W, H):
    sc = synth.tiny_scene(P=120, width=W, height=H, seed=47, scale=0.3)
    rs, ten = scene_to_torch(sc, bg=(1.0, 0.0, 0.5), antialiasing=True)
    _compare(oracle, sc, rs, ten, True, (1.0, 0.0, 0.5))


def test_single_gaussian_and_full_screen_splat(oracle):
    sc = synth.tiny_scene(P=1, width=64, height=48, seed=51)
    rs, ten = scene_to_torch(sc, antialiasing=False)
    _compare(oracle, sc, rs, ten, False, (0.0, 0.0, 0.0))
    # one enormous, nearly opaque splat in front of many small ones: touches every tile
    sc = synth.tiny_scene(P=200, width=160, height=96, seed=53)
    sc["scales"][0] = 50.0
    sc["opacities"].reshape(-1)[0] = 0.99
    sc["means3D"][0] = sc["means3D"].mean(axis=0)
    rs, ten = scene_to_torch(sc, antialiasing=True)
    f, out = _compare(oracle, sc, rs, ten, True, (0.0, 0.0, 0.0))
    tiles = ((160 + 15) // 16) * ((96 + 15) // 16)
    it = read_internals(out, 200, 160, 96)
    assert it["bin"][0, 3] == tiles or f.radii[0] == 0


def test_everything_culled(oracle):
    sc = synth.tiny_scene(P=64, width=48, height=32, seed=57)
    sc["means3D"][:, 2] = -np.abs(sc["means3D"][:, 2]) - 50.0     # behind the camera wherever it looks from
    cam = sc["camera"]
    view = np.asarray(cam["viewmatrix"], np.float32)
    # move the points to the far side along the camera's viewing axis
    z_axis = view[:3, 2]
    sc["means3D"] = (np.asarray(cam["campos"], np.float32)[None] - 10.0 * z_axis[None] +
                     0.1 * sc["means3D"]).astype(np.float32)
    rs, ten = scene_to_torch(sc, bg=(0.3, 0.3, 0.3), antialiasing=False)
    f = oracle.forward_from_scene(sc, bg=(0.3, 0.3, 0.3))
    out = run_forward_raw(rs, ten)
    assert f.R == 0 and out["num_rendered"] == 0
    assert np.array_equal(out["radii"].cpu().numpy(), f.radii)
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    assert (out["last_contr_gauss"].cpu().numpy() == -1).all()
    H, W = rs.image_height, rs.image_width
    got = _grads_gpu(rs, ten, torch.ones((3, H, W), device="cuda:0"), None, torch.ones((4, H, W), device="cuda:0"),
                     {}, out)
    for k, v in got.items():
        assert np.all(v == 0), k


def test_hidden_gaussians_get_exactly_zero_gradients(oracle):
    """A wall of opaque splats saturates every pixel before the Gaussians behind it are reached:
    those are rendered (radius > 0, instances in the tile lists) but no pixel ever gets to them,
    so the per-Gaussian backward treats them like culled ones.  Their gradients must be exactly
    zero, as the checker's are, and everybody else's must still agree."""
    sc = synth.tiny_scene(P=400, width=96, height=64, seed=61)
    cam = sc["camera"]
    view = np.asarray(cam["viewmatrix"], np.float32)   # row-vector convention: p_view = p @ view[:3,:3] + view[3,:3]
    campos = np.asarray(cam["campos"], np.float32)
    z_axis = view[:3, 2]
    centre = sc["means3D"].mean(axis=0)
    depth = float((centre - campos) @ z_axis)
    # the first 40 Gaussians become a wall half way between the camera and the scene
    rng = np.random.default_rng(5)
    x_axis, y_axis = view[:3, 0], view[:3, 1]
    uv = rng.uniform(-1.0, 1.0, (40, 2)).astype(np.float32) * 0.35 * depth
    sc["means3D"][:40] = campos[None] + 0.5 * depth * z_axis[None] + uv[:, :1] * x_axis[None] + uv[:, 1:] * y_axis[None]
    sc["scales"][:40] = 0.5 * depth
    sc["opacities"].reshape(-1)[:40] = 0.99
    rs, ten = scene_to_torch(sc, antialiasing=True)
    f, out = _compare(oracle, sc, rs, ten, True, (0.0, 0.0, 0.0))
    count = out["gauss_count"].cpu().numpy()
    radii = out["radii"].cpu().numpy()
    hidden = (radii > 0) & (count == 0)
    assert hidden.sum() > 20, "the scene no longer hides anything: the test is vacuous"
    H, W = rs.image_height, rs.image_width
    rng = np.random.default_rng(7)
    t = lambda a: torch.tensor(a, device="cuda:0")
    got = _grads_gpu(rs, ten, t(rng.standard_normal((3, H, W)).astype(np.float32)),
                     t(rng.standard_normal((1, H, W)).astype(np.float32)),
                     t(rng.standard_normal((4, H, W)).astype(np.float32)), {}, out)
    for k, v in got.items():
        if v.shape[0] == hidden.shape[0]:
            assert np.all(v[hidden] == 0), k


def test_tile_grid_too_large_for_the_rectangle_count_path(oracle):
    """Tile ranges and the sort's histograms normally come from the tile rectangles, in the LDS of
    one workgroup; a 640 x 64 tile grid does not fit there, so this image takes the other path
    (histogram sweep over the keys, range scan over the sorted keys).  Same bars."""
    sc = synth.tiny_scene(P=300, width=10240, height=1024, seed=71, scale=0.05)
    rs, ten = scene_to_torch(sc, antialiasing=True)
    f, out = _compare(oracle, sc, rs, ten, True, (0.0, 0.0, 0.0))
    assert f.R > 100000


@pytest.mark.parametrize("W,H", [(400, 300), (800, 600), (2560, 1440), (3840, 2208)])
def test_every_digit_split_of_the_fused_emission(oracle, W, H):
    """Tile ids of 9, 11, 14 and 16 bits: the emission partitions on 4, 5, 7 and 8 high bits and the
    segmented pass orders 5, 6, 7 and 8 low bits (13 and 15 bits are the 1080p and 4K cases of the
    full-size tests).  Keys, lists and ranges bit for bit as always."""
    sc = synth.tiny_scene(P=3000, width=W, height=H, seed=81, scale=0.12)
    rs, ten = scene_to_torch(sc, antialiasing=False)
    f, out = _compare(oracle, sc, rs, ten, False, (0.0, 0.0, 0.0), backward=False)
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    assert f.R > 4 * 4096 and (f.ranges[:, 1] > f.ranges[:, 0]).sum() > ntiles // 4


def test_many_one_tile_gaussians_per_emission_chunk(oracle):
    """Splats of a pixel or two: fewer than two tiles per Gaussian, so one emission chunk of 4096
    instances takes its Gaussians from ten and more blocks of 256 (a chunk usually spans two or three)."""
    sc = synth.tiny_scene(P=60000, width=800, height=600, seed=83, scale=0.002)
    rs, ten = scene_to_torch(sc, antialiasing=False)
    f, out = _compare(oracle, sc, rs, ten, False, (0.0, 0.0, 0.0))
    assert f.R < 2 * (f.radii > 0).sum()


@pytest.mark.parametrize("which", [1, 2], ids=["instance_sort", "depth_order"])
def test_sort_error_word_is_surfaced_without_debug(which):
    """A look-back that trips its spin limit leaves an error word on the device: the instance sort's in the
    binning buffer, the depth order's in the geometry buffer.  With debug the forward call waits and fails;
    without it the call returns (it must not wait), and the next call into the rasterizer after the GPU has
    passed that point raises.  The error is injected through the test hook, into either word."""
    import synth
    from _gsplat_lib import lib, sort_errors
    from gpu_util import run_forward_raw, scene_to_torch
    L = lib()
    sc = synth.tiny_scene(P=300, width=64, height=48, seed=3)
    sort_errors.poll(block=True)
    try:
        L.gsr_test_inject_sort_error(which)
        rs, ten = scene_to_torch(sc, debug=True)
        with pytest.raises(RuntimeError, match="look-back"):
            run_forward_raw(rs, ten)
        sort_errors.poll(block=True)          # the debug call reported it itself: nothing is left pending
    finally:
        L.gsr_test_inject_sort_error(0)
    try:
        L.gsr_test_inject_sort_error(which)
        rs, ten = scene_to_torch(sc, debug=False)
        out = run_forward_raw(rs, ten)        # returns: the host does not wait for the word
        assert out["num_rendered"] > 0
    finally:
        L.gsr_test_inject_sort_error(0)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="look-back"):
        run_forward_raw(rs, ten)              # the next call finds the word of the one before
    out = run_forward_raw(rs, ten)            # and the state is clean again
    torch.cuda.synchronize()
    sort_errors.poll(block=True)


def test_depth_order_error_of_the_batch_call_reaches_the_views_host_word():
    """gsr_preprocess_batch makes the front-to-back orders of all the views of a step; a tripped look-back
    there sets the error word in that view's geometry buffer, and the view's gsr_forward_stage2 hands it to
    the host word together with its own sort's.  The word is set by hand behind the batch call."""
    import ctypes as C
    import synth
    from _gsplat_lib import lib, sort_errors
    from diff_gaussian_rasterization import _C as host
    from test_gpu_params_mode import _arc_settings, _raw_scene
    L = lib()
    W, H, P = 96, 64, 400
    sc = synth.tiny_scene(P=P, width=W, height=H, seed=31)
    raw = _raw_scene(sc, seed=5)
    settings = _arc_settings(2, W, H, debug=False)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda:0")
    ten = dict(xyz=t(sc["means3D"]), dc=t(raw["features_dc"]), rest=t(raw["features_rest"]), opacity=t(raw["opacity"]),
               scaling=t(raw["scaling"]), rotation=t(raw["rotation"]), reps=t(sc["reps"]))
    e = torch.Tensor([])
    sort_errors.poll(block=True)
    prep = host.preprocess_batch(ten["xyz"], ten["opacity"], ten["scaling"], ten["rotation"], 1.0, ten["dc"], ten["rest"],
                                 3, ten["reps"], True, settings, debug=False)
    assert prep[1]["depth_ordered"] and prep[1]["num_rendered"] > 0
    # where the depth order's error word lives: the sort area is the last array of the geometry buffer
    geom = prep[1]["geom"]
    from _gsplat_lib import GsrLayout
    lay = GsrLayout()
    assert L.gsr_layout(P, W, H, prep[1]["num_rendered"], C.byref(lay)) == 0
    off = lay.depth_sort + L.gsr_sort_temp_error_offset()
    assert int(geom[off:off + 4].view(torch.int32).item()) == 0
    geom[off:off + 4].view(torch.int32).fill_(1)

    def fwd(rs, pr):
        return host.rasterize_gaussians(
            rs.bg, ten["xyz"], e, ten["opacity"], ten["scaling"], ten["rotation"], rs.scale_modifier, e, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, ten["dc"], rs.sh_degree, rs.campos,
            False, rs.antialiasing, rs.debug, ten["reps"], sh_rest=ten["rest"], raw_params=True, prepared=pr)

    fwd(settings[0], prep[0])                  # the other view is clean
    torch.cuda.synchronize()
    sort_errors.poll(block=True)
    fwd(settings[1], prep[1])                  # returns; the word arrives behind it
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="look-back"):
        sort_errors.poll(block=True)
    sort_errors.poll(block=True)


def test_prefiltered_violation_is_reported():
    """`prefiltered=True` promises that no point fails the near-plane test.  Where the reference prints and
    traps on the device (cuda_rasterizer/auxiliary.h:181-185), the forward call raises; a scene that keeps
    the promise renders exactly as without the flag.  Both host layers."""
    import synth
    from diff_gaussian_rasterization import _C
    from gpu_util import run_forward_raw, scene_to_torch
    sc = synth.tiny_scene(P=300, width=64, height=48, seed=3)
    sc["means3D"][::15, 2] = -10.0                                 # twenty Gaussians behind the camera
    rs, ten = scene_to_torch(sc, debug=False)
    ref = run_forward_raw(rs, ten)
    assert int((ref["radii"] == 0).sum()) > 0
    pv = np.concatenate([sc["means3D"], np.ones((300, 1), np.float32)], 1) @ sc["camera"]["viewmatrix"]
    front = pv[:, 2] > 0.2
    assert 0 < front.sum() < 300
    ext = _C._hip_ext
    for layer in ("torch extension", "ctypes"):
        try:
            if layer == "ctypes":
                _C._hip_ext = None
            assert _C.host_layer() == layer
            with pytest.raises(RuntimeError, match="prefiltered"):
                run_forward_raw(rs._replace(prefiltered=True), ten)
            kept = {k: v[torch.tensor(front, device=v.device)].contiguous() for k, v in ten.items()}
            a = run_forward_raw(rs._replace(prefiltered=True), kept)
            b = run_forward_raw(rs, kept)
            torch.cuda.synchronize()
            assert a["num_rendered"] == b["num_rendered"] and torch.equal(a["color"], b["color"])
        finally:
            _C._hip_ext = ext


def test_sort_error_offset_is_inside_the_temp_area():
    from _gsplat_lib import lib
    L = lib()
    assert 0 < L.gsr_sort_temp_error_offset() < L.gsr_sort_temp_bytes(1)


def test_rasterizer_with_64_bit_sort_status_words(oracle):
    """More than 2^30 instances (the reference allows 2^31 - 1, rasterizer_impl.cu:338) take the plain emission
    and radix passes with 64-bit look-back words instead of the fused emission; the test hook selects that path
    on a scene the checker can render."""
    from _gsplat_lib import lib
    L = lib()
    sc = synth.tiny_scene(P=300, width=75, height=45, seed=9, scale=0.1)
    try:
        L.gsr_test_force_wide_sort_status(1)
        rs, ten = scene_to_torch(sc, antialiasing=True)
        _compare(oracle, sc, rs, ten, True, (0, 0, 0))
    finally:
        L.gsr_test_force_wide_sort_status(0)


@pytest.mark.parametrize("extra", [0, 1], ids=["2^25_gaussians_packed", "2^25+1_gaussians_as_pairs"])
def test_instance_packing_at_its_limit(oracle, extra):
    """Between the key emission and the segmented pass an instance is one word, (low tile digit << 25) |
    Gaussian index at 1080p, as long as every index fits below the digit: 2^25 Gaussians is the last model
    size that packs, one more takes (tile id, index) pairs.  The Gaussians that reach the screen have the
    highest indices (all index bits in use); the rest of the model sits at the camera position and is
    culled.  Lists, ranges and images against the checker at both sizes."""
    W, H, nvis = 1920, 1080, 4000
    P = (1 << 25) + extra
    vis = synth.street_scene(nvis, W, H, seed=5, sh_coeffs=1)
    cam = vis["camera"]

    def grown(a, fill):
        out = np.empty((P,) + a.shape[1:], np.float32)
        out[:P - nvis] = np.asarray(fill, np.float32)
        out[P - nvis:] = a
        return out

    sc = dict(means3D=grown(vis["means3D"], np.asarray(cam["campos"], np.float32)),   # view depth 0: culled
              scales=grown(vis["scales"], 0.01), rotations=grown(vis["rotations"], (1.0, 0.0, 0.0, 0.0)),
              opacities=grown(vis["opacities"], 0.5), shs=grown(vis["shs"], 0.0), reps=grown(vis["reps"], 0.0),
              camera=cam, sh_degree=0)
    bg = (0.2, 0.1, 0.3)
    f = oracle.forward_from_scene(sc, bg=bg, antialiasing=True)
    assert f.R > 0 and int((f.radii > 0).sum()) > nvis // 4 and f.point_list.min() >= P - nvis
    rs, ten = scene_to_torch(sc, bg=bg, antialiasing=True, debug=False)
    out = run_forward_raw(rs, ten)
    torch.cuda.synchronize()
    assert out["num_rendered"] == f.R
    assert np.array_equal(out["radii"].cpu().numpy(), f.radii)
    from _gsplat_lib import GsrLayout, lib
    import ctypes
    lay = GsrLayout()
    assert lib().gsr_layout(P, W, H, f.R, ctypes.byref(lay)) == 0
    v = (lay.vals0, lay.vals1)[lay.sorted_buffer]
    point_list = out["binning"][v:v + 4 * f.R].cpu().numpy().view(np.uint32)
    assert np.array_equal(point_list, f.point_list)
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    img = out["img"].cpu().numpy()
    assert np.array_equal(img[lay.ranges:lay.ranges + ntiles * 8].view(np.uint32).reshape(ntiles, 2), f.ranges)
    assert np.array_equal(img[lay.n_contrib:lay.n_contrib + W * H * 4].view(np.uint32), f.n_contrib)
    keys = torch.empty(f.R, dtype=torch.int64, device="cuda:0")
    assert lib().gsr_sorted_keys_u64(P, W, H, f.R, out["geom"].data_ptr(), out["binning"].data_ptr(),
                                     out["img"].data_ptr(), keys.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(keys.cpu().numpy().view(np.uint64), f.keys)
    # which path ran: only the pair path stores the sorted tile ids beside the index list
    k = (lay.keys0, lay.keys1)[lay.sorted_buffer]
    stored = out["binning"][k:k + 4 * f.R].cpu().numpy().view(np.uint32)
    assert np.array_equal(stored, (f.keys >> np.uint64(32)).astype(np.uint32)) == (extra == 1)
    assert np.array_equal(out["gauss_count"].cpu().numpy(), f.gauss_count)
    assert np.array_equal(out["last_contr_gauss"].cpu().numpy(), f.last_contr)
    assert np.abs(out["color"].cpu().numpy() - f.out_color).max() <= ABS_TOL
    assert np.abs(out["out_reps"].cpu().numpy() - f.out_reps).max() <= ABS_TOL * 10
