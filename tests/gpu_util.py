"""Helpers for the GPU parity tests: run a synthetic scene through the
operator package (which goes through the C ABI of libgsplat_amd.so) and read
the scratch buffers back for bit-exact comparison with the CPU checker."""
import ctypes as C
import math

import numpy as np
import torch


def scene_to_torch(sc, device="cuda:0", bg=(0.0, 0.0, 0.0), antialiasing=False, debug=True, scale_modifier=1.0):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    cam = sc["camera"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=device)
    rs = GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
        tanfovy=cam["tanfovy"], bg=t(bg), scale_modifier=scale_modifier, viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), sh_degree=sc["sh_degree"], campos=t(cam["campos"]), prefiltered=False,
        debug=debug, antialiasing=antialiasing)
    ten = {k: t(sc[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs", "reps")}
    return rs, ten


def run_forward_raw(rs, ten, colors_precomp=None, cov3D_precomp=None, with_reps=True):
    """Calls the pybind-level mirror and returns its 13-tuple as a dict."""
    from diff_gaussian_rasterization import _C
    e = torch.Tensor([])
    out = _C.rasterize_gaussians(
        rs.bg, ten["means3D"], e if colors_precomp is None else colors_precomp, ten["opacities"],
        ten["scales"] if cov3D_precomp is None else e, ten["rotations"] if cov3D_precomp is None else e,
        rs.scale_modifier, e if cov3D_precomp is None else cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
        rs.tanfovy, rs.image_height, rs.image_width, ten["shs"] if colors_precomp is None else e, rs.sh_degree,
        rs.campos, rs.prefiltered, rs.antialiasing, rs.debug, ten["reps"] if with_reps else e)
    names = ("num_rendered", "color", "radii", "geom", "binning", "img", "invdepth", "accum_alpha", "gauss_sum",
             "gauss_count", "last_contr_gauss", "out_depths", "out_reps")
    return dict(zip(names, out))


def read_internals(out, P, W, H, packed_pairs=None):
    """Views into the scratch buffers (numpy copies).  packed_pairs=False: the forward ran with
    gsr_test_packed_instances(0), so the sorted tile ids are in the binning buffer and are compared too."""
    from _gsplat_lib import GsrLayout, lib
    R = out["num_rendered"]
    lay = GsrLayout()
    assert lib().gsr_layout(P, W, H, R, C.byref(lay)) == 0
    geom, img, binning = out["geom"].cpu().numpy(), out["img"].cpu().numpy(), out["binning"].cpu().numpy()
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    res = {}
    res["rec"] = geom[lay.rec:lay.rec + P * 64].view(np.float32).reshape(P, 16)
    res["cov3D"] = geom[lay.cov3D:lay.cov3D + P * 24].view(np.float32).reshape(P, 6)
    # {x0 | y0<<16, w | h<<16} of every Gaussian, followed by {depth bits, tiles} of every Gaussian
    res["bin"] = np.hstack([geom[lay.tiles:lay.tiles + P * 8].view(np.uint32).reshape(P, 2),
                            geom[lay.tiles + P * 8:lay.tiles + P * 16].view(np.uint32).reshape(P, 2)])
    res["clamped"] = geom[lay.clamped:lay.clamped + P]
    res["final_T"] = img[lay.final_T:lay.final_T + W * H * 4].view(np.float32)
    res["n_contrib"] = img[lay.n_contrib:lay.n_contrib + W * H * 4].view(np.uint32)
    res["ranges"] = img[lay.ranges:lay.ranges + ntiles * 8].view(np.uint32).reshape(ntiles, 2)
    if R > 0:
        v = (lay.vals0, lay.vals1)[lay.sorted_buffer]
        res["point_list"] = binning[v:v + R * 4].view(np.uint32)
        # the reference's 64-bit keys (tile << 32 | depth bits), rebuilt by the library
        import torch
        keys = torch.empty(R, dtype=torch.int64, device=out["geom"].device)
        assert lib().gsr_sorted_keys_u64(P, W, H, R, out["geom"].data_ptr(), out["binning"].data_ptr(),
                                         out["img"].data_ptr(), keys.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        res["keys"] = keys.cpu().numpy().view(np.uint64)
        if packed_pairs is False:   # the sorted tile ids are stored when the instances were sorted as pairs
            k = (lay.keys0, lay.keys1)[lay.sorted_buffer]
            assert np.array_equal((res["keys"] >> np.uint64(32)).astype(np.uint32), binning[k:k + R * 4].view(np.uint32))
    else:
        res["keys"] = np.zeros(0, np.uint64)
        res["point_list"] = np.zeros(0, np.uint32)
    return res


def rel_err(got, ref, floor=1e-2):
    """Element-wise error measure for sums whose order is not defined (float atomics): the largest
    |got - ref| / (|ref| + floor * max|ref|).  rel_err <= 1e-4 says that EVERY element obeys
    |got - ref| <= 1e-4 |ref| + 1e-6 max|ref|: an element far below the tensor's maximum cannot be
    wrong by more than a millionth of that maximum plus 1e-4 of itself."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    top = float(np.abs(ref).max())
    return float((np.abs(got - ref) / (np.abs(ref) + floor * top + 1e-300)).max())


def worst(got, ref, floor=1e-2):
    """Where rel_err is attained: (flat index, got, ref, max|ref|), for assertion messages."""
    got, ref = np.asarray(got, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    top = float(np.abs(ref).max())
    i = int(np.argmax(np.abs(got - ref) / (np.abs(ref) + floor * top + 1e-300)))
    return i, float(got[i]), float(ref[i]), top
