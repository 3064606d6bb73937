"""TEST INFRASTRUCTURE: numpy front-end of the CPU checker (libgs_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_bp = C.POINTER(C.c_uint8)


class GsoScene(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("P", "D", "M", "DIM", "W", "H")] + \
               [(n, _fp) for n in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                   "cov3D_precomp", "view", "proj", "campos", "reps")] + \
               [(n, C.c_float) for n in ("scale_modifier", "tanfovx", "tanfovy")] + \
               [(n, C.c_int) for n in ("prefiltered", "antialiasing")]


class GsoGeom(C.Structure):
    _fields_ = [("radii", _ip), ("means2D", _fp), ("depths", _fp), ("cov3D", _fp), ("conic_opacity", _fp),
                ("rgb", _fp), ("clamped", _bp), ("tiles_touched", _up), ("point_offsets", _up), ("out_depths", _fp)]


class GsoBin(C.Structure):
    _fields_ = [("keys_unsorted", _u64p), ("vals_unsorted", _up), ("keys", _u64p), ("point_list", _up)]


class GsoImage(C.Structure):
    _fields_ = [("ranges", _up), ("final_T", _fp), ("n_contrib", _up), ("out_color", _fp), ("invdepth", _fp),
                ("accum_alpha", _fp), ("out_reps", _fp), ("last_contr", _ip), ("gauss_sum", _fp), ("gauss_count", _ip)]


class GsoGrads(C.Structure):
    _fields_ = [(n, _fp) for n in ("dL_dpix", "dL_dinvdepth", "dL_dreps_pix", "dL_dmean2D", "dL_dconic",
                                   "dL_dopacity", "dL_dcolor", "dL_dinvdepths", "dL_dmean3D", "dL_dcov3D", "dL_dsh",
                                   "dL_dscale", "dL_drot", "dL_dreps")]


def build(force=False):
    """Compile the checker in place (make, g++)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "gs_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_forward_geom.restype = C.c_int64
        _lib.gso_forward_geom.argtypes = [C.POINTER(GsoScene), C.POINTER(GsoGeom)]
        _lib.gso_forward_render.restype = C.c_int
        _lib.gso_forward_render.argtypes = [C.POINTER(GsoScene), C.POINTER(GsoGeom), C.POINTER(GsoBin),
                                            C.POINTER(GsoImage), C.c_int64]
        _lib.gso_backward.restype = C.c_int
        _lib.gso_backward.argtypes = [C.POINTER(GsoScene), C.POINTER(GsoGeom), C.POINTER(GsoBin), C.POINTER(GsoImage),
                                      C.c_int64, C.POINTER(GsoGrads)]
        _lib.gso_mark_visible.argtypes = [C.c_int, _fp, _fp, _fp, _bp]
        _lib.gso_sort_pairs.argtypes = [_u64p, _up, _u64p, _up, C.c_int64, C.c_int]
        _lib.gso_expf.argtypes = [_fp, _fp, C.c_int64]
        _lib.gso_sh_eval.argtypes = [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _bp]
        _lib.gso_activate.argtypes = [C.c_int] + [_fp] * 6
        _lib.gso_activate_bwd.argtypes = [C.c_int] + [_fp] * 6
        _lib.gso_adam_update.argtypes = [_fp, _fp, _fp, _fp, _bp, C.c_int64, C.c_int] + [C.c_double] * 6
        _lib.gso_dist2.argtypes = [C.c_int, _fp, _fp]
        _lib.gso_dist2_strided.argtypes = [C.c_int, _fp, _fp, C.c_int]
        _lib.gso_ssim_fwd.argtypes = [C.c_int] * 4 + [C.c_float] * 2 + [_fp] * 6
        _lib.gso_ssim_bwd.argtypes = [C.c_int] * 4 + [_fp] * 7
        _lib.gso_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _c(a, dtype=np.float32):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else None


class Forward:
    """Holds inputs, every intermediate and every output of one forward pass."""

    def __init__(self, means3D, opacities, camera, bg, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, reps=None, sh_degree=0, scale_modifier=1.0, antialiasing=False,
                 want_invdepth=True):
        self.keep = k = {}
        k["means3D"] = _c(means3D)
        P = self.P = 0 if k["means3D"] is None else k["means3D"].shape[0]
        k["opacities"] = _c(opacities)
        k["shs"] = _c(shs)
        k["colors_precomp"] = _c(colors_precomp)
        k["scales"] = _c(scales)
        k["rotations"] = _c(rotations)
        k["cov3D_precomp"] = _c(cov3D_precomp)
        k["reps"] = _c(reps)
        k["bg"] = _c(bg)
        k["view"] = _c(camera["viewmatrix"])
        k["proj"] = _c(camera["projmatrix"])
        k["campos"] = _c(camera["campos"])
        self.W, self.H = int(camera["image_width"]), int(camera["image_height"])
        self.M = 0 if k["shs"] is None else k["shs"].shape[1]
        self.DIM = 0 if k["reps"] is None else k["reps"].shape[1]
        W, H, DIM = self.W, self.H, self.DIM
        s = self.scene = GsoScene()
        s.P, s.D, s.M, s.DIM, s.W, s.H = P, int(sh_degree), self.M, DIM, W, H
        for n in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp",
                  "view", "proj", "campos", "reps"):
            setattr(s, n, _f(k[n]))
        s.scale_modifier, s.tanfovx, s.tanfovy = float(scale_modifier), float(camera["tanfovx"]), float(camera["tanfovy"])
        s.prefiltered, s.antialiasing = 0, int(bool(antialiasing))
        self.grid = ((W + 15) // 16, (H + 15) // 16)
        ntiles = self.grid[0] * self.grid[1]

        z = np.zeros
        self.radii = z(P, np.int32)
        self.means2D = z((P, 2), np.float32)
        self.depths = z(P, np.float32)
        self.cov3D = z((P, 6), np.float32)
        self.conic_opacity = z((P, 4), np.float32)
        self.rgb = z((P, 3), np.float32)
        self.clamped = z((P, 3), np.uint8)
        self.tiles_touched = z(P, np.uint32)
        self.point_offsets = z(P, np.uint32)
        self.out_depths = z(P, np.float32)
        g = self.geom = GsoGeom()
        g.radii = self.radii.ctypes.data_as(_ip)
        g.means2D, g.depths, g.cov3D = _f(self.means2D), _f(self.depths), _f(self.cov3D)
        g.conic_opacity, g.rgb = _f(self.conic_opacity), _f(self.rgb)
        g.clamped = self.clamped.ctypes.data_as(_bp)
        g.tiles_touched = self.tiles_touched.ctypes.data_as(_up)
        g.point_offsets = self.point_offsets.ctypes.data_as(_up)
        g.out_depths = _f(self.out_depths)

        self.ranges = z((ntiles, 2), np.uint32)
        self.final_T = z(H * W, np.float32)
        self.n_contrib = z(H * W, np.uint32)
        self.out_color = z((3, H, W), np.float32)
        self.invdepth = z((1, H, W), np.float32) if want_invdepth else None
        self.accum_alpha = z((1, H, W), np.float32)
        self.out_reps = z((DIM, H, W), np.float32)
        self.last_contr = np.full((1, H, W), -1, np.int32)
        self.gauss_sum = z(P, np.float32)
        self.gauss_count = z(P, np.int32)
        im = self.image = GsoImage()
        im.ranges = self.ranges.ctypes.data_as(_up)
        im.final_T = _f(self.final_T)
        im.n_contrib = self.n_contrib.ctypes.data_as(_up)
        im.out_color, im.invdepth, im.accum_alpha = _f(self.out_color), _f(self.invdepth), _f(self.accum_alpha)
        im.out_reps = _f(self.out_reps) if DIM else None
        im.last_contr = self.last_contr.ctypes.data_as(_ip)
        im.gauss_sum = _f(self.gauss_sum)
        im.gauss_count = self.gauss_count.ctypes.data_as(_ip)
        self.R = 0
        self.bin = GsoBin()

    def run(self):
        L = lib()
        if self.P == 0:
            return self
        self.R = R = int(L.gso_forward_geom(C.byref(self.scene), C.byref(self.geom)))
        n = max(R, 1)
        self.keys_unsorted = np.zeros(n, np.uint64)
        self.vals_unsorted = np.zeros(n, np.uint32)
        self.keys = np.zeros(n, np.uint64)
        self.point_list = np.zeros(n, np.uint32)
        b = self.bin
        b.keys_unsorted = self.keys_unsorted.ctypes.data_as(_u64p)
        b.vals_unsorted = self.vals_unsorted.ctypes.data_as(_up)
        b.keys = self.keys.ctypes.data_as(_u64p)
        b.point_list = self.point_list.ctypes.data_as(_up)
        rc = L.gso_forward_render(C.byref(self.scene), C.byref(self.geom), C.byref(b), C.byref(self.image), R)
        assert rc == 0
        for n_ in ("keys_unsorted", "vals_unsorted", "keys", "point_list"):
            setattr(self, n_, getattr(self, n_)[:R])
        return self

    def backward(self, dL_dcolor, dL_dinvdepth=None, dL_dreps=None):
        """Returns a dict of every gradient the reference's backward produces."""
        P, M, DIM, H, W = self.P, self.M, self.DIM, self.H, self.W
        z = np.zeros
        keep = dict(dpix=_c(dL_dcolor), dinv=_c(dL_dinvdepth),
                    drep=_c(dL_dreps) if DIM else None)
        if DIM and keep["drep"] is None:
            keep["drep"] = z((DIM, H, W), np.float32)
        out = dict(dL_dmean2D=z((P, 3), np.float32), dL_dconic=z((P, 2, 2), np.float32),
                   dL_dopacity=z((P, 1), np.float32), dL_dcolor=z((P, 3), np.float32),
                   dL_dmean3D=z((P, 3), np.float32), dL_dcov3D=z((P, 6), np.float32),
                   dL_dsh=z((P, M, 3), np.float32), dL_dscale=z((P, 3), np.float32),
                   dL_drot=z((P, 4), np.float32), dL_dreps=z((P, DIM), np.float32))
        out["dL_dinvdepths"] = z((P, 1), np.float32) if keep["dinv"] is not None else None
        gr = GsoGrads()
        gr.dL_dpix, gr.dL_dinvdepth, gr.dL_dreps_pix = _f(keep["dpix"]), _f(keep["dinv"]), _f(keep["drep"])
        for n_ in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_dinvdepths", "dL_dmean3D", "dL_dcov3D",
                   "dL_dsh", "dL_dscale", "dL_drot", "dL_dreps"):
            a = out[n_]
            setattr(gr, n_, _f(a) if (a is not None and a.size) else None)
        if P:
            rc = lib().gso_backward(C.byref(self.scene), C.byref(self.geom), C.byref(self.bin), C.byref(self.image),
                                    self.R, C.byref(gr))
            assert rc == 0
        return out


def forward_from_scene(scene, bg=(0.0, 0.0, 0.0), antialiasing=False, use_cov_precomp=None, colors_precomp=None,
                       scale_modifier=1.0, sh_degree=None, want_invdepth=True, with_reps=True):
    """Convenience: run the checker on a dict from synth.*_scene()."""
    kw = dict(means3D=scene["means3D"], opacities=scene["opacities"], camera=scene["camera"],
              bg=np.asarray(bg, np.float32), scale_modifier=scale_modifier, antialiasing=antialiasing,
              sh_degree=scene["sh_degree"] if sh_degree is None else sh_degree, want_invdepth=want_invdepth)
    if with_reps:
        kw["reps"] = scene["reps"]
    if colors_precomp is not None:
        kw["colors_precomp"] = colors_precomp
    else:
        kw["shs"] = scene["shs"]
    if use_cov_precomp is not None:
        kw["cov3D_precomp"] = use_cov_precomp
    else:
        kw["scales"], kw["rotations"] = scene["scales"], scene["rotations"]
    return Forward(**kw).run()


def mark_visible(means3D, view, proj):
    m = _c(means3D)
    out = np.zeros(m.shape[0], np.uint8)
    lib().gso_mark_visible(m.shape[0], _f(m), _f(_c(view)), _f(_c(proj)), out.ctypes.data_as(_bp))
    return out.astype(bool)


def sort_pairs(keys, vals, end_bit=64):
    k = np.ascontiguousarray(keys, np.uint64)
    v = np.ascontiguousarray(vals, np.uint32)
    ko, vo = np.empty_like(k), np.empty_like(v)
    lib().gso_sort_pairs(k.ctypes.data_as(_u64p), v.ctypes.data_as(_up), ko.ctypes.data_as(_u64p),
                         vo.ctypes.data_as(_up), k.size, int(end_bit))
    return ko, vo


def expf(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().gso_expf(_f(x), _f(y), x.size)
    return y


def activate(opacity_logit, log_scaling, rotation_raw):
    """(opacity, scales, rotations) as the kernels' raw-parameter mode computes them."""
    lo, ls, rr = _c(opacity_logit).reshape(-1), _c(log_scaling), _c(rotation_raw)
    P = ls.shape[0]
    o, s, r = np.zeros(P, np.float32), np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32)
    lib().gso_activate(P, _f(lo), _f(ls), _f(rr), _f(o), _f(s), _f(r))
    return o, s, r


def activate_bwd(opacity_logit, log_scaling, rotation_raw, g_opacity, g_scales, g_rotations):
    """Chain rule through the activations: gradients w.r.t. the raw parameters."""
    lo, ls, rr = _c(opacity_logit).reshape(-1), _c(log_scaling), _c(rotation_raw)
    go, gs, gr = (np.array(a, np.float32, copy=True, order="C") for a in (g_opacity, g_scales, g_rotations))
    go = go.reshape(-1)
    lib().gso_activate_bwd(ls.shape[0], _f(lo), _f(ls), _f(rr), _f(go), _f(gs), _f(gr))
    return go, gs, gr


def adam_update(param, grad, exp_avg, exp_avg_sq, visible, lr, beta1, beta2, eps, step=0):
    """One Adam step on copies of the [N, M] arrays; step = 0: no bias correction
    (the sparse variant).  Returns (param, exp_avg, exp_avg_sq)."""
    p, m, v = (np.array(a, np.float32, copy=True, order="C") for a in (param, exp_avg, exp_avg_sq))
    g = _c(grad)
    N = p.shape[0]
    M = int(p.size // max(N, 1))
    bc1 = 1.0 - beta1 ** step if step > 0 else 1.0
    bc2s = float(np.sqrt(1.0 - beta2 ** step)) if step > 0 else 1.0
    vis = None if visible is None else np.ascontiguousarray(visible, np.uint8)
    lib().gso_adam_update(_f(p), _f(g), _f(m), _f(v), None if vis is None else vis.ctypes.data_as(_bp), N, M, lr, beta1,
                          beta2, eps, bc1, bc2s)
    return p, m, v


def densify_stats(radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D):
    """train.py:298-312 and scene/gaussian_model.py:202-214 restated in numpy (float32):
    returns the updated (xyz_gradient_accum, denom, max_radii2D)."""
    vis = np.nonzero(np.asarray(radii) > 0)[0]
    acc, den, mx = (np.array(a, np.float32, copy=True) for a in (xyz_gradient_accum, denom, max_radii2D))
    mx[vis] = np.maximum(mx[vis], np.asarray(radii)[vis].astype(np.float32))
    g = np.asarray(viewspace_grad, np.float32)[vis, :2]
    acc[vis] += np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1], dtype=np.float32).reshape(-1, 1)
    den[vis] += np.float32(1)
    return acc, den, mx


def sh_eval(deg, pos, campos, shs):
    pos, campos, shs = _c(pos), _c(campos), _c(shs)
    N, M = shs.shape[0], shs.shape[1]
    rgb = np.zeros((N, 3), np.float32)
    cl = np.zeros((N, 3), np.uint8)
    lib().gso_sh_eval(N, int(deg), M, _f(pos), _f(campos), _f(shs), _f(rgb), cl.ctypes.data_as(_bp))
    return rgb, cl


def dist2(points, stride=1):
    """stride > 1: only every stride-th point of the Morton order is searched, the rest are -1."""
    p = _c(points)
    out = np.zeros(p.shape[0], np.float32)
    lib().gso_dist2_strided(p.shape[0], _f(p), _f(out), int(stride))
    return out


def ssim_fwd(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2, train=True):
    a, b = _c(img1), _c(img2)
    B, CH, H, W = a.shape
    m = np.zeros_like(a)
    d = [np.zeros_like(a) for _ in range(3)] if train else [None] * 3
    lib().gso_ssim_fwd(B, CH, H, W, C1, C2, _f(a), _f(b), _f(m), _f(d[0]), _f(d[1]), _f(d[2]))
    return m, d[0], d[1], d[2]


def ssim_bwd(img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    a, b = _c(img1), _c(img2)
    B, CH, H, W = a.shape
    out = np.zeros_like(a)
    lib().gso_ssim_bwd(B, CH, H, W, _f(a), _f(b), _f(_c(dL_dmap)), _f(_c(dm_dmu1)), _f(_c(dm_dsigma1_sq)),
                       _f(_c(dm_dsigma12)), _f(out))
    return out


def num_threads():
    return lib().gso_num_threads()
