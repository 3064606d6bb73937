"""Host side of the rasterizer operator: the three functions the reference's
pybind module exports (submodules/diff-gaussian-rasterization/ext.cpp:15-19,
rasterize_points.cu:35-365), with the same argument order and return tuples,
implemented over the C ABI of libgsplat_amd.so.
"""
import ctypes as C

import torch

from _gsplat_lib import (GsrBackwardIO, GsrBackwardView, GsrForwardOut, GsrForwardView, GsrScene, check, dev_f32, lib,
                         profile_events, ptr, sort_errors, stream_of)

SUPPORTED_DIMS = (0, 4, 8, 16)   # feature channels the blend kernels are instantiated for (the reference: one, at compile time)

# The compiled PyTorch-ROCm host extension (csrc/torch/ext.cpp): the reference's three pybind functions
# over the C ABI.  The calls below with the reference's own arguments go through it; the ctypes
# code in this file serves them only if the extension has not been built, and always serves the
# arguments the reference does not have (split SH tensors, raw parameters, prepared views, gradient
# buffers to write into).
try:
    import os as _os
    if _os.environ.get("GSPLAT_AMD_LIB"):
        # a library override (A/B runs of kernel variants): the extension is linked against the in-tree
        # library, so everything goes through ctypes and the one library that was asked for
        raise ImportError("library override")
    from . import _hip_ext
except ImportError:
    _hip_ext = None


def host_layer():
    """Which host layer serves the reference-signature calls: 'torch extension' or 'ctypes'."""
    return "torch extension" if _hip_ext is not None else "ctypes"


def _events_address():
    ev = profile_events()
    return C.addressof(ev) if ev is not None else 0


def _scene(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
           tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, antialiasing, debug, reps, keep,
           sh_rest=None, raw=False):
    dev = means3D.device
    P = means3D.size(0)
    s = GsrScene()
    t = dict(bg=bg, means3D=means3D, shs=sh, colors_precomp=colors, opacities=opacity, scales=scales,
             rotations=rotations, cov3D_precomp=cov3D_precomp, reps=reps, viewmatrix=viewmatrix,
             projmatrix=projmatrix, campos=campos, shs_rest=sh_rest)
    for name, ten in t.items():
        c = dev_f32(ten, dev, name)
        keep.append(c)
        setattr(s, name, ptr(c))
    s.P, s.D, s.W, s.H = P, int(degree), int(W), int(H)
    s.M = int(sh.size(1)) if (sh is not None and sh.numel() != 0) else 0
    if sh_rest is not None and sh_rest.numel() != 0:
        s.M += int(sh_rest.size(1))
    s.raw_params = int(bool(raw))
    s.DIM = int(reps.size(1)) if (reps is not None and reps.numel() != 0) else 0
    s.scale_modifier, s.tanfovx, s.tanfovy = float(scale_modifier), float(tan_fovx), float(tan_fovy)
    s.prefiltered, s.antialiasing, s.debug = int(bool(prefiltered)), int(bool(antialiasing)), int(bool(debug))
    ev = profile_events()
    if ev is not None:
        s.profile_events = C.cast(ev, C.POINTER(C.c_void_p))
    return s


def _check_shapes(means3D, reps):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if reps is not None and reps.numel() != 0:
        if reps.dim() != 2 or reps.size(0) != means3D.size(0) or reps.size(1) not in SUPPORTED_DIMS:
            raise RuntimeError("reps must have dimensions (num_points, NUM_DIM) with NUM_DIM in %s" %
                               (SUPPORTED_DIMS,))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, antialiasing, debug, reps=None, sh_rest=None, raw_params=False, prepared=None):
    """rasterize_points.cu:35-206.  Returns (num_rendered, color, radii,
    geomBuffer, binningBuffer, imgBuffer, invdepth, accum_alpha, gauss_sum,
    gauss_count, last_contr_gauss, out_depths, out_reps).

    Beyond the reference: `sh_rest` (then `sh` is the [P,1,3] DC tensor and
    `sh_rest` the [P,M-1,3] remainder, uncatenated) and `raw_params` (opacity
    logits, log-scales and unnormalised rotations; the kernels apply the
    activations of scene/gaussian_model.py:41-53); `prepared` = this view's
    entry of what `preprocess_batch` returned for the same parameters (the
    preprocess kernel is then skipped)."""
    _check_shapes(means3D, reps)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor (there is no CPU path)")
    if _hip_ext is not None and sh_rest is None and not raw_params and prepared is None:
        addr, slot = sort_errors.take()
        try:
            res = _hip_ext.rasterize_gaussians(
                background, means3D, colors, opacity, scales, rotations, float(scale_modifier), cov3D_precomp,
                viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), int(image_height), int(image_width), sh,
                int(degree), campos, bool(prefiltered), bool(antialiasing), bool(debug),
                reps if reps is not None else torch.Tensor([]), _events_address(), addr)
        except Exception:
            sort_errors.release(slot)
            raise
        sort_errors.submitted(slot)
        return res
    L = lib()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    keep = []
    with torch.cuda.device(dev):
        s = _scene(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                   projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, prefiltered, antialiasing, debug, reps,
                   keep, sh_rest=sh_rest, raw=raw_params)
        DIM = s.DIM
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        out_color = torch.empty((3, H, W), **f32)
        out_invdepth = torch.empty((1, H, W), **f32)
        radii = prepared["radii"] if prepared else torch.empty((P,), **i32)
        accum_alpha = torch.empty((1, H, W), **f32)
        gauss_sum = torch.empty((P,), **f32)
        gauss_count = torch.empty((P,), **i32)
        last_contr = torch.empty((1, H, W), **i32)
        out_depths = prepared["out_depths"] if prepared else torch.empty((P,), **f32)
        out_reps = torch.empty((DIM, H, W), **f32)
        geom = prepared["geom"] if prepared else torch.empty((L.gsr_geom_bytes(P),), **u8)
        if prepared and (radii.numel() != P or geom.numel() < L.gsr_geom_bytes(P) or geom.device != dev):
            raise RuntimeError("`prepared` does not belong to these parameters")
        img = torch.empty((L.gsr_image_bytes(W, H),), **u8)
        o = GsrForwardOut()
        o.out_color, o.out_invdepth, o.radii = ptr(out_color), ptr(out_invdepth), ptr(radii)
        o.accum_alpha, o.gauss_sum, o.gauss_count = ptr(accum_alpha), ptr(gauss_sum), ptr(gauss_count)
        o.last_contr_gauss, o.out_depths, o.out_reps = ptr(last_contr), ptr(out_depths), ptr(out_reps)
        o.geom_buffer, o.geom_bytes = ptr(geom), geom.numel()
        o.image_buffer, o.image_bytes = ptr(img), img.numel()
        o.preprocessed = (2 if prepared.get("depth_ordered") else 1) if prepared else 0
        st = stream_of(dev)
        if prepared and "num_rendered" in prepared:   # the batch call has run the prefix sums too
            R = int(prepared["num_rendered"])
            o.num_visible = int(prepared["num_visible"])
        else:
            R = C.c_int64(0)
            check(L.gsr_forward_stage1(C.byref(s), C.byref(o), st, C.byref(R)), "rasterize_gaussians (stage 1)")
            R = int(R.value)
        binning = torch.empty((L.gsr_binning_bytes(R) if R > 0 else 0,), **u8)
        # the sort's error word comes back through a pinned host word, looked at by later calls (no wait here)
        o.sort_error_host, slot = sort_errors.take()
        try:
            check(L.gsr_forward_stage2(C.byref(s), C.byref(o), R, ptr(binning), binning.numel(), st),
                  "rasterize_gaussians (stage 2)")
        except Exception:
            sort_errors.release(slot)    # the call reported its failure itself
            raise
        sort_errors.submitted(slot)
    return (R, out_color, radii, geom, binning, img, out_invdepth, accum_alpha, gauss_sum, gauss_count, last_contr,
            out_depths, out_reps)


def rasterize_gaussians_backward(background, means3D, radii, colors, opacities, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_invdepth, dL_dout_reps, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, antialiasing, debug, reps=None, sh_rest=None, raw_params=False,
                                 into=None, accumulate=False):
    """rasterize_points.cu:208-345.  Returns (dL_dmeans2D, dL_dcolors,
    dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
    dL_dreps), plus dL_dsh_rest as a tenth element when `sh_rest` is given.

    Beyond the reference: `sh_rest` / `raw_params` as in the forward;
    `into` = dict of preallocated gradient tensors (keys means3D, opacity, sh,
    sh_rest, scales, rotations, reps) that the kernels write directly, adding
    to their contents when `accumulate` is set."""
    L = lib()
    sort_errors.poll()
    if _hip_ext is not None and sh_rest is None and not raw_params and into is None and not accumulate:
        e = torch.Tensor([])
        return _hip_ext.rasterize_gaussians_backward(
            background, means3D, radii, colors, opacities, scales, rotations, float(scale_modifier), cov3D_precomp,
            viewmatrix, projmatrix, float(tan_fovx), float(tan_fovy), dL_dout_color,
            dL_dout_invdepth if dL_dout_invdepth is not None else e, dL_dout_reps if dL_dout_reps is not None else e,
            sh, int(degree), campos, geomBuffer, int(R), binningBuffer, imageBuffer, bool(antialiasing), bool(debug),
            reps if reps is not None else e, _events_address())
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    keep = []
    with torch.cuda.device(dev):
        s = _scene(background, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                   viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, False, antialiasing, debug,
                   reps, keep, sh_rest=sh_rest, raw=raw_params)
        M, DIM = s.M, s.DIM
        split = sh_rest is not None and sh_rest.numel() != 0
        into = into or {}
        if accumulate and not all(k in into for k in ("means3D", "opacity", "sh", "scales", "rotations")):
            raise RuntimeError("accumulate needs the gradient tensors to add into")
        f32 = dict(dtype=torch.float32, device=dev)
        has_sr = scales is not None and scales.numel() != 0
        e = torch.empty

        def buf(key, shape, fill_zero=False):
            t = into.get(key)
            if t is not None:
                if t.numel() != int(torch.Size(shape).numel()) or t.dtype != torch.float32 or t.device != dev or \
                        not t.is_contiguous():
                    raise RuntimeError(f"gradient buffer '{key}' must be a contiguous float32 tensor of shape {shape}")
                return t
            return torch.zeros(shape, **f32) if fill_zero else e(shape, **f32)

        # the two intermediate gradients are not stored when the caller writes into its own buffers
        # (forward_params never returns them)
        lean = bool(into) and raw_params
        dL_dmeans3D, dL_dmeans2D = buf("means3D", (P, 3)), e((P, 3), **f32)
        dL_dcolors = None if lean else e((P, 3), **f32)
        dL_dopacity, dL_dcov3D = buf("opacity", (P, 1)), (None if lean else e((P, 6), **f32))
        dL_dsh = buf("sh", (P, 1 if split else M, 3))
        dL_dsh_rest = buf("sh_rest", (P, M - 1, 3)) if split else None
        dL_dscales = buf("scales", (P, 3), not has_sr)
        dL_drotations = buf("rotations", (P, 4), not has_sr)
        dL_dreps = buf("reps", (P, DIM))
        if P != 0:
            gc = dev_f32(dL_dout_color, dev, "dL_dout_color")
            gd = dev_f32(dL_dout_invdepth, dev, "dL_dout_invdepth")
            gr = dev_f32(dL_dout_reps, dev, "dL_dout_reps") if DIM else None
            if DIM and gr is None:
                gr = torch.zeros((DIM, H, W), **f32)
            work = torch.empty((L.gsr_backward_work_bytes_dim(P, DIM),), dtype=torch.uint8, device=dev)
            io = GsrBackwardIO()
            io.dL_dout_color, io.dL_dout_invdepth, io.dL_dout_reps = ptr(gc), ptr(gd), ptr(gr)
            io.dL_dmeans2D, io.dL_dcolors, io.dL_dopacity = ptr(dL_dmeans2D), ptr(dL_dcolors), ptr(dL_dopacity)
            io.dL_dmeans3D, io.dL_dcov3D, io.dL_dsh = ptr(dL_dmeans3D), ptr(dL_dcov3D), ptr(dL_dsh)
            io.dL_dscales = ptr(dL_dscales) if has_sr else None
            io.dL_drotations = ptr(dL_drotations) if has_sr else None
            io.dL_dreps = ptr(dL_dreps)
            io.work, io.work_bytes = ptr(work), work.numel()
            io.dL_dsh_rest = ptr(dL_dsh_rest)
            io.accumulate = int(bool(accumulate))
            check(L.gsr_backward(C.byref(s), ptr(radii), ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer),
                                 int(R), C.byref(io), stream_of(dev)), "rasterize_gaussians_backward")
    res = (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dreps)
    return res + (dL_dsh_rest,) if split else res


def rasterize_gaussians_backward_blend(background, means3D, radii, opacities, scales, rotations, scale_modifier,
                                       viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_invdepth,
                                       dL_dout_reps, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                       antialiasing, debug, reps, sh_rest, raw_params, want_means2D=True,
                                       means2D_out=None):
    """First half of the backward for one view (no counterpart in the reference):
    the blend backward into a fresh per-Gaussian accumulator buffer.  Returns
    (dL_dmeans2D, work); `work` goes to backward_geometry_batch later, which
    can also fill dL_dmeans2D (want_means2D=False leaves it uninitialised here;
    `means2D_out`, a [P,3] float32 tensor, is then returned in its place)."""
    L = lib()
    sort_errors.poll()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    keep = []
    with torch.cuda.device(dev):
        s = _scene(background, means3D, None, opacities, scales, rotations, scale_modifier, None, viewmatrix,
                   projmatrix, tan_fovx, tan_fovy, H, W, sh, degree, campos, False, antialiasing, debug, reps, keep,
                   sh_rest=sh_rest, raw=raw_params)
        f32 = dict(dtype=torch.float32, device=dev)
        if means2D_out is not None and not want_means2D:
            if means2D_out.shape != (P, 3) or means2D_out.dtype != torch.float32 or means2D_out.device != dev or \
                    not means2D_out.is_contiguous():
                raise RuntimeError(f"means2D_out must be a contiguous float32 [P, 3] tensor on {dev}")
            dL_dmeans2D = means2D_out
        else:
            dL_dmeans2D = torch.empty((P, 3), **f32)
        work = torch.empty((L.gsr_backward_work_bytes_dim(P, s.DIM),), dtype=torch.uint8, device=dev)
        if P != 0:
            gc = dev_f32(dL_dout_color, dev, "dL_dout_color")
            gd = dev_f32(dL_dout_invdepth, dev, "dL_dout_invdepth")
            gr = dev_f32(dL_dout_reps, dev, "dL_dout_reps") if s.DIM else None
            if s.DIM and gr is None:
                gr = torch.zeros((s.DIM, H, W), **f32)
            io = GsrBackwardIO()
            io.dL_dout_color, io.dL_dout_invdepth, io.dL_dout_reps = ptr(gc), ptr(gd), ptr(gr)
            io.dL_dmeans2D = ptr(dL_dmeans2D) if want_means2D else None
            io.work, io.work_bytes = ptr(work), work.numel()
            check(L.gsr_backward_blend(C.byref(s), ptr(radii), ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer),
                                       int(R), C.byref(io), stream_of(dev)), "rasterize_gaussians_backward_blend")
    return dL_dmeans2D, work


def _blocks(t, P, dev):
    """A block-state array for P Gaussians (or None), checked."""
    if t is not None and (t.dtype != torch.uint8 or t.device != dev or not t.is_contiguous()
                          or t.numel() != (P + 63) // 64):
        raise RuntimeError(f"a block-state array must be a contiguous uint8 tensor of (P + 63) // 64 bytes on {dev}")
    return t


def backward_geometry_batch(means3D, opacities, scales, rotations, scale_modifier, sh, sh_rest, degree, reps, raw_params,
                            views, into, accumulate, debug=False, first=0, count=None, row_blocks=None):
    """Second half of the backward for a batch of views: `views` is a list of
    dicts (viewmatrix, projmatrix, campos, tanfovx, tanfovy, W, H, antialiasing,
    has_invdepth, radii, geom, work and optionally dL_dmeans2D, a [P,3] tensor
    that receives that view's screen-space gradient); the parameter gradients summed over the
    views are written (or, with `accumulate`, added) into the tensors of `into`
    (keys means3D, opacity, sh, sh_rest, scales, rotations, reps).  `first`, `count`:
    only the Gaussians [first, first + count) (first a multiple of 256); rows
    outside the range are not touched.  `row_blocks` and a view's
    `means2D_blocks` (uint8 [(P + 63) // 64]): per block of 64 Gaussians whether
    the gradient tensors / that view's dL_dmeans2D hold anything but zeros there;
    blocks that are zero and stay zero are not rewritten (GsrBackwardIO.row_blocks)."""
    L = lib()
    dev = means3D.device
    P = means3D.size(0)
    count = P - first if count is None else count
    if P == 0 or not views or count == 0:
        return
    keep = []
    with torch.cuda.device(dev):
        v0 = views[0]
        s = _scene(None, means3D, None, opacities, scales, rotations, scale_modifier, None, v0["viewmatrix"],
                   v0["projmatrix"], v0["tanfovx"], v0["tanfovy"], v0["H"], v0["W"], sh, degree, v0["campos"], False,
                   False, debug, reps, keep, sh_rest=sh_rest, raw=raw_params)
        arr = (GsrBackwardView * len(views))()
        for k, v in enumerate(views):
            for name in ("viewmatrix", "projmatrix", "campos"):
                c = dev_f32(v[name], dev, name)
                keep.append(c)
                setattr(arr[k], name, ptr(c))
            arr[k].tanfovx, arr[k].tanfovy = float(v["tanfovx"]), float(v["tanfovy"])
            arr[k].W, arr[k].H = int(v["W"]), int(v["H"])
            arr[k].antialiasing, arr[k].has_invdepth = int(bool(v["antialiasing"])), int(bool(v["has_invdepth"]))
            arr[k].radii, arr[k].geom_buffer, arr[k].work = ptr(v["radii"]), ptr(v["geom"]), ptr(v["work"])
            arr[k].dL_dmeans2D = ptr(v.get("dL_dmeans2D"))
            arr[k].means2D_blocks = ptr(_blocks(v.get("means2D_blocks"), P, dev)) \
                if v.get("dL_dmeans2D") is not None else None
        io = GsrBackwardIO()
        for key, field in (("means3D", "dL_dmeans3D"), ("opacity", "dL_dopacity"), ("sh", "dL_dsh"),
                           ("sh_rest", "dL_dsh_rest"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                           ("reps", "dL_dreps")):
            t = into.get(key)
            if t is not None and (t.dtype != torch.float32 or t.device != dev or not t.is_contiguous()):
                raise RuntimeError(f"gradient buffer '{key}' must be a contiguous float32 tensor on {dev}")
            setattr(io, field, ptr(t))
        io.accumulate = int(bool(accumulate))
        io.row_blocks = ptr(_blocks(row_blocks, P, dev))
        check(L.gsr_backward_geometry_batch_range(C.byref(s), len(views), arr, C.byref(io), int(first), int(count),
                                                  stream_of(dev)), "backward_geometry_batch")


def preprocess_batch(means3D, opacities, scales, rotations, scale_modifier, sh, sh_rest, degree, reps, raw_params,
                     views, debug=False):
    """The forward preprocess of several views in one pass over the Gaussians
    (no counterpart in the reference).  `views` is a list of
    GaussianRasterizationSettings-like objects (viewmatrix, projmatrix, campos,
    tanfovx, tanfovy, image_width, image_height, antialiasing); returns one dict
    per view (radii, out_depths, geom) to hand to `rasterize_gaussians(...,
    prepared=...)` while the parameters are unchanged."""
    L = lib()
    dev = means3D.device
    P = means3D.size(0)
    out = []
    if not views:
        return out
    keep = []
    with torch.cuda.device(dev):
        v0 = views[0]
        s = _scene(None, means3D, None, opacities, scales, rotations, scale_modifier, None, v0.viewmatrix,
                   v0.projmatrix, v0.tanfovx, v0.tanfovy, v0.image_height, v0.image_width, sh, degree, v0.campos, False,
                   False, debug, reps, keep, sh_rest=sh_rest, raw=raw_params)
        arr = (GsrForwardView * len(views))()
        nbytes = L.gsr_geom_bytes(P)
        for k, v in enumerate(views):
            for name in ("viewmatrix", "projmatrix", "campos"):
                c = dev_f32(getattr(v, name), dev, name)
                keep.append(c)
                setattr(arr[k], name, ptr(c))
            arr[k].tanfovx, arr[k].tanfovy = float(v.tanfovx), float(v.tanfovy)
            arr[k].W, arr[k].H = int(v.image_width), int(v.image_height)
            arr[k].antialiasing = int(bool(v.antialiasing))
            d = dict(radii=torch.empty((P,), dtype=torch.int32, device=dev),
                     out_depths=torch.empty((P,), dtype=torch.float32, device=dev),
                     geom=torch.empty((nbytes,), dtype=torch.uint8, device=dev))
            arr[k].radii, arr[k].out_depths = d["radii"].data_ptr(), d["out_depths"].data_ptr()
            arr[k].geom_buffer, arr[k].geom_bytes = d["geom"].data_ptr(), nbytes
            out.append(d)
        if P != 0:
            check(L.gsr_preprocess_batch(C.byref(s), len(views), arr, stream_of(dev)), "preprocess_batch")
        for k, d in enumerate(out):
            d["num_rendered"], d["num_visible"] = int(arr[k].num_rendered), int(arr[k].num_visible)
            d["depth_ordered"] = bool(arr[k].depth_ordered)
            d["depth_key_bits"] = {0: 0, 1: 32, 2: 27}[int(arr[k].depth_ordered)]   # which keys ordered it
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    """rasterize_points.cu:347-365."""
    if _hip_ext is not None:
        return _hip_ext.mark_visible(means3D, viewmatrix, projmatrix)
    L = lib()
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            m = dev_f32(means3D, dev, "means3D")
            v = dev_f32(viewmatrix, dev, "viewmatrix")
            pm = dev_f32(projmatrix, dev, "projmatrix")
            check(L.gsr_mark_visible(P, ptr(m), ptr(v), ptr(pm), present.data_ptr(), stream_of(dev)), "mark_visible")
    return present
