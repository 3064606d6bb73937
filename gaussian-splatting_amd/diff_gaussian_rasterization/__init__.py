"""Drop-in operator package for the reference's `diff_gaussian_rasterization`
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
same names, same call signatures, same 6-tuple result and 10 gradients, so
gaussian_renderer/__init__.py and train.py run unchanged.  The device work is
done by hand-written gfx950 kernels in libgsplat_amd.so; there is no CPU path.
"""
import os
import sys
from typing import NamedTuple

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)

import warnings

import torch
import torch.nn as nn

from . import _C

_warned = set()


def _warn_once(key, message):
    if key not in _warned:
        _warned.add(key)
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


class GaussianRasterizationSettings(NamedTuple):
    """__init__.py:263-276 of the reference."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    antialiasing: bool


class _RasterizeGaussians(torch.autograd.Function):
    """__init__.py:59-261 of the reference: forward returns (color, radii,
    invdepths, gauss_sum, gauss_count, out_reps); backward returns gradients
    for the 10 forward inputs (None for the settings)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, reps):
        rs = raster_settings
        (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, accum_alpha, gauss_sum,
         gauss_count, last_contr_gauss, depths, out_reps) = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.antialiasing, rs.debug, reps)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.out_shapes = (color.shape, out_reps.shape)
        # outputs the loss does not use arrive in backward as None instead of freshly zero-filled tensors:
        # no fill kernels, and an unused inverse-depth output switches its gradient path off in the kernels
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer, reps)
        return color, radii, invdepths, gauss_sum, gauss_count, out_reps

    @staticmethod
    def backward(ctx, grad_out_color, _radii, grad_out_depth, _sum, _count, grad_out_reps):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer, binningBuffer,
         imgBuffer, reps) = ctx.saved_tensors
        grad_out_color, grad_out_depth, grad_out_reps = _given_grads(ctx, means3D, grad_out_color, grad_out_depth,
                                                                     grad_out_reps)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_reps) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, grad_out_reps, sh,
            rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.antialiasing,
            rs.debug, reps)

        def given(t, g):  # optional inputs arrive as empty tensors; they take no gradient
            return g if t.numel() != 0 else None

        return (grad_means3D, grad_means2D, given(sh, grad_sh), given(colors_precomp, grad_colors_precomp),
                grad_opacities.view_as(opacities), given(scales, grad_scales), given(rotations, grad_rotations),
                given(cov3Ds_precomp, grad_cov3Ds_precomp), None, given(reps, grad_reps))


def _given_grads(ctx, like, grad_color, grad_depth, grad_reps):
    """Incoming gradients with materialisation off: the colour gradient is needed by the kernels
    (zeros if the loss did not use the image), the inverse-depth one is optional (None switches
    that path off), the reps one is needed when there are reps channels."""
    color_shape, reps_shape = ctx.out_shapes
    if grad_color is None:
        grad_color = torch.zeros(color_shape, dtype=torch.float32, device=like.device)
    if grad_depth is None:
        grad_depth = torch.Tensor([])
    if grad_reps is None:
        grad_reps = torch.zeros(reps_shape, dtype=torch.float32, device=like.device) if reps_shape[0] else torch.Tensor([])
    return grad_color, grad_depth, grad_reps


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, reps):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, reps)


class ParamGradSink:
    """Gradient buffers that the raw-parameter backward fills directly.

    Autograd would hand back seven fresh gradient tensors per view and add them
    into `.grad` with one elementwise kernel each (1.5 GB of parameters at 6 M
    Gaussians).  With a sink the backward kernels write (first view of a step)
    or add (later views) straight into persistent buffers, which are installed
    as the parameters' `.grad`.  Call `begin_step()` before the first
    backward of every optimiser step instead of zeroing the gradients.

    The buffers are views into ONE flat allocation (`flat`, tensor after tensor,
    each starting on a 256-byte boundary), so a multi-GPU caller can hand the
    whole gradient, or the rows of a range of Gaussians, to a collective without
    gathering them first.  The sink belongs to the parameter objects it was built
    from: build a new one after densification replaced them."""

    NAMES = ("means3D", "sh", "sh_rest", "opacity", "scales", "rotations", "reps")
    ROW_ALIGN = 256   # Gaussian ranges of a chunked flush start at multiples of this

    def __init__(self, xyz, features_dc, features_rest, opacity, scaling, rotation, reps=None, defer=False,
                 keep_zero_rows=False):
        """defer: a view's backward only runs the blend backward (and returns the
        screen-space gradient); the per-Gaussian backward of all the views of a
        step runs once, in `flush()`, reading every parameter and writing every
        gradient a single time.  Call `flush()` before using the gradients: the
        parameters' and also the `means2D.grad` of the deferred views, which
        `flush()` fills.

        keep_zero_rows (with defer): the gradient rows of a Gaussian that no view of the
        step reached are zero (see `touched_rows`), and in an occluded scene that is
        nearly all of them, step after step.  The sink then remembers per block of 64
        Gaussians whether the buffers hold anything but zeros there, and `flush()`
        does not write zeros over zeros (`GsrBackwardIO.row_blocks`).  The results
        are the same complete gradients.  Two conditions come with it: (1) whoever
        else writes into the gradient buffers (another loss term's backward adding to
        `.grad`, a collective that adds other GPUs' rows) calls `mark_rows_dirty`
        afterwards (`gsdp.reduce_flush` does); (2) the `means2D.grad` tensors of the
        views come from a pool that the next step reuses, so they are valid until the
        next step's `flush()` (train.py reads them right after `backward()`)."""
        self.params = dict(means3D=xyz, sh=features_dc, sh_rest=features_rest, opacity=opacity, scales=scaling,
                           rotations=rotation, reps=reps)
        live = [(k, p) for k, p in self.params.items() if p is not None and p.numel() != 0]
        offs, total = {}, 0
        for k, p in live:
            offs[k] = total
            total += (p.numel() + 63) // 64 * 64
        like = live[0][1]
        self.flat = torch.zeros((total,), dtype=torch.float32, device=like.device)
        self.grads = {k: self.flat[offs[k]:offs[k] + p.numel()].view(p.shape) for k, p in live}
        self.attach()
        self.fresh = True
        self.defer = bool(defer) and "sh_rest" in self.grads and \
            features_dc.size(1) + features_rest.size(1) == 16
        if defer and not self.defer:
            _warn_once("defer", "ParamGradSink(defer=True): the batched per-Gaussian backward needs split SH tensors "
                       "with 16 coefficients in all (SH degree 3 allocated); this model has not, so every view's "
                       "backward runs the one-view kernel (same gradients, slower)")
        self.pending = []
        self.prepared = []
        self.hit_counts = []
        # block states, see keep_zero_rows: `flat` starts as zeros, so every block starts clean
        self.row_blocks = None
        self.m2_pool, self.m2_next = [], 0
        if keep_zero_rows and self.defer and like.is_cuda:
            self.row_blocks = torch.zeros(((like.size(0) + 63) // 64,), dtype=torch.uint8, device=like.device)

    def mark_rows_dirty(self, rows=None):
        """Tell the sink that something other than its own backward wrote non-zero values into the
        gradient rows `rows` (int64 indices; None: any row).  Only matters with keep_zero_rows."""
        if self.row_blocks is None:
            return
        if rows is None:
            self.row_blocks.fill_(1)
        elif rows.numel() != 0:
            self.row_blocks.index_fill_(0, rows.to(self.row_blocks.device, torch.int64) >> 6, 1)

    def take_means2D(self):
        """(dL_dmeans2D [P,3], its block states) for the next deferred view of this step: from the
        pool (slot k serves the k-th backward of every step), or None without keep_zero_rows."""
        if self.row_blocks is None:
            return None
        if self.m2_next == len(self.m2_pool):
            P, dev = self.num_gaussians(), self.flat.device
            self.m2_pool.append((torch.zeros((P, 3), dtype=torch.float32, device=dev),
                                 torch.zeros_like(self.row_blocks)))
        slot = self.m2_pool[self.m2_next]
        self.m2_next += 1
        return slot

    def attach(self):
        """(Re-)install the buffers as the parameters' `.grad`: `optimizer.zero_grad(set_to_none=True)`
        (train.py) detaches them, after which an optimizer would see no gradient at all."""
        for k, g in self.grads.items():
            p = self.params[k]
            if p.shape != g.shape or p.device != g.device:
                raise RuntimeError(f"ParamGradSink: parameter '{k}' changed shape or device since the sink was built "
                                   "(build a new sink after densification)")
            if p.grad is not g:
                p.grad = g

    def begin_step(self):
        """The next backward overwrites the buffers, the ones after it accumulate."""
        self.pending = []
        self.prepared = []
        self.hit_counts = []
        self.m2_next = 0
        self.fresh = True
        self.attach()

    def prepare_views(self, settings_list):
        """Run the forward preprocess of the views that are going to be rendered from the current
        parameter values (e.g. all the views of this optimiser step) in one pass over the
        Gaussians: parameters and SH rows are read once instead of once per view.  The following
        `forward_params(..., grad_sink=self)` calls whose `raster_settings` is one of these
        objects pick their share up; results are identical to unprepared calls.  Only valid
        while the parameters keep their values (`begin_step()` drops what was not used).
        Needs 16 SH coefficients; otherwise does nothing."""
        p = self.params
        if not settings_list:
            return
        if "sh_rest" not in self.grads or p["sh"].size(1) + p["sh_rest"].size(1) != 16:
            _warn_once("prepare", "ParamGradSink.prepare_views: the batched forward preprocess needs split SH tensors "
                       "with 16 coefficients in all (SH degree 3 allocated); this model has not, so every view runs "
                       "its own preprocess (same results, slower)")
            return
        groups = {}   # one call per (scale modifier, degree): the rest is per view
        for rs in settings_list:
            groups.setdefault((float(rs.scale_modifier), int(rs.sh_degree), bool(rs.debug)), []).append(rs)
        empty = torch.Tensor([])
        for (mod, degree, debug), views in groups.items():
            res = _C.preprocess_batch(p["means3D"], p["opacity"], p["scales"], p["rotations"], mod, p["sh"],
                                      p["sh_rest"], degree, empty if p["reps"] is None else p["reps"], True, views,
                                      debug)
            self.prepared.extend(zip(views, res))

    def take_prepared(self, rs):
        for k, (v, d) in enumerate(self.prepared):
            if v is rs:
                del self.prepared[k]
                return d
        return None

    def num_gaussians(self):
        return self.params["means3D"].size(0)

    def chunk_bounds(self, chunks):
        """(first, count) of `chunks` contiguous Gaussian ranges that cover all the Gaussians; every
        range starts at a multiple of ROW_ALIGN (fewer ranges when there are too few Gaussians)."""
        P = self.num_gaussians()
        blocks = (P + self.ROW_ALIGN - 1) // self.ROW_ALIGN
        chunks = max(1, min(int(chunks), blocks))
        out = []
        for c in range(chunks):
            b0, b1 = blocks * c // chunks, blocks * (c + 1) // chunks
            first, end = b0 * self.ROW_ALIGN, min(P, b1 * self.ROW_ALIGN)
            if end > first:
                out.append((first, end - first))
        return out or [(0, 0)]

    def _geometry(self, cfg, views, first, count, accumulate):
        """The per-Gaussian backward of `views` for the Gaussians [first, first + count)."""
        scale_modifier, degree, debug = cfg
        p = self.params
        empty = torch.Tensor([])
        _C.backward_geometry_batch(p["means3D"], p["opacity"], p["scales"], p["rotations"], scale_modifier, p["sh"],
                                   p["sh_rest"], degree, empty if p["reps"] is None else p["reps"], True, views,
                                   self.grads, accumulate, debug, first=first, count=count, row_blocks=self.row_blocks)

    def flush(self, chunks=1, on_chunk=None):
        """Run the per-Gaussian backward of the views whose backward was deferred.

        chunks / on_chunk: the Gaussians are processed as `chunks` contiguous ranges, one launch
        after the other on the current stream, and `on_chunk(first, count)` is called right after
        the launch that completes the gradient rows [first, first + count) of every parameter
        (they are final once the work enqueued so far has run).  A data-parallel caller starts
        the exchange of those rows there, so that it overlaps the remaining ranges.  `on_chunk`
        is also called when nothing was deferred (the gradients are then already complete; if no
        backward ran at all in this step they are set to zero first)."""
        groups = []   # views that share scale modifier / SH degree go through one launch
        for v in self.pending:
            for cfg, lst in groups:
                if cfg == v["cfg"]:
                    lst.append(v)
                    break
            else:
                groups.append((v["cfg"], [v]))
        if not groups and on_chunk is not None and self.fresh:
            self.flat.zero_()
            if self.row_blocks is not None:
                self.row_blocks.zero_()
            self.fresh = False
        for first, count in self.chunk_bounds(chunks if (groups or on_chunk is not None) else 1):
            accumulate = not self.fresh
            for cfg, views in groups:
                self._geometry(cfg, views, first, count, accumulate)
                accumulate = True
            if on_chunk is not None:
                on_chunk(first, count)
        for v in self.pending:   # the screen-space gradients the batch kernel wrote, one tensor per view
            m2 = v["means2D"]
            m2.grad = v["dL_dmeans2D"] if m2.grad is None else m2.grad + v["dL_dmeans2D"]
        if groups:
            self.fresh = False
        self.pending = []

    def tensors(self):
        return list(self.grads.values())

    def touched_rows(self):
        """uint8 [P]: 1 for the Gaussians whose gradient rows can be non-zero in this step, i.e. that
        some pixel of some view rendered since `begin_step()` reached while it still accepted
        contributions (`gauss_count` != 0, forward.cu:423-432).  Every gradient of a Gaussian comes
        from such a pixel (backward.cu:591-611 walks the same pairs), so the rows of all other
        Gaussians are exactly zero: a data-parallel caller exchanges only the union of these rows."""
        P = self.num_gaussians()
        dev = self.flat.device
        counts = [c.view(-1) for c in self.hit_counts]
        if dev.type == "cuda" and all(c.is_cuda and c.dtype == torch.int32 and c.is_contiguous() and c.numel() == P
                                      and c.data_ptr() % 16 == 0 for c in counts):
            # one kernel over the views' counts (gso_mark_nonzero)
            import ctypes
            from _gsplat_lib import check, lib
            t = torch.empty(((P + 3) // 4 * 4,), dtype=torch.uint8, device=dev)[:P]
            ptrs = (ctypes.c_void_p * max(1, len(counts)))(*[c.data_ptr() for c in counts])
            with torch.cuda.device(dev):
                check(lib().gso_mark_nonzero(P, len(counts), ptrs, t.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "gso_mark_nonzero")
            return t
        t = torch.zeros((P,), dtype=torch.uint8, device=dev)
        for c in counts:
            t |= c.ne(0)
        return t


class _RasterizeGaussiansParams(torch.autograd.Function):
    """The same operator on the model's raw parameters: the SH tensors stay
    split (no torch.cat, scene/gaussian_model.py:487-490) and the opacity,
    scaling and rotation activations (scene/gaussian_model.py:41-53) run inside
    the preprocess kernels, forward and backward."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings, reps,
                sink):
        rs = raster_settings
        empty = torch.Tensor([])
        prepared = sink.take_prepared(rs) if sink is not None else None
        (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, invdepths, accum_alpha, gauss_sum,
         gauss_count, last_contr_gauss, depths, out_reps) = _C.rasterize_gaussians(
            rs.bg, xyz, empty, opacity, scaling, rotation, rs.scale_modifier, empty, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, features_dc, rs.sh_degree, rs.campos,
            rs.prefiltered, rs.antialiasing, rs.debug, reps, sh_rest=features_rest, raw_params=True,
            prepared=prepared)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.out_shapes = (color.shape, out_reps.shape)
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        ctx.means2D = means2D
        ctx.gauss_count = gauss_count if sink is not None else None
        ctx.save_for_backward(xyz, scaling, rotation, radii, features_dc, features_rest, opacity, geomBuffer,
                              binningBuffer, imgBuffer, reps)
        return color, radii, invdepths, gauss_sum, gauss_count, out_reps

    @staticmethod
    def backward(ctx, grad_out_color, _radii, grad_out_depth, _sum, _count, grad_out_reps):
        rs = ctx.raster_settings
        (xyz, scaling, rotation, radii, features_dc, features_rest, opacity, geomBuffer, binningBuffer, imgBuffer,
         reps) = ctx.saved_tensors
        grad_out_color, grad_out_depth, grad_out_reps = _given_grads(ctx, xyz, grad_out_color, grad_out_depth,
                                                                     grad_out_reps)
        sink = ctx.sink
        empty = torch.Tensor([])
        if sink is not None:
            sink.hit_counts.append(ctx.gauss_count)
        if sink is not None and sink.defer:
            slot = sink.take_means2D()
            g_means2D, work = _C.rasterize_gaussians_backward_blend(
                rs.bg, xyz, radii, opacity, scaling, rotation, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
                rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, grad_out_reps, features_dc, rs.sh_degree,
                rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.antialiasing, rs.debug, reps,
                features_rest, True, want_means2D=False, means2D_out=None if slot is None else slot[0])
            # nothing per Gaussian runs here (this may be a side stream with other views' kernels in flight):
            # flush() processes every pending view, however many, in launches of up to eight views
            sink.pending.append(dict(
                cfg=(float(rs.scale_modifier), int(rs.sh_degree), bool(rs.debug)), viewmatrix=rs.viewmatrix, projmatrix=rs.projmatrix, campos=rs.campos, tanfovx=rs.tanfovx,
                tanfovy=rs.tanfovy, W=rs.image_width, H=rs.image_height, antialiasing=rs.antialiasing,
                has_invdepth=grad_out_depth is not None and grad_out_depth.numel() != 0, radii=radii,
                geom=geomBuffer, work=work, dL_dmeans2D=g_means2D, means2D=ctx.means2D,
                means2D_blocks=None if slot is None else slot[1]))
            return (None,) * 10
        into, accumulate = None, False
        if sink is not None:
            into, accumulate = sink.grads, not sink.fresh
        res = _C.rasterize_gaussians_backward(
            rs.bg, xyz, radii, empty, opacity, scaling, rotation, rs.scale_modifier, empty, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, grad_out_reps, features_dc,
            rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.antialiasing,
            rs.debug, reps, sh_rest=features_rest, raw_params=True, into=into, accumulate=accumulate)
        g_means2D, _gc, g_opacity, g_means3D, _gcov, g_dc, g_scales, g_rot, g_reps = res[:9]
        g_rest = res[9] if len(res) > 9 else None     # a degree-0 model has no second SH tensor
        if sink is not None:
            sink.fresh = False
            sink.mark_rows_dirty()   # the one-view kernel wrote (or added to) the rows of everything it rendered
            return (None, g_means2D, None, None, None, None, None, None, None, None)
        return (g_means3D, g_means2D, g_dc, g_rest, g_opacity.view_as(opacity), g_scales, g_rot, None,
                g_reps if reps.numel() != 0 else None, None)


class GaussianRasterizer(nn.Module):
    """__init__.py:278-340 of the reference."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, reps=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        reps = empty if reps is None else reps   # the reference requires reps; without them DIM = 0
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, reps)

    def forward_params(self, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, reps=None,
                       grad_sink=None):
        """Render from the model's raw parameters (`GaussianModel._xyz`,
        `_features_dc`, `_features_rest`, `_opacity`, `_scaling`, `_rotation`,
        `_reps`): equal to `forward(means3D=xyz, shs=cat(dc, rest),
        opacities=sigmoid(opacity), scales=exp(scaling),
        rotations=normalize(rotation), reps=reps)` with the activations and the
        concatenation done inside the kernels.  Same 6-tuple result.  With a
        `ParamGradSink` the parameter gradients are written into its buffers."""
        features_rest = torch.Tensor([]) if features_rest is None else features_rest
        reps = torch.Tensor([]) if reps is None else reps
        return _RasterizeGaussiansParams.apply(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation,
                                               self.raster_settings, reps, grad_sink)
