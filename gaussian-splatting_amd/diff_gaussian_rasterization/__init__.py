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

import torch
import torch.nn as nn

from . import _C


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
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities,
                              geomBuffer, binningBuffer, imgBuffer, reps)
        return color, radii, invdepths, gauss_sum, gauss_count, out_reps

    @staticmethod
    def backward(ctx, grad_out_color, _radii, grad_out_depth, _sum, _count, grad_out_reps):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, opacities, geomBuffer, binningBuffer,
         imgBuffer, reps) = ctx.saved_tensors
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


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, reps):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, reps)


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
