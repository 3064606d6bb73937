"""Drop-in for the reference's `fused_ssim` package
(submodules/fused-ssim/fused_ssim/__init__.py:8-41) over the C ABI."""
import os
import sys

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG_ROOT not in sys.path:
    sys.path.insert(0, _PKG_ROOT)

import torch

from _gsplat_lib import check, lib, ptr, stream_of

allowed_padding = ["same", "valid"]


def fusedssim(C1, C2, img1, img2, train=True):
    """ssim.cu:368-404: returns (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)."""
    if not img1.is_cuda:
        raise RuntimeError("img1 must be a GPU tensor (there is no CPU path)")
    dev = img1.device
    a, b = img1.contiguous().float(), img2.contiguous().float()
    B, CH, H, W = a.shape
    m = torch.empty_like(a)
    if train:
        d1, d2, d3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    else:
        d1 = d2 = d3 = torch.empty(0, device=dev)
    with torch.cuda.device(dev):
        check(lib().gss_ssim_forward(B, CH, H, W, float(C1), float(C2), ptr(a), ptr(b), ptr(m), ptr(d1), ptr(d2),
                                     ptr(d3), stream_of(dev)), "fusedssim")
    return m, d1, d2, d3


def fusedssim_backward(C1, C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    """ssim.cu:406-444."""
    dev = img1.device
    a, b, g = img1.contiguous().float(), img2.contiguous().float(), dL_dmap.contiguous().float()
    B, CH, H, W = a.shape
    out = torch.empty_like(a)
    with torch.cuda.device(dev):
        check(lib().gss_ssim_backward(B, CH, H, W, float(C1), float(C2), ptr(a), ptr(b), ptr(g), ptr(out),
                                      ptr(dm_dmu1), ptr(dm_dsigma1_sq), ptr(dm_dsigma12), stream_of(dev)),
              "fusedssim_backward")
    return out


class FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = fusedssim(C1, C2, img1, img2, train)
        if padding == "valid":
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        ctx.save_for_backward(img1.detach(), img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        ctx.C1, ctx.C2, ctx.padding = C1, C2, padding
        return ssim_map

    @staticmethod
    def backward(ctx, opt_grad):
        img1, img2, dm_dmu1, dm_dsigma1_sq, dm_dsigma12 = ctx.saved_tensors
        dL_dmap = opt_grad
        if ctx.padding == "valid":
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        grad = fusedssim_backward(ctx.C1, ctx.C2, img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)
        return None, None, grad, None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    C1 = 0.01 ** 2
    C2 = 0.03 ** 2
    assert padding in allowed_padding
    return FusedSSIMMap.apply(C1, C2, img1, img2, padding, train).mean()
