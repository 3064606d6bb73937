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


class _L1SSIMLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        if not image.is_cuda:
            raise RuntimeError("image must be a GPU tensor (there is no CPU path)")
        dev = image.device
        a, b = image.contiguous().float(), gt.contiguous().float()
        if a.shape != b.shape or a.dim() < 2:
            raise RuntimeError("image and gt must have the same shape [..., H, W]")
        H, W = a.shape[-2], a.shape[-1]
        planes = a.numel() // (H * W)
        L = lib()
        loss = torch.empty((3,), dtype=torch.float32, device=dev)
        d1, d2, d3 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        partial = torch.empty((L.gss_loss_partial_floats(planes, H, W),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.gss_l1_ssim_loss_forward(planes, H, W, 0.01 ** 2, 0.03 ** 2, float(lambda_dssim), ptr(a), ptr(b),
                                             ptr(loss), ptr(d1), ptr(d2), ptr(d3), ptr(partial), stream_of(dev)),
                  "l1_ssim_loss")
        ctx.save_for_backward(a, b, d1, d2, d3)
        ctx.lam, ctx.dims = float(lambda_dssim), (planes, H, W)
        ctx.mark_non_differentiable(loss)
        ctx.set_materialize_grads(False)
        return loss[0], loss

    @staticmethod
    def backward(ctx, grad_loss, _parts):
        a, b, d1, d2, d3 = ctx.saved_tensors
        if grad_loss is None:
            return None, None, None
        planes, H, W = ctx.dims
        dev = a.device
        g = grad_loss.contiguous().float()
        out = torch.empty_like(a)
        with torch.cuda.device(dev):
            check(lib().gss_l1_ssim_loss_backward(planes, H, W, ctx.lam, ptr(g), ptr(a), ptr(b), ptr(out), ptr(d1),
                                                  ptr(d2), ptr(d3), stream_of(dev)), "l1_ssim_loss backward")
        return out, None, None


def l1_ssim_loss(image, gt, lambda_dssim=0.2, return_parts=False):
    """(1 - lambda_dssim) * l1_loss(c, gt) + lambda_dssim * (1 - fused_ssim(c, gt)) with
    c = image.clamp(0, 1): the photometric loss of train.py:222-230 on the unclamped render
    (gaussian_renderer/__init__.py:179 clamps it), forward in two kernels and backward in one
    instead of fused-ssim plus a dozen elementwise PyTorch kernels.  No counterpart in the
    reference's fused_ssim package.  `image`, `gt`: [C, H, W] or [B, C, H, W].  With
    `return_parts` also returns a detached tensor (loss, L1 mean, ssim mean) for logging."""
    loss, parts = _L1SSIMLoss.apply(image, gt, lambda_dssim)
    return (loss, parts) if return_parts else loss


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, b_scalar):
        if not a.is_cuda:
            raise RuntimeError("the first argument must be a GPU tensor (there is no CPU path)")
        dev = a.device
        x = a.contiguous().float()
        y = None
        if b is not None:
            if b.shape != a.shape:
                raise RuntimeError("l1_loss: both tensors must have the same shape")
            y = b.to(dev).contiguous().float()
        n = x.numel()
        if n == 0:
            raise RuntimeError("l1_loss of an empty tensor")
        L = lib()
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        partial = torch.empty((L.gss_l1_partial_floats(n),), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(L.gss_l1_loss_forward(n, ptr(x), ptr(y) if y is not None else None, float(b_scalar), ptr(loss),
                                        ptr(partial), stream_of(dev)), "l1_loss")
        ctx.save_for_backward(x, y)
        ctx.b_scalar = float(b_scalar)
        ctx.set_materialize_grads(False)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        x, y = ctx.saved_tensors
        if grad_loss is None:
            return None, None, None
        dev = x.device
        g = grad_loss.contiguous().float()
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            check(lib().gss_l1_loss_backward(x.numel(), ptr(g), ptr(x), ptr(y) if y is not None else None, ctx.b_scalar,
                                             ptr(out), stream_of(dev)), "l1_loss backward")
        return out, None, None


def l1_loss(network_output, gt):
    """utils/loss_utils.py:23-24, `torch.abs(network_output - gt).mean()`, forward in two kernels
    and backward in one (train.py:249 applies it to the rendered feature channels).  `gt` is a
    tensor of the same shape or a number; the gradient goes to `network_output` only."""
    if torch.is_tensor(gt):
        return _L1Loss.apply(network_output, gt, 0.0)
    return _L1Loss.apply(network_output, None, float(gt))
