"""Fused Adam for the Gaussian parameter groups (one HIP kernel per tensor:
parameter, gradient and both moments are read once, 28 bytes per element).

`FusedGaussianAdam` subclasses torch.optim.Adam, keeps its param_groups and
state layout ({"step", "exp_avg", "exp_avg_sq"}), so the densification code
that edits the optimizer state (scene/gaussian_model.py:325-458) and
state_dict()/load_state_dict() work unchanged:

    self.optimizer = FusedGaussianAdam(l, lr=0.0, eps=1e-15)      # scene/gaussian_model.py:113
    ...
    gaussians.optimizer.step()                                     # dense: torch.optim.Adam's update
    gaussians.optimizer.step(radii > 0, radii.shape[0])            # sparse: train.py:339-343

The sparse step is the update of SparseGaussianAdam (graphdeco-inria/
diff-gaussian-rasterization, branch 3dgs_accel, which the reference tries to
import but does not vendor): Adam without bias correction, applied to the rows
flagged visible only.  The class is deliberately not exported from
diff_gaussian_rasterization: the reference switches its renderer to a branch
that does not work with the fork's rasterizer as soon as that import succeeds
(train.py:38-42, gaussian_renderer/__init__.py:138-150).
"""
import math

import torch

from _gsplat_lib import check, lib, stream_of


class FusedGaussianAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps)

    @torch.no_grad()
    def step(self, visibility=None, N=None, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        vis = None
        if visibility is not None:
            vis = visibility.to(torch.uint8) if visibility.dtype != torch.uint8 else visibility
            vis = vis.contiguous()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedGaussianAdam needs contiguous float32 GPU parameters (there is no CPU path)")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                rows = p.shape[0] if p.dim() > 0 else 1
                width = p.numel() // max(rows, 1)
                if vis is not None:
                    if N is not None and rows != N:
                        continue   # a group that is not per-Gaussian takes no sparse step
                    if vis.numel() != rows:
                        raise RuntimeError("visibility must have one entry per Gaussian")
                    bc1, bc2s = 1.0, 1.0
                else:
                    st["step"] += 1
                    t = float(st["step"])
                    bc1, bc2s = 1.0 - beta1 ** t, math.sqrt(1.0 - beta2 ** t)
                with torch.cuda.device(p.device):
                    check(L.gso_adam_update(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                            st["exp_avg_sq"].data_ptr(), None if vis is None else vis.data_ptr(),
                                            rows, width, lr, beta1, beta2, eps, bc1, bc2s, stream_of(p.device)),
                          "adam update")
        return loss


@torch.no_grad()
def add_densification_stats(radii, viewspace_grad, xyz_gradient_accum, denom, max_radii2D):
    """One fused pass for what train.py:298-312 and GaussianModel.add_densification_stats
    (scene/gaussian_model.py:202-214) do after every iteration, without the host-blocking
    `nonzero()` of the visibility mask: for Gaussians with radii > 0,
    max_radii2D = max(max_radii2D, radii), xyz_gradient_accum += |viewspace_grad[:, :2]|, denom += 1.
    All tensors are updated in place; `viewspace_grad` is `viewspace_point_tensor.grad` ([P,3])."""
    P = radii.shape[0]
    for t, shape in ((viewspace_grad, (P, 3)), (xyz_gradient_accum, (P, 1)), (denom, (P, 1)), (max_radii2D, (P,))):
        if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError(f"expected a contiguous float32 GPU tensor of shape {shape}, got {tuple(t.shape)} {t.dtype}")
    if radii.dtype != torch.int32 or not radii.is_contiguous():
        raise RuntimeError("radii must be the contiguous int32 tensor the rasterizer returned")
    with torch.cuda.device(radii.device):
        check(lib().gso_densify_stats(P, radii.data_ptr(), viewspace_grad.data_ptr(), xyz_gradient_accum.data_ptr(),
                                      denom.data_ptr(), max_radii2D.data_ptr(), stream_of(radii.device)),
              "densify stats")
