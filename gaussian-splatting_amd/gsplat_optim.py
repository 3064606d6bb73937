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
import ctypes as C
import math

import torch
import torch.nn as nn

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


class RowCompactor:
    """The kept rows of a boolean mask, computed once and applied to any number of tensors:
    `RowCompactor(mask)(t)` equals `t[mask]` for tensors whose first dimension is the mask's
    length.  One host block (to learn how many rows are kept) instead of one per tensor."""

    def __init__(self, mask):
        if mask.dim() != 1 or not mask.is_cuda:
            raise RuntimeError("mask must be a 1-D GPU tensor (there is no CPU path)")
        L = lib()
        self.N = mask.shape[0]
        self.device = mask.device
        keep = (mask != 0).to(torch.uint8).contiguous() if mask.dtype not in (torch.uint8, torch.bool) else \
            mask.contiguous().view(torch.uint8)
        self.index = torch.empty((max(self.N, 1),), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            temp = torch.empty((L.gso_compact_temp_bytes(self.N),), dtype=torch.uint8, device=self.device)
            kept = C.c_int64(0)
            check(L.gso_compact_map(self.N, keep.data_ptr() if self.N else None, self.index.data_ptr(), temp.data_ptr(),
                                    temp.numel(), C.byref(kept), stream_of(self.device)), "compact map")
        self.kept = int(kept.value)

    def __call__(self, t):
        if t.shape[0] != self.N or t.device != self.device:
            raise RuntimeError("tensor does not match the mask")
        if t.element_size() != 4 or not t.is_contiguous():
            return t[self.index[:self.kept].long()]   # other layouts: plain indexing with the same list
        out = torch.empty((self.kept,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        M = t.numel() // max(self.N, 1)
        if self.kept and M:
            with torch.cuda.device(self.device):
                check(lib().gso_gather_rows(t.data_ptr(), out.data_ptr(), self.index.data_ptr(), self.kept, M,
                                            stream_of(self.device)), "gather rows")
        return out


@torch.no_grad()
def prune_optimizer(optimizer, mask):
    """GaussianModel._prune_optimizer (scene/gaussian_model.py:427-441) with the mask resolved once:
    keeps the rows of `mask` in every group's parameter and in its Adam moments, re-registers the
    state under the new nn.Parameter and returns {group name: new parameter}.  Also returns the
    compactor as the second value, for the per-Gaussian statistics tensors that prune_points
    filters next (scene/gaussian_model.py:419-424)."""
    comp = RowCompactor(mask)
    optimizable = {}
    for group in optimizer.param_groups:
        p = group["params"][0]
        stored = optimizer.state.get(p, None)
        new_p = nn.Parameter(comp(p.data).requires_grad_(True))
        if stored is not None:
            stored["exp_avg"] = comp(stored["exp_avg"])
            stored["exp_avg_sq"] = comp(stored["exp_avg_sq"])
            del optimizer.state[p]
            optimizer.state[new_p] = stored
        group["params"][0] = new_p
        optimizable[group["name"]] = new_p
    return optimizable, comp


@torch.no_grad()
def cat_tensors_to_optimizer(optimizer, tensors_dict):
    """GaussianModel.cat_tensors_to_optimizer (scene/gaussian_model.py:325-346): appends the new
    rows to every group's parameter, zero moments for them, returns {group name: new parameter}."""
    optimizable = {}
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        ext = tensors_dict[group["name"]]
        p = group["params"][0]
        stored = optimizer.state.get(p, None)
        new_p = nn.Parameter(torch.cat((p.data, ext), dim=0).requires_grad_(True))
        if stored is not None:
            stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
            stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del optimizer.state[p]
            optimizer.state[new_p] = stored
        group["params"][0] = new_p
        optimizable[group["name"]] = new_p
    return optimizable


# ---------------------------------------------------------------- densification (clone / split / prune)
def _f32(t, shape, what):
    if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise RuntimeError(f"{what}: expected a contiguous float32 GPU tensor of shape {tuple(shape)}, "
                           f"got {tuple(t.shape)} {t.dtype} on {t.device}")
    return t


@torch.no_grad()
def densify_select(xyz_gradient_accum, denom, scaling, grad_threshold, size_threshold, split, n_grads=None):
    """The clone (split=False, scene/gaussian_model.py:263-270) or split (split=True, :356-364) selection as a
    bool mask over the N rows of `scaling` (raw log-scales), in one kernel: gradient = accum / denom with
    NaN -> 0 for the first `n_grads` rows (default: all), 0 for the rest; size_threshold = percent_dense * extent."""
    N = scaling.shape[0]
    n_grads = N if n_grads is None else int(n_grads)
    _f32(scaling, (N, 3), "scaling")
    sel = torch.empty((N,), dtype=torch.uint8, device=scaling.device)
    with torch.cuda.device(scaling.device):
        check(lib().gso_densify_select(N, n_grads, xyz_gradient_accum.data_ptr() if n_grads else None,
                                       denom.data_ptr() if n_grads else None, scaling.data_ptr(), float(grad_threshold),
                                       float(size_threshold), int(bool(split)), sel.data_ptr(), stream_of(scaling.device)),
              "densify select")
    return sel.view(torch.bool)


@torch.no_grad()
def prune_mask(opacity, max_radii2D, scaling, min_opacity, max_screen_size, extent, also=None):
    """The prune decision of scene/gaussian_model.py:237-244 (+ the rows of `also`) as a bool mask, one kernel."""
    N = opacity.shape[0]
    m = torch.empty((N,), dtype=torch.uint8, device=opacity.device)
    al = None if also is None else also.contiguous().view(torch.uint8)
    with torch.cuda.device(opacity.device):
        check(lib().gso_prune_mask(N, opacity.contiguous().data_ptr(), max_radii2D.data_ptr(), scaling.data_ptr(),
                                   float(min_opacity), float(max_screen_size) if max_screen_size else 0.0,
                                   0.1 * float(extent), None if al is None else al.data_ptr(), m.data_ptr(),
                                   stream_of(opacity.device)), "prune mask")
    return m.view(torch.bool)


@torch.no_grad()
def split_children(parents, n, xyz, scaling, rotation, normals):
    """Positions and log-scales of the K * n children of the K parents (row indices, int32, ascending)
    as scene/gaussian_model.py:367-377 forms them, `normals` = the [K*n, 3] standard normal samples;
    also returns parent_of_child [K*n] (int32) for copying the other tensors' rows."""
    K = parents.shape[0]
    dev = xyz.device
    new_xyz = torch.empty((K * n, 3), dtype=torch.float32, device=dev)
    new_scaling = torch.empty((K * n, 3), dtype=torch.float32, device=dev)
    poc = torch.empty((K * n,), dtype=torch.int32, device=dev)
    if K:
        _f32(normals, (K * n, 3), "normals")
        with torch.cuda.device(dev):
            check(lib().gso_split_children(K, int(n), parents.data_ptr(), xyz.data_ptr(), scaling.data_ptr(),
                                           rotation.data_ptr(), normals.data_ptr(), new_xyz.data_ptr(),
                                           new_scaling.data_ptr(), poc.data_ptr(), stream_of(dev)), "split children")
    return new_xyz, new_scaling, poc


def _take_rows(t, rows):
    """t[rows] for an int32 row list through the row-gather kernel."""
    K = rows.shape[0]
    out = torch.empty((K,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    M = t.numel() // max(t.shape[0], 1)
    if K and M:
        if t.element_size() != 4 or not t.is_contiguous():
            return t[rows.long()]
        with torch.cuda.device(t.device):
            check(lib().gso_gather_rows(t.data_ptr(), out.data_ptr(), rows.data_ptr(), K, M, stream_of(t.device)),
                  "gather rows")
    return out


@torch.no_grad()
def densify_and_prune(optimizer, stats, max_grad, min_opacity, extent, max_screen_size, radii, percent_dense=0.01,
                      n_split=2, normals=None):
    """GaussianModel.densify_and_prune (scene/gaussian_model.py:216-252) on an optimizer whose param groups carry
    the reference's names ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "reps"; one tensor each):
    clone the small Gaussians with a large view-space gradient, split the large ones into `n_split` children,
    prune the transparent and the oversized, with the Adam moments following every row.  `stats` is a dict with
    xyz_gradient_accum [N,1], denom [N,1], max_radii2D [N]; `radii` the last forward's radii (the reference's
    tmp_radii).  `normals`: the standard normal samples for the split, [n_split * (number split), 3] or a
    callable (rows) -> tensor; default torch.randn.  The selections, the split and the prune mask are one
    kernel each, every row list is resolved once (one host block per stage) and applied to all tensors.
    Returns ({group name: new parameter}, new stats dict, radii of the surviving rows)."""
    by_name = {g["name"]: g for g in optimizer.param_groups}
    P = lambda name: by_name[name]["params"][0]
    accum, den = stats["xyz_gradient_accum"], stats["denom"]
    N0 = P("xyz").shape[0]
    size_thr = percent_dense * extent
    tmp_radii = radii

    def append(rows_src, overrides):
        ext = {}
        for name in by_name:
            ext[name] = overrides[name] if name in overrides else _take_rows(P(name).data, rows_src)
        return cat_tensors_to_optimizer(optimizer, ext)

    # clone (:254-283)
    sel = densify_select(accum, den, P("scaling").data, max_grad, size_thr, split=False)
    comp = RowCompactor(sel)
    rows = comp.index[:comp.kept]
    if comp.kept:
        append(rows, {})
        tmp_radii = torch.cat((tmp_radii, _take_rows(tmp_radii, rows)))
    # split (:347-392): decided on the grown model, gradients padded with zeros for the clones
    sel = densify_select(accum, den, P("scaling").data, max_grad, size_thr, split=True, n_grads=N0)
    comp = RowCompactor(sel)
    rows = comp.index[:comp.kept]
    also = None
    if comp.kept:
        K = comp.kept
        z = normals(K * n_split) if callable(normals) else normals
        if z is None:
            z = torch.randn((K * n_split, 3), device=sel.device)
        new_xyz, new_scaling, poc = split_children(rows, n_split, P("xyz").data, P("scaling").data, P("rotation").data,
                                                   z.contiguous())
        append(poc, {"xyz": new_xyz, "scaling": new_scaling})
        tmp_radii = torch.cat((tmp_radii, _take_rows(tmp_radii, poc)))
        also = torch.cat((sel, torch.zeros(K * n_split, dtype=torch.bool, device=sel.device)))
    # densification_postfix restarts the statistics at the new size whether or not rows were appended
    # (:318-320), max_radii2D included: the screen-size test below therefore reads zeros, as it does in the
    # reference, and the returned statistics are zeros of the final size.
    N2 = P("xyz").shape[0]
    dev = P("xyz").device
    max_radii2D = torch.zeros((N2,), device=dev)
    # prune: the parents a split replaced (:388-392) and the transparent / oversized (:237-247), decided per
    # row, so one compaction serves both (the reference prunes twice; the surviving rows and their order are
    # the same)
    mask = prune_mask(P("opacity").data.view(-1), max_radii2D, P("scaling").data, min_opacity, max_screen_size, extent,
                      also=also)
    optimizable, comp = prune_optimizer(optimizer, ~mask)
    n = comp.kept
    new_stats = dict(xyz_gradient_accum=torch.zeros((n, 1), device=dev), denom=torch.zeros((n, 1), device=dev),
                     max_radii2D=torch.zeros((n,), device=dev))
    return optimizable, new_stats, comp(tmp_radii)
