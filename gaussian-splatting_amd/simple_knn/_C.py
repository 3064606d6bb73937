"""distCUDA2 (submodules/simple-knn/spatial.cu:15-26, ext.cpp) over the C ABI."""
import torch

from _gsplat_lib import check, lib, ptr, stream_of


def distCUDA2(points):
    """points [P,3] float32 on the GPU -> [P] mean squared distance to the 3
    nearest neighbours."""
    if not points.is_cuda:
        raise RuntimeError("points must be a GPU tensor (there is no CPU path)")
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    L = lib()
    dev = points.device
    P = points.size(0)
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=dev)
    if P == 0:
        return means
    with torch.cuda.device(dev):
        temp = torch.empty((L.gsk_dist2_temp_bytes(P),), dtype=torch.uint8, device=dev)
        check(L.gsk_dist2(P, ptr(pts), ptr(means), ptr(temp), temp.numel(), stream_of(dev)), "distCUDA2")
    return means
