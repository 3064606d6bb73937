"""Independent check on the CPU checker: a dense float64 PyTorch restatement
of 3D Gaussian splatting written from the textbook formulas (EWA projection
J R Sigma R^T J^T, front-to-back alpha compositing), differentiated by autograd.
It shares no code with oracle/ or with the kernels, costs O(P * H * W), and is
only usable on tiny scenes.
"""
import math

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_color(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] unit."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R  # standard rotation matrix, rows as listed


def sym6(c):
    return torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                       dim=1).reshape(-1, 3, 3)


class _RefAAFactor(torch.autograd.Function):
    """sqrt(max(2.5e-5, det(cov) / det(cov + 0.3 I))) with the reference's
    backward (backward.cu:234-249), which evaluates the closed-form derivative
    at the DILATED diagonal entries.  The forward value is the textbook one."""

    @staticmethod
    def forward(ctx, a, b, c):
        h = 0.3
        ratio = (a * c - b * b) / ((a + h) * (c + h) - b * b)
        f = torch.sqrt(torch.clamp(ratio, min=0.000025))
        ctx.save_for_backward(a, b, c, ratio, f)
        return f

    @staticmethod
    def backward(ctx, g):
        a, b, c, ratio, f = ctx.saved_tensors
        w = 0.3
        x, y, z = a + w, c + w, b
        inside = torch.where(ratio <= 0.000025, torch.zeros_like(g), g / (2 * f))
        den = w * w + w * (x + y) + x * y - z * z
        k = inside / (den * den)
        return w * (w * y + y * y + z * z) * k, -2.0 * w * z * (w + x + y) * k, w * (w * x + x * x + z * z) * k


def dense_render(inp, cam, bg, sh_degree, antialiasing=False, scale_modifier=1.0, ref_quirks=False):
    """inp: dict of float64 tensors (leaf, requires_grad as wanted):
         means3D, means2D (zeros [P,3]), opacities [P,1], reps [P,DIM] or None,
         either shs [P,M,3] or colors_precomp [P,3],
         either scales+rotations or cov3D_precomp [P,6].
    ref_quirks=False gives the true derivative of the forward; ref_quirks=True
    reproduces the two places where the reference's hand-written backward
    deviates from it (frustum-clamped means, antialiasing factor).
    Returns dict(color [3,H,W], invdepth [1,H,W], reps [DIM,H,W], radii, n_contrib, final_T, gauss_sum, gauss_count,
                 tiles_touched)."""
    dt = torch.float64
    W, H = cam["image_width"], cam["image_height"]
    V = torch.tensor(np.asarray(cam["viewmatrix"], np.float64))
    PM = torch.tensor(np.asarray(cam["projmatrix"], np.float64))
    campos = torch.tensor(np.asarray(cam["campos"], np.float64))
    tanx, tany = float(np.float32(cam["tanfovx"])), float(np.float32(cam["tanfovy"]))
    fx, fy = W / (2 * tanx), H / (2 * tany)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    bg = torch.tensor(np.asarray(bg, np.float64))

    m = inp["means3D"]
    P = m.shape[0]
    hom = torch.cat([m, torch.ones(P, 1, dtype=dt)], dim=1)
    pv = hom @ V
    ph = hom @ PM
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * pw[:, None] + inp["means2D"][:, :2]
    depth = pv[:, 2]
    visible = depth > 0.2

    if inp.get("cov3D_precomp") is not None:
        Sigma = sym6(inp["cov3D_precomp"])
    else:
        R = quat_R(inp["rotations"])
        S = torch.diag_embed(inp["scales"] * scale_modifier)
        Mx = R @ S
        Sigma = Mx @ Mx.transpose(1, 2)

    tz = pv[:, 2]
    limx, limy = 1.3 * tanx, 1.3 * tany
    txc = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    if ref_quirks:
        # backward.cu:185-186, 313-315: a clamped coordinate is treated as a
        # constant, although lim * t.z still depends on t.z.
        okx = ((pv[:, 0] / tz) >= -limx) & ((pv[:, 0] / tz) <= limx)
        oky = ((pv[:, 1] / tz) >= -limy) & ((pv[:, 1] / tz) <= limy)
        txc = torch.where(okx, pv[:, 0], txc.detach())
        tyc = torch.where(oky, pv[:, 1], tyc.detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * txc / (tz * tz), zero, fy / tz, -fy * tyc / (tz * tz)], dim=1).reshape(-1, 2, 3)
    Rwc = V[:3, :3].T
    JW = J @ Rwc
    cov2 = JW @ Sigma @ JW.transpose(1, 2)
    a, b, c = cov2[:, 0, 0], cov2[:, 0, 1], cov2[:, 1, 1]
    det0 = a * c - b * b
    a = a + 0.3
    c = c + 0.3
    det = a * c - b * b
    if not antialiasing:
        aa = torch.ones_like(det)
    elif ref_quirks:
        aa = _RefAAFactor.apply(a - 0.3, b, c - 0.3)
    else:
        aa = torch.sqrt(torch.clamp(det0 / det, min=0.000025))
    conic = torch.stack([c / det, -b / det, a / det], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)
    pd = pix.detach()
    x0 = torch.clamp(torch.trunc((pd[:, 0] - radius) / 16), 0, gx).long()
    y0 = torch.clamp(torch.trunc((pd[:, 1] - radius) / 16), 0, gy).long()
    x1 = torch.clamp(torch.trunc((pd[:, 0] + radius + 15) / 16), 0, gx).long()
    y1 = torch.clamp(torch.trunc((pd[:, 1] + radius + 15) / 16), 0, gy).long()
    tiles = (x1 - x0) * (y1 - y0)
    alive = visible & (tiles > 0) & (det != 0)
    radii = torch.where(alive, radius, torch.zeros_like(radius)).int()
    tiles = torch.where(alive, tiles, torch.zeros_like(tiles))

    if inp.get("colors_precomp") is not None:
        col = inp["colors_precomp"]
    else:
        d = m - campos
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp_min(sh_color(sh_degree, inp["shs"], d) + 0.5, 0.0)
    opac = inp["opacities"][:, 0] * aa
    reps = inp.get("reps")
    DIM = 0 if reps is None else reps.shape[1]

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxf, pyf = xs.reshape(-1).to(dt), ys.reshape(-1).to(dt)
    ptx, pty = (xs // 16).reshape(-1), (ys // 16).reshape(-1)
    N = H * W
    T = torch.ones(N, dtype=dt)
    done = torch.zeros(N, dtype=torch.bool)
    Cc = torch.zeros(3, N, dtype=dt)
    Re = torch.zeros(DIM, N, dtype=dt)
    Dd = torch.zeros(N, dtype=dt)
    position = torch.zeros(N, dtype=torch.long)   # index within the pixel's tile list
    n_contrib = torch.zeros(N, dtype=torch.long)
    gsum = torch.zeros(P, dtype=dt)
    gcnt = torch.zeros(P, dtype=torch.long)

    # (depth as f32 bits, then index) is the reference's sort order within a tile
    d32 = depth.detach().to(torch.float32).numpy()
    order = np.lexsort((np.arange(P), d32))
    for g in order:
        if not bool(alive[g]):
            continue
        in_tile = (ptx >= x0[g]) & (ptx < x1[g]) & (pty >= y0[g]) & (pty < y1[g])
        position = position + in_tile.long()
        act = in_tile & ~done
        dx = pix[g, 0] - pxf
        dy = pix[g, 1] - pyf
        power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
        act = act & (power <= 0)
        gsum[g] = (T.detach() * act).sum()
        gcnt[g] = act.sum()
        alpha = torch.clamp(opac[g] * torch.exp(power), max=0.99)
        act = act & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        term = act & (test_T < 0.0001)
        done = done | term
        act = act & ~term
        w = torch.where(act, alpha * T, torch.zeros_like(T))
        Cc = Cc + col[g][:, None] * w[None, :]
        if DIM:
            Re = Re + reps[g][:, None] * w[None, :]
        Dd = Dd + (1.0 / depth[g]) * w
        T = torch.where(act, test_T, T)
        n_contrib = torch.where(act, position, n_contrib)
    color = Cc + T[None, :] * bg[:, None]
    return dict(color=color.reshape(3, H, W), invdepth=Dd.reshape(1, H, W), reps=Re.reshape(DIM, H, W),
                radii=radii, n_contrib=n_contrib.reshape(H, W), final_T=T.reshape(H, W), gauss_sum=gsum,
                gauss_count=gcnt, tiles_touched=tiles)
