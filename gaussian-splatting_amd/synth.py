"""Deterministic synthetic scenes and cameras for tests and benchmarks.

Pure numpy (PCG64 streams are identical on every machine for one numpy
version), so the build container and the GPU box see the same bits.

Camera matrices follow the reference conventions:
  * world->view and projection are built as in utils/graphics_utils.py:38-71
    (getWorld2View2, getProjectionMatrix) and then transposed, i.e. stored
    "row-vector" style, scene/cameras.py:86-89;
  * full_proj = view @ proj, campos = inverse(view)[3, :3].
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """utils/graphics_utils.py:38-49 (getWorld2View2).  R is camera-to-world."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + np.asarray(translate)) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:51-71 (getProjectionMatrix)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def look_at(eye, target, up=(0.0, -1.0, 0.0)):
    """World->camera rotation rows (x right, y down, z forward) and translation."""
    eye = np.asarray(eye, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    Rwc = np.stack([r, d, f])  # rows
    t = -Rwc @ eye
    return Rwc.T, t  # (camera-to-world R as the reference stores it, t)


def make_camera(width, height, fovx, eye, target, znear=0.01, zfar=100.0):
    """Returns the dict of raster settings a Camera of the reference would give."""
    fovy = focal2fov(width / (2 * math.tan(fovx / 2)), height)
    R, t = look_at(eye, target)
    view = world2view(R, t).transpose().copy()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose().copy()
    full = (view @ proj).astype(np.float32)
    campos = np.linalg.inv(view)[3, :3].astype(np.float32).copy()
    return dict(
        image_width=int(width), image_height=int(height),
        tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
        viewmatrix=np.ascontiguousarray(view, dtype=np.float32),
        projmatrix=np.ascontiguousarray(full, dtype=np.float32),
        campos=campos,
    )


def _common_attrs(rng, P, sh_coeffs, dim):
    rot = rng.standard_normal((P, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.standard_normal(P)))
    opac = np.clip(opac, 0.02, 0.98).astype(np.float32)
    sh = np.zeros((P, sh_coeffs, 3), dtype=np.float32)
    sh[:, 0, :] = ((rng.random((P, 3)) - 0.5) / SH_C0).astype(np.float32)
    if sh_coeffs > 1:
        sh[:, 1:, :] = (0.05 * rng.standard_normal((P, sh_coeffs - 1, 3))).astype(np.float32)
    # the first four feature channels are the same numbers whatever the width
    reps = rng.standard_normal((P, 4)).astype(np.float32)
    if dim != 4:
        more = np.random.default_rng(977).standard_normal((P, max(dim, 4))).astype(np.float32)
        reps = np.concatenate([reps, more[:, :dim - 4]], axis=1) if dim > 4 else np.ascontiguousarray(reps[:, :dim])
    return rot, opac, sh, reps


def ball_scene(P=100_000, seed=1, sh_coeffs=16, dim=4, width=800, height=800):
    """NeRF-synthetic-like: points uniform in [-1.3,1.3]^3 (scene/dataset_readers.py:319),
    isotropic scales near the mean neighbour distance, camera on a sphere of
    radius 4.031 looking at the origin."""
    rng = np.random.default_rng(seed)
    xyz = ((rng.random((P, 3)) * 2.6) - 1.3).astype(np.float32)
    nn = 0.554 * (2.6 ** 3 / P) ** (1.0 / 3.0)
    s = nn * np.exp(0.3 * rng.standard_normal((P, 1)))
    scales = np.repeat(s, 3, axis=1).astype(np.float32)
    rot, opac, sh, reps = _common_attrs(rng, P, sh_coeffs, dim)
    az, el, rad = 0.7, 0.45, 4.031
    eye = (rad * math.cos(el) * math.sin(az), -rad * math.sin(el), -rad * math.cos(el) * math.cos(az))
    cam = make_camera(width, height, 0.6911112, eye, (0.0, 0.0, 0.0))
    return dict(means3D=xyz, scales=scales, rotations=rot, opacities=opac.reshape(P, 1), shs=sh, reps=reps,
                camera=cam, sh_degree=int(round(math.sqrt(sh_coeffs))) - 1)


def street_camera(width, height, cam_index=0, n_cams=1, fovx_deg=60.0):
    """Camera `cam_index` of `n_cams` on a 30 degree arc around the scene centre (0, 0, 20), looking at it."""
    ang = 0.0 if n_cams <= 1 else math.radians(-15.0 + 30.0 * cam_index / (n_cams - 1))
    eye = (20.0 * math.sin(ang), -0.3, 20.0 - 20.0 * math.cos(ang))
    return make_camera(width, height, math.radians(fovx_deg), eye, (0.0, 0.5, 20.0))


def street_scene(P, width, height, seed=2, sh_coeffs=16, dim=4, fovx_deg=60.0, cam_index=0, n_cams=1,
                 translucent=False, opacity_logit_mean=-2.0, scale_factor=0.5):
    """Outdoor-capture-like: 70 % of the points in a slab in front of the
    camera, 30 % Gaussian clutter around it; log-normal anisotropic scales.

    translucent: the same geometry with smaller and much more transparent splats
    (opacity = sigmoid(N(opacity_logit_mean, 1)), scales times scale_factor), as in a
    scene early in training or one made of thin structures: a pixel then blends
    many Gaussians before it saturates, so the blend kernels walk a large part of
    every tile list instead of the first few per cent."""
    rng = np.random.default_rng(seed)
    n_slab = int(0.7 * P)
    slab = np.empty((n_slab, 3))
    slab[:, 0] = rng.uniform(-12, 12, n_slab)
    slab[:, 1] = rng.uniform(-1.5, 3.0, n_slab)
    slab[:, 2] = rng.uniform(2.0, 40.0, n_slab)
    clutter = rng.standard_normal((P - n_slab, 3)) * 3.0
    clutter[:, 2] = np.abs(clutter[:, 2]) * 4.0 + 1.5
    xyz = np.concatenate([slab, clutter]).astype(np.float32)
    xyz = xyz[rng.permutation(P)]
    base = 0.02 * np.exp(0.8 * rng.standard_normal((P, 1)))
    scales = (base * np.exp(0.35 * rng.standard_normal((P, 3)))).astype(np.float32)
    rot, opac, sh, reps = _common_attrs(rng, P, sh_coeffs, dim)
    if translucent:
        rng2 = np.random.default_rng(seed + 1000)
        scales = (scales * np.float32(scale_factor)).astype(np.float32)
        opac = (1.0 / (1.0 + np.exp(-(opacity_logit_mean + rng2.standard_normal(P))))).astype(np.float32)
        opac = np.clip(opac, 0.005, 0.98).astype(np.float32)
    cam = street_camera(width, height, cam_index, n_cams, fovx_deg)
    return dict(means3D=xyz, scales=scales, rotations=rot, opacities=opac.reshape(P, 1), shs=sh, reps=reps,
                camera=cam, sh_degree=int(round(math.sqrt(sh_coeffs))) - 1)


def tiny_scene(P=64, width=40, height=24, seed=7, sh_coeffs=16, dim=4, spread=1.0, scale=0.18):
    """A handful of large overlapping splats: every code path in a few tiles."""
    rng = np.random.default_rng(seed)
    xyz = (rng.standard_normal((P, 3)) * spread).astype(np.float32)
    xyz[:, 2] = (rng.random(P) * 6.0 - 1.0).astype(np.float32)  # some behind the near plane
    scales = (scale * np.exp(0.5 * rng.standard_normal((P, 3)))).astype(np.float32)
    rot, opac, sh, reps = _common_attrs(rng, P, sh_coeffs, dim)
    cam = make_camera(width, height, 0.9, (0.1, -0.05, -2.5), (0.0, 0.0, 1.0))
    return dict(means3D=xyz, scales=scales, rotations=rot, opacities=opac.reshape(P, 1), shs=sh, reps=reps,
                camera=cam, sh_degree=int(round(math.sqrt(sh_coeffs))) - 1)


def deep_scene(P=480, width=56, height=40, seed=31, sh_coeffs=16, dim=4, high_share=0.15):
    """Long lists on a few tiles: most splats are nearly transparent (opacity 0.01 .. 0.06), so a pixel
    blends dozens of them, and every few entries comes one with opacity 0.9 .. 1.0 (alpha up to the
    0.99 clamp), the case in which the backward's T / (1 - alpha) amplifies an error in alpha most.
    Everything sits in front of the camera, depths interleaved."""
    sc = tiny_scene(P=P, width=width, height=height, seed=seed, sh_coeffs=sh_coeffs, dim=dim, spread=0.8, scale=0.10)
    rng = np.random.default_rng(seed + 500)
    sc["means3D"][:, 2] = (0.5 + 3.0 * rng.random(P)).astype(np.float32)
    op = (0.01 + 0.05 * rng.random(P)).astype(np.float32)
    high = rng.random(P) < high_share
    op[high] = (0.9 + 0.1 * rng.random(int(high.sum()))).astype(np.float32)
    sc["opacities"] = op.reshape(P, 1)
    return sc
