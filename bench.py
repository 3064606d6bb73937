"""Benchmark of the rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one training iteration of the hot path on one view of a synthetic
6M-Gaussian street scene at 1920x1080 (SH degree 3, 4 `reps` channels,
antialiasing on): activations -> rasterizer forward -> L1 + fused-ssim loss ->
rasterizer backward, exactly the calls train.py:188-253 makes into the three
operator packages.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank renders its own camera of an N-view batch
from a full replica and the parameter gradients are all-reduced with RCCL:
weak scaling, value = views per second over the whole job.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      for the kernel that took the most time, from HIP events recorded
                inside the timed region on the launching stream;
  cpu_baseline  the CPU checker (oracle/) timed on this host's cores on a
                bounded sample of the same workload (N = 1 only);
  kernels       per-kernel mean milliseconds of the timed region;
  render_fps / sort_GBps   the other two parts of the headline metric.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, datasheet


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_params(sc, dev):
    """Raw (pre-activation) leaf parameters, as scene/gaussian_model.py keeps them."""
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    op = np.clip(sc["opacities"], 1e-4, 1 - 1e-4)
    return dict(
        xyz=t(sc["means3D"]).requires_grad_(True),
        features_dc=t(sc["shs"][:, :1]).contiguous().requires_grad_(True),
        features_rest=t(sc["shs"][:, 1:]).contiguous().requires_grad_(True),
        scaling=t(np.log(sc["scales"])).requires_grad_(True),
        rotation=t(sc["rotations"]).requires_grad_(True),
        opacity=t(np.log(op / (1 - op))).requires_grad_(True),
        reps=t(sc["reps"]).requires_grad_(True),
    )


def make_settings(cam, dev, sh_degree):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    return GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
        tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), sh_degree=sh_degree, campos=t(cam["campos"]), prefiltered=False, debug=False,
        antialiasing=True)


def render(params, rs):
    """gaussian_renderer/__init__.py:49-169: activations + rasterizer call."""
    from diff_gaussian_rasterization import GaussianRasterizer
    means2D = torch.zeros_like(params["xyz"], requires_grad=True)
    shs = torch.cat((params["features_dc"], params["features_rest"]), dim=1)
    out = GaussianRasterizer(rs)(
        means3D=params["xyz"], means2D=means2D, opacities=torch.sigmoid(params["opacity"]), shs=shs,
        scales=torch.exp(params["scaling"]), rotations=torch.nn.functional.normalize(params["rotation"]),
        reps=params["reps"])
    return out


def train_loss(out, gt, lambda_dssim=0.2, lambda_rep=0.01):
    """train.py:222-251 without the depth prior: L1 + (1 - ssim) + rep regulariser."""
    from fused_ssim import fused_ssim
    image = out[0].clamp(0, 1)
    l1 = torch.abs(image - gt).mean()
    ssim_v = fused_ssim(image.unsqueeze(0), gt.unsqueeze(0))
    loss = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim_v)
    return loss + lambda_rep * torch.abs(out[5] - 10.0).mean()


class KernelTimer:
    """HIP events recorded by the library around its kernels, one set per step."""

    def __init__(self, steps):
        import _gsplat_lib as G
        self.G = G
        self.n = G.NUM_PROFILE_EVENTS
        self.events = [[torch.cuda.Event(enable_timing=True) for _ in range(self.n)] for _ in range(steps)]
        for row in self.events:
            for e in row:
                e.record()  # creates the underlying hipEvent_t
        torch.cuda.synchronize()
        self.arrays = []
        for row in self.events:
            arr = (C.c_void_p * self.n)(*[e.cuda_event for e in row])
            self.arrays.append(arr)

    def arm(self, step):
        self.G.set_profile_events(self.arrays[step] if step is not None else None)

    def means_ms(self, fwd_only=False):
        pairs = {"preprocess": (0, 1), "scan": (1, 2), "depth_order": (3, 4), "emit_keys": (4, 5), "sort": (5, 6),
                 "tile_ranges": (6, 7), "render_fwd": (7, 8), "finish_stats": (8, 9)}
        if not fwd_only:
            pairs.update({"bwd_clear": (10, 11), "render_bwd": (11, 12), "preprocess_bwd": (12, 13)})
        out = {}
        for k, (a, b) in pairs.items():
            out[k] = float(np.mean([row[a].elapsed_time(row[b]) for row in self.events]))
        return out


def algorithmic_bytes(kernel, P, visible, R, HW, passes):
    """Bytes one launch must move at least, per the per-unit figures of the
    hot-path scope table (reference data layout)."""
    return {
        "preprocess": 315.0 * visible + 20.0 * (P - visible),
        "depth_order": P * (28.0 + 8.0 + 4 * 24.0),
        "emit_keys": 20.0 * P + 12.0 * R,
        "sort": R * (8.0 + passes * 24.0),
        "tile_ranges": 8.0 * R,
        "render_fwd": 28.0 * R + 48.0 * HW,
        "render_bwd": 60.0 * R + 40.0 * HW + 56.0 * R,
        "preprocess_bwd": 575.0 * visible,
    }.get(kernel)


def cpu_baseline(P_full, W, H, seed, stride=8):
    """The CPU checker on every `stride`-th Gaussian of the same scene and camera."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gs_oracle
    import synth
    sc = synth.street_scene(P_full, W, H, seed=seed)
    sub = {k: (v[::stride].copy() if isinstance(v, np.ndarray) else v) for k, v in sc.items()}
    gt = np.random.default_rng(0).random((3, H, W)).astype(np.float32)
    t0 = time.time()
    f = gs_oracle.forward_from_scene(sub, antialiasing=True)
    img = np.clip(f.out_color, 0, 1)[None]
    m, d1, d2, d3 = gs_oracle.ssim_fwd(img, gt[None])
    dl = np.full(img.shape, -0.2 / img.size, np.float32)
    g_ssim = gs_oracle.ssim_bwd(img, gt[None], dl, d1, d2, d3)[0]
    g_l1 = 0.8 * np.sign(img[0] - gt) / img[0].size
    f.backward((g_ssim + g_l1).astype(np.float32), None, np.full((4, H, W), 0.01 / (4 * H * W), np.float32))
    dt = time.time() - t0
    return {"value": 1.0 / (dt * stride), "unit": "it/s", "cores": gs_oracle.num_threads(), "kind": "port",
            "sample": f"every {stride}th Gaussian ({sub['means3D'].shape[0]} of {P_full}), same camera {W}x{H}, "
                      f"fwd+ssim+bwd took {dt:.2f} s (R={f.R}); value = 1/({stride} x seconds), linear in R"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=6_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true", help="time rendering only (render FPS)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}")
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import synth
    import gsdp
    P, W, H = args.gaussians, args.width, args.height
    seed = 2
    t0 = time.time()
    sc = synth.street_scene(P, W, H, seed=seed, cam_index=rank, n_cams=max(world, 1))
    params = build_params(sc, dev)
    rs = make_settings(sc["camera"], dev, sc["sh_degree"])
    gt = torch.rand((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(1 + rank))
    leaves = list(params.values())
    del sc
    log(f"[rank {rank}] scene ready in {time.time() - t0:.1f}s")

    def step(k=None, timer=None):
        if timer is not None:
            timer.arm(k)
        if args.fwd_only:
            with torch.no_grad():
                return render(params, rs)
        for p in leaves:
            p.grad = None
        out = render(params, rs)
        loss = train_loss(out, gt)
        loss.backward()
        if world > 1:
            gsdp.GradReducer().start([p.grad for p in leaves]).wait()
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync()
    timer = KernelTimer(args.steps)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(k, timer)
    sync()
    dt = time.perf_counter() - t0
    timer.arm(None)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kernels = timer.means_ms(args.fwd_only)
    visible = int((out[1] > 0).sum().item())
    # number of instances of the last step
    import _gsplat_lib as G
    R = None
    try:
        R = int(out[0].grad_fn.num_rendered) if out[0].grad_fn is not None else None
    except Exception:
        R = None
    if R is None:
        with torch.no_grad():
            from diff_gaussian_rasterization import _C as RC
            e = torch.Tensor([])
            shs = torch.cat((params["features_dc"], params["features_rest"]), dim=1)
            R = RC.rasterize_gaussians(rs.bg, params["xyz"], e, torch.sigmoid(params["opacity"]),
                                       torch.exp(params["scaling"]), torch.nn.functional.normalize(params["rotation"]),
                                       1.0, e, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, H, W, shs,
                                       rs.sh_degree, rs.campos, False, True, False, params["reps"])[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    passes = ((gx * gy).bit_length() + 7) // 8   # the instance sort only orders the tile bits

    # render FPS (forward only) and the sort rate, measured after the timed region
    render_fps = None
    if not args.fwd_only and rank == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nfr = 10
        with torch.no_grad():
            for _ in range(nfr):
                render(params, rs)
        torch.cuda.synchronize()
        render_fps = nfr / (time.perf_counter() - t1)

    if rank == 0:
        dom = max((k for k in kernels if algorithmic_bytes(k, P, visible, R, H * W, passes)), key=lambda k: kernels[k])
        ab = algorithmic_bytes(dom, P, visible, R, H * W, passes)
        achieved = ab / (kernels[dom] * 1e-3) / 1e9
        res = {
            "metric": "render FPS (forward only)" if args.fwd_only else "train it/s (hot path: fwd + L1/fused-ssim + bwd)",
            "value": world * args.steps / dt,
            "unit": "frames/s" if args.fwd_only else "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"street scene, {P} Gaussians SH3 + 4 reps, {W}x{H}, antialiasing, "
                                   f"1 view per GPU per step" + (", forward only" if args.fwd_only else
                                                                   ", fwd + fused-ssim loss + bwd, no optimizer"),
                       "gaussians": P, "visible": visible, "num_rendered": R, "image": [W, H],
                       "parallelism": f"view-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_launch": ab, "ms_per_launch": kernels[dom]},
            "kernels_ms": kernels,
            "sort_GBps": (R * 12.0 / (kernels["sort"] * 1e-3) / 1e9) if R else None,
            "render_fps": render_fps,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(P, W, H, seed)
            except Exception as ex:  # the checker is optional for the measurement itself
                res["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
