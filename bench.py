"""Benchmark of the rasterizer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One iteration ("it") is what one iteration of the reference's train.py:188-253
does with the three operator packages on one view of a synthetic 6M-Gaussian
street scene at 1920x1080 (SH degree 3, 4 `reps` channels, antialiasing on):
activations -> rasterizer forward -> L1 + fused-ssim loss -> rasterizer
backward.  A step is a batch of V views per GPU (default 8, different cameras)
whose parameter gradients accumulate, followed with N > 1 by the RCCL sum
all-reduce of the 63 floats per Gaussian (of every Gaussian, or with --reduce
auto / sparse of the Gaussians that some view of the step, on any rank,
reached: the other rows are zero everywhere).  The forward preprocess and the
per-Gaussian backward of the V views each run as one batch kernel per step, and
the views in between are dealt to a few HIP streams (default 3) so that their
kernels overlap.  Every rank holds a full replica and renders its own cameras:
weak scaling, value = views per second over the whole job.  The step goes through `GaussianRasterizer.forward_params` (raw
parameters; activations, SH concatenation and gradient accumulation inside the
kernels); with N = 1 the same iteration through the unchanged reference
signature (torch activations + torch.cat in front of `forward`) is timed too
and reported as `reference_api_it_s`.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      for the kernel that took the most time, from HIP events recorded
                inside the timed region on the launching stream;
  kernels       per-kernel mean milliseconds and achieved algorithmic GB/s;
  cpu_baseline  the CPU checker (oracle/) timed on this host's cores on one
                view of the same workload (N = 1 only);
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
        opacity=t(np.log(op / (1 - op))).requires_grad_(True),
        scaling=t(np.log(sc["scales"])).requires_grad_(True),
        rotation=t(sc["rotations"]).requires_grad_(True),
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


def render_reference_api(params, rs):
    """gaussian_renderer/__init__.py:49-169: activations + rasterizer call, unchanged signature."""
    from diff_gaussian_rasterization import GaussianRasterizer
    means2D = torch.zeros_like(params["xyz"], requires_grad=True)
    shs = torch.cat((params["features_dc"], params["features_rest"]), dim=1)
    return GaussianRasterizer(rs)(
        means3D=params["xyz"], means2D=means2D, opacities=torch.sigmoid(params["opacity"]), shs=shs,
        scales=torch.exp(params["scaling"]), rotations=torch.nn.functional.normalize(params["rotation"]),
        reps=params["reps"])


def render_params(params, rs, sink=None):
    """The same render from the raw parameters (activations and concatenation in the kernels)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    means2D = torch.empty_like(params["xyz"], requires_grad=True)   # a gradient sink: its values are never read
    return GaussianRasterizer(rs).forward_params(
        params["xyz"], means2D, params["features_dc"], params["features_rest"], params["opacity"], params["scaling"],
        params["rotation"], params["reps"], grad_sink=sink)


def train_loss(out, gt, lambda_dssim=0.2, lambda_rep=0.01, fused=False):
    """train.py:222-251 without the depth prior: L1 + (1 - ssim) + rep regulariser.  fused: the photometric
    part through fused_ssim.l1_ssim_loss (clamp, L1, ssim and their backward in three kernels); otherwise as
    the reference composes it from PyTorch operations around fused_ssim."""
    from fused_ssim import fused_ssim, l1_ssim_loss
    if fused:
        loss = l1_ssim_loss(out[0], gt, lambda_dssim)
    else:
        image = out[0].clamp(0, 1)
        l1 = torch.abs(image - gt).mean()
        ssim_v = fused_ssim(image.unsqueeze(0), gt.unsqueeze(0))
        loss = (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ssim_v)
    if fused:
        from fused_ssim import l1_loss
        return loss + lambda_rep * l1_loss(out[5], 10.0)
    return loss + lambda_rep * torch.abs(out[5] - 10.0).mean()


class KernelTimer:
    """HIP events recorded by the library around its kernels, one set per rendered view."""

    PAIRS = {"preprocess": (0, 1), "scan": (1, 2), "host_sync_gap": (2, 3), "tile_count": (3, 18), "depth_order": (18, 4), "emit_keys": (4, 5), "sort": (5, 6),
             "tile_ranges": (6, 7), "render_fwd": (7, 8), "finish_stats": (8, 9), "bwd_clear": (10, 11),
             "render_bwd": (11, 12), "preprocess_bwd": (12, 13), "preprocess_bwd_batch": (14, 15),
             "preprocess_batch": (16, 17), "depth_order_batch": (19, 20)}
    # one launch (one set of launches) per step, recorded in one event set of it
    PER_STEP = ("preprocess_bwd_batch", "preprocess_batch", "depth_order_batch")

    def __init__(self, sets):
        import _gsplat_lib as G
        self.G = G
        self.n = G.NUM_PROFILE_EVENTS
        self.events = [[torch.cuda.Event(enable_timing=True) for _ in range(self.n)] for _ in range(sets)]
        for row in self.events:
            for e in row:
                e.record()  # creates the underlying hipEvent_t
        torch.cuda.synchronize()
        self.arrays = [(C.c_void_p * self.n)(*[e.cuda_event for e in row]) for row in self.events]

    def arm(self, i):
        self.G.set_profile_events(self.arrays[i] if i is not None else None)

    def means_ms(self, fwd_only=False, batch_sets=None, fwd_batch_sets=None):
        """batch_sets / fwd_batch_sets: indices of the event sets that were armed when the batched
        backward / the batched forward preprocess ran."""
        out = {}
        for k, (a, b) in self.PAIRS.items():
            if fwd_only and 10 <= a < 16:
                continue
            rows = self.events
            if k in self.PER_STEP:
                sets = batch_sets if k == "preprocess_bwd_batch" else fwd_batch_sets
                if not sets:
                    continue
                rows = [self.events[i] for i in sets]
            out[k] = float(np.mean([row[a].elapsed_time(row[b]) for row in rows]))
        return out


def deferred_flag(use_params, sink, args):
    return bool(use_params and sink is not None and sink.defer and not args.fwd_only)


def algorithmic_bytes(kernel, P, visible, R, HW, passes, V=1, walked=None, from_rects=True, reached=None,
                      reached_any=None, packed=False):
    """Bytes one launch must move at least: per-unit figures times the units one
    launch processes.  Preprocess and blend figures of the one-view kernels are
    those of the reference data layout; the binning stages and the two batch
    kernels (one launch for the V views of a step) are counted in this
    implementation's own layout, which is less than V times the one-view
    figure: that saving is what they are for.  The units of the two blend
    kernels are the list entries a launch walks (`walked`: per tile up to the
    last instance one of its pixels blended), not all R sorted instances: at
    this depth complexity a tile is saturated after a few per cent of its list."""
    walked = R if walked is None else walked
    reached = visible if reached is None else reached              # mean over the views: Gaussians a pixel reached
    reached_any = visible if reached_any is None else reached_any  # ... in at least one of the step's views
    return {
        "preprocess": 315.0 * visible + 20.0 * (P - visible),
        "depth_order": P * (16.0 + 8.0 + 4.0 + 4 * 16.0),       # bin read, pair write, histogram read, 4 passes
        # the same for the V views of a step in shared launches, on 27-bit keys: three passes
        "depth_order_batch": V * P * (16.0 + 8.0 + 4.0 + 3 * 16.0),
        # gather + sbin write + sbin read; an instance leaves as one word (low tile digit | Gaussian index)
        # when the indices fit below the digit, else as a (tile id, index) pair
        "emit_keys": P * (4.0 + 16.0 + 16.0 + 16.0) + (4.0 if packed else 8.0) * R,
        # the digit histograms and the tile ranges come from the tile rectangles (16 B per Gaussian) when the
        # tile grid fits one workgroup's LDS; otherwise from a histogram sweep and a range scan over the keys
        "tile_count": P * 16.0 if from_rects else None,
        # two-digit tile ids with rectangle counts: the emission partitions by the high digit, one pass is left
        # (packed instances: one word read, the Gaussian index written)
        "sort": R * ((0.0 if from_rects else 4.0) + (1 if from_rects and passes == 2 else passes) *
                     (8.0 if packed else 16.0)),
        "tile_ranges": None if from_rects else 4.0 * R,
        "render_fwd": 28.0 * walked + 48.0 * HW,
        "render_bwd": 60.0 * walked + 40.0 * HW + 56.0 * walked,
        "preprocess_bwd": 575.0 * visible,
        # parameters (60 B) and SH row (192 B) read once; per view the 64-byte gradient record, radius and clamp
        # flags in, the screen-space gradient (12 B) out; the 63 parameter gradients written once
        # (the kernel skips, per view, the Gaussians no pixel reached: 12 bytes of radius and hit count are all it
        # reads of them; parameters and SH row only of those reached in some view)
        "preprocess_bwd_batch": reached_any * (60.0 + 192.0) + V * (reached * 65.0 + P * 12.0 + P * 12.0) + P * 252.0,
        # parameters and SH row read once; per view the 64-byte record, depth and clamp flags of the rendered
        # Gaussians and radius, tile rectangle and statistics slot (28 B) of all of them out
        "preprocess_batch": P * 60.0 + visible * 192.0 + V * (visible * 69.0 + P * 28.0),
    }.get(kernel)


# What bounds each kernel (the blend kernels stop early per pixel and are limited by
# VALU / LDS issue, not by the bytes they move; the others stream through HBM).
BOUND = {"tile_count": "hbm", "preprocess_bwd_batch": "hbm", "preprocess_batch": "hbm", "preprocess": "hbm", "depth_order": "hbm", "depth_order_batch": "hbm", "emit_keys": "hbm", "sort": "hbm", "tile_ranges": "hbm",
         "render_fwd": "valu+lds", "render_bwd": "valu+lds+atomics", "preprocess_bwd": "hbm"}


def cpu_baseline(P_full, W, H, seed, translucent=False):
    """The CPU checker on one whole view of the same scene (camera 0)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gs_oracle
    import synth
    sc = synth.street_scene(P_full, W, H, seed=seed, translucent=translucent)
    gt = np.random.default_rng(0).random((3, H, W)).astype(np.float32)
    t0 = time.time()
    f = gs_oracle.forward_from_scene(sc, antialiasing=True)
    img = np.clip(f.out_color, 0, 1)[None]
    m, d1, d2, d3 = gs_oracle.ssim_fwd(img, gt[None])
    dl = np.full(img.shape, -0.2 / img.size, np.float32)
    g_ssim = gs_oracle.ssim_bwd(img, gt[None], dl, d1, d2, d3)[0]
    g_l1 = 0.8 * np.sign(img[0] - gt) / img[0].size
    f.backward((g_ssim + g_l1).astype(np.float32), None, np.full((4, H, W), 0.01 / (4 * H * W), np.float32))
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "it/s", "cores": gs_oracle.num_threads(), "kind": "port",
            "sample": f"one whole view (camera 0, {P_full} Gaussians, {W}x{H}, R={f.R}): forward + ssim + backward "
                      f"of the OpenMP checker took {dt:.2f} s"}


def sort_u64_standalone(n, tile_bits, dev, reps=5):
    """gsr_sort_pairs_u64_u32 on n (tile << 32 | depth bits, Gaussian index) pairs as the reference's
    duplicateWithKeys emits them (rasterizer_impl.cu:93-104, sorted on 32 + tile_bits bits at :373-380):
    milliseconds per sort (HIP events on the launching stream), payload GB/s = n * 12 B / t, and the
    sort's own traffic rate with 24 B per pair and pass plus the histogram read."""
    import _gsplat_lib as G
    L = G.lib()
    g = torch.Generator(device=dev).manual_seed(7)
    tiles = torch.randint(0, 1 << tile_bits, (n,), device=dev, generator=g, dtype=torch.int64)
    depth = (torch.rand(n, device=dev, generator=g) * 40.0 + 0.2).view(torch.int32).to(torch.int64)
    keys_src = (tiles << 32) | depth
    vals_src = torch.arange(n, device=dev, dtype=torch.int32)
    k0, v0 = torch.empty_like(keys_src), torch.empty_like(vals_src)
    k1, v1 = torch.empty_like(keys_src), torch.empty_like(vals_src)
    temp = torch.empty(L.gsr_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
    res = C.c_int(0)
    end_bit = 32 + tile_bits
    st = torch.cuda.current_stream().cuda_stream
    ms = []
    for i in range(reps + 1):
        k0.copy_(keys_src)
        v0.copy_(vals_src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        G.check(L.gsr_sort_pairs_u64_u32(k0.data_ptr(), v0.data_ptr(), k1.data_ptr(), v1.data_ptr(), n, end_bit,
                                         temp.data_ptr(), temp.numel(), st, C.byref(res)), "sort")
        e1.record()
        torch.cuda.synchronize()
        if i:
            ms.append(e0.elapsed_time(e1))
    ko = k0 if res.value == 0 else k1
    mask = (1 << end_bit) - 1
    assert bool(((ko[1:] & mask) >= (ko[:-1] & mask)).all()), "stand-alone sort: output not sorted"
    err = temp[L.gsr_sort_temp_error_offset():L.gsr_sort_temp_error_offset() + 4].view(torch.int32)
    assert int(err.item()) == 0, "stand-alone sort: look-back error word set"
    t = float(np.mean(ms))
    passes = (end_bit + 7) // 8
    return {"pairs": n, "key_bits": end_bit, "passes": passes, "ms": t, "payload_GBps": n * 12.0 / (t * 1e-3) / 1e9,
            "hbm_GBps": n * (8.0 + passes * 24.0) / (t * 1e-3) / 1e9,
            "frac_of_hbm_peak": n * (8.0 + passes * 24.0) / (t * 1e-3) / 1e9 / HBM_PEAK_GBPS}


def roofline_entry(dom, ent, achieved_hbm, ab, traffic, ms):
    """The dominant kernel's roofline.  A blend kernel moves a few per cent of what HBM could in its time:
    what bounds it is vector-instruction issue, so that is the roofline reported for it (with the HBM
    floor beside it); the streaming kernels are priced against HBM."""
    hbm = {"achieved": achieved_hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved_hbm / HBM_PEAK_GBPS}
    if BOUND.get(dom, "hbm").startswith("valu") and "valu" in ent:
        v = ent["valu"]
        # note: this fraction is the share of the issue slots the kernel's own instruction stream fills; an
        # optimisation that removes instructions lowers it together with the time (compare ms_per_launch and
        # wave_insts_per_launch between builds)
        return {"kernel": dom, "bound": "valu", "achieved": v["achieved_Ginst_s"], "peak": v["peak_Ginst_s"],
                "unit": "G wave64-instructions/s", "frac": v["frac"], "traffic": traffic,
                "wave_insts_per_launch": v["wave_insts_per_launch"], "simd_busy_frac": v.get("simd_busy_frac"),
                "algorithmic_bytes_per_launch": ab, "ms_per_launch": ms, "limited_by": BOUND.get(dom),
                "hbm_floor": hbm, "source": "SQ_INSTS_VALU per launch from the committed rocprofv3 --pmc passes of this "
                                            "command on this scene (profiles/pmc_traffic*.json), duration live from HIP events"}
    return dict({"kernel": dom, "bound": "hbm", "traffic": traffic, "algorithmic_bytes_per_launch": ab,
                 "ms_per_launch": ms, "limited_by": BOUND.get(dom)}, **hbm)


def pmc_file(scene):
    return os.path.join(ROOT, "profiles", "pmc_traffic.json" if scene == "street" else f"pmc_traffic_{scene}.json")


def pmc_is_stale(P, W, H, scene="street"):
    """True when the committed counter summary was collected on a library built from other sources than
    the ones at hand (or says nothing about its build): its instruction counts and traffic then describe
    other kernels, and no roofline fraction is derived from them."""
    try:
        from _gsplat_lib import source_hash
        with open(pmc_file(scene)) as f:
            d = json.load(f)
        if d.get("_workload") != [P, W, H]:
            return False          # no summary for this workload: nothing is used
        return d.get("_build") != source_hash()
    except Exception:
        return False


def load_pmc_valu(P, W, H, scene="street"):
    """Per kernel: wave64 vector instructions per launch, LDS atomics per launch and the shader clock, from
    the committed PMC summary of this command (empty if none matches the workload)."""
    path = pmc_file(scene)
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("_workload") != [P, W, H]:
            return {}
        return d.get("_valu", {})
    except Exception:
        return {}


def load_pmc_traffic(P, W, H, scene="street"):
    """HBM bytes per launch from the rocprofv3 --pmc passes of this command, if a summary was
    committed for this workload (the counters cannot be read from inside the process)."""
    path = pmc_file(scene)
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("_workload") != [P, W, H]:
            return {}
        return {k: v for k, v in d.items() if not k.startswith("_")}
    except Exception:
        return {}


def launch_ranks(n):
    """Start `n` copies of this command as ranks 0..n-1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, rendezvous on 127.0.0.1), wait for them, return the worst exit code.  If a rank fails the
    others are ended (by their PIDs).  With fewer GPUs than ranks (a rehearsal on a smaller box) the ranks
    share the devices and exchange through gloo, because RCCL refuses two ranks on one device."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    ndev = torch.cuda.device_count()
    if ndev < n and "GSPLAT_BENCH_BACKEND" not in env:
        log(f"launcher: {ndev} GPU(s) for {n} ranks: the ranks share devices and exchange through gloo (rehearsal)")
        env["GSPLAT_BENCH_BACKEND"] = "gloo"
    # Only the result line may reach this process's stdout: what else the ranks or their libraries print
    # there (gloo announces its connections on stdout) goes to stderr.
    import threading

    def forward(stream):
        for line in stream:
            out = sys.stdout if line.startswith('{"metric"') else sys.stderr
            out.write(line)
            out.flush()

    procs, readers = [], []
    for r in range(n):
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                             env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True)
        t = threading.Thread(target=forward, args=(p.stdout,), daemon=True)
        t.start()
        procs.append(p)
        readers.append(t)
    worst = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                rc = p.wait(timeout=1.0)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                log(f"launcher: a rank exited with code {rc}; ending the others")
                for q in alive:
                    q.terminate()
    for t in readers:
        t.join(timeout=10.0)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views-per-gpu", type=int, default=8, help="views rendered per GPU per step (gradient accumulation)")
    ap.add_argument("--scene", choices=("street", "translucent"), default="street",
                    help="street: the headline workload (opaque splats: a tile saturates after ~3 %% of its list); "
                         "translucent: same geometry, splats half the size and mostly transparent, so that the blend "
                         "kernels walk ~40 %% of every tile list (reported as extras.translucent by the default run)")
    ap.add_argument("--gaussians", type=int, default=6_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--dim", type=int, default=4, choices=(4, 8, 16),
                    help="feature (reps) channels per Gaussian; the reference's build has 4 (config.h:16), its CLI default is 16")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--api", choices=("params", "reference"), default="params",
                    help="params: forward_params (raw parameters); reference: unchanged reference signature")
    ap.add_argument("--no-batch-forward", action="store_true",
                    help="run the forward preprocess per view instead of once per step for all its views")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the views of a step are dealt to round-robin, the current stream being one of them "
                         "(kernels of different views overlap; 4 = the hardware queues a process has); 1 = one view "
                         "after the other")
    ap.add_argument("--torch-loss", action="store_true",
                    help="compose the photometric loss from PyTorch operations around fused_ssim, as train.py does, "
                         "instead of fused_ssim.l1_ssim_loss (the reference-signature runs always do)")
    ap.add_argument("--rewrite-zero-rows", action="store_true",
                    help="ParamGradSink without keep_zero_rows: every gradient row is written in every step, also "
                         "the all-zero rows of Gaussians no view reached that were zero before")
    ap.add_argument("--no-defer", action="store_true",
                    help="run the per-Gaussian backward after every view instead of once per step for all its views")
    ap.add_argument("--reduce-chunks", type=int, default=6,
                    help="n_gpus > 1: ranges of Gaussians the per-Gaussian backward runs over, the all-reduce of each "
                         "range's gradient rows starting as soon as its launch is enqueued (1 = one all-reduce after it)")
    ap.add_argument("--reduce", choices=("auto", "dense", "sparse"), default="auto",
                    help="n_gpus > 1: what the gradient exchange carries.  dense: every row; sparse: the rows of the "
                         "Gaussians some view of the step (any rank) reached; auto: sparse while those are at most "
                         "35 %% of the rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timeline", action="store_true",
                    help="print (stderr) when each stage of each view of the last timed step began and ended on the GPU, "
                         "in ms from the start of that step: how the views on their streams overlap")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-view / reference-API / FPS measurements")
    ap.add_argument("--fwd-only", action="store_true", help="time rendering only (render FPS)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly (`python bench.py --gpus N`): become the launcher of N ranks, one per GPU.  Nothing in
        # this process has touched the GPU yet (device_count() does not initialise it); the ranks are child
        # processes and this one only waits for them and passes rank 0's JSON line through.
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}")
    # one rank per GPU; for rehearsing the multi-rank path on a box with fewer GPUs the ranks wrap around the
    # devices and GSPLAT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    dev = torch.device(f"cuda:{local_rank % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    backend = os.environ.get("GSPLAT_BENCH_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import synth
    import gsdp
    from diff_gaussian_rasterization import ParamGradSink
    P, W, H, V = args.gaussians, args.width, args.height, max(1, args.views_per_gpu)
    seed = 2
    t0 = time.time()
    sc = synth.street_scene(P, W, H, seed=seed, translucent=args.scene == "translucent", dim=args.dim)
    params = build_params(sc, dev)
    sh_degree = sc["sh_degree"]
    del sc
    n_cams = world * V
    views = []
    for v in range(V):
        ci = rank * V + v
        cam = synth.street_camera(W, H, ci, n_cams)
        gt = torch.rand((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(1 + ci))
        views.append((make_settings(cam, dev, sh_degree), gt))
    leaves = list(params.values())
    use_params = args.api == "params"
    fused_loss = use_params and not args.torch_loss
    sink = ParamGradSink(*leaves, defer=not args.no_defer, keep_zero_rows=not args.rewrite_zero_rows) \
        if use_params else None
    log(f"[rank {rank}] scene ready in {time.time() - t0:.1f}s")

    def one_view(rs, gt):
        if args.fwd_only:
            with torch.no_grad():
                return render_params(params, rs, sink) if use_params else render_reference_api(params, rs)
        out = render_params(params, rs, sink) if use_params else render_reference_api(params, rs)
        train_loss(out, gt, fused=fused_loss).backward()
        return out

    reducer_ms = []
    side_streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams - 1)] if args.streams > 1 else []
    batch_fwd = use_params and not args.no_batch_forward and V > 1

    host_wait = [0.0]

    reducer = gsdp.GradReducer(sparse={"auto": "auto", "dense": "never", "sparse": "always"}[args.reduce])
    reduce_modes = []
    exchange = {"reducer": reducer, "ms": reducer_ms, "modes": reduce_modes}   # what step() exchanges through

    def step(k=None, timer=None, overlap=True):
        # the views are dealt to the current stream and the side streams in turn (only when the per-Gaussian
        # backward is deferred to the flush: otherwise every view's backward writes the shared gradient buffers)
        side = ([torch.cuda.current_stream()] + side_streams) if (overlap and side_streams and
                                                                  (sink is None or sink.defer or args.fwd_only)) else []
        if use_params:
            sink.begin_step()
        else:
            for p in leaves:
                p.grad = None
        out = None
        if batch_fwd:
            if timer is not None:
                timer.arm(k * V)
            tp = time.perf_counter()
            sink.prepare_views([rs for rs, _ in views])   # the forward preprocess of all V views, once
            host_wait[0] += time.perf_counter() - tp      # the one call of a step in which the host blocks
        if len(side) > 1:
            ready = torch.cuda.Event()
            ready.record()
            for st in side[1:]:
                st.wait_event(ready)
        for v, (rs, gt) in enumerate(views):
            if timer is not None:
                timer.arm(k * V + v)
            if len(side) > 1:
                with torch.cuda.stream(side[v % len(side)]):
                    out = one_view(rs, gt)
            else:
                out = one_view(rs, gt)
        if len(side) > 1:
            for st in side[1:]:
                torch.cuda.current_stream().wait_stream(st)
        if args.fwd_only:
            return out
        if world == 1:
            if use_params:
                sink.flush()   # deferred mode: the per-Gaussian backward of all V views, once
            return out
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        if use_params:
            # Few Gaussians reached by the views of this step (all ranks): only their gradient rows are
            # exchanged.  Otherwise the per-Gaussian backward runs range by range; each range's rows go to
            # RCCL (its own stream) as soon as the launch is enqueued and are exchanged while the later ranges
            # are computed.  e1: the compute stream's own work of the step ends, from there it waits for the
            # exchange.
            mode = gsdp.reduce_flush(sink, exchange["reducer"], chunks=args.reduce_chunks, after_flush=e1.record)
        else:
            exchange["reducer"].start([p.grad for p in leaves])
            e1.record()
            exchange["reducer"].wait()
            mode = "dense"
        e2.record()
        if k is not None:
            exchange["ms"].append((e0, e1, e2))
            exchange["modes"].append(mode)
        return out

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warm-up steps, run exactly like the timed ones (event sets armed: they are recorded again
    # later).  The allocator and the HIP runtime reach their steady state in the third step of a process and
    # the first step that records the profiling events pays a one-off cost, so at least three are run.
    timer = KernelTimer(args.steps * V)
    for _ in range(max(args.warmup, 3)):
        out = step(0, timer)
    reducer_ms.clear()
    reduce_modes.clear()
    sync()
    host_wait[0] = 0.0
    t0 = time.perf_counter()
    marks = []
    for k in range(args.steps):
        out = step(k, timer)
        marks.append(time.perf_counter())   # host-side launch progress (no sync): shows when the host is the limit
    sync()
    dt = time.perf_counter() - t0
    host_wait_ms = host_wait[0] / args.steps * 1e3
    log(f"[rank {rank}] host launch time per step (ms): " +
        " ".join(f"{(b - a) * 1e3:.1f}" for a, b in zip([t0] + marks[:-1], marks)))
    timer.arm(None)
    if args.timeline and rank == 0:
        k = args.steps - 1
        rows = timer.events[k * V:(k + 1) * V]
        origin = rows[0][16] if batch_fwd else rows[0][0]
        at = lambda row, slot: origin.elapsed_time(row[slot])
        if batch_fwd:
            log("timeline (ms): batched preprocess %.2f-%.2f, batched depth order %.2f-%.2f" %
                (at(rows[0], 16), at(rows[0], 17), at(rows[0], 19), at(rows[0], 20)))
        for v, row in enumerate(rows):
            lane = v % max(1, args.streams)
            log("timeline view %d lane %d: binning %.2f-%.2f  render_fwd %.2f-%.2f  stats %.2f  | backward %.2f-%.2f "
                "(render_bwd %.2f-%.2f)" % (v, lane, at(row, 3), at(row, 6), at(row, 7), at(row, 8), at(row, 9),
                                            at(row, 10), at(row, 12), at(row, 11), at(row, 12)))
        if deferred_flag(use_params, sink, args):
            log("timeline: batched per-Gaussian backward %.2f-%.2f" % (at(rows[V - 1], 14), at(rows[V - 1], 15)))
    if world > 1:
        tmax = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # The same step with every gradient row exchanged ("dense"), when the timed region did not already: on a
    # scene whose views reach few Gaussians the default policy exchanges a few MB, and the scaling that shows
    # says nothing about a model whose every Gaussian is seen.  A few steps, timed the same way (barrier +
    # synchronize on both sides, max over the ranks).
    dense_leg = None
    if world > 1 and not args.fwd_only and use_params and args.reduce != "dense":
        dense_steps = max(2, min(args.steps, 6))
        exchange.update(reducer=gsdp.GradReducer(sparse="never"), ms=[], modes=[])
        step(0, None)
        exchange["ms"].clear()
        exchange["modes"].clear()
        sync()
        t1 = time.perf_counter()
        for k in range(dense_steps):
            step(k, None)
        sync()
        ddt = time.perf_counter() - t1
        tmax = torch.tensor([ddt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        ddt = float(tmax.item())
        dense_leg = {"steps": dense_steps, "ms_per_step": ddt / dense_steps * 1e3,
                     "it_s": world * V * dense_steps / ddt,
                     "flush_ms": float(np.mean([a.elapsed_time(b) for a, b, _ in exchange["ms"]])),
                     "exposed_ms": float(np.mean([b.elapsed_time(c) for _, b, c in exchange["ms"]])),
                     "steps_dense": exchange["modes"].count("dense")}
        exchange.update(reducer=reducer, ms=reducer_ms, modes=reduce_modes)

    # one all-reduce of the whole gradient buffer on its own (every rank takes part)
    allreduce_alone_ms = None
    if world > 1 and not args.fwd_only:
        buf = sink.flat if use_params else torch.zeros(sum(p.numel() for p in leaves), device=dev)
        ts = []
        for _ in range(4):
            buf.zero_()
            sync()
            t1 = time.perf_counter()
            gsdp.GradReducer().start([buf]).wait()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
        allreduce_alone_ms = float(np.mean(ts[1:]))

    collective_version = None
    if world > 1 and backend == "nccl":
        try:
            collective_version = "RCCL (torch.cuda.nccl.version) " + ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as ex:
            collective_version = "unknown: " + repr(ex)

    deferred = use_params and sink.defer and not args.fwd_only

    def kernel_means(tm, nsteps):
        km = tm.means_ms(args.fwd_only, [k * V + V - 1 for k in range(nsteps)] if deferred else None,
                         [k * V for k in range(nsteps)] if batch_fwd else None)
        if batch_fwd:   # these slots are empty when the views were preprocessed (and their prefix sums run) together
            for k in ("preprocess", "scan", "host_sync_gap"):
                km.pop(k)
        return km

    # Per-kernel durations.  With the views on several streams the kernels of different views overlap in the
    # timed region and an event pair there spans other views' kernels too: those numbers are kept
    # (`kernels_in_timed_region_ms`), and the durations that the roofline figures use are measured, live and on
    # the same workload, in a pass with one view after the other that follows the timed region.
    overlapping = len(side_streams) >= 1
    kernels_overlapped, one_stream_it_s = None, None
    if overlapping:
        kernels_overlapped = kernel_means(timer, args.steps)
        iso_steps = min(args.steps, 5)
        timer = KernelTimer(iso_steps * V)
        exchange.update(ms=[], modes=[])      # the exchange figures are those of the timed region only
        step(0, timer, overlap=False)
        sync()
        t1 = time.perf_counter()
        for k in range(iso_steps):
            step(k, timer, overlap=False)
        sync()
        one_stream_it_s = world * V * iso_steps / (time.perf_counter() - t1)
        exchange.update(ms=reducer_ms, modes=reduce_modes)
        timer.arm(None)
        kernels = kernel_means(timer, iso_steps)
    else:
        kernels = kernel_means(timer, args.steps)
    if deferred:
        kernels["means2d_grad"] = kernels.pop("preprocess_bwd")   # in deferred mode that slot times the gradient copy
    gx, gy = (W + 15) // 16, (H + 15) // 16
    passes = ((gx * gy).bit_length() + 7) // 8   # the instance sort only orders the tile bits
    from_rects = ((gx + 1) * (gy + 1) + passes * 256 + 17) * 4 <= 150 * 1024   # as tile_counts_fit in the library
    tile_bits = (gx * gy).bit_length()
    low_digit_bits = tile_bits // 2 + tile_bits % 2
    packed = from_rects and passes == 2 and P <= (1 << (32 - low_digit_bits))   # as fused_sort_packs in the library

    # workload counters of this rank's views (outside the timed region)
    vis_l, R_l, pairs_l, reached_l = [], [], [], []
    reached_any_t = torch.zeros(P, dtype=torch.bool, device=dev)
    with torch.no_grad():
        for rs, _ in views:
            o = render_params(params, rs)
            vis_l.append(int((o[1] > 0).sum().item()))
            pairs_l.append(int(o[4].sum(dtype=torch.int64).item()))   # (pixel, Gaussian) pairs the forward blend tested
            hit = (o[4] > 0) & (o[1] > 0)                             # Gaussians some pixel reached in this view
            reached_l.append(int(hit.sum().item()))
            reached_any_t |= hit
        from diff_gaussian_rasterization import _C as RC
        e = torch.Tensor([])
        import ctypes
        from _gsplat_lib import GsrLayout, lib as gs_lib
        walked_l = []
        for rs, _ in views:
            o = RC.rasterize_gaussians(
                rs.bg, params["xyz"], e, params["opacity"], params["scaling"], params["rotation"], 1.0, e,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, H, W, params["features_dc"], rs.sh_degree,
                rs.campos, False, True, False, params["reps"], sh_rest=params["features_rest"], raw_params=True)
            R_l.append(int(o[0]))
            # list entries a tile walks: up to the last instance any of its pixels blended (what the backward blend
            # stages; the forward stops within one staging round of it).  The rest of each list is sorted but never read.
            lay = GsrLayout()
            assert gs_lib().gsr_layout(P, W, H, R_l[-1], ctypes.byref(lay)) == 0
            n = o[5][lay.n_contrib:lay.n_contrib + W * H * 4].view(torch.int32).reshape(H, W)
            n = torch.nn.functional.pad(n, (0, gx * 16 - W, 0, gy * 16 - H))
            walked_l.append(int(n.reshape(gy, 16, gx, 16).amax(dim=(1, 3)).sum(dtype=torch.int64).item()))
    visible, R, pairs = float(np.mean(vis_l)), float(np.mean(R_l)), float(np.mean(pairs_l))
    walked = float(np.mean(walked_l))
    reached, reached_any = float(np.mean(reached_l)), float(reached_any_t.sum().item())

    extras = {}
    if rank == 0 and world == 1 and not args.fwd_only and not args.no_extras:
        rs0, gt0 = views[0]

        def timed(fn, n):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return n / (time.perf_counter() - t1)

        def it_params():
            sink.begin_step() if use_params else None
            o = render_params(params, rs0, sink)
            train_loss(o, gt0, fused=fused_loss).backward()
            sink.flush()

        def it_reference():
            for p in leaves:
                p.grad = None
            train_loss(render_reference_api(params, rs0), gt0).backward()

        def fps_params():
            with torch.no_grad():
                render_params(params, rs0)

        def fps_reference():
            with torch.no_grad():
                render_reference_api(params, rs0)

        def reference_breakdown(n=10):
            """Where an iteration through the unchanged reference signature spends its time, from events
            on the stream: the model's activations and torch.cat (scene/gaussian_model.py:470-502), the
            rasterizer forward (the extension call: allocations, kernels and the one host block), the loss as
            train.py:222-251 composes it, its backward, and the backward through the rasterizer, split into
            the library's own kernels (its profile events) and what autograd does around them (activation
            backward, the split of the concatenated SH gradient, accumulation into .grad)."""
            from diff_gaussian_rasterization import GaussianRasterizer
            tm = KernelTimer(1)
            acc = {}
            for i in range(n + 1):
                for p in leaves:
                    p.grad = None
                e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
                tm.arm(0)
                e[0].record()
                means2D = torch.zeros_like(params["xyz"], requires_grad=True)
                shs = torch.cat((params["features_dc"], params["features_rest"]), dim=1)
                op, sc_, rot = (torch.sigmoid(params["opacity"]), torch.exp(params["scaling"]),
                                torch.nn.functional.normalize(params["rotation"]))
                e[1].record()
                o = GaussianRasterizer(rs0)(means3D=params["xyz"], means2D=means2D, opacities=op, shs=shs, scales=sc_,
                                            rotations=rot, reps=params["reps"])
                e[2].record()
                loss = train_loss(o, gt0)
                e[3].record()
                g_img, g_rep = torch.autograd.grad(loss, [o[0], o[5]])
                e[4].record()
                torch.autograd.backward([o[0], o[5]], [g_img, g_rep])
                e[5].record()
                torch.cuda.synchronize()
                tm.arm(None)
                if i == 0:
                    continue
                ev = tm.events[0]
                cur = {"activations_and_cat": e[0].elapsed_time(e[1]), "rasterizer_forward": e[1].elapsed_time(e[2]),
                       "loss_forward": e[2].elapsed_time(e[3]), "loss_backward": e[3].elapsed_time(e[4]),
                       "backward_through_rasterizer": e[4].elapsed_time(e[5]),
                       "rasterizer_forward_kernels": ev[0].elapsed_time(ev[2]) + ev[3].elapsed_time(ev[9]),
                       "rasterizer_backward_kernels": ev[10].elapsed_time(ev[13])}
                for k, v in cur.items():
                    acc[k] = acc.get(k, 0.0) + v / n
            acc["autograd_around_the_rasterizer_backward"] = (acc["backward_through_rasterizer"] -
                                                              acc["rasterizer_backward_kernels"])
            acc["iteration"] = sum(acc[k] for k in ("activations_and_cat", "rasterizer_forward", "loss_forward",
                                                    "loss_backward", "backward_through_rasterizer"))
            acc["library_kernels_share"] = (acc["rasterizer_forward_kernels"] + acc["rasterizer_backward_kernels"]) / \
                acc["iteration"]
            return {k: round(v, 4) for k, v in acc.items()}

        if use_params:
            extras["single_view_it_s"] = timed(it_params, 20)
        saved = [p.grad for p in leaves]
        extras["reference_api_it_s"] = timed(it_reference, 20)
        extras["reference_api_breakdown_ms"] = reference_breakdown()
        for p, g in zip(leaves, saved):
            p.grad = g
        extras["render_fps"] = timed(fps_params, 20)
        extras["render_fps_reference_api"] = timed(fps_reference, 20)
        if use_params:
            def fps_batch():   # the V cameras of a step rendered as a batch: one preprocess pass, views on the side streams
                with torch.no_grad():
                    sink.begin_step()
                    sink.prepare_views([rs for rs, _ in views])
                    lanes = [torch.cuda.current_stream()] + side_streams
                    ready = torch.cuda.Event()
                    ready.record()
                    for st in side_streams:
                        st.wait_event(ready)
                    for v, (rs, _) in enumerate(views):
                        with torch.cuda.stream(lanes[v % len(lanes)]):
                            render_params(params, rs, sink)
                    for st in side_streams:
                        torch.cuda.current_stream().wait_stream(st)
            extras["render_fps_batched"] = V * timed(fps_batch, 10)

        # optimizer step over the same 6 M x 63 floats (not part of `value`): fused Adam kernel, dense and on
        # the visible rows only, next to torch.optim.Adam as the reference configures it (gaussian_model.py:113)
        from gsplat_optim import FusedGaussianAdam
        groups = lambda: [{"params": [p], "lr": 1e-4} for p in leaves]
        vis0 = render_params(params, rs0)[1] > 0
        for name, opt, fn in (("fused_dense", FusedGaussianAdam(groups(), lr=0.0, eps=1e-15), lambda o: o.step()),
                              ("fused_sparse", FusedGaussianAdam(groups(), lr=0.0, eps=1e-15),
                               lambda o: o.step(vis0, vis0.numel())),
                              ("torch_adam", torch.optim.Adam(groups(), lr=0.0, eps=1e-15), lambda o: o.step())):
            extras.setdefault("optimizer_step_ms", {})[name] = 1e3 / timed(lambda: fn(opt), 10)
            del opt

        # the loss of one view (not among `kernels`: the library's events only bracket the rasterizer):
        # fused photometric loss + L1 on the feature channels, forward and backward, and the same composed
        # from PyTorch operations around fused_ssim as train.py does
        def loss_ms(fused):
            img = torch.rand((3, H, W), device=dev).requires_grad_(True)
            rp = torch.rand((4, H, W), device=dev).requires_grad_(True)
            fake = (img, None, None, None, None, rp)

            def f():
                img.grad = None
                rp.grad = None
                train_loss(fake, gt0, fused=fused).backward()
            return 1e3 / timed(f, 50)

        extras["loss_ms_per_view"] = {"fused": loss_ms(True), "composed_in_torch": loss_ms(False)}

        # the stand-alone 64-bit sort on as many pairs as a view has instances
        extras["sort_u64"] = sort_u64_standalone(int(R), (gx * gy).bit_length(), dev)

        # simple-knn initialisation (scene/gaussian_model.py:533) over the scene's 6 M points
        from simple_knn._C import distCUDA2
        pts = params["xyz"].detach()
        extras["dist2_ms"] = {"points": P, "ms": 1e3 / timed(lambda: distCUDA2(pts), 3)}

        # densification statistics after a view: the fused kernel next to the reference's formulation
        from gsplat_optim import add_densification_stats
        o = render_params(params, rs0)
        radii0 = o[1]
        g2 = torch.randn((P, 3), device=dev)
        acc, den, mx = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros((P,), device=dev)

        def stats_reference():
            vf = (radii0 > 0).nonzero()
            mx[vf] = torch.max(mx[vf], radii0[vf].float())
            acc[vf] += torch.norm(g2[vf, :2], dim=-1, keepdim=True)
            den[vf] += 1

        extras["densify_stats_ms"] = {
            "fused": 1e3 / timed(lambda: add_densification_stats(radii0, g2, acc, den, mx), 10),
            "torch_reference_formulation": 1e3 / timed(stats_reference, 10)}

    if rank == 0 and world == 1 and not args.fwd_only and not args.no_extras and args.scene == "street" \
            and use_params:
        # The same step on the translucent variant of the scene, where the blend kernels dominate: a child
        # process (this one keeps its GPU state and waits), a few steps, its JSON line condensed.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--scene", "translucent", "--steps", "6", "--warmup", "2",
               "--no-extras", "--no-cpu-baseline", "--gaussians", str(P), "--width", str(W), "--height", str(H),
               "--views-per-gpu", str(V), "--streams", str(args.streams)]
        try:
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
            d = json.loads(cp.stdout.decode().strip().splitlines()[-1])
            extras["translucent"] = {
                "it_s": d["value"], "ms_per_step": d["ms_per_step"], "one_stream_it_s": d.get("one_stream_it_s"),
                "visible_mean": d["config"]["visible_mean"], "num_rendered_mean": d["config"]["num_rendered_mean"],
                "instances_walked_mean": d["config"]["instances_walked_mean"],
                "pairs_tested_mean": d["config"]["pairs_tested_mean"],
                "kernels_ms": {k: round(v["ms"], 4) for k, v in d["kernels"].items()},
                "roofline": d["roofline"]}
        except Exception as ex:
            extras["translucent"] = {"error": repr(ex)}

    if rank == 0:
        # the committed counter passes are of the 4-channel kernels
        pmc_scene = args.scene if args.dim == 4 else f"{args.scene}_dim{args.dim}"
        pmc_stale = pmc_is_stale(P, W, H, pmc_scene)
        traffic = {} if pmc_stale else load_pmc_traffic(P, W, H, pmc_scene)
        per_kernel = {}
        launches = lambda k: 1 if k in KernelTimer.PER_STEP else V          # per step

        def alg(k):
            return algorithmic_bytes(k, P, visible, R, H * W, passes, V, walked, from_rects, reached, reached_any,
                                     packed)

        for k, ms in kernels.items():
            ab = alg(k)
            ent = {"ms": ms, "launches_per_step": launches(k)}
            if k == "depth_order" and "depth_order_batch" in kernels:
                # the batched preprocess call has ordered every view already: stage 2 has nothing to do here
                ent["note"] = "empty interval: done for all the views of a step by depth_order_batch"
                per_kernel[k] = ent
                continue
            if ab:
                ent.update({"algorithmic_GBps": ab / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "bound": BOUND.get(k)})
                if k in traffic:
                    ent["hbm_traffic_bytes"] = traffic[k]
            per_kernel[k] = ent
        # the kernel with the largest share of a step
        dom = max((k for k in kernels if alg(k) and "note" not in per_kernel[k]), key=lambda k: kernels[k] * launches(k))
        ab = alg(dom)
        achieved = ab / (kernels[dom] * 1e-3) / 1e9
        # Vector-instruction issue: wave64 instructions a launch executes (SQ_INSTS_VALU of the committed PMC passes
        # of this command) over the live duration, against 1024 SIMDs x 0.5 instructions per clock (a wave64
        # instruction issues over 2 cycles on a 32-lane SIMD) at the nominal 2.4 GHz.  (The clock the counter
        # passes ran at is reported beside it; it belongs to the box of that collection, not to this run, so it
        # does not enter the peak: a slower box there must not flatter the fraction here.)
        valu = {} if pmc_stale else load_pmc_valu(P, W, H, pmc_scene)
        for k, ent in per_kernel.items():
            if k in valu:
                v = valu[k]
                peak = 1024 * 0.5 * 2.4      # G wave-instructions / s
                rate = v["valu_insts"] / (ent["ms"] * 1e-3) / 1e9
                ent["valu"] = {"wave_insts_per_launch": v["valu_insts"], "achieved_Ginst_s": rate, "peak_Ginst_s": peak,
                               "frac": rate / peak, "clock_GHz_of_the_pmc_passes": v.get("clock_GHz"),
                               # how full the vector pipes were: the cycles vector instructions kept their SIMD
                               # (SQ_ACTIVE_INST_VALU, units of 4 cycles) over 1024 SIMDs x the launch's cycles
                               "simd_busy_frac": (v["valu_active_simd_cycles"] / (1024.0 * v["launch_cycles"])
                                                  if v.get("valu_active_simd_cycles") and v.get("launch_cycles") else None),
                               "lds_atomic_insts": v.get("lds_atomic_insts"),
                               "insts_per_tested_pair": v["valu_insts"] * 64.0 / pairs if k.startswith("render") and pairs else None}
        views_total = world * V * args.steps
        res = {
            "metric": "render FPS (forward only)" if args.fwd_only else
                      "train it/s (1 it = 1 view: fwd + L1/fused-ssim loss + bwd, as one iteration of the reference)",
            "value": views_total / dt,
            "unit": "frames/s" if args.fwd_only else "it/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # of a step, the time the host spends blocked in the batched preprocess call (waiting for the GPU to
            # get there); the rest of the step it is enqueueing: the closer this gets to zero, the closer the host
            # is to being the limit
            "host_blocked_ms_per_step": host_wait_ms if batch_fwd else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene if args.scene != 'street' else 'street scene'}"
                                   f"{' street scene (small, mostly transparent splats)' if args.scene == 'translucent' else ''}, {P} Gaussians SH3 + {args.dim} reps, {W}x{H}, antialiasing; a step = {V} views "
                                   f"per GPU (different cameras, gradients accumulate)" +
                                   (", forward only" if args.fwd_only else ", each fwd + fused-ssim loss + bwd, then the "
                                    "sum of the gradients over the GPUs when n_gpus > 1 (see allreduce); no optimizer"),
                       "api": "forward_params (raw parameters)" if use_params else "reference signature",
                       "gradient_buffers": ("persistent; blocks of rows that are zero and stay zero are not rewritten"
                                            if use_params and sink.row_blocks is not None else "every row written"),
                       "streams": 1 + len(side_streams),
                       "loss": ("fused_ssim.l1_ssim_loss + fused_ssim.l1_loss on the reps" if fused_loss else
                                "PyTorch L1 / clamp around fused_ssim + reps L1 (as train.py)"),
                       "gaussians": P, "views_per_gpu_per_step": V, "visible_mean": visible, "num_rendered_mean": R,
                       "pairs_tested_mean": pairs, "tile_list_mean": R / (gx * gy),
                       "instances_walked_mean": walked, "reached_mean": reached, "reached_in_any_view": reached_any,
                       "image": [W, H], "parallelism": f"view-sharded x{world}, full replicas" if world > 1 else "single GPU",
                       "collective_backend": (backend if world > 1 else None)},
            "roofline": roofline_entry(dom, per_kernel[dom], achieved, ab, traffic.get(dom), kernels[dom]),
            "kernels": per_kernel,
            "kernels_measured_in": ("a pass with one view after the other, after the timed region (same process, same "
                                    "workload)" if overlapping else "the timed region"),
            # What the reference does with duplicateWithKeys + SortPairs (R pairs of u64 key + u32 value) is done
            # here by the depth ordering of the Gaussians, the emission that partitions on the high tile digit and
            # the segmented pass on the low one: payload rate R x 12 B over ALL of that (emission included: a
            # lower bound for the sort alone).  `sort_u64` (extras) is the stand-alone 64-bit sort of the C ABI.
            "sort_GBps": (R * 12.0 / ((kernels["sort"] + kernels["depth_order"] + kernels.get("depth_order_batch", 0.0) / V +
                                       kernels["emit_keys"]) * 1e-3) / 1e9) if R else None,
            "sort_hbm_GBps": (algorithmic_bytes("sort", P, visible, R, H * W, passes, V, walked, from_rects,
                                                packed=packed) /
                              (kernels["sort"] * 1e-3) / 1e9) if R else None,
        }
        if pmc_stale:
            # the counter summary under profiles/ belongs to another build of the kernels: nothing is derived from it
            res["pmc_stale"] = True
            if BOUND.get(dom, "hbm").startswith("valu"):
                res["roofline"]["frac"] = None
                res["roofline"]["note"] = ("vector-issue fraction withheld: profiles/pmc_traffic*.json was collected "
                                           "on other kernel sources (re-run tools/collect_pmc.sh and tools/pmc_summary.py)")
        if overlapping:
            res["kernels_in_timed_region_ms"] = kernels_overlapped
            res["one_stream_it_s"] = one_stream_it_s
        res.update(extras)
        # the like-for-like numbers next to the headline: the same iteration one view at a time, and through the
        # unchanged reference signature (what train.py / render.py get as a drop-in)
        for k in ("single_view_it_s", "reference_api_it_s", "render_fps", "render_fps_reference_api"):
            if k in extras:
                res["config"][k] = extras[k]
        if world > 1 and reducer_ms:
            nbytes = float(sum(p.grad.numel() * 4 for p in leaves))
            res["allreduce"] = {
                "bytes": nbytes, "chunks": args.reduce_chunks if use_params else 1,
                # per step on the compute stream: the per-Gaussian backward (chunked, exchange running beside it) ...
                "flush_ms": float(np.mean([a.elapsed_time(b) for a, b, _ in reducer_ms])),
                # ... and what is left of the exchange after it, which nothing overlaps
                "exposed_ms": float(np.mean([b.elapsed_time(c) for _, b, c in reducer_ms])),
                # one all-reduce of the whole flat gradient buffer with nothing else running
                "standalone_ms": allreduce_alone_ms,
                "bus_GBps": (2.0 * (world - 1) / world * nbytes / (allreduce_alone_ms * 1e-3) / 1e9)
                            if allreduce_alone_ms else None,
                # what the timed steps exchanged: every row ("dense") or the rows of the Gaussians that the
                # views of the step, over all ranks, reached ("sparse"; rows, of how many, at the last probe)
                "policy": args.reduce,
                "steps_dense": reduce_modes.count("dense"), "steps_sparse": reduce_modes.count("sparse"),
                "union_rows": reducer.last_union[0] if reducer.last_union else None,
                "union_fraction": (reducer.last_union[0] / max(1, reducer.last_union[1])) if reducer.last_union else None,
                "bytes_sparse": (reducer.last_union[0] * (nbytes // max(1, P)) + P) if reducer.last_union else None,
                # the same step with every row exchanged (a model whose every Gaussian some view sees): THE number
                # to read multi-GPU scaling from when `policy` let the timed steps go sparse
                "dense": dense_leg,
                "ranks_seen": dist.get_world_size(), "backend": backend, "collective_library": collective_version}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(P, W, H, seed, args.scene == "translucent")
            except Exception as ex:  # the checker is optional for the measurement itself
                res["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
