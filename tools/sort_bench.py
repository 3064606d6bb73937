"""Times gsr_sort_pairs_u64_u32 on rasterizer-like keys (runs of equal depth
bits with consecutive tile ids, as key emission produces them)."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
from _gsplat_lib import ptr  # noqa: E402
import _gsplat_lib  # noqa: E402


def make_keys(n, ntiles=8160, mean_run=10, seed=0, dev="cuda:0"):
    g = torch.Generator(device=dev).manual_seed(seed)
    G = n // mean_run + 1
    runs = torch.randint(1, 2 * mean_run, (G,), device=dev, generator=g)
    gid = torch.repeat_interleave(torch.arange(G, device=dev), runs)[:n]
    start = torch.cumsum(runs, 0) - runs
    within = torch.arange(gid.numel(), device=dev) - start[gid]
    depth = (torch.rand(G, device=dev, generator=g) * 38 + 2).to(torch.float32).view(torch.int32).to(torch.int64)
    tile0 = torch.randint(0, ntiles, (G,), device=dev, generator=g)
    tile = (tile0[gid] + within) % ntiles
    keys = (tile << 32) | depth[gid]
    return keys.contiguous(), gid.to(torch.int32).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=42_000_000)
    ap.add_argument("--end-bit", type=int, default=45)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--random", action="store_true")
    ap.add_argument("--libs", type=str, default="")
    a = ap.parse_args()
    dev = "cuda:0"
    libs = [x for x in a.libs.split(",") if x] or [_gsplat_lib.LIB_PATH]
    Ls = {os.path.basename(x): C.CDLL(x) for x in libs}
    for L in Ls.values():
        L.gsr_sort_temp_bytes.restype = C.c_size_t
        L.gsr_sort_temp_bytes.argtypes = [C.c_int64]
        L.gsr_sort_pairs_u64_u32.restype = C.c_int
        L.gsr_sort_pairs_u64_u32.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
    if a.random:
        keys = torch.randint(0, 2 ** 62, (a.n,), device=dev, dtype=torch.int64) & ((1 << a.end_bit) - 1)
        vals = torch.arange(a.n, device=dev, dtype=torch.int32)
    else:
        keys, vals = make_keys(a.n)
    n = keys.numel()
    temp = torch.empty(max(L.gsr_sort_temp_bytes(n) for L in Ls.values()), dtype=torch.uint8, device=dev)
    k0, v0, k1, v1 = (torch.empty_like(keys), torch.empty_like(vals), torch.empty_like(keys), torch.empty_like(vals))
    res = C.c_int(0)
    st = torch.cuda.current_stream().cuda_stream
    times = {k: [] for k in Ls}
    for it in range(a.iters + 2):
      for name, L in Ls.items():
          k0.copy_(keys)
          v0.copy_(vals)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          rc = L.gsr_sort_pairs_u64_u32(ptr(k0), ptr(v0), ptr(k1), ptr(v1), n, a.end_bit, ptr(temp), temp.numel(), st,
                                        C.byref(res))
          e1.record()
          torch.cuda.synchronize()
          assert rc == 0
          if it >= 2:
              times[name].append(e0.elapsed_time(e1))
    ko = k0 if res.value == 0 else k1
    mask = (1 << a.end_bit) - 1
    mk = ko & mask
    ok = bool((mk[1:] >= mk[:-1]).all())
    passes = (a.end_bit + 7) // 8
    for name, tt in times.items():
      ms = sorted(tt)[len(tt) // 2]
      print(f"{name}: min {min(tt):.3f} n={n} end_bit={a.end_bit} sorted={ok} median {ms:.3f} ms  payload {n * 12 / ms / 1e6:.1f} GB/s  "
          f"traffic {(n * (8 + passes * 24)) / ms / 1e6:.1f} GB/s ({(n * (8 + passes * 24)) / ms / 1e6 / 8000:.3f} of 8 TB/s)")


if __name__ == "__main__":
    main()
