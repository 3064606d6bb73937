"""Achievable HBM bandwidth with plain PyTorch kernels (copy, fill, sum) for calibration."""
import time
import torch
dev = "cuda:0"
n = 512 * 1024 * 1024  # floats: 2 GiB
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
a.fill_(1.0)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
nbytes = n * 4
print("copy  %.0f GB/s (read+write)" % (2 * nbytes / t(lambda: b.copy_(a)) / 1e9))
print("fill  %.0f GB/s (write)" % (nbytes / t(lambda: b.fill_(2.0)) / 1e9))
print("sum   %.0f GB/s (read)" % (nbytes / t(lambda: a.sum()) / 1e9))
print("add   %.0f GB/s (2 read + write)" % (3 * nbytes / t(lambda: torch.add(a, b, out=b)) / 1e9))
