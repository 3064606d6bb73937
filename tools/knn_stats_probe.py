import os, sys, ctypes as C, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd"))
import synth
from simple_knn._C import distCUDA2
import _gsplat_lib
P = 6_000_000
pts = torch.tensor(synth.street_scene(P, 1920, 1080, seed=2)["means3D"], dtype=torch.float32, device="cuda")
d = distCUDA2(pts); torch.cuda.synchronize()
L = C.CDLL(os.environ["GSPLAT_AMD_LIB"])
out = (C.c_ulonglong * 8)()
L.gsk_debug_stats(out)
waves = P // 64
print("box visits per wave %.1f, lanes wanting per visit %.1f, share of visits for <= 4 lanes %.2f, <= 16 lanes %.2f" % (out[0] / waves, out[1] / max(1, out[0]), out[2] / max(1, out[0]), out[3] / max(1, out[0])))
# how sparse: distribution of the result
q = torch.quantile(d[::64].float(), torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda"))
print("dist2 quantiles 50/90/99/99.9 %:", q.tolist())
