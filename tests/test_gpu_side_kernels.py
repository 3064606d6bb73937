"""GPU parity for the stand-alone sort, simple-knn and fused-ssim kernels
(through the C ABI) against the CPU checker.  Sort and knn are exact; ssim is
the same arithmetic in the same order, so also exact up to the rounding of
identical operations (asserted at 1e-6 absolute)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu_sort(keys, vals, end_bit):
    from _gsplat_lib import lib, ptr
    L = lib()
    n = keys.size
    dev = "cuda:0"
    k0 = torch.tensor(keys.view(np.int64), device=dev)
    v0 = torch.tensor(vals.view(np.int32), device=dev)
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    temp = torch.empty(L.gsr_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
    res = C.c_int(0)
    rc = L.gsr_sort_pairs_u64_u32(ptr(k0), ptr(v0), ptr(k1), ptr(v1), n, end_bit, ptr(temp), temp.numel(),
                                  torch.cuda.current_stream().cuda_stream, C.byref(res))
    assert rc == 0
    torch.cuda.synchronize()
    ko, vo = (k0, v0) if res.value == 0 else (k1, v1)
    return ko.cpu().numpy().view(np.uint64), vo.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,end_bit", [(1, 64), (255, 13), (4096, 44), (4097, 47), (100_003, 45), (1_500_000, 47),
                                       (70_000, 64), (9000, 8), (9000, 3)])
def test_sort_pairs_bit_exact_and_stable(oracle, n, end_bit):
    rng = np.random.default_rng(n + end_bit)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64)
    if end_bit < 64:
        # few distinct values in the sorted bits => many ties => stability is visible in the values
        keys = (keys & np.uint64((1 << end_bit) - 1)) if n < 10000 else (keys % np.uint64(50_000))
        keys |= (rng.integers(0, 2 ** 15, size=n, dtype=np.int64).astype(np.uint64) << np.uint64(48)) if end_bit <= 47 else np.uint64(0)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = _gpu_sort(keys, vals, end_bit)
    kr, vr = oracle.sort_pairs(keys, vals, end_bit)
    assert np.array_equal(ko, kr)
    assert np.array_equal(vo, vr)
    mask = np.uint64((1 << end_bit) - 1) if end_bit < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    mk = ko & mask
    assert np.all(mk[:-1] <= mk[1:])                                   # sortedness
    same = mk[:-1] == mk[1:]
    assert np.all(vo[:-1][same] < vo[1:][same])                        # stability
    assert np.array_equal(np.sort(vo), vals)                           # permutation


def test_sort_empty():
    from _gsplat_lib import lib
    res = C.c_int(5)
    assert lib().gsr_sort_pairs_u64_u32(None, None, None, None, 0, 44, None, 0, None, C.byref(res)) == 0


@pytest.mark.parametrize("P", [1, 3, 4, 700, 5000, 60_000])
def test_dist2_exact(oracle, P):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(P)
    pts = (rng.standard_normal((P, 3)) * np.array([1.0, 0.3, 2.0]) + np.array([0.5, -2.0, 4.0])).astype(np.float32)
    if P >= 700:
        pts[::50] = pts[1::50][: len(pts[::50])]  # exact duplicates
    got = distCUDA2(torch.tensor(pts, device="cuda:0")).cpu().numpy()
    ref = oracle.dist2(pts)
    assert np.array_equal(got, ref)
    if 4 <= P <= 5000:  # brute force, independent of the box scheme
        d = ((pts[:, None, :].astype(np.float32) - pts[None, :, :]) ** 2)
        d2 = (d[..., 0] + d[..., 1]) + d[..., 2]
        np.fill_diagonal(d2, np.inf)
        part = np.sort(d2, axis=1)[:, :3].astype(np.float32)
        bf = ((part[:, 0] + part[:, 1]) + part[:, 2]) / np.float32(3.0)
        assert np.allclose(got, bf, rtol=1e-6, atol=0)


@pytest.mark.parametrize("shape", [(1, 3, 37, 53), (2, 1, 64, 64), (1, 3, 200, 333), (1, 1, 5, 7)])
@pytest.mark.parametrize("padding", ["same", "valid"])
def test_fused_ssim_value_and_grad(oracle, shape, padding):
    from fused_ssim import fused_ssim, fusedssim
    if padding == "valid" and min(shape[2:]) <= 10:
        pytest.skip("valid padding needs more than 10 pixels")
    rng = np.random.default_rng(sum(shape))
    a = rng.random(shape).astype(np.float32)
    b = np.clip(a + 0.2 * rng.standard_normal(shape), 0, 1).astype(np.float32)
    ta = torch.tensor(a, device="cuda:0", requires_grad=True)
    tb = torch.tensor(b, device="cuda:0")
    m, d1, d2, d3 = fusedssim(0.01 ** 2, 0.03 ** 2, ta.detach(), tb, True)
    rm, r1, r2, r3 = oracle.ssim_fwd(a, b)
    for g, r in ((m, rm), (d1, r1), (d2, r2), (d3, r3)):
        assert np.abs(g.cpu().numpy() - r).max() <= 1e-6 * max(1.0, np.abs(r).max())
    val = fused_ssim(ta, tb, padding=padding)
    val.backward()
    ref_map = rm if padding == "same" else rm[:, :, 5:-5, 5:-5]
    assert abs(float(val) - float(ref_map.mean())) <= 1e-6
    dl = np.full(shape, 1.0 / ref_map.size, np.float32)
    if padding == "valid":
        z = np.zeros(shape, np.float32)
        z[:, :, 5:-5, 5:-5] = dl[:, :, 5:-5, 5:-5]
        dl = z
    rg = oracle.ssim_bwd(a, b, dl, r1, r2, r3)
    assert np.abs(ta.grad.cpu().numpy() - rg).max() <= 1e-6 * max(1e-3, np.abs(rg).max())
    m2, e1, _, _ = fusedssim(0.01 ** 2, 0.03 ** 2, ta.detach(), tb, False)
    assert e1.numel() == 0 and torch.equal(m2, m)


@pytest.mark.parametrize("shape", [(3, 64, 96), (3, 45, 75), (1, 3, 130, 131), (3, 1080, 1920)])
@pytest.mark.parametrize("lam", [0.2, 0.0, 1.0])
def test_l1_ssim_loss_equals_the_composed_formulation(shape, lam):
    """fused_ssim.l1_ssim_loss against clamp + l1 + fused_ssim composed in PyTorch (train.py:222-230;
    fused_ssim itself is pinned by the reference's conv2d SSIM fixture)."""
    from fused_ssim import fused_ssim, l1_ssim_loss
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    img = (torch.rand(shape, device="cuda", generator=g) * 1.6 - 0.3)      # a fifth of the values get clamped
    img.view(-1)[::97] = 0.0
    img.view(-1)[5::89] = 1.0                                              # values on the clamp bounds pass the gradient
    gt = torch.rand(shape, device="cuda", generator=g)
    gt.view(-1)[7::101] = img.view(-1)[7::101].clamp(0, 1)                 # exact ties: sign(0) = 0
    a = img.clone().requires_grad_(True)
    c = a.clamp(0, 1)
    c4, g4 = (c, gt) if c.dim() == 4 else (c.unsqueeze(0), gt.unsqueeze(0))
    ref = (1.0 - lam) * (c - gt).abs().mean() + lam * (1.0 - fused_ssim(c4, g4))
    (ref * 1.7).backward()
    b = img.clone().requires_grad_(True)
    got, parts = l1_ssim_loss(b, gt, lam, return_parts=True)
    (got * 1.7).backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert abs(float(parts[1]) - float((c - gt).abs().mean())) <= 2e-6
    assert float(parts[0]) == float(got)
    scale = float(a.grad.abs().max())
    assert float((b.grad - a.grad).abs().max()) <= 2e-6 * scale
    assert torch.equal(b.grad == 0, a.grad == 0)


@pytest.mark.parametrize("shape,target", [((4, 37, 53), "tensor"), ((4, 1080, 1920), 10.0), ((1, 5), "tensor"),
                                          ((3, 300, 500), -0.25)])
def test_l1_loss_equals_torch(shape, target):
    """fused_ssim.l1_loss against torch.abs(a - b).mean() (utils/loss_utils.py:23-24) and its autograd
    gradient, with a tensor and with a number as the target; exact zeros in a - b get sign 0."""
    import fused_ssim
    g = torch.Generator(device="cuda:0").manual_seed(11)
    a = torch.randn(shape, device="cuda:0", generator=g) * 3
    b = torch.randn(shape, device="cuda:0", generator=g) if target == "tensor" else target
    if target == "tensor":
        a.view(-1)[::7] = b.view(-1)[::7]              # ties: the gradient there is zero
    a1 = a.clone().requires_grad_(True)
    a2 = a.clone().requires_grad_(True)
    got = fused_ssim.l1_loss(a1, b)
    ref = torch.abs(a2 - b).mean()
    assert abs(got.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    (2.5 * got).backward()
    (2.5 * ref).backward()
    assert torch.allclose(a1.grad, a2.grad, rtol=1e-6, atol=0)
    if target == "tensor":
        assert (a1.grad.view(-1)[::7] == 0).all()


def test_fused_losses_match_the_reference_composition():
    """fused_ssim.l1_ssim_loss and fused_ssim.l1_loss against values and autograd gradients of the
    reference's own utils/loss_utils.l1_loss + ssim composed as train.py:222-230, 248-251 does
    (tests/golden/train_loss_ref.npz, written by tests/golden/make_fixtures.py from the reference)."""
    import os
    import fused_ssim
    from gpu_util import rel_err
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_loss_ref.npz"))
    lam = float(fx["lambda_dssim"])
    for i in range(2):
        raw = torch.tensor(fx[f"raw_{i}"], device="cuda:0", requires_grad=True)
        gt = torch.tensor(fx[f"gt_{i}"], device="cuda:0")
        loss, parts = fused_ssim.l1_ssim_loss(raw, gt, lam, return_parts=True)
        loss.backward()
        assert abs(float(loss) - float(fx[f"loss_{i}"])) <= 2e-6
        assert abs(float(parts[1]) - float(fx[f"l1_{i}"])) <= 2e-6
        assert abs(float(parts[2]) - float(fx[f"ssim_{i}"])) <= 2e-6
        assert rel_err(raw.grad.cpu().numpy(), fx[f"grad_{i}"]) <= 1e-4
        rep = torch.tensor(fx[f"rep_{i}"], device="cuda:0", requires_grad=True)
        l1 = fused_ssim.l1_loss(rep, 10.0)
        l1.backward()
        assert abs(float(l1) - float(fx[f"l1rep_{i}"])) <= 1e-6 * max(1.0, float(fx[f"l1rep_{i}"]))
        assert np.allclose(rep.grad.cpu().numpy(), fx[f"gradrep_{i}"], rtol=1e-6, atol=0)


def test_fused_ssim_matches_the_reference_conv2d_ssim_fixture():
    """The HIP fused-ssim kernels straight against tests/golden/ssim_ref.npz: values and autograd gradients
    of the reference's own utils/loss_utils.ssim (conv2d formulation) on three image pairs, written by
    tests/golden/make_fixtures.py from the reference.  The check submodules/fused-ssim/tests/test.py:82-91
    makes between the two, without the CPU checker in between."""
    import os
    from fused_ssim import fused_ssim
    from gpu_util import rel_err
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim_ref.npz"))
    for i in range(3):
        a = torch.tensor(fx[f"img1_{i}"], device="cuda:0", requires_grad=True)
        b = torch.tensor(fx[f"img2_{i}"], device="cuda:0")
        val = fused_ssim(a, b)
        val.backward()
        assert abs(float(val) - float(fx[f"ssim_{i}"])) <= 2e-6
        assert rel_err(a.grad.cpu().numpy(), fx[f"grad_{i}"]) <= 1e-4
        assert abs(float(fused_ssim(a.detach(), b, train=False)) - float(fx[f"ssim_{i}"])) <= 2e-6


@pytest.mark.parametrize("n,end_bit", [(4097, 47), (100_003, 45), (300_000, 64)])
def test_sort_with_64_bit_status_words(oracle, n, end_bit):
    """Sorts of 2^30 pairs and more publish their look-back prefixes in 64-bit status words (a 32-bit word has
    30 bits for the count); the test hook selects them at a size the checker can sort too."""
    from _gsplat_lib import lib
    L = lib()
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64) % np.uint64(30_000)
    if end_bit == 64:
        keys = rng.integers(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + (keys & np.uint64(1))
    vals = np.arange(n, dtype=np.uint32)
    small = L.gsr_sort_temp_bytes(n)
    try:
        L.gsr_test_force_wide_sort_status(1)
        assert L.gsr_sort_temp_bytes(n) > small
        ko, vo = _gpu_sort(keys, vals, end_bit)
    finally:
        L.gsr_test_force_wide_sort_status(0)
    kr, vr = oracle.sort_pairs(keys, vals, end_bit)
    assert np.array_equal(ko, kr) and np.array_equal(vo, vr)
