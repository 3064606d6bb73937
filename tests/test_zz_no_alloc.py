"""The sort entry point that allocates nothing (its temp area is a device array inside the library)
against the entry point with a caller-provided temp buffer: bit-identical keys, values and result
buffer, on two calls in a row, the second with a smaller input (the array keeps the first call's
contents)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(n, end_bit, seed):
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64) % np.uint64(40_000)
    keys |= rng.integers(0, 2 ** 15, size=n, dtype=np.int64).astype(np.uint64) << np.uint64(48)   # bits the sort ignores
    return keys, np.arange(n, dtype=np.uint32)


def _sort(keys, vals, end_bit, noalloc):
    from _gsplat_lib import lib, ptr
    L = lib()
    n = keys.size
    dev = "cuda:0"
    k0 = torch.tensor(keys.view(np.int64), device=dev)
    v0 = torch.tensor(vals.view(np.int32), device=dev)
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    res = C.c_int(-1)
    st = torch.cuda.current_stream().cuda_stream
    if noalloc:
        rc = L.gsr_sort_pairs_u64_u32_noalloc(ptr(k0), ptr(v0), ptr(k1), ptr(v1), n, end_bit, st, C.byref(res))
    else:
        temp = torch.empty(L.gsr_sort_temp_bytes(n), dtype=torch.uint8, device=dev)
        rc = L.gsr_sort_pairs_u64_u32(ptr(k0), ptr(v0), ptr(k1), ptr(v1), n, end_bit, ptr(temp), temp.numel(), st,
                                      C.byref(res))
    assert rc == 0, rc
    torch.cuda.synchronize()
    ko, vo = (k0, v0) if res.value == 0 else (k1, v1)
    return ko.cpu().numpy().view(np.uint64), vo.cpu().numpy().view(np.uint32), res.value


def test_noalloc_sort_matches_the_sort_with_a_temp_buffer():
    from _gsplat_lib import lib
    L = lib()
    nmax = L.gsr_sort_noalloc_max_pairs()
    assert nmax >= 1 << 20
    # two calls in a row on the same GPU, the second smaller; then the largest size allowed
    for n, end_bit, seed in ((1_200_003, 47, 1), (70_001, 45, 2), (nmax, 44, 3), (5, 64, 4)):
        keys, vals = _inputs(n, end_bit, seed)
        got = _sort(keys, vals, end_bit, noalloc=True)
        ref = _sort(keys, vals, end_bit, noalloc=False)
        assert got[2] == ref[2]
        assert np.array_equal(got[0], ref[0])
        assert np.array_equal(got[1], ref[1])
        err = C.c_uint32(7)     # the error word of the sort just made, from the library's own array
        assert L.gsr_sort_noalloc_error(torch.cuda.current_stream().cuda_stream, C.byref(err)) == 0 and err.value == 0


def test_noalloc_sort_rejects_what_does_not_fit():
    from _gsplat_lib import lib
    L = lib()
    nmax = L.gsr_sort_noalloc_max_pairs()
    res = C.c_int(0)
    t = torch.zeros(8, dtype=torch.int64, device="cuda:0")   # never touched: the size check comes first
    rc = L.gsr_sort_pairs_u64_u32_noalloc(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), nmax + 1, 44, None,
                                          C.byref(res))
    assert rc == -7
    assert b"scratch" in L.gsr_status_string(rc)
    assert L.gsr_sort_pairs_u64_u32_noalloc(None, None, None, None, 0, 44, None, C.byref(res)) == 0
    # 64-bit status words (test hook) take twice the room: the largest size then no longer fits the array,
    # and that is a limit, not a runtime error
    try:
        L.gsr_test_force_wide_sort_status(1)
        rc = L.gsr_sort_pairs_u64_u32_noalloc(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), nmax, 64, None,
                                              C.byref(res))
    finally:
        L.gsr_test_force_wide_sort_status(0)
    assert rc == -7
    assert L.gsr_sort_noalloc_error(None, None) == -1
