"""ctypes binding of libgsplat_amd.so (the C ABI in include/gsplat_amd.h).

There is no CPU fallback: if the shared library is missing, importing any of
the operator packages raises.  PyTorch is used for device memory and streams
only; nothing but raw device pointers and sizes crosses into the library.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPLAT_AMD_LIB", os.path.join(_HERE, "libgsplat_amd.so"))

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class GsrScene(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("P", "D", "M", "DIM", "W", "H")] + \
               [(n, _vp) for n in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                   "cov3D_precomp", "reps", "viewmatrix", "projmatrix", "campos")] + \
               [(n, C.c_float) for n in ("scale_modifier", "tanfovx", "tanfovy")] + \
               [(n, C.c_int) for n in ("prefiltered", "antialiasing", "debug")] + \
               [("profile_events", C.POINTER(C.c_void_p)), ("raw_params", C.c_int), ("shs_rest", _vp)]


class GsrForwardOut(C.Structure):
    _fields_ = [(n, _vp) for n in ("out_color", "out_invdepth", "radii", "accum_alpha", "gauss_sum", "gauss_count",
                                   "last_contr_gauss", "out_depths", "out_reps")] + \
               [("geom_buffer", _vp), ("geom_bytes", C.c_size_t), ("image_buffer", _vp), ("image_bytes", C.c_size_t),
                ("num_visible", C.c_int64), ("preprocessed", C.c_int), ("sort_error_host", _vp)]


class GsrBackwardIO(C.Structure):
    _fields_ = [(n, _vp) for n in ("dL_dout_color", "dL_dout_invdepth", "dL_dout_reps", "dL_dmeans2D", "dL_dcolors",
                                   "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations",
                                   "dL_dreps", "work")] + \
               [("work_bytes", C.c_size_t), ("dL_dsh_rest", _vp), ("accumulate", C.c_int), ("row_blocks", _vp)]


class GsrBackwardView(C.Structure):
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("W", C.c_int), ("H", C.c_int), ("antialiasing", C.c_int), ("has_invdepth", C.c_int), ("radii", _vp),
                ("geom_buffer", _vp), ("work", _vp), ("dL_dmeans2D", _vp), ("means2D_blocks", _vp)]


class GsrForwardView(C.Structure):
    _fields_ = [("viewmatrix", _vp), ("projmatrix", _vp), ("campos", _vp), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("W", C.c_int), ("H", C.c_int), ("antialiasing", C.c_int), ("radii", _vp), ("out_depths", _vp),
                ("geom_buffer", _vp), ("geom_bytes", C.c_size_t), ("num_rendered", C.c_int64), ("num_visible", C.c_int64),
                ("depth_ordered", C.c_int)]


class GsrLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("rec", "cov3D", "tiles", "clamped", "block_sums", "total", "depth_sort", "final_T",
                                          "n_contrib", "ranges", "keys0", "keys1", "vals0", "vals1")] + \
               [("sorted_buffer", C.c_int)]


_lib = None


def lib():
    """Load libgsplat_amd.so once.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(hipcc, --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.gsr_geom_bytes.restype = C.c_size_t
    L.gsr_geom_bytes.argtypes = [C.c_int]
    L.gsr_image_bytes.restype = C.c_size_t
    L.gsr_image_bytes.argtypes = [C.c_int, C.c_int]
    L.gsr_binning_bytes.restype = C.c_size_t
    L.gsr_binning_bytes.argtypes = [C.c_int64]
    L.gsr_backward_work_bytes.restype = C.c_size_t
    L.gsr_backward_work_bytes.argtypes = [C.c_int]
    L.gsr_backward_work_bytes_dim.restype = C.c_size_t
    L.gsr_backward_work_bytes_dim.argtypes = [C.c_int, C.c_int]
    L.gsr_sort_temp_bytes.restype = C.c_size_t
    L.gsr_sort_temp_bytes.argtypes = [C.c_int64]
    L.gsr_sort_temp_error_offset.restype = C.c_size_t
    L.gsr_sort_temp_error_offset.argtypes = []
    L.gsr_test_force_wide_sort_status.restype = None
    L.gsr_test_force_wide_sort_status.argtypes = [C.c_int]
    L.gsr_test_forward_reach_masks.restype = None
    L.gsr_test_forward_reach_masks.argtypes = [C.c_int]
    L.gsr_test_packed_instances.restype = None
    L.gsr_test_packed_instances.argtypes = [C.c_int]
    L.gsr_test_wide_depth_keys.restype = None
    L.gsr_test_wide_depth_keys.argtypes = [C.c_int]
    L.gsr_test_inject_sort_error.restype = None
    L.gsr_test_inject_sort_error.argtypes = [C.c_int]
    L.gsr_sort_noalloc_error.restype = C.c_int
    L.gsr_sort_noalloc_error.argtypes = [_vp, C.POINTER(C.c_uint32)]
    L.gsr_forward_stage1.restype = C.c_int
    L.gsr_forward_stage1.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrForwardOut), _vp, C.POINTER(C.c_int64)]
    L.gsr_forward_stage2.restype = C.c_int
    L.gsr_forward_stage2.argtypes = [C.POINTER(GsrScene), C.POINTER(GsrForwardOut), C.c_int64, _vp, C.c_size_t, _vp]
    L.gsr_backward.restype = C.c_int
    L.gsr_backward.argtypes = [C.POINTER(GsrScene), _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(GsrBackwardIO), _vp]
    L.gsr_backward_blend.restype = C.c_int
    L.gsr_backward_blend.argtypes = L.gsr_backward.argtypes
    L.gsr_backward_geometry_batch.restype = C.c_int
    L.gsr_backward_geometry_batch.argtypes = [C.POINTER(GsrScene), C.c_int, C.POINTER(GsrBackwardView),
                                              C.POINTER(GsrBackwardIO), _vp]
    L.gsr_backward_geometry_batch_range.restype = C.c_int
    L.gsr_backward_geometry_batch_range.argtypes = [C.POINTER(GsrScene), C.c_int, C.POINTER(GsrBackwardView),
                                                    C.POINTER(GsrBackwardIO), C.c_int, C.c_int, _vp]
    L.gsr_preprocess_batch.restype = C.c_int
    L.gsr_preprocess_batch.argtypes = [C.POINTER(GsrScene), C.c_int, C.POINTER(GsrForwardView), _vp]
    L.gsr_mark_visible.restype = C.c_int
    L.gsr_mark_visible.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.gsr_sort_pairs_u64_u32.restype = C.c_int
    L.gsr_sort_pairs_u64_u32.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, C.c_size_t, _vp,
                                         C.POINTER(C.c_int)]
    L.gsr_sort_noalloc_max_pairs.restype = C.c_int64
    L.gsr_sort_noalloc_max_pairs.argtypes = []
    L.gsr_sort_pairs_u64_u32_noalloc.restype = C.c_int
    L.gsr_sort_pairs_u64_u32_noalloc.argtypes = [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, C.POINTER(C.c_int)]
    L.gsr_layout.restype = C.c_int
    L.gsr_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(GsrLayout)]
    L.gsr_sorted_keys_u64.restype = C.c_int
    L.gsr_sorted_keys_u64.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, _vp]
    L.gsk_dist2_temp_bytes.restype = C.c_size_t
    L.gsk_dist2_temp_bytes.argtypes = [C.c_int]
    L.gsk_dist2.restype = C.c_int
    L.gsk_dist2.argtypes = [C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]
    L.gss_ssim_forward.restype = C.c_int
    L.gss_ssim_forward.argtypes = [C.c_int] * 4 + [C.c_float] * 2 + [_vp] * 7
    L.gss_ssim_backward.restype = C.c_int
    L.gss_ssim_backward.argtypes = [C.c_int] * 4 + [C.c_float] * 2 + [_vp] * 8
    L.gss_loss_partial_floats.restype = C.c_size_t
    L.gss_loss_partial_floats.argtypes = [C.c_int] * 3
    L.gss_l1_ssim_loss_forward.restype = C.c_int
    L.gss_l1_ssim_loss_forward.argtypes = [C.c_int] * 3 + [C.c_float] * 3 + [_vp] * 8
    L.gss_l1_ssim_loss_backward.restype = C.c_int
    L.gss_l1_ssim_loss_backward.argtypes = [C.c_int] * 3 + [C.c_float] + [_vp] * 8
    L.gss_l1_partial_floats.restype = C.c_size_t
    L.gss_l1_partial_floats.argtypes = [C.c_int64]
    L.gss_l1_loss_forward.restype = C.c_int
    L.gss_l1_loss_forward.argtypes = [C.c_int64, _vp, _vp, C.c_float, _vp, _vp, _vp]
    L.gss_l1_loss_backward.restype = C.c_int
    L.gss_l1_loss_backward.argtypes = [C.c_int64, _vp, _vp, _vp, C.c_float, _vp, _vp]
    L.gso_adam_update.restype = C.c_int
    L.gso_adam_update.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int] + [C.c_double] * 6 + [_vp]
    L.gso_densify_stats.restype = C.c_int
    L.gso_densify_stats.argtypes = [C.c_int] + [_vp] * 6
    L.gso_densify_select.restype = C.c_int
    L.gso_densify_select.argtypes = [C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_float, C.c_float, C.c_int, _vp, _vp]
    L.gso_mark_nonzero.restype = C.c_int
    L.gso_mark_nonzero.argtypes = [C.c_int64, C.c_int, C.POINTER(C.c_void_p), _vp, _vp]
    L.gso_prune_mask.restype = C.c_int
    L.gso_prune_mask.argtypes = [C.c_int64, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]
    L.gso_split_children.restype = C.c_int
    L.gso_split_children.argtypes = [C.c_int64, C.c_int] + [_vp] * 9
    L.gso_compact_temp_bytes.restype = C.c_size_t
    L.gso_compact_temp_bytes.argtypes = [C.c_int64]
    L.gso_compact_map.restype = C.c_int
    L.gso_compact_map.argtypes = [C.c_int64, _vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_int64), _vp]
    L.gso_gather_rows.restype = C.c_int
    L.gso_gather_rows.argtypes = [_vp, _vp, _vp, C.c_int64, C.c_int, _vp]
    L.gsr_status_string.restype = C.c_char_p
    L.gsr_status_string.argtypes = [C.c_int]
    L.gsr_struct_bytes.restype = C.c_size_t
    L.gsr_struct_bytes.argtypes = [C.c_int]
    # the structures above mirror include/gsplat_amd.h field by field: a library built from another header fails here
    for which, cls in enumerate((GsrScene, GsrForwardOut, GsrBackwardIO, GsrBackwardView, GsrForwardView, GsrLayout)):
        if L.gsr_struct_bytes(which) != C.sizeof(cls):
            raise RuntimeError(f"{LIB_PATH}: {cls.__name__} is {L.gsr_struct_bytes(which)} bytes in the library and "
                               f"{C.sizeof(cls)} in this binding (rebuild the library: make -C csrc)")
    _lib = L
    return L


EXPORTED = ("gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes", "gsr_backward_work_bytes", "gsr_backward_work_bytes_dim",
            "gsr_forward_stage1", "gsr_forward_stage2", "gsr_backward", "gsr_backward_blend",
            "gsr_backward_geometry_batch", "gsr_backward_geometry_batch_range", "gsr_preprocess_batch", "gsr_mark_visible", "gsr_sort_temp_bytes", "gsr_sort_temp_error_offset", "gsr_test_inject_sort_error", "gsr_test_force_wide_sort_status", "gsr_test_forward_reach_masks", "gsr_test_packed_instances", "gsr_test_wide_depth_keys",
            "gsr_sort_pairs_u64_u32", "gsr_sort_noalloc_max_pairs", "gsr_sort_pairs_u64_u32_noalloc", "gsr_sort_noalloc_error", "gsr_layout", "gsr_sorted_keys_u64", "gsk_dist2_temp_bytes", "gsk_dist2", "gss_ssim_forward",
            "gss_ssim_backward", "gss_loss_partial_floats", "gss_l1_ssim_loss_forward", "gss_l1_ssim_loss_backward",
            "gss_l1_partial_floats", "gss_l1_loss_forward", "gss_l1_loss_backward",
            "gso_adam_update", "gso_densify_stats", "gso_densify_select", "gso_mark_nonzero", "gso_prune_mask", "gso_split_children", "gso_compact_temp_bytes", "gso_compact_map",
            "gso_gather_rows", "gsr_status_string", "gsr_struct_bytes")


NUM_PROFILE_EVENTS = 21
EVENT_NAMES = ("stage1_begin", "preprocess", "scan", "stage2_begin", "depth_order", "emit", "sort", "ranges", "render",
               "finish", "bwd_begin", "bwd_clear", "bwd_render", "bwd_preprocess", "batch_begin", "batch",
               "fwd_batch_begin", "fwd_batch", "tile_count", "fwd_batch_sort_begin", "fwd_batch_sort")

# Optional per-thread hook used by bench.py: a ctypes array of hipEvent_t handles
# that the next rasterizer calls record around their kernels.
_profile_events = None


def set_profile_events(arr):
    """arr: None, or (c_void_p * NUM_PROFILE_EVENTS) of hipEvent_t handles."""
    global _profile_events
    _profile_events = arr


def profile_events():
    return _profile_events


class SortErrorWords:
    """Where the sorts' error words of every forward call land without making the host wait.

    The look-back spins of the radix sorts are bounded; a tripped limit leaves an error word on the
    device and an invalid instance list.  Each `rasterize_gaussians` call hands stage 2 one word of
    pinned host memory (GsrForwardOut.sort_error_host), preset to all ones; the last kernel of the
    forward stores there the error word of the instance sort OR-ed with that of the view's depth
    order (whichever call made the order: stage 2 itself or gsr_preprocess_batch).  Every later call into the rasterizer looks at the words
    that have arrived and raises if one is set, so the failure surfaces at the next call after the
    GPU got there instead of never.  No events, no copies, nothing the stream waits on."""

    SLOTS = 1024
    UNSET = -1   # all ones: never stored by the library

    def __init__(self):
        self.words = None
        self.view = None     # numpy view of the pinned words
        self.pending = []
        self.free = list(range(self.SLOTS))

    def poll(self, block=False):
        if not self.pending:
            return
        if block:
            torch.cuda.synchronize()
        keep, bad = [], False
        for slot in self.pending:
            v = int(self.view[slot])
            if v == self.UNSET:
                keep.append(slot)
            else:
                bad = bad or v != 0
                self.free.append(slot)
        self.pending = keep
        if bad:
            raise RuntimeError("rasterize_gaussians: " + lib().gsr_status_string(-6).decode() +
                               " in an earlier forward call (-6): its instance list and everything rendered from it "
                               "are invalid")

    def take(self):
        """Address of a preset host word for the next stage-2 call, and its slot."""
        if self.words is None:
            self.words = torch.full((self.SLOTS,), self.UNSET, dtype=torch.int32).pin_memory()
            self.view = self.words.numpy()
        self.poll()
        if not self.free:
            self.poll(block=True)
        slot = self.free.pop()
        self.view[slot] = self.UNSET
        return self.words.data_ptr() + 4 * slot, slot

    def submitted(self, slot):
        self.pending.append(slot)

    def release(self, slot):
        self.free.append(slot)


sort_errors = SortErrorWords()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().gsr_status_string(rc).decode()} ({rc})")


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the
    reference passes options as empty tensors whose data_ptr() is null)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def stream_of(device):
    return torch.cuda.current_stream(device).cuda_stream


def dev_f32(t, device, what):
    """Contiguous float32 view of an input on `device` (empty tensors pass through)."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError(f"{what} must be on {device}, got {t.device}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what} must be float32, got {t.dtype}")
    return t.contiguous()


def source_hash():
    """Hash of everything libgsplat_amd.so is built from (kernel sources, headers, Makefile).  Profiles
    and counter summaries collected on one build carry it, so that a reader can tell whether they still
    describe the sources at hand."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hip")) + glob.glob(os.path.join(here, "csrc", "*.h")) +
                   [os.path.join(here, "csrc", "Makefile"),
                    os.path.join(os.path.dirname(here), "include", "gsplat_amd.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
