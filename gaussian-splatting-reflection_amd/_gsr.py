"""ctypes binding of libgsr_hip.so (C ABI declared in include/gsr_hip.h).

This is the only place the product path touches native code.  There is NO CPU fallback: if the
library is missing or cannot be loaded, importing the rasterizer packages fails loudly.
torch is used for device memory, streams and autograd plumbing only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB") or os.path.join(_HERE, "libgsr_hip.so")   # GSR_LIB: development A/B builds only

c_void_p, c_int, c_float, c_size_t, c_uint32, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t,
                                                         ctypes.c_uint32, ctypes.c_char_p)
ALLOC_FN = ctypes.CFUNCTYPE(c_void_p, c_void_p, c_int, c_size_t)

GSR_BUF_GEOM, GSR_BUF_BINNING, GSR_BUF_IMAGE = 0, 1, 2


class GsrError(RuntimeError):
    pass


class GatherGroup(ctypes.Structure):
    """gsr_gather_group of include/gsr_hip.h."""
    _fields_ = [("src_offset", ctypes.c_uint64), ("dst_offset", ctypes.c_uint64), ("width", c_uint32)]


class AdamSegment(ctypes.Structure):
    """gsr_adam_segment of include/gsr_hip.h."""
    _fields_ = [("begin", ctypes.c_uint64), ("end", ctypes.c_uint64), ("lr", c_float), ("lr2", c_float), ("period", c_uint32),
                ("split", c_uint32)]


class ReflForward(ctypes.Structure):
    """gsr_refl_forward of include/gsr_hip.h."""
    _fields_ = [("cam", c_void_p), ("cubemap", c_void_p), ("fail_value", c_void_p), ("L", c_uint32), ("cubemap_rgba", c_void_p),
                ("out_final", c_void_p), ("out_refl_color", c_void_p), ("out_normal_world", c_void_p), ("sort_keys", c_void_p),
                ("scratch", c_void_p), ("scratch_floats", c_size_t), ("async_sort", c_int)]


GSR_ABI_VERSION = 101      # GSR_ABI_VERSION of include/gsr_hip.h this binding was written against


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            f"`python {os.path.join(_HERE, 'csrc', 'build.py')}` (hipcc, gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    P = c_void_p
    lib.gsr_last_error.restype = c_char_p
    lib.gsr_version.restype = c_int
    if lib.gsr_version() != GSR_ABI_VERSION:
        # an entry point never changes its signature, but a library built from another header may lack (or, once, have changed) symbols
        # this binding calls: refuse it instead of passing shifted arguments
        raise ImportError(f"{LIB_PATH} reports ABI version {lib.gsr_version()}, this binding was written against {GSR_ABI_VERSION} "
                          f"(include/gsr_hip.h): rebuild with `python {os.path.join(_HERE, 'csrc', 'build.py')} --force`")
    lib.gsr_surfel_forward.restype = c_int
    lib.gsr_surfel_forward.argtypes = [ALLOC_FN, P, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, P, P, c_float, P, P, P, P, P,
                                       c_float, c_float, c_int, P, P, P, P, P, c_int, P]
    lib.gsr_surfel_backward.restype = c_int
    lib.gsr_surfel_backward.argtypes = [c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, c_float, P, P, P, P, P, c_float, c_float,
                                        P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P]
    lib.gsr_surfel_backward_accum.restype = c_int
    lib.gsr_surfel_backward_accum.argtypes = lib.gsr_surfel_backward.argtypes[:-2] + [c_int, c_int, P]
    lib.gsr_surfel_backward_ex.restype = c_int
    lib.gsr_surfel_backward_ex.argtypes = lib.gsr_surfel_backward.argtypes[:-2] + [c_int, P, c_int, P]
    lib.gsr_surfel_forward_refl.restype = c_int
    lib.gsr_surfel_forward_refl.argtypes = lib.gsr_surfel_forward.argtypes[:-2] + [ctypes.POINTER(ReflForward), c_int, P]
    lib.gsr_gauss_forward.restype = c_int
    lib.gsr_gauss_forward.argtypes = [ALLOC_FN, P, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, P, P, c_float, P, P, P, P, P,
                                      c_float, c_float, c_int, P, P, P, P, c_int, P, c_int, P]
    lib.gsr_gauss_backward.restype = c_int
    lib.gsr_gauss_backward.argtypes = [c_int, c_int, c_int, c_int, P, c_int, c_int, P, P, P, P, P, P, P, c_float, P, P, P, P, P, c_float,
                                       c_float, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, P]
    lib.gsr_gauss_backward_accum.restype = c_int
    lib.gsr_gauss_backward_accum.argtypes = lib.gsr_gauss_backward.argtypes[:-2] + [c_int, c_int, P]
    lib.gsr_mark_visible.restype = c_int
    lib.gsr_mark_visible.argtypes = [c_int, P, P, P, P, P]
    lib.gsr_debug_fetch.restype = c_int
    lib.gsr_debug_fetch.argtypes = [c_int, c_char_p, c_int, c_int, c_int, c_int, P, P, P, P, P]
    lib.gsr_cubemap_forward.restype = c_int
    lib.gsr_cubemap_forward.argtypes = [P, P, P, P, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, P]
    lib.gsr_cubemap_backward.restype = c_int
    lib.gsr_cubemap_backward.argtypes = [P, P, P, P, P, P, c_uint32, c_uint32, c_uint32, c_uint32, c_uint32, P]
    lib.gsr_deferred_reflection_forward.restype = c_int
    lib.gsr_deferred_reflection_forward.argtypes = [P, P, P, P, P, P, c_uint32, c_int, c_int, P, P, P, P]
    lib.gsr_deferred_reflection_backward.restype = c_int
    lib.gsr_deferred_reflection_scratch_floats.restype = c_size_t
    lib.gsr_deferred_reflection_scratch_floats.argtypes = [c_uint32, c_int, c_int, c_int]
    lib.gsr_deferred_reflection_backward.argtypes = [P, P, P, P, P, P, c_uint32, c_int, c_int, P, P, P, P, P, P, P, P, P, c_size_t, P]
    lib.gsr_deferred_reflection_backward_accum.restype = c_int
    lib.gsr_deferred_reflection_backward_accum.argtypes = lib.gsr_deferred_reflection_backward.argtypes[:-1] + [c_int, P]
    lib.gsr_deferred_reflection_backward_ex.restype = c_int
    lib.gsr_deferred_reflection_backward_ex.argtypes = lib.gsr_deferred_reflection_backward.argtypes[:-1] + [c_int, c_int, P, P]
    lib.gsr_deferred_reflection_backward_keys.restype = c_int
    lib.gsr_deferred_reflection_backward_keys.argtypes = lib.gsr_deferred_reflection_backward.argtypes[:-1] + [c_int, c_int, P, P, c_int, P]
    lib.gsr_deferred_reflection_forward_ex.restype = c_int
    lib.gsr_deferred_reflection_forward_ex.argtypes = lib.gsr_deferred_reflection_forward.argtypes[:-1] + [P, P]
    lib.gsr_deferred_reflection_forward_keys.restype = c_int
    lib.gsr_deferred_reflection_forward_keys.argtypes = lib.gsr_deferred_reflection_forward.argtypes[:-1] + [P, P, P]
    lib.gsr_side_join.restype = c_int
    lib.gsr_side_join.argtypes = [P]
    lib.gsr_normal_world_forward.restype = c_int
    lib.gsr_normal_world_forward.argtypes = [P, P, c_int, c_int, P, P]
    lib.gsr_normal_world_backward.restype = c_int
    lib.gsr_normal_world_backward.argtypes = [P, P, c_int, c_int, P, P, P]
    lib.gsr_ssim_l1_forward.restype = c_int
    lib.gsr_ssim_l1_scratch_floats.restype = c_size_t
    lib.gsr_ssim_l1_scratch_floats.argtypes = [c_int, c_int, c_int]
    lib.gsr_ssim_l1_forward.argtypes = [P, P, c_int, c_int, c_int, c_float, c_float, P, P, P, P, P, P, P]
    lib.gsr_ssim_l1_backward.restype = c_int
    lib.gsr_ssim_l1_backward.argtypes = [P, P, c_int, c_int, c_int, P, P, P, P, P, P]
    lib.gsr_normal_loss_scratch_floats.restype = c_size_t
    lib.gsr_normal_loss_scratch_floats.argtypes = []
    lib.gsr_normal_loss_forward.restype = c_int
    lib.gsr_normal_loss_forward.argtypes = [P, P, P, c_int, c_int, P, P, P]
    lib.gsr_normal_loss_backward.restype = c_int
    lib.gsr_normal_loss_backward.argtypes = [P, P, P, c_int, c_int, P, P, P, P]
    lib.gsr_surface_forward.restype = c_int
    lib.gsr_surface_forward.argtypes = [P, P, c_float, c_int, c_int, P, P, P]
    lib.gsr_surface_backward.restype = c_int
    lib.gsr_surface_backward.argtypes = [P, P, c_float, c_int, c_int, P, P, P, P, P]
    lib.gsr_densification_stats.restype = c_int
    lib.gsr_densification_stats.argtypes = [c_int, P, P, P, P, P, P, P, P, P]
    lib.gsr_gather_rows.restype = c_int
    lib.gsr_gather_rows.argtypes = [P, P, P, ctypes.c_uint64, ctypes.POINTER(GatherGroup), c_int, P]
    lib.gsr_split_children.restype = c_int
    lib.gsr_split_children.argtypes = [c_int, c_int, c_int, P, P, P, P, P, P, P, P]
    lib.gsr_adam_step.restype = c_int
    lib.gsr_adam_step.argtypes = [P, P, P, P, ctypes.c_uint64, ctypes.POINTER(AdamSegment), c_int, c_float, c_float, c_float, c_int, P]
    lib.gsr_adam_step_range.restype = c_int
    lib.gsr_adam_step_range.argtypes = lib.gsr_adam_step.argtypes[:-1] + [ctypes.c_uint64, ctypes.c_uint64, P]
    lib.gsr_set_option.restype = c_int
    lib.gsr_set_option.argtypes = [c_char_p, c_int]
    lib.gsr_profile_enable.restype = c_int
    lib.gsr_profile_enable.argtypes = [c_int]
    lib.gsr_profile_collect.restype = c_int
    lib.gsr_profile_collect.argtypes = [P, P]
    return lib


lib = _load()


def compiled_binding():
    """The optional compiled torch/pybind binding (csrc/gsr_torch_binding.cpp -> _gsr_C.so): marshaling in C++ over the same C
    ABI, about half the host cost per call of the ctypes path (tests/host_overhead.py at C1: 25 vs 49 us per backward).
    GSR_BINDING=pybind requires it (an error if it is not built), GSR_BINDING=ctypes never uses it; unset: used when present and
    loadable (it is built against the running torch; both bindings drive the same HIP library, so this choice is not a
    fallback away from native code)."""
    want = os.environ.get("GSR_BINDING", "auto")
    if want == "ctypes":
        return None
    import importlib.util
    path = os.path.join(_HERE, "_gsr_C.so")
    if not os.path.exists(path):
        if want == "pybind":
            raise ImportError(f"GSR_BINDING=pybind but {path} is not built: run `python {os.path.join(_HERE, 'csrc', 'build.py')} --binding`")
        return None
    try:
        spec = importlib.util.spec_from_file_location("_gsr_C", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    except Exception:
        if want == "pybind":
            raise
        return None


PYBIND = compiled_binding()

EXPORTED = ["gsr_last_error", "gsr_version", "gsr_surfel_forward", "gsr_surfel_backward", "gsr_surfel_backward_accum", "gsr_surfel_backward_ex",
            "gsr_surfel_forward_refl", "gsr_deferred_reflection_backward_keys", "gsr_deferred_reflection_forward_keys",
            "gsr_deferred_reflection_backward_accum", "gsr_deferred_reflection_backward_ex", "gsr_deferred_reflection_forward_ex", "gsr_side_join", "gsr_normal_world_forward", "gsr_normal_world_backward", "gsr_gauss_forward", "gsr_gauss_backward", "gsr_gauss_backward_accum",
            "gsr_mark_visible", "gsr_debug_fetch", "gsr_cubemap_forward", "gsr_cubemap_backward", "gsr_deferred_reflection_forward",
            "gsr_deferred_reflection_scratch_floats", "gsr_deferred_reflection_backward", "gsr_ssim_l1_scratch_floats", "gsr_ssim_l1_forward", "gsr_ssim_l1_backward", "gsr_normal_loss_scratch_floats", "gsr_normal_loss_forward", "gsr_normal_loss_backward", "gsr_adam_step", "gsr_adam_step_range", "gsr_densification_stats", "gsr_gather_rows", "gsr_split_children", "gsr_surface_forward", "gsr_surface_backward", "gsr_profile_enable",
            "gsr_profile_collect", "gsr_set_option"]

STAGES = ["preprocess", "scan_readback", "emit_keys", "sort", "tile_ranges", "render_fwd", "render_bwd", "preprocess_bwd", "refl_fwd",
          "refl_bwd", "cubemap_fwd", "cubemap_bwd", "loss_fwd", "loss_bwd", "adam", "surface_fwd", "surface_bwd", "refl_bwd_tail"]


def set_option(name, value):
    check(lib.gsr_set_option(name.encode(), int(value)), f"gsr_set_option({name})")


_side_held = {}          # device index -> [(tensor the side stream of THAT device still reads or writes, stream it was allocated / last used on)]


def side_hold(*tensors):
    """Keeps device tensors alive that work on the library's side stream still reads (see side_join), and remembers the stream that is
    current now — the one the caching allocator will hand their blocks back to."""
    for t in tensors:
        if t is None:
            continue
        dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
        _side_held.setdefault(dev, []).append((t, torch.cuda.current_stream(t.device).cuda_stream))


def side_join(device=None):
    """Makes the current stream of `device` wait for everything the library has put on its side stream (the texel-gradient tail
    of deferred_reflection(..., async_tail=True)), then lets go of the scratch tensors held for it.  No host synchronisation.
    Call before anything reads the cubemap / fail-value gradient sink: gsr_dist.FlatGrads and gsr_train.FlatAdam do."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    # always ask the library (a stream-wait on an event that has completed, or was never recorded, costs nothing): a second
    # consumer stream must be ordered behind the tail too, and another device's held tensors are none of this call's business
    held = _side_held.pop(dev.index, [])
    with torch.cuda.device(dev):
        cur = torch.cuda.current_stream(dev).cuda_stream
        check(lib.gsr_side_join(cur), "gsr_side_join")
        # The held tensors go back to the caching allocator, which re-issues a block to work on the stream it was ALLOCATED on — not
        # necessarily the stream joining here (an all-reduce or optimizer stream, bench.py --view-streams).  Every such stream is ordered
        # behind the tail as well, so whatever reuses the memory cannot run while the tail still reads or writes it.
        for s in {s for _, s in held if s != cur}:
            check(lib.gsr_side_join(s), "gsr_side_join")
    del held


def profile_enable(on=True):
    lib.gsr_profile_enable(1 if on else 0)


def profile_collect():
    """Returns {stage: (total_ms, launches)} measured with hipEvents on the launch stream since the last call."""
    ms = (ctypes.c_float * len(STAGES))()
    n = (ctypes.c_int * len(STAGES))()
    lib.gsr_profile_collect(ctypes.cast(ms, c_void_p), ctypes.cast(n, c_void_p))
    return {STAGES[i]: (float(ms[i]), int(n[i])) for i in range(len(STAGES))}


def check(rc, what):
    if rc < 0:
        msg = lib.gsr_last_error()
        raise GsrError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
    return rc


def ptr(t):
    """Device pointer of a tensor, or NULL for None / empty tensors (the reference passes
    `.contiguous().data<float>()`, which is nullptr for empty tensors, and tests `== nullptr`)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, name):
    # CHECK_INPUT of DSR rasterize_points.cu:27-28
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def f32c(t, name):
    """Contiguous float32 view (reference: `.contiguous().data<float>()` throws on other dtypes)."""
    if t.numel() and t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")
    return t.contiguous()


class Workspace:
    """The three opaque byte buffers of one forward call (geomBuffer, binningBuffer, imgBuffer of
    DSR rasterize_points.cu:102-108), allocated by torch when the library asks for them."""

    def __init__(self, device):
        self.device = device
        self.bufs = [torch.empty(0, dtype=torch.uint8, device=device) for _ in range(3)]
        self.error = None

        def _alloc(user, which, nbytes):
            try:
                t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
                self.bufs[which] = t
                return t.data_ptr()
            except Exception as e:  # surfaced as GSR_E_ALLOC by the library
                self.error = e
                return 0

        self.cb = ALLOC_FN(_alloc)


def debug_fetch(variant, name, P, R, W, H, geom, binning, img, dtype, shape):
    """Parity-test helper: copy a named workspace array out (see gsr_debug_fetch in gsr_hip.h)."""
    out = torch.empty(shape, dtype=dtype, device=geom.device)
    check(lib.gsr_debug_fetch(variant, name.encode(), P, R, W, H, ptr(geom), ptr(binning), ptr(img), ptr(out) if out.numel() else None,
                              stream_ptr(geom.device)), f"gsr_debug_fetch({name})")
    return out
