"""Loader for libgsplat_hip.so, the C-ABI library declared in include/gsplat_hip.h.

The product path has no CPU fallback: if the library is missing or a call fails this module
raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C gaussian_splatting_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GSPLAT_HIP_LIB selects another build of the same library (tools only: the instrumented
# libgsplat_hip_stats.so of `make stats`); there is still no fallback if it is missing
LIB_PATH = os.environ.get("GSPLAT_HIP_LIB") or os.path.join(_HERE, "libgsplat_hip.so")

GS_F32 = 0
GS_F64 = 1
GS_SORT_PREFIX = 1024
GS_CUT_HIST_BINS = 8192   # include/gsplat_hip.h
GS_BACKWARD_DEFAULT = -1  # per-call argument of the render-backward entry points: the process default
GS_BACKWARD_COMPAT = 0   # render backward bug-compatible with render_backward.cu:185 (default)
GS_BACKWARD_EXACT = 1    # the exact gradient of the forward pass


def set_backward_mode(mode):
    """"compat" (default: the reference's arithmetic including SURVEY.md Q1) or "exact" (global splat index
    in the transmittance update: the mathematically exact gradient).  Sets the process-wide DEFAULT: the
    render-backward entry points take the mode per call (ABI 5) and the fused frames pass the default that
    was in force at their FORWARD, so changing it never affects a backward that is already queued"""
    code = {"compat": GS_BACKWARD_COMPAT, "exact": GS_BACKWARD_EXACT}.get(mode, mode)
    check(lib().gs_set_backward_mode(int(code)))


def get_backward_mode():
    return int(lib().gs_get_backward_mode())

# every entry point include/gsplat_hip.h declares
EXPORTS = [
    "gs_last_error", "gs_abi_version",
    "gs_camera_projection", "gs_camera_projection_backward",
    "gs_compute_sigma_world", "gs_compute_sigma_world_backward",
    "gs_compute_projection_jacobian", "gs_compute_projection_jacobian_backward",
    "gs_compute_conic", "gs_compute_conic_backward",
    "gs_precompute_rgb_from_sh", "gs_precompute_rgb_from_sh_backward",
    "gs_tile_workspace_ints", "gs_tile_count", "gs_tile_emit_sort", "gs_tile_emit_sort_bounded", "gs_tile_sort_flagged",
    "gs_preprocess_workspace_ints", "gs_preprocess_forward", "gs_preprocess_backward",
    "gs_pack_splats", "gs_render_tiles", "gs_render_tiles_packed", "gs_render_tiles_prefix", "gs_render_tiles_prefix_phased",
    "gs_render_tiles_backward",
    "gs_render_tiles_backward_packed", "gs_render_tiles_backward_slab", "gs_render_backward_prologue",
    "gs_render_segment_workspace_bytes",
    "gs_set_backward_mode", "gs_get_backward_mode", "gs_render_depth", "gs_halo_workspace_ints", "gs_halo_plan", "gs_halo_gather_sum",
    "gs_band_project", "gs_halo_plan_masked", "gs_preprocess_forward_list",
    "gs_band_frontend_workspace_ints", "gs_band_frontend", "gs_band_gather_sum", "gs_preprocess_backward_gathered",
    "gs_render_tiles_prefix_phased_m", "gs_render_tiles_cut_m", "gs_render_tiles_backward_slab_m",
    "gs_adam_step", "gs_accumulate_grad_stats", "gs_stream_copy", "gs_ssim_l1_workspace_bytes", "gs_ssim_l1_loss",
    "gs_densify_move",
    "gs_cut_workspace_ints", "gs_cut_sample_stride", "gs_cut_supported", "gs_preprocess_forward_cut", "gs_tile_count_cut",
    "gs_tile_emit_sort_cut", "gs_cut_debug_views", "gs_render_tiles_cut",
    "gs_band_row_costs", "gs_band_assemble",
]

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: the HIP extension is not built. "
                "Run `make -C gaussian_splatting_amd/csrc` (there is no CPU fallback).")
        # One HIP runtime per process: the library needs libamdhip64.so.7 and must bind to the copy
        # PyTorch ships (same SONAME), whose streams and allocations it works on.  Loaded before torch
        # it would pull in /opt/rocm's runtime instead, and the second runtime to initialise finds no
        # device ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.gs_last_error.restype = ctypes.c_char_p
        _lib.gs_preprocess_workspace_ints.restype = ctypes.c_size_t
        _lib.gs_tile_workspace_ints.restype = ctypes.c_size_t
        _lib.gs_halo_workspace_ints.restype = ctypes.c_size_t
        _lib.gs_band_frontend_workspace_ints.restype = ctypes.c_size_t
        _lib.gs_ssim_l1_workspace_bytes.restype = ctypes.c_size_t
        _lib.gs_render_segment_workspace_bytes.restype = ctypes.c_size_t
        _lib.gs_cut_workspace_ints.restype = ctypes.c_size_t
    return _lib


def current_stream():
    """torch's current stream on the current device as a ctypes pointer for the C ABI.  The raw-handle query
    (~1 us); torch.cuda.current_stream() builds a Stream object and costs ~15 us of host time per call."""
    import torch

    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def check(status):
    if status != 0:
        raise RuntimeError(lib().gs_last_error().decode())


# ---- optional per-entry-point timing (bench.py) --------------------------------------------------------
# When enabled, every C-ABI call is bracketed by events recorded on torch's current stream -- the
# stream the kernels are launched on -- so the elapsed time of one entry point is the GPU time of
# the kernels it enqueues.
_timers = None
_event_pool = []   # timing events are recycled: creating one costs ~5 us of host time per call
# modules that time their own C-ABI calls the same way (the native frame orchestration,
# csrc/frame_hip.cpp): enable_timing / reserve_events / collect_timing are forwarded to them
timing_providers = []


def _event():
    import torch

    return _event_pool.pop() if _event_pool else torch.cuda.Event(enable_timing=True)


def reserve_events(n):
    """create the timing events ahead of the timed region"""
    import torch

    while len(_event_pool) < n:
        _event_pool.append(torch.cuda.Event(enable_timing=True))
    for p in timing_providers:
        p.reserve_events(n)


_only = None   # name of the one entry point to time, or None for all


def enable_timing(on=True, only=None):
    """only: time just this entry point (two events per frame instead of two per C-ABI call)"""
    global _timers, _only
    _timers = {} if on else None
    _only = only if on else None
    for p in timing_providers:
        p.enable_timing(bool(on), only or "")


def collect_timing():
    """-> {entry point: [ms, ...]} for the calls made since enable_timing(); synchronises."""
    import torch

    torch.cuda.synchronize()
    out = {}
    for name, pairs in (_timers or {}).items():
        out[name] = [a.elapsed_time(b) for a, b in pairs]
        for a, b in pairs:
            _event_pool.append(a)
            _event_pool.append(b)
    if _timers is not None:
        _timers.clear()
    for p in timing_providers:
        for name, ms in p.collect_timing().items():
            out.setdefault(name, []).extend(ms)
    return out


def timed_region(name, fn):
    """run fn() and, when timing is enabled, book its GPU time (events on the current stream) under
    `name` next to the C-ABI entry points -- used for the RCCL collectives of the multi-GPU frame"""
    if _timers is None or (_only is not None and _only != name):
        return fn()
    a, b = _event(), _event()
    a.record()
    out = fn()
    b.record()
    _timers.setdefault(name, []).append((a, b))
    return out


def call(name, *args):
    fn = getattr(lib(), name)
    if _timers is None or (_only is not None and _only != name):
        check(fn(*args))
        return
    a, b = _event(), _event()
    a.record()
    check(fn(*args))
    b.record()
    _timers.setdefault(name, []).append((a, b))
