"""Fast path with the contract of splat_py.rasterize.rasterize (reference: splat_py/rasterize.py:18-112):

    image, culling_mask, uv = rasterize(gaussians, camera_T_world, camera, near_thresh, far_thresh,
                                        cull_mask_padding, mh_dist, use_sh_precompute, background_rgb)

Two autograd nodes instead of six-plus-glue:

    _Preprocess   gs_preprocess_forward + tile binning + per-tile sort   (parameters -> uv, conic,
                  opacity, colour; non-differentiable: packed records, tile lists, culling mask)
    _Render       gs_render_tiles_prefix (_packed) / gs_render_tiles_backward_slab

so `uv` is still an autograd intermediate between two nodes: `uv.retain_grad()` followed by
`uv.grad` gives the render-backward grad_uv exactly as the trainer expects (trainer.py:360,379).
One small device->host read per frame (V and S, to size the outputs); the reference has ~25
synchronisation points.  fp32, SH-precompute colour mode; anything else is routed
to the reference-shaped path in splat_py.rasterize (still HIP kernels, never a CPU fallback).

The four stages are plain functions (preprocess_forward, render_forward, render_backward,
preprocess_backward) that gaussian_splatting_amd.sharded composes differently for multi-GPU frames.
"""
import ctypes
import os
from types import SimpleNamespace

import torch

from . import _hip
from .splat_py import rasterize as _reference_shaped
from .splat_py.structs import TILE_EDGE_LENGTH_PX
from .splat_py.utils import compute_rays_in_world_frame


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    # every stage asks once and passes it on
    return _hip.current_stream()


def _cf(x):
    return ctypes.c_float(float(x))


# Prefix mode of the per-tile sort (include/gsplat_hip.h: gs_tile_emit_sort): order only the 1024
# nearest entries of long tile lists and repair, on the device, the tiles that needed more.  Exact;
# off only for callers that want the complete sorted lists back (return_aux=True).
SORT_PREFIX = True
# mean tile-list length from which the backward starts its tiles longest-first (render_backward)
LPT_MIN_MEAN_LIST = 256
# Enqueue the render on the speculative tile lists before waiting for the frame's host read.
EARLY_RENDER = True

# Depth cut (csrc/binning.hip "depth cut"; include/gsplat_hip.h): the count pass walks the Gaussians in depth buckets,
# every tile learns the last bucket up to which it holds <= 1024 entries, and only those are emitted and sorted; a
# truncated tile that runs out of list unsaturated is repaired on the device from its complete list.  Exact.
# "auto": whole frames in the LDS-histogram regime whose lists averaged DEPTH_CUT_MIN_MEAN_LIST entries or more in
# an earlier frame of the same shape (the partition of the Gaussians by depth costs ~0.03 ms: it must buy more than
# that); True / False force it.  Callers that want the complete lists back (return_aux) never get it.
DEPTH_CUT = {"0": False, "1": True}.get(os.environ.get("GSPLAT_DEPTH_CUT", "auto"), "auto")   # (env: A/B runs of bench.py)
DEPTH_CUT_MIN_MEAN_LIST = 1280
_mean_list_hint = {}   # frame shape -> complete instance count of the latest frame

last_tile_flags = None   # int32[T] of the latest prefix-mode frame of the Python path (see last_flags())
# Frames without hooks go through the native orchestration (csrc/frame_hip.cpp: the same C-ABI calls and
# autograd structure in C++, one Python call per frame).  False = always the Python orchestration below.
NATIVE = True
_native_mod = None
_native_failed = False


def native():
    """the gsplat_frame module, or None when NATIVE is off / it is not built (the Python orchestration of
    the same HIP kernels is then used; this is a host-side speed-up, not a compute fallback)"""
    global _native_mod, _native_failed
    if not NATIVE or _native_failed:
        return None
    if _native_mod is None:
        try:
            from . import splat_cuda_native
            _native_mod = splat_cuda_native.load_frame()
            _hip.timing_providers.append(_native_mod)
        except ImportError as e:
            import warnings
            _native_failed = True
            warnings.warn(f"native frame orchestration unavailable ({e}); using the Python orchestration")
            return None
    return _native_mod


def last_flags(clear=False):
    """tile_flags (int32[T]; 1 = the prefix sort ran out and the tile was repaired) of the latest
    prefix-mode frame, whichever orchestration ran it; None if there was none since the last clear"""
    global last_tile_flags
    out = last_tile_flags
    m = _native_mod
    if m is not None:
        nat = m.last_tile_flags(clear)
        out = nat if nat is not None else out
    if clear:
        last_tile_flags = None
    return out


_capacity_hint = {}   # (device, N, tiles, band) -> instance capacity guessed from the previous frame
_pinned = {}

SLAB_WIDTH = 9   # render gradients per visible Gaussian: rgb 3 | opacity 1 | uv 2 | conic 3
SLAB_RGB, SLAB_OPACITY, SLAB_UV, SLAB_CONIC = slice(0, 3), slice(3, 4), slice(4, 6), slice(6, 9)


_PINNED_RING = 4   # host read buffers per (device, size): frames in flight on other streams / threads keep their own


def _pinned_ints(dev, n):
    """a pinned int32[n] for the frame's host read, from a small ring: a frame whose read is still
    pending (another stream or thread) does not see its record overwritten by the next frame"""
    ring = _pinned.get((dev.index, n))
    if ring is None:
        ring = _pinned[(dev.index, n)] = [[torch.empty(n, dtype=torch.int32, pin_memory=True) for _ in range(_PINNED_RING)], 0]
    ring[1] = (ring[1] + 1) % _PINNED_RING
    return ring[0][ring[1]]


# ---- frame counters (bench.py --moving-camera reports them) ---------------------------------------------
_counters = {"frames": 0, "speculative_frames": 0, "capacity_misses": 0, "S_min": None, "S_max": None,
             "depth_cut_frames": 0}
_flag_log = []   # tile_flags of recent prefix-mode frames (device tensors: summed only when counters() is asked)


def reset_counters():
    _counters.update(frames=0, speculative_frames=0, capacity_misses=0, S_min=None, S_max=None, depth_cut_frames=0)
    _flag_log.clear()
    if _native_mod is not None:
        _native_mod.reset_counters()


def counters():
    """frames seen since reset_counters(): how many were enqueued on a guessed capacity, how many of those
    had to repeat emit + sort + render because the instance count exceeded the guess, the range of S, and
    the tiles the prefix sort had to repair (flag-and-redo on the device).  Synchronises."""
    out = dict(_counters)
    out["prefix_repaired_tiles"] = int(sum(int(f.sum()) for f in _flag_log)) if _flag_log else 0
    out["prefix_frames_logged"] = len(_flag_log)
    out["orchestration"] = "python"
    if _native_mod is not None:
        nat = _native_mod.counters()
        if nat["frames"]:
            out["depth_cut_backoffs"] = nat.get("depth_cut_backoffs", 0)
            for k in ("prefix_frames_without_repair_launches", "prefix_late_repairs", "long_list_misses"):
                out[k] = nat.get(k, 0)
            for k in ("frames", "speculative_frames", "capacity_misses", "prefix_repaired_tiles", "prefix_frames_logged",
                      "depth_cut_frames"):
                out[k] += nat[k]
            lo = [x for x in (out["S_min"], nat["S_min"]) if x is not None]
            hi = [x for x in (out["S_max"], nat["S_max"]) if x is not None]
            out["S_min"], out["S_max"] = (min(lo) if lo else None), (max(hi) if hi else None)
            out["orchestration"] = "native (csrc/frame_hip.cpp)" if nat["frames"] == out["frames"] else "mixed"
    return out


class _Arena:
    """one allocation cut into 1-D blocks of the given element counts, each starting 16-byte aligned"""

    def __init__(self, dtype, device, sizes, zero=False):
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append((off, n))
            off += (n + 3) & ~3
        self.buf = (torch.zeros if zero else torch.empty)(off, dtype=dtype, device=device)

    def blocks(self):
        return [self.buf[o:o + n] for o, n in self.offsets]


# ---------------------------------------------------------------------------------------------------
# stages
# ---------------------------------------------------------------------------------------------------
def want_depth_cut(hint_key, N, ntx, row0, row1, whole):
    """the "auto" policy of DEPTH_CUT for one frame"""
    if DEPTH_CUT is False or not whole:
        return False
    if not _hip.lib().gs_cut_supported(ntx, row0, row1, N):
        return False
    if DEPTH_CUT is True:
        return True
    n_tiles = (row1 - row0) * ntx
    known = _mean_list_hint.get(hint_key, 0)
    if known < DEPTH_CUT_MIN_MEAN_LIST * n_tiles:
        return False
    # "auto" cut and "auto" segments exclude each other per shape: a cut frame takes the unsegmented backward, so a
    # small dense frame would change backward kernels (and the last bits of its gradients) whenever the cut policy
    # switches; such shapes keep the segments.  What counts is the choice segments_for() STORED for the shape when
    # there is one; segments forced on (SEGMENTS = True) keep the auto cut off -- a cut frame cannot honour them
    # (csrc/frame_hip.cpp want_depth_cut does the same)
    if SEGMENTS != "auto":
        return not bool(SEGMENTS)
    if hint_key in _segment_choice:
        return not _segment_choice[hint_key]
    return not want_segments(known, n_tiles)


def preprocess_forward(xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, width, height, near_thresh,
                       far_thresh, cull_mask_padding, mh_dist, tile_rows, sort_prefix, plan=None, plan_ints=0,
                       defer=False, depth_cut=False):
    """Per-Gaussian stage, binning and per-tile sort of one frame.  `plan`, if given, is called as
    plan(f) after the per-Gaussian stage is enqueued and fills the device record f.record
    (plan_ints int32) that rides on the frame's one host read (its host copy is left in f.host:
    multi-GPU, the split sizes of the gradient exchange); it may set f.subset = (index list, device
    count) to restrict the binning to those rows.  defer=True returns before the host read: the caller may enqueue the render on
    the speculative lists (f.sorted_buf, f.keys_buf, f.capacity) and calls preprocess_finish(f)."""
    dev = xyz.device
    N = xyz.shape[0]
    n_sh = 1 if sh is None else sh.shape[2] + 1
    i32 = dict(dtype=torch.int32, device=dev)
    ntx = (width + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
    nty = (height + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
    T = ntx * nty
    row0, row1 = tile_rows if tile_rows is not None else (0, nty)
    f = SimpleNamespace(N=N, n_sh=n_sh, ntx=ntx, nty=nty, T=T, row0=row0, row1=row1, width=width, height=height,
                        mh_dist=mh_dist, sort_prefix=sort_prefix)

    stream = f.stream = _stream()
    f.hint_key = (dev.index, N, T, row0, row1)
    if depth_cut:
        assert plan is None and not defer, "the depth cut is a single-GPU, whole-frame path in the Python orchestration"
        return _preprocess_forward_cut(f, xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, near_thresh,
                                       far_thresh, cull_mask_padding)
    # two allocations for the stage's buffers (host time matters at ~1 ms per frame): an int32 arena
    # for the bookkeeping and a float32 arena for the per-Gaussian outputs, blocks 16-byte aligned
    n_ws = _hip.lib().gs_preprocess_workspace_ints(N)
    n_tc = _hip.lib().gs_tile_workspace_ints(T)
    iar = _Arena(torch.int32, dev, (n_ws, 1, N, N, n_tc, T + 2 + plan_ints, (N + 3) // 4))
    f.ws, f.count, f.rank, f.vis_idx, f.tile_counts, f.ranges_buf, mask_bytes = iar.blocks()
    f.culling_mask = mask_bytes.view(torch.bool)[:N]   # kernel writes 0/1 bytes
    far = _Arena(torch.float32, dev, (3, 2 * N, 3 * N, 3 * N, N, 3 * N, 12 * N))
    f.center, uv, xyz_cam, conic, opa, rgbr, packed = far.blocks()
    f.uv, f.xyz_cam, f.conic = uv.view(N, 2), xyz_cam.view(N, 3), conic.view(N, 3)
    f.opacity_act, f.rgb_render, f.packed = opa.view(N, 1), rgbr.view(N, 3), packed.view(N, 12)
    f.cut = None
    _hip.call("gs_preprocess_forward", _p(xyz), _p(quaternion), _p(scale), _p(opacity), _p(rgb), _p(sh), n_sh,
              _p(camera_T_world), _p(K), N, width, height, _cf(near_thresh), _cf(far_thresh),
              _cf(cull_mask_padding), _cf(mh_dist), row0, row1, _p(f.ws), _p(f.center), _p(f.count),
              _p(f.culling_mask), _p(f.rank), _p(f.vis_idx), _p(f.uv), _p(f.xyz_cam), _p(f.conic),
              _p(f.opacity_act), _p(f.rgb_render), _p(f.packed), stream)

    # the plan's record lives right behind (S, V) at the tail of ranges_buf: one host read gets both
    f.subset = (None, None)
    f.record = f.ranges_buf[T + 2:]
    if plan is not None:
        plan(f)
    subset, subset_n = f.subset
    _hip.call("gs_tile_count", _p(f.uv), _p(f.conic), N, _p(f.count), _p(subset), _p(subset_n), ntx, nty,
              _cf(mh_dist), row0, row1, _p(f.tile_counts), _p(f.ranges_buf), None, stream)

    def emit_sort(capacity):
        sorted_buf = torch.empty(capacity, **i32)
        keys = torch.empty(capacity, dtype=torch.int64, device=dev)
        if capacity > 0:
            _hip.call("gs_tile_emit_sort", _p(f.uv), _p(f.xyz_cam), _p(f.conic), N, _p(f.count), _p(subset),
                      _p(subset_n), ntx, nty, _cf(mh_dist), row0, row1, _p(f.ranges_buf), _p(f.tile_counts), _p(keys),
                      ctypes.c_int64(capacity), _p(sorted_buf), sort_prefix, stream)
        return sorted_buf, keys

    # The frame's only device->host read: (S, V) [+ the plan's record], to size the outputs.  If a
    # previous frame of the same shape is known, the emit + sort are enqueued first with a capacity
    # guessed from it, so the GPU keeps working while the host waits for the integers; the kernels
    # never write beyond the capacity and the step is repeated only if S turned out larger.
    f.emit_sort = emit_sort
    guess = _capacity_hint.get(f.hint_key)
    f.host_buf = _pinned_ints(dev, 2 + plan_ints)

    def read_back(non_blocking):
        f.host_buf.copy_(f.ranges_buf[T:], non_blocking=non_blocking)

    f.ranges = f.ranges_buf[:T + 1]
    f.speculative = guess is not None
    if f.speculative:
        read_back(True)
        f.ready = torch.cuda.Event()
        f.ready.record()
        f.capacity = guess
        f.sorted_buf, f.keys_buf = emit_sort(guess)
    else:
        read_back(False)
        f.capacity = int(f.host_buf[0])
        f.sorted_buf, f.keys_buf = emit_sort(f.capacity)
    f.deferred = defer
    if not defer:
        preprocess_finish(f)
    return f


_DEPTH_HIST = {}


def _depth_hist(dev):
    """the depth histogram gs_preprocess_forward_cut fills and returns to zero: one per (device, stream), zeroed once"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _DEPTH_HIST:
        _DEPTH_HIST[key] = torch.zeros(_hip.GS_CUT_HIST_BINS, dtype=torch.int32, device=dev)
    return _DEPTH_HIST[key]


def _preprocess_forward_cut(f, xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, near_thresh, far_thresh,
                            cull_mask_padding):
    """preprocess_forward with the depth-bucketed binning (DEPTH_CUT).  The Python orchestration of this mode reads
    the frame's counts before it emits (no speculative capacity): it serves tests and tools; frames without hooks
    run through csrc/frame_hip.cpp."""
    lib = _hip.lib()
    dev = xyz.device
    N, T, ntx, nty, row0, row1 = f.N, f.T, f.ntx, f.nty, f.row0, f.row1
    stream = f.stream
    stride = lib.gs_cut_sample_stride(N)
    n_ws = lib.gs_preprocess_workspace_ints(N)
    n_tc = lib.gs_tile_workspace_ints(T)
    n_cut = lib.gs_cut_workspace_ints(N, T)
    iar = _Arena(torch.int32, dev, (n_ws, 1, N, N, n_tc, T + 3, (N + 3) // 4, n_cut, T + 1))
    f.ws, f.count, f.rank, f.vis_idx, f.tile_counts, f.ranges_buf, mask_bytes, cut_ws, full_ranges = iar.blocks()
    f.culling_mask = mask_bytes.view(torch.bool)[:N]
    far = _Arena(torch.float32, dev, (3, 2 * N, 3 * N, 3 * N, N, 3 * N, 12 * N, 8 * N))
    f.center, uv, xyz_cam, conic, opa, rgbr, packed, bin_rec = far.blocks()
    f.uv, f.xyz_cam, f.conic = uv.view(N, 2), xyz_cam.view(N, 3), conic.view(N, 3)
    f.opacity_act, f.rgb_render, f.packed = opa.view(N, 1), rgbr.view(N, 3), packed.view(N, 12)
    _hip.call("gs_preprocess_forward_cut", _p(xyz), _p(quaternion), _p(scale), _p(opacity), _p(rgb), _p(sh), f.n_sh,
              _p(camera_T_world), _p(K), N, f.width, f.height, _cf(near_thresh), _cf(far_thresh),
              _cf(cull_mask_padding), _cf(f.mh_dist), row0, row1, _p(f.ws), _p(f.center), _p(f.count),
              _p(f.culling_mask), _p(f.rank), _p(f.vis_idx), _p(f.uv), _p(f.xyz_cam), _p(f.conic),
              _p(f.opacity_act), _p(f.rgb_render), _p(f.packed), _p(bin_rec), _p(cut_ws), _p(_depth_hist(dev)), stride,
              stream)
    _hip.call("gs_tile_count_cut", _p(bin_rec), N, _p(f.count), ntx, nty, _cf(f.mh_dist), row0, row1,
              _p(f.tile_counts), _p(cut_ws), _p(f.ranges_buf), _p(full_ranges), None, stream)
    f.subset = (None, None)
    f.record = f.ranges_buf[T + 3:]
    f.host_buf = _pinned_ints(dev, 3)
    f.host_buf.copy_(f.ranges_buf[T:T + 3])
    S, V, S_full = (int(x) for x in f.host_buf.tolist())
    sorted_buf = torch.empty(S, dtype=torch.int32, device=dev)
    keys = torch.empty(S, dtype=torch.int64, device=dev)
    if S > 0:
        _hip.call("gs_tile_emit_sort_cut", _p(bin_rec), N, ntx, nty, _cf(f.mh_dist), row0, row1, _p(f.ranges_buf),
                  _p(f.tile_counts), _p(cut_ws), _p(keys), ctypes.c_int64(S), _p(sorted_buf), stream)
    c = _counters
    c["frames"] += 1
    c["depth_cut_frames"] += 1
    c["S_min"] = S_full if c["S_min"] is None else min(c["S_min"], S_full)
    c["S_max"] = S_full if c["S_max"] is None else max(c["S_max"], S_full)
    _mean_list_hint[f.hint_key] = S_full
    f.ranges = f.ranges_buf[:T + 1]
    f.speculative, f.deferred, f.capacity = False, False, S
    f.sorted_buf = f.sorted_g = sorted_buf
    f.keys_buf = f.keys = keys
    f.S, f.V, f.host = S, V, []
    f.cut = SimpleNamespace(bin_rec=bin_rec, cut_ws=cut_ws, full_ranges=full_ranges, S_full=S_full, N=N,
                            tile_counts=f.tile_counts, mh_dist=f.mh_dist, flags=None, overflow_sorted=None)
    return f


def preprocess_finish(f):
    """Waits for the frame's host read and fixes the sizes (f.S, f.V, f.sorted_g, f.keys, f.host).
    -> True if the speculative capacity was too small and emit + sort were repeated (anything
    enqueued on the speculative lists in between has to be repeated too)."""
    redone = False
    if f.speculative:
        f.ready.synchronize()
    S, V = int(f.host_buf[0]), int(f.host_buf[1])
    c = _counters
    c["frames"] += 1
    c["speculative_frames"] += int(f.speculative)
    c["S_min"] = S if c["S_min"] is None else min(c["S_min"], S)
    c["S_max"] = S if c["S_max"] is None else max(c["S_max"], S)
    if S > f.capacity:
        f.sorted_buf, f.keys_buf = f.emit_sort(S)
        f.capacity = S
        redone = True
        c["capacity_misses"] += 1
    # the capacity only sizes two buffers (12 B per instance; the kernels write S entries whatever it is), so
    # the guess is the largest count seen for this frame shape plus a margin: views of a training run
    # differ by tens of percent in S, and a miss costs a repeated emit + sort + render
    _capacity_hint[f.hint_key] = max(_capacity_hint.get(f.hint_key, 0), int(S * 1.25) + 4096)
    _mean_list_hint[f.hint_key] = S
    f.host = f.host_buf.tolist()[2:]
    f.S, f.V = S, V
    f.sorted_g = f.sorted_buf[:S]
    f.keys = f.keys_buf[:S]   # prefix mode: the repair pass of the render sorts flagged tiles from these
    return redone


def preprocess_backward(xyz, quaternion, scale, camera_T_world, K, f, slab, v_base=0, i0=0, i1=None):
    """Dense parameter gradients of the Gaussians [i0, i1) from the render-gradient slab
    (row of visible Gaussian v at slab[v - v_base]).  xyz, quaternion, scale: the full tensors."""
    i1 = f.N if i1 is None else i1
    n = i1 - i0
    dev = xyz.device
    n_extra = 3 * (f.n_sh - 1)
    gx, gq, gs, go, gc, gsh = _Arena(torch.float32, dev, (3 * n, 4 * n, 3 * n, n, 3 * n, n_extra * n)).blocks()
    grad_xyz, grad_q, grad_scale = gx.view(n, 3), gq.view(n, 4), gs.view(n, 3)
    grad_opacity, grad_rgb = go.view(n, 1), gc.view(n, 3)
    grad_sh = gsh.view(n, 3, f.n_sh - 1) if f.n_sh > 1 else None
    if n > 0:
        _hip.call("gs_preprocess_backward", _p(xyz[i0:i1]), _p(quaternion[i0:i1]), _p(scale[i0:i1]), f.n_sh,
                  _p(camera_T_world), _p(K), _p(f.center), _p(f.rank[i0:i1]), _p(f.opacity_act), _p(slab),
                  int(v_base), n, _p(grad_xyz), _p(grad_q), _p(grad_scale), _p(grad_opacity), _p(grad_rgb),
                  _p(grad_sh), _stream())
    return grad_xyz, grad_q, grad_scale, grad_opacity, grad_rgb, grad_sh


# Depth segments of the backward (csrc/render.hip "depth segments"): the forward leaves, per (tile, 128-entry
# segment of its list, pixel), the state a backward walk has at the segment boundary, and the backward runs one
# workgroup per (tile, segment).  "auto": for frames / bands of fewer than SEGMENT_MAX_TILES tiles whose lists
# average SEGMENT_MIN_MEAN_LIST entries or more -- a multi-GPU rank's band is two workgroups per CU otherwise.
# Measured at workload D, forward + backward: 1/8 band 0.259 -> 0.205 ms (backward 0.173 -> 0.093, the forward pays
# 0.086 -> 0.112 for the extra state), 1/4 band 0.321 -> 0.298, half frame 0.463 -> 0.489: on below 1500 tiles.
# True / False force it.
SEGMENTS = "auto"
SEGMENT_MAX_TILES = 1500
SEGMENT_MIN_MEAN_LIST = 192
_EMPTY = {}


def _empty(dev, dtype=torch.float32):
    key = (dev, dtype)
    if key not in _EMPTY:
        _EMPTY[key] = torch.empty(0, dtype=dtype, device=dev)
    return _EMPTY[key]


def want_segments(n_instances, n_tiles):
    if SEGMENTS == "auto":
        return 0 < n_tiles < SEGMENT_MAX_TILES and n_instances >= SEGMENT_MIN_MEAN_LIST * n_tiles
    return bool(SEGMENTS) and n_tiles > 0


_segment_choice = {}   # frame shape (the capacity hint's key) -> "auto"'s decision, fixed by the first exact count


def segments_for(hint_key, n_instances, exact_count, n_tiles):
    """want_segments, decided once per frame shape: a speculative frame only knows a capacity (1.25 S + 4096), so
    near the threshold the first frame and the later frames of one scene would otherwise pick different backward
    kernels (gradients differing in the last bits from frame to frame)."""
    if SEGMENTS != "auto":
        return want_segments(n_instances, n_tiles)
    if hint_key in _segment_choice:
        return _segment_choice[hint_key]
    on = want_segments(n_instances, n_tiles)
    if exact_count:
        _segment_choice[hint_key] = on
    return on


def render_forward(packed, rgb, ranges, sorted_g, keys, background_rgb, height, width, tile_rows, sort_prefix,
                   image_rows=None, segments=None, cut=None):
    """cut: the frame's depth-cut record (preprocess_forward(depth_cut=True).cut) -> gs_render_tiles_cut.
    -> image, splat counts, final weights, tile costs, segment state.  image_rows > height: the image buffer
    gets that many rows (the multi-GPU gather wants equal-sized bands); the kernels only see the first `height`.
    tile costs: int32[n_tiles], how long each tile took (prefix mode; empty otherwise) -- the launch-order
    hint render_backward hands back to the library.  segment state: the workspace for the depth-segmented
    backward (empty when segments are off; `segments` None = the module's policy), for render_backward."""
    dev = packed.device
    ntx = (width + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
    nty = (height + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
    row0, row1 = tile_rows if tile_rows is not None else (0, nty)
    # one allocation: image [H,W,3] | final weight [H,W] | splat count [H,W] (int32 view).  Rows outside
    # [row0, row1) are not written by the kernel: zero-fill only when sharded
    alloc = torch.empty if tile_rows is None else torch.zeros
    P = height * width
    PI = (image_rows if image_rows is not None else height) * width
    buf = alloc(3 * PI + 2 * P, dtype=torch.float32, device=dev)
    image = buf[:3 * PI].view(-1, width, 3)
    fw = buf[3 * PI:3 * PI + P].view(height, width)
    nsp = buf[3 * PI + P:].view(torch.int32).view(height, width)
    stream = _stream()
    if segments is None:
        segments = want_segments(sorted_g.shape[0], (row1 - row0) * ntx)
    seg = _empty(dev)
    if segments:
        seg = torch.empty(_hip.lib().gs_render_segment_workspace_bytes(width, height, row0, row1) // 4,
                          dtype=torch.float32, device=dev)
    seg_p = _p(seg) if segments else None
    if cut is not None:
        # depth-cut lists (preprocess_forward(depth_cut=True)): kept prefixes first, flagged tiles repaired from
        # their complete lists in the overflow buffers -- one call, the host never looks at the flags
        assert not segments, "depth segments and the depth cut are not combined"
        scratch = torch.empty(2 * ntx * nty, dtype=torch.int32, device=dev)
        flags, cost = scratch[:ntx * nty], scratch[ntx * nty:]
        # (never empty: a frame without a visible Gaussian still hands the backward a non-null overflow list)
        okeys = torch.empty(max(cut.S_full, 1), dtype=torch.int64, device=dev)
        osorted = torch.empty(max(cut.S_full, 1), dtype=torch.int32, device=dev)
        _hip.call("gs_render_tiles_cut", _p(packed), _p(rgb), _p(ranges), _p(sorted_g), ctypes.c_int64(sorted_g.shape[0]),
                  _p(cut.full_ranges), _p(cut.bin_rec), cut.N, _cf(cut.mh_dist), _p(cut.tile_counts), _p(cut.cut_ws),
                  _p(okeys), _p(osorted), ctypes.c_int64(cut.S_full), _p(background_rgb), width, height, row0, row1,
                  _p(flags), _p(nsp), _p(fw), _p(image), _p(cost), None, stream)
        cut.flags, cut.overflow_sorted = flags, osorted
        global last_tile_flags
        last_tile_flags = flags
        if len(_flag_log) < 512:
            _flag_log.append(flags)
        return image, nsp, fw, cost, seg
    if sort_prefix and sorted_g.shape[0] > sort_prefix:
        # provisional render from the ordered prefixes; tiles that ran out of prefix are flagged,
        # sorted in full and rendered again -- one call, the host never looks at the flags
        scratch = torch.empty(2 * ntx * nty, dtype=torch.int32, device=dev)
        flags, cost = scratch[:ntx * nty], scratch[ntx * nty:]
        _hip.call("gs_render_tiles_prefix", _p(packed), _p(rgb), _p(ranges), _p(sorted_g), _p(keys),
                  ctypes.c_int64(sorted_g.shape[0]), _p(background_rgb), width, height, row0, row1, _p(flags),
                  _p(nsp), _p(fw), _p(image), _p(cost), seg_p, stream)
        last_tile_flags = flags
        if len(_flag_log) < 512:
            _flag_log.append(flags)
    else:
        cost = _empty(dev, torch.int32)
        _hip.call("gs_render_tiles_packed", _p(packed), _p(rgb), None, _p(ranges), _p(sorted_g), _p(background_rgb),
                  width, height, 1, row0, row1, _p(nsp), _p(fw), _p(image), _hip.GS_F32, seg_p, stream)
    return image, nsp, fw, cost, seg


def render_backward(packed, rgb, ranges, sorted_g, background_rgb, nsp, fw, grad_image, height, width, tile_rows,
                    V, tile_cost=None, backward_mode=None, seg_state=None, cut=None):
    """-> the slab [V, 9] of accumulated render gradients (rgb 3 | opacity 1 | uv 2 | conic 3).
    tile_cost: render_forward's fourth output (the tiles are then started longest-first).
    seg_state: render_forward's fifth output; non-empty -> one workgroup per (tile, depth segment).
    backward_mode: _hip.GS_BACKWARD_COMPAT / _EXACT, per call (ABI 5); None = the process default at the call"""
    nty = (height + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
    row0, row1 = tile_rows if tile_rows is not None else (0, nty)
    # (cleared by the call itself, in the launch that also orders the tiles: zero_slab_rows)
    slab = torch.empty(max(V, 1), SLAB_WIDTH, dtype=torch.float32, device=packed.device)
    cost = order = None
    seg_on = seg_state is not None and seg_state.numel() > 0
    # longest-first only pays where a workgroup lives long enough for the kernel's tail to matter: lists of a
    # few hundred entries per tile (workload B, 52 per tile: the order kernel's 6 us are not won back)
    if (not seg_on and tile_cost is not None and tile_cost.numel() > 0
            and sorted_g.shape[0] >= LPT_MIN_MEAN_LIST * tile_cost.numel()):
        cost = tile_cost
        order = torch.empty(tile_cost.numel() + 8, dtype=torch.int32, device=packed.device)
    # one prologue launch (clear the slab + order the tiles), then the render kernel alone in its entry
    _hip.call("gs_render_backward_prologue", _p(slab), ctypes.c_int64(slab.shape[0]), _p(cost) if cost is not None else None,
              _p(order) if order is not None else None, width, height, row0, row1, _stream())
    _hip.call("gs_render_tiles_backward_slab", _p(packed), _p(rgb), _p(ranges), _p(sorted_g), _p(background_rgb),
              _p(nsp), _p(fw), _p(grad_image), width, height, row0, row1, _p(slab), ctypes.c_int64(0),
              None, _p(order) if order is not None else None,
              _p(seg_state) if seg_on else None,
              _p(cut.flags) if cut is not None else None, _p(cut.full_ranges) if cut is not None else None,
              _p(cut.overflow_sorted) if cut is not None else None,
              _hip.GS_BACKWARD_DEFAULT if backward_mode is None else int(backward_mode), _stream())
    return slab[:V]


def _as_slab(g_uv, g_conic, g_opa, g_rgb, V, dev):
    """the four render gradients as one [V, 9] slab: the slab they are views of when they come
    straight from _Render.backward, a packed copy otherwise (outputs nobody consumed count as 0)"""
    parts = ((g_rgb, SLAB_RGB), (g_opa, SLAB_OPACITY), (g_uv, SLAB_UV), (g_conic, SLAB_CONIC))
    base = g_rgb._base if g_rgb is not None else None
    if (base is not None and base.dim() == 2 and base.shape[1] == SLAB_WIDTH and base.is_contiguous()
            and all(g is not None and g._base is base and g.shape[0] == V and g.stride() == (SLAB_WIDTH, 1)
                    and g.storage_offset() == base.storage_offset() + sl.start for g, sl in parts)):
        return base
    slab = torch.zeros(max(V, 1), SLAB_WIDTH, dtype=torch.float32, device=dev)
    for g, sl in parts:
        if g is not None:
            slab[:V, sl] = g
    return slab


# ---------------------------------------------------------------------------------------------------
# autograd nodes
# ---------------------------------------------------------------------------------------------------
class _Preprocess(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, width, height, near_thresh,
                far_thresh, cull_mask_padding, mh_dist, tile_rows, sort_prefix=0, background_rgb=None, cut_box=None):
        # cut_box: a list; when the frame takes the depth cut its record (what _Render's two passes need) is left in it
        ntx = (width + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
        nty = (height + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
        row0, row1 = tile_rows if tile_rows is not None else (0, nty)
        cut = (cut_box is not None and background_rgb is not None and sort_prefix != 0 and
               want_depth_cut((xyz.device.index, xyz.shape[0], ntx * nty, row0, row1), xyz.shape[0], ntx, row0, row1,
                              tile_rows is None))
        f = preprocess_forward(xyz, quaternion, scale, opacity, rgb, sh, camera_T_world, K, width, height,
                               near_thresh, far_thresh, cull_mask_padding, mh_dist, tile_rows, sort_prefix,
                               defer=(not cut) and EARLY_RENDER and background_rgb is not None and sort_prefix != 0,
                               depth_cut=cut)
        if cut:
            cut_box.append(f.cut)
            pre = render_forward(f.packed, f.rgb_render, f.ranges, f.sorted_g, f.keys, background_rgb, height, width,
                                 tile_rows, sort_prefix, segments=False, cut=f.cut)
        elif f.deferred:
            # the render (the _Render node's forward) is enqueued on the speculative tile lists before
            # the host waits for the frame's counts, so a short frame leaves no bubble on the GPU;
            # a too small capacity repeats it
            def render(exact_count):
                seg = segments_for(f.hint_key, f.sorted_buf.shape[0] if not exact_count else f.S, exact_count,
                                   (f.row1 - f.row0) * f.ntx)
                return render_forward(f.packed, f.rgb_render, f.ranges, f.sorted_buf, f.keys_buf, background_rgb,
                                      height, width, tile_rows, sort_prefix, segments=seg)

            # (only gs_render_tiles_prefix takes the capacity and skips segments beyond it)
            pre = render(False) if (f.speculative and f.capacity > sort_prefix) else None
            if preprocess_finish(f) or pre is None:
                pre = render(True)
        else:
            pre = ()
        V = f.V
        ctx.save_for_backward(xyz, quaternion, scale, camera_T_world, K)
        ctx.set_materialize_grads(False)   # no zero tensors for the auxiliary outputs in backward
        ctx.f = SimpleNamespace(N=f.N, V=V, n_sh=f.n_sh, center=f.center, rank=f.rank, opacity_act=f.opacity_act)
        uv_v, conic_v, opa_v, rgb_v = f.uv[:V], f.conic[:V], f.opacity_act[:V], f.rgb_render[:V]
        aux = (f.packed, f.xyz_cam[:V], f.culling_mask, f.ranges, f.sorted_g, f.vis_idx[:V], f.keys) + tuple(pre)
        ctx.mark_non_differentiable(*aux)
        return (uv_v, conic_v, opa_v, rgb_v) + aux

    @staticmethod
    def backward(ctx, g_uv, g_conic, g_opa, g_rgb, *unused):
        xyz, quaternion, scale, camera_T_world, K = ctx.saved_tensors
        slab = _as_slab(g_uv, g_conic, g_opa, g_rgb, ctx.f.V, xyz.device)
        grads = preprocess_backward(xyz, quaternion, scale, camera_T_world, K, ctx.f, slab)
        return grads + (None,) * 12


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, conic, opacity, rgb, packed, ranges, sorted_g, background_rgb, height, width, tile_rows,
                slab_sync=None, keys=None, sort_prefix=0, rendered=None, cut=None):
        # rendered: (image, nsp, fw, cost, seg) when _Preprocess.forward already enqueued this node's kernels
        image, nsp, fw, cost, seg = rendered if rendered else render_forward(
            packed, rgb, ranges, sorted_g, keys, background_rgb, height, width, tile_rows, sort_prefix)
        ctx.cut = cut   # depth-cut frame: flags, complete ranges and overflow lists for the backward
        ctx.save_for_backward(packed, rgb, ranges, sorted_g, background_rgb, nsp, fw, cost, seg)
        ctx.set_materialize_grads(False)
        ctx.dims = (height, width, tile_rows, uv.shape[0])
        ctx.slab_sync = slab_sync
        # the gradient mode of THIS frame is the default at its forward: a later set_backward_mode() cannot
        # race the backward the autograd engine runs on its own thread
        ctx.backward_mode = _hip.get_backward_mode()
        return image

    @staticmethod
    def backward(ctx, grad_image):
        packed, rgb, ranges, sorted_g, background_rgb, nsp, fw, cost, seg = ctx.saved_tensors
        height, width, tile_rows, V = ctx.dims
        if grad_image is None:
            return (None,) * 16
        slab = render_backward(packed, rgb, ranges, sorted_g, background_rgb, nsp, fw, grad_image.contiguous(),
                               height, width, tile_rows, V, cost, ctx.backward_mode, seg, cut=ctx.cut)
        if ctx.slab_sync is not None:
            ctx.slab_sync(slab.view(-1))   # multi-GPU: sum the partial gradients of all bands in place
        # the four gradients are views of the one slab; _Preprocess.backward recognises that
        return (slab[:, SLAB_UV], slab[:, SLAB_CONIC], slab[:, SLAB_OPACITY], slab[:, SLAB_RGB]) + (None,) * 12


class _GatherRows(torch.autograd.Function):
    """rows `index` (unique, int64) of a dense parameter tensor; the gradient is an index_copy into zeros
    (torch's own indexing backward sorts the indices to accumulate duplicates: there are none)"""

    @staticmethod
    def forward(ctx, dense, index):
        ctx.save_for_backward(index)
        ctx.shape = dense.shape
        return dense.index_select(0, index)

    @staticmethod
    def backward(ctx, grad):
        (index,) = ctx.saved_tensors
        return torch.zeros(ctx.shape, dtype=grad.dtype, device=grad.device).index_copy_(0, index, grad), None


class _RenderSH(torch.autograd.Function):
    """Per-pixel SH colour (rasterize.py:100-110 with use_sh_precompute=False; render.cu:283-333,
    render_backward.cu:422-488): the N_SH in {4, 9, 16} render kernels on the fused frame's packed records and
    tile lists.  coeffs: [V, 3, N_SH] colour coefficients of the visible Gaussians, rays: [H, W, 3]."""

    @staticmethod
    def forward(ctx, uv, conic, opacity, coeffs, packed, ranges, sorted_g, rays, background_rgb, height, width,
                tile_rows):
        dev = packed.device
        nty = (height + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX
        row0, row1 = tile_rows if tile_rows is not None else (0, nty)
        alloc = torch.empty if tile_rows is None else torch.zeros
        P = height * width
        buf = alloc(5 * P, dtype=torch.float32, device=dev)
        image, fw = buf[:3 * P].view(height, width, 3), buf[3 * P:4 * P].view(height, width)
        nsp = buf[4 * P:].view(torch.int32).view(height, width)
        _hip.call("gs_render_tiles_packed", _p(packed), _p(coeffs), _p(rays), _p(ranges), _p(sorted_g),
                  _p(background_rgb), width, height, int(coeffs.shape[2]), row0, row1, _p(nsp), _p(fw), _p(image),
                  _hip.GS_F32, None, _stream())
        ctx.save_for_backward(packed, coeffs, ranges, sorted_g, rays, background_rgb, nsp, fw)
        ctx.set_materialize_grads(False)
        ctx.dims = (height, width, row0, row1, uv.shape[0])
        ctx.backward_mode = _hip.get_backward_mode()   # the frame's mode is the default at its forward
        return image

    @staticmethod
    def backward(ctx, grad_image):
        packed, coeffs, ranges, sorted_g, rays, background_rgb, nsp, fw = ctx.saved_tensors
        height, width, row0, row1, V = ctx.dims
        if grad_image is None:
            return (None,) * 12
        n_sh = int(coeffs.shape[2])
        (gc, go, gu, gk) = _Arena(torch.float32, packed.device, (3 * n_sh * V, V, 2 * V, 3 * V), zero=True).blocks()
        g_rgb, g_opa, g_uv, g_conic = gc.view(V, 3, n_sh), go.view(V, 1), gu.view(V, 2), gk.view(V, 3)
        _hip.call("gs_render_tiles_backward_packed", _p(packed), _p(coeffs), _p(rays), _p(ranges), _p(sorted_g),
                  _p(background_rgb), _p(nsp), _p(fw), _p(grad_image.contiguous()), width, height, n_sh, row0, row1,
                  _p(g_rgb), _p(g_opa), _p(g_uv), _p(g_conic), _hip.GS_F32, int(ctx.backward_mode), _stream())
        return (g_uv, g_conic, g_opa, g_rgb) + (None,) * 8


def _rasterize_per_pixel_sh(g, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding, mh_dist,
                            background_rgb, tile_rows):
    """use_sh_precompute=False on the fused frame: its per-Gaussian stage, binning and full per-tile sort, the
    colour coefficients of the visible Gaussians gathered once, and the per-pixel-SH render kernels.  Same
    (image, culling_mask, uv) contract and the same kernels as the reference-shaped path of this colour mode
    (which spends ~4 ms per frame of workload B in its host glue); against it the frame differs as the
    precompute mode's does: by the fused per-Gaussian stage's explicit fp32 world->camera transform."""
    out = _Preprocess.apply(
        g.xyz.contiguous(), g.quaternion.contiguous(), g.scale.contiguous(), g.opacity.contiguous(),
        g.rgb.contiguous(), None, camera_T_world.contiguous(), camera.K.contiguous(), int(camera.width),
        int(camera.height), near_thresh, far_thresh, cull_mask_padding, mh_dist, tile_rows, 0, None)
    uv, conic, opacity, _rgb_unused, packed, _xyz_cam, culling_mask, ranges, sorted_g, vis_idx = out[:10]
    index = vis_idx.long()
    # coefficient 0 is the rgb parameter (rasterize.py:103: cat of rgb and sh)
    coeffs = torch.cat((_GatherRows.apply(g.rgb, index).unsqueeze(2), _GatherRows.apply(g.sh, index)), dim=2)
    rays = compute_rays_in_world_frame(camera, camera_T_world)
    image = _RenderSH.apply(uv, conic, opacity, coeffs.contiguous(), packed, ranges, sorted_g, rays,
                            background_rgb.contiguous(), int(camera.height), int(camera.width), tile_rows)
    return image, culling_mask, uv


def render_depth(gaussians, alpha_threshold, camera_T_world, camera, near_thresh, cull_mask_padding, mh_dist):
    """splat_py.depth.render_depth (depth.py:17-88) on the fused frame's stages: [H, W, 1], the distance of the
    first Gaussian at which a pixel's accumulated alpha passes alpha_threshold, -1 where it never does.  No
    gradient.  The reference's depth path has no far threshold (depth.py:33-41): none is applied."""
    g = gaussians
    if not g.xyz.is_cuda or g.xyz.dtype != torch.float32:
        from .splat_py.depth import render_depth as reference_shaped
        return reference_shaped(g, alpha_threshold, camera_T_world, camera, near_thresh, cull_mask_padding, mh_dist)
    with torch.no_grad():
        W, H = int(camera.width), int(camera.height)
        f = preprocess_forward(g.xyz.contiguous(), g.quaternion.contiguous(), g.scale.contiguous(),
                               g.opacity.contiguous(), g.rgb.contiguous(), None, camera_T_world.contiguous(),
                               camera.K.contiguous(), W, H, near_thresh, 3.0e38, cull_mask_padding, mh_dist, None, 0)
        depth = torch.full((H, W, 1), -1.0, dtype=torch.float32, device=g.xyz.device)
        _hip.call("gs_render_depth", _p(f.packed), _p(f.xyz_cam), _p(f.ranges), _p(f.sorted_g), W, H,
                  _cf(alpha_threshold), _p(depth), _stream())
        return depth


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def validate(gaussians, camera_T_world, camera, background_rgb):
    """The checks of the reference's extension (src/checks.cuh:5-14, render.cu:204-246) for the tensors the
    fused path hands to the C ABI as raw pointers: device, dtype, shape.  (Contiguity is established by
    the caller with .contiguous().)  Raises RuntimeError like the reference's TORCH_CHECKs."""
    g = gaussians
    dev = g.xyz.device
    N = g.xyz.shape[0]
    named = (("xyz", g.xyz, (N, 3)), ("quaternion", g.quaternion, (N, 4)), ("scale", g.scale, (N, 3)),
             ("opacity", g.opacity, (N, 1)), ("rgb", g.rgb, (N, 3)), ("camera_T_world", camera_T_world, (4, 4)),
             ("K", camera.K, (3, 3)), ("background_rgb", background_rgb, (3,)))
    for name, t, shape in named:
        _require(t.is_cuda and t.device == dev, f"{name} is not a CUDA tensor on {dev}")
        _require(t.dtype == torch.float32, f"{name} is not a float tensor")
        _require(tuple(t.shape) == shape, f"{name} must have shape {list(shape)}, got {list(t.shape)}")
    if g.sh is not None:
        t = g.sh
        _require(t.is_cuda and t.device == dev, f"sh is not a CUDA tensor on {dev}")
        _require(t.dtype == torch.float32, "sh is not a float tensor")
        _require(t.dim() == 3 and t.shape[0] == N and t.shape[1] == 3 and t.shape[2] in (3, 8, 15),
                 f"sh must have shape [{N}, 3, 3|8|15], got {list(t.shape)}")


def supported(gaussians, camera_T_world, camera, use_sh_precompute):
    if not gaussians.xyz.is_cuda or gaussians.xyz.dtype != torch.float32:
        return False
    if gaussians.sh is not None and not use_sh_precompute:
        return False   # per-pixel SH evaluation: not the two-node frame (rasterize() routes it, see there)
    return True


def rasterize(gaussians, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding, mh_dist,
              use_sh_precompute, background_rgb, tile_rows=None, return_aux=False, grad_sync=None, slab_sync=None,
              frame_hook=None):
    """tile_rows=(row0, row1) restricts binning and rendering to those tile rows; slab_sync(flat) is
    called on the flat [9 V] render-gradient slab in the backward (grad_sync is the generic
    per-tensor form used by the reference-shaped path): the hooks gaussian_splatting_amd.sharded
    uses; all default to the single-GPU behaviour.  frame_hook(dict) receives the frame's tile ranges."""
    hooks = return_aux or grad_sync or slab_sync or frame_hook
    if (gaussians.sh is not None and not use_sh_precompute and not hooks and gaussians.xyz.is_cuda
            and gaussians.xyz.dtype == torch.float32):
        # per-pixel SH evaluation (N_SH in {4, 9, 16} render kernels) on the fused frame's stages: single-GPU frames;
        # the multi-GPU hooks of this colour mode stay on the reference-shaped path
        validate(gaussians, camera_T_world, camera, background_rgb)
        return _rasterize_per_pixel_sh(gaussians, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding,
                                       mh_dist, background_rgb, tile_rows)
    if not supported(gaussians, camera_T_world, camera, use_sh_precompute):
        return _reference_shaped.rasterize(gaussians, camera_T_world, camera, near_thresh, far_thresh,
                                           cull_mask_padding, mh_dist, use_sh_precompute, background_rgb,
                                           tile_rows=tile_rows, grad_sync=grad_sync)
    g = gaussians
    validate(g, camera_T_world, camera, background_rgb)
    nat = native() if not (return_aux or grad_sync or slab_sync or frame_hook) else None
    if nat is not None:
        nat.set_modes(bool(SORT_PREFIX), bool(EARLY_RENDER))
        nat.set_segments(0 if SEGMENTS == "auto" else (1 if SEGMENTS else -1))
        nat.set_depth_cut(0 if DEPTH_CUT == "auto" else (1 if DEPTH_CUT else -1), int(DEPTH_CUT_MIN_MEAN_LIST))
        row0, row1 = tile_rows if tile_rows is not None else (0, -1)
        return nat.rasterize(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, camera_T_world, camera.K,
                             int(camera.width), int(camera.height), near_thresh, far_thresh, cull_mask_padding, mh_dist,
                             background_rgb, row0, row1)
    sh = g.sh.contiguous() if g.sh is not None else None
    sort_prefix = _hip.GS_SORT_PREFIX if (SORT_PREFIX and not return_aux) else 0
    cut_box = [] if not (return_aux or grad_sync or slab_sync or frame_hook) else None
    out = _Preprocess.apply(
        g.xyz.contiguous(), g.quaternion.contiguous(), g.scale.contiguous(), g.opacity.contiguous(),
        g.rgb.contiguous(), sh, camera_T_world.contiguous(), camera.K.contiguous(), int(camera.width),
        int(camera.height), near_thresh, far_thresh, cull_mask_padding, mh_dist, tile_rows, sort_prefix,
        background_rgb.contiguous(), cut_box)
    uv, conic, opacity, rgb, packed, xyz_cam, culling_mask, ranges, sorted_g, vis_idx, keys = out[:11]
    if frame_hook is not None:   # multi-GPU cost-balanced bands: the band's tile ranges
        frame_hook(dict(ranges=ranges, ntx=(int(camera.width) + TILE_EDGE_LENGTH_PX - 1) // TILE_EDGE_LENGTH_PX))
    image = _Render.apply(uv, conic, opacity, rgb, packed, ranges, sorted_g, background_rgb.contiguous(),
                          int(camera.height), int(camera.width), tile_rows, slab_sync, keys, sort_prefix,
                          tuple(out[11:]), cut_box[0] if cut_box else None)
    if return_aux:
        return image, culling_mask, uv, dict(conic=conic, opacity=opacity, rgb=rgb, packed=packed,
                                             xyz_camera_frame=xyz_cam, tile_ranges=ranges,
                                             sorted_gaussians=sorted_g, vis_idx=vis_idx)
    return image, culling_mask, uv
