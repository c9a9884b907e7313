"""Experiment: can an HBM-bound per-Gaussian kernel run concurrently with the LDS-atomic-bound binning kernels?
Launches gs_preprocess_forward (0.2 ms alone at D, proxy for a separate SH-colour pass) on one stream and
gs_tile_count + gs_tile_emit_sort (0.38 ms alone) on another, on inputs of a finished frame, and compares the
wall time of the pair with the two run back to back.  usage: python scripts/exp_overlap.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
N, W, H, deg = WORKLOADS["D"]
g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
d = DEFAULTS
f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H, d["near_thresh"],
                             d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None, _hip.GS_SORT_PREFIX)
S = f.S
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
cf = lambda x: ctypes.c_float(float(x))
lib = _hip.lib()
# a second set of outputs for the concurrent preprocess (same inputs)
n_ws = lib.gs_preprocess_workspace_ints(N)
ib = torch.empty(n_ws + 1 + 2 * N + (N + 3) // 4 + 16, dtype=torch.int32, device=dev)
fb = torch.empty(3 + 35 * N + 64, dtype=torch.float32, device=dev)
ws2, cnt2, rank2, vis2, mask2 = ib[:n_ws], ib[n_ws:n_ws + 1], ib[n_ws + 4:n_ws + 4 + N], ib[n_ws + 4 + N:n_ws + 4 + 2 * N], ib[n_ws + 4 + 2 * N:]
o = 4
def cut(n):
    global o
    t = fb[o:o + n]; o += (n + 3) & ~3
    return t
center2, uv2, xc2, co2, op2, rg2, pk2 = fb[:3], cut(2 * N), cut(3 * N), cut(3 * N), cut(N), cut(3 * N), cut(12 * N)
tws = torch.empty(lib.gs_tile_workspace_ints(f.T), dtype=torch.int32, device=dev)
ranges = torch.empty(f.T + 2, dtype=torch.int32, device=dev)
keys = torch.empty(S, dtype=torch.int64, device=dev)
out = torch.empty(S, dtype=torch.int32, device=dev)


def pre(stream):
    _hip.check(lib.gs_preprocess_forward(p(g.xyz), p(g.quaternion), p(g.scale), p(g.opacity), p(g.rgb), p(g.sh), 16, p(T),
               p(cam.K), N, W, H, cf(d["near_thresh"]), cf(d["far_thresh"]), cf(d["cull_mask_padding"]), cf(d["mh_dist"]), 0,
               f.nty, p(ws2), p(center2), p(cnt2), p(mask2), p(rank2), p(vis2), p(uv2), p(xc2), p(co2), p(op2), p(rg2), p(pk2),
               stream))


def binning(stream):
    _hip.check(lib.gs_tile_count(p(f.uv), p(f.conic), N, p(f.count), None, None, f.ntx, f.nty, cf(d["mh_dist"]), 0, f.nty,
                                 p(tws), p(ranges), None, stream))
    _hip.check(lib.gs_tile_emit_sort(p(f.uv), p(f.xyz_cam), p(f.conic), N, p(f.count), None, None, f.ntx, f.nty,
                                     cf(d["mh_dist"]), 0, f.nty, p(ranges), p(tws), p(keys), ctypes.c_int64(S), p(out),
                                     _hip.GS_SORT_PREFIX, stream))


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(sb.cuda_stream)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("preprocess alone          %.3f ms" % timeit(lambda: pre(pa)))
print("binning alone             %.3f ms" % timeit(lambda: binning(pa)))
print("back to back, one stream  %.3f ms" % timeit(lambda: (pre(pa), binning(pa))))


def both():
    pre(pa)
    binning(pb)
    sa.wait_stream(sb)   # join, so that consecutive pairs do not overlap each other


print("concurrent, two streams   %.3f ms" % timeit(both))
