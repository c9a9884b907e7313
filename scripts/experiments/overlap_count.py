"""Does the VALU-bound tile count overlap with an HBM-bound per-Gaussian kernel when the two run on separate
streams?  (What a split of k_preprocess into a geometry kernel and an SH-colour kernel would buy: the count only
needs the geometry.)  Stand-in for the HBM-bound kernel: gs_preprocess_backward on the same frame's buffers.
usage: python scripts/experiments/overlap_count.py [--workload D]"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_splatting_amd import _hip, fused
from gaussian_splatting_amd.fused import _cf, _p
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="D")
ap.add_argument("--reps", type=int, default=50)
a = ap.parse_args()
N, W, H, deg = WORKLOADS[a.workload]
g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H,
                             DEFAULTS["near_thresh"], DEFAULTS["far_thresh"], DEFAULTS["cull_mask_padding"],
                             DEFAULTS["mh_dist"], None, 0)
ntx, nty = (W + 15) // 16, (H + 15) // 16
slab = torch.randn(f.V, 9, device="cuda")
n_extra = 3 * (f.n_sh - 1)
out = [torch.empty(N, k, device="cuda") for k in (3, 4, 3, 1, 3)] + [torch.empty(N, 3, f.n_sh - 1, device="cuda")]


def count(stream):
    _hip.call("gs_tile_count", _p(f.uv), _p(f.conic), N, _p(f.count), None, None, ntx, nty, _cf(DEFAULTS["mh_dist"]),
              0, nty, _p(f.tile_counts), _p(f.ranges_buf), None, stream)


def hbm(stream):
    _hip.call("gs_preprocess_backward", _p(g.xyz), _p(g.quaternion), _p(g.scale), f.n_sh, _p(T), _p(cam.K), _p(f.center),
              _p(f.rank), _p(f.opacity_act), _p(slab), 0, N, *[_p(t) for t in out], stream)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa, pb = ctypes.c_void_p(sa.cuda_stream), ctypes.c_void_p(sb.cuda_stream)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps * 1e3


def both_one_stream():
    count(pa)
    hbm(pa)


def both_two_streams():
    count(pa)
    hbm(pb)


print(f"workload {a.workload}: count alone {timed(lambda: count(pa)):.4f} ms, HBM-bound kernel alone {timed(lambda: hbm(pa)):.4f} ms, "
      f"one stream {timed(both_one_stream):.4f} ms, two streams {timed(both_two_streams):.4f} ms")
