"""The two secondary render paths of SURVEY 8(f2)/(f3) at workload B, by themselves (for rocprofv3):
depth renderer (forward) and per-pixel-SH colour mode (forward + backward) through the reference-shaped API.
usage: python scripts/secondary_paths.py [--steps 10] [--what depth,sh]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip
from gaussian_splatting_amd.splat_py.depth import render_depth
from gaussian_splatting_amd.splat_py.rasterize import rasterize
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--workload", default="B")
ap.add_argument("--what", default="depth,sh")
a = ap.parse_args()
N, W, H, deg = WORKLOADS[a.workload]
g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
gi = make_grad_image(W, H, seed=1, device="cuda")
bg = torch.zeros(3, device="cuda")
params = [p for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh) if p is not None]


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _hip.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    t = _hip.collect_timing()
    _hip.enable_timing(False)
    return round(ms, 4), {k: round(sum(v) / a.steps, 4) for k, v in sorted(t.items())}


if "depth" in a.what:
    print("depth forward", timed(lambda: render_depth(g, 0.5, T, cam, DEFAULTS["near_thresh"],
                                                      DEFAULTS["cull_mask_padding"], DEFAULTS["mh_dist"])))
    from gaussian_splatting_amd import fused as _fused
    print("depth forward, fused frame", timed(lambda: _fused.render_depth(g, 0.5, T, cam, DEFAULTS["near_thresh"],
                                                                          DEFAULTS["cull_mask_padding"], DEFAULTS["mh_dist"])))
if "sh" in a.what:
    for p in params:
        p.requires_grad_(True)

    def per_pixel_sh():
        for p in params:
            p.grad = None
        image, _, _ = rasterize(g, T, cam, use_sh_precompute=False, background_rgb=bg, **DEFAULTS)
        image.backward(gi)

    print("per-pixel SH forward+backward", timed(per_pixel_sh))

    from gaussian_splatting_amd import fused

    def per_pixel_sh_fused():
        for p in params:
            p.grad = None
        image, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=False, background_rgb=bg, **DEFAULTS)
        image.backward(gi)

    print("per-pixel SH forward+backward, fused frame", timed(per_pixel_sh_fused))

    def precomputed():
        for p in params:
            p.grad = None
        image, _, _ = rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
        image.backward(gi)

    print("precomputed SH forward+backward (same API)", timed(precomputed))
