"""Randomised stress of the fused path: prefix-sort mode (default; every other frame with the depth-segmented
backward forced on, every other with it off) against full-sort mode (return_aux=True, unsegmented) -- images
must be bit-identical, gradients equal up to summation order -- over
random scene sizes, image shapes, SH degrees, opacities (faint scenes flag and repair tiles) and list
lengths (all sort classes).  usage: python scripts/stress_fused.py [--n 200] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import fused
from gaussian_splatting_amd.synthetic import make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = random.Random(a.seed)
NAMES = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")
stats = dict(frames=0, flagged_frames=0, max_list=0, classes=set(), segmented_frames=0)
for it in range(a.n):
    N = int(10 ** rng.uniform(2.3, 4.8))
    W, H = rng.randint(17, 420), rng.randint(17, 300)
    deg = rng.randint(0, 3)
    seed = rng.randint(0, 10 ** 6)
    shift = rng.choice([0.0, 0.0, -3.0, -5.0, 2.0])
    near, far, pad = rng.choice([(0.3, 500.0, 100), (2.0, 25.0, 20), (5.0, 12.0, 0)])
    bgv = rng.choice([0.0, 0.5])
    out = {}
    # the default path (native orchestration, prefix sort) alternately with and without depth segments; the
    # reference run (Python orchestration, full sort) never segments
    seg_default = bool(it % 2)
    for mode in ("prefix", "full"):
        fused.SEGMENTS = seg_default if mode == "prefix" else False
        g, cam, T = make_scene(N, W, H, deg, seed=seed, device="cuda")
        with torch.no_grad():
            g.opacity.add_(shift)
        for k in NAMES:
            if getattr(g, k) is not None:
                getattr(g, k).requires_grad_(True)
        bg = torch.full((3,), bgv, device="cuda")
        if mode == "prefix":
            fused.last_flags(clear=True)
            img, mask, uv = fused.rasterize(g, T, cam, near, far, pad, 3.0, True, bg)
            flags = fused.last_flags()
        else:
            img, mask, uv, aux = fused.rasterize(g, T, cam, near, far, pad, 3.0, True, bg, return_aux=True)
            counts = aux["tile_ranges"][1:] - aux["tile_ranges"][:-1]
        img.backward(make_grad_image(W, H, seed=seed % 97, device="cuda"))
        out[mode] = (img.detach(), {k: getattr(g, k).grad for k in NAMES if getattr(g, k) is not None})
    assert torch.equal(out["prefix"][0], out["full"][0]), (it, N, W, H, deg, seed, shift)
    for k, ref in out["full"][1].items():
        got = out["prefix"][1][k]
        scale = ref.abs().max().clamp(min=1e-30)
        err = float((got - ref).abs().max() / scale)
        assert err < 2e-5 and torch.isfinite(got).all(), (it, k, err, N, W, H, deg, seed, shift)
    mx = int(counts.max()) if counts.numel() else 0
    stats["frames"] += 1
    stats["segmented_frames"] += int(seg_default)
    stats["max_list"] = max(stats["max_list"], mx)
    stats["classes"] |= {c for c, lo, hi in (("<=64", 0, 64), ("<=256", 64, 256), ("<=1024", 256, 1024),
                                              ("<=4096", 1024, 4096), ("<=8192", 4096, 8192),
                                              (">8192", 8192, 10 ** 9)) if bool(((counts > lo) & (counts <= hi)).any())}
    if flags is not None and int(flags.sum()) > 0:
        stats["flagged_frames"] += 1
fused.SEGMENTS = "auto"
print("ok", {k: (sorted(v) if isinstance(v, set) else v) for k, v in stats.items()})
