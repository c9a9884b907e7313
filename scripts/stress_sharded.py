"""Randomised stress of the multi-GPU owner mode with simulated ranks on one device (every rank's real
kernels, the all_to_all routed in-process): image bands must tile the single-GPU image bit-exactly and
the owned-slice gradients must equal the single-GPU gradients up to summation order.
usage: python scripts/stress_sharded.py [--n 30] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import fused
from gaussian_splatting_amd.sharded import ShardedRasterizer, owned_slice, owner_range
from gaussian_splatting_amd.synthetic import make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=30)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--workload", default="", help="run ONE frame of this bench workload (e.g. D) instead of random scenes")
ap.add_argument("--world", type=int, default=8)
a = ap.parse_args()
rng = random.Random(a.seed)
NAMES = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")
rows_moved = 0
if a.workload:
    a.n = 1
for it in range(a.n):
    G = rng.choice([2, 3, 4, 5, 8])
    N = int(10 ** rng.uniform(2.5, 4.6))
    W, H = rng.randint(17, 420), rng.randint(17, 300)
    deg = rng.randint(0, 3)
    seed = rng.randint(0, 10 ** 6)
    shift = rng.choice([0.0, -3.0, 2.0])
    args = rng.choice([(0.3, 500.0, 100, 3.0), (2.0, 25.0, 20, 3.0), (5.0, 12.0, 0, 3.0)])
    if a.workload:
        from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS
        N, W, H, deg = WORKLOADS[a.workload]
        G, seed, shift = a.world, 0, 0.0
        args = (DEFAULTS["near_thresh"], DEFAULTS["far_thresh"], DEFAULTS["cull_mask_padding"], DEFAULTS["mh_dist"])
    bg = torch.full((3,), rng.choice([0.0, 0.5]), device="cuda")
    gi = make_grad_image(W, H, seed=seed % 89, device="cuda")

    def scene():
        g, cam, T = make_scene(N, W, H, deg, seed=seed, device="cuda")
        with torch.no_grad():
            g.opacity.add_(shift)
        return g, cam, T

    g, cam, T = scene()
    for k in NAMES:
        if getattr(g, k) is not None:
            getattr(g, k).requires_grad_(True)
    ref_img, ref_mask, _ = fused.rasterize(g, T, cam, *args, True, bg)
    ref_img.backward(gi)
    ref_img = ref_img.detach()
    ref = {k: getattr(g, k).grad for k in NAMES if getattr(g, k) is not None}
    sent = {}

    def run(rank, a2a):
        g2, cam2, T2 = scene()
        owned = owned_slice(g2, G, rank)
        rast = ShardedRasterizer(H, G, rank, grad_mode="owner", all_to_all=a2a)
        img, mask, uv = rast.rasterize(g2, T2, cam2, *args, True, bg, owned=owned)
        img.backward(gi)
        return img.detach(), mask, owned, rast

    def recorder(rank):
        def a2a(recv, send, recv_splits, send_splits):
            sent[rank] = (send.clone(), list(send_splits))
            recv.zero_()
        return a2a

    def router(rank):
        def a2a(recv, send, recv_splits, send_splits):
            off = 0
            for s in range(G):
                buf, splits = sent[s]
                lo = sum(splits[:rank])
                assert splits[rank] == recv_splits[s]
                recv[off:off + recv_splits[s]] = buf[lo:lo + splits[rank]]
                off += recv_splits[s]
        return a2a

    for r in range(G):
        run(r, recorder(r))
    total = torch.zeros_like(ref_img)
    for r in range(G):
        img, mask, owned, rast = run(r, router(r))
        total += img
        assert torch.equal(mask, ref_mask)
        i0, i1 = owner_range(N, G, r)
        for k, full in ref.items():
            got = getattr(owned, k).grad
            if got.numel():
                err = float((got - full[i0:i1]).abs().max() / full.abs().max().clamp(min=1e-30))
                assert err < 2e-5 and torch.isfinite(got).all(), (it, G, r, k, err, N, W, H, deg, seed)
        rows_moved += sum(rast.last_plan.send_splits)
    assert torch.equal(total, ref_img), (it, G, N, W, H, deg, seed)
print("ok", a.n, "frames, rows exchanged:", rows_moved)
