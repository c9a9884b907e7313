"""Per-rank compute of a G-way tile-row split measured on ONE GPU: renders only the band of rank
`--rank` of `--world` (no collectives).  usage: python scripts/band_cost.py --world 8 --rank 3"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip, fused
from gaussian_splatting_amd.sharded import ShardedRasterizer, band_of, owned_slice
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--workload", default="D")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--grad-mode", default="replicated", choices=["replicated", "owner"],
                help="owner: the sparse-exchange path with the all_to_all replaced by a local fill (compute only)")
ap.add_argument("--native", type=int, default=1, help="0: the Python orchestration of the sharded frame")
ap.add_argument("--compact", type=int, default=1, help="0: replicated per-Gaussian stage (native path)")
ap.add_argument("--fused", type=int, default=1, help="0: the three-call band pipeline of rounds 3-5 (A/B)")
a = ap.parse_args()
if fused.native() is not None:
    fused.native().set_band_fused(bool(a.fused))
from gaussian_splatting_amd import sharded as _sh
_sh.NATIVE = bool(a.native)
_sh.BAND_COMPACT = bool(a.compact)
N, W, H, deg = WORKLOADS[a.workload]
g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
params = [p for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh) if p is not None]
for p in params:
    p.requires_grad_(True)
gi = make_grad_image(W, H, seed=1, device="cuda")
bg = torch.zeros(3, device="cuda")
rows = band_of((H + 15) // 16, a.world, a.rank)


moved = {}
if a.grad_mode == "owner":
    def fake_a2a(recv, send, recv_splits, send_splits):
        moved["send_rows"], moved["recv_rows"] = sum(send_splits), sum(recv_splits)
        moved["send_rows_remote"] = sum(send_splits) - send_splits[a.rank]
        recv.zero_()

    for p in params:
        p.requires_grad_(False)
    owned = owned_slice(g, a.world, a.rank)
    rast = ShardedRasterizer(H, a.world, a.rank, grad_mode="owner", all_to_all=fake_a2a)


def step():
    if a.grad_mode == "owner":
        for p in (owned.xyz, owned.rgb, owned.opacity, owned.scale, owned.quaternion, owned.sh):
            if p is not None:
                p.grad = None
        img, _, _ = rast.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, owned=owned, **DEFAULTS)
        img.backward(gi)
        return
    for p in params:
        p.grad = None
    img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows, **DEFAULTS)
    img.backward(gi)


for _ in range(3):
    step()
torch.cuda.synchronize()
_hip.enable_timing(True)
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.steps * 1e3
t = _hip.collect_timing()
print(f"world {a.world} rank {a.rank} rows {rows} native {a.native} compact {a.compact} fused {a.fused}: {ms:.3f} ms/step",
      {k: round(sum(v) / len(v), 4) for k, v in sorted(t.items())}, moved)
