"""cProfile of the host side of band-mode frames.  usage: python scripts/host_profile.py [--world 8]"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import fused, sharded
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--workload", default="D")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--rccl", action="store_true", help="world size 1 over real RCCL collectives instead of the local fills")
a = ap.parse_args()
N, W, H, deg = WORKLOADS[a.workload]
g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
gi = make_grad_image(W, H, seed=1, device="cuda")
bg = torch.zeros(3, device="cuda")
if a.rccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    owned = sharded.owned_slice(g, 1, 0)
    rast = sharded.ShardedRasterizer(H, 1, 0, grad_mode="owner")
    holders = owned

    def fwd():
        return rast.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, owned=owned, **DEFAULTS)
elif a.world > 1:
    rank = a.world // 2
    owned = sharded.owned_slice(g, a.world, rank)
    rast = sharded.ShardedRasterizer(H, a.world, rank, grad_mode="owner", all_to_all=lambda r, s, rs, ss: r.zero_())
    holders = owned

    def fwd():
        return rast.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, owned=owned, **DEFAULTS)
else:
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        getattr(g, k).requires_grad_(True)
    holders = g

    def fwd():
        return fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)


def step():
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        getattr(holders, k).grad = None
    img, _, _ = fwd()
    img.backward(gi)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("SORT", "tottime"))
print(f"per step: {st.total_tt / a.steps * 1e3:.3f} ms host")
st.print_stats(28)
