"""Where does one training iteration go?  Host time to enqueue and GPU time (sync after each stage) of
rasterize / loss+backward / Adam / statistics at a given Gaussian count.  usage: python scripts/iter_profile.py [N]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gaussian_splatting_amd import fused  # noqa: E402
from gaussian_splatting_amd.densify import DensifyConfig, DensityController  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, make_scene  # noqa: E402
from gaussian_splatting_amd.train_ops import Adam, ssim_l1_loss  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 700_000
dev = torch.device("cuda", 0)
W, H = 1297, 840
g, cam, T = make_scene(N, W, H, 3, seed=5, device=dev)
names = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")
for k in names:
    getattr(g, k).requires_grad_(True)
opt = Adam([{"params": getattr(g, k), "lr": 1e-3} for k in names])
ctrl = DensityController(g, opt, DensifyConfig())
poses = bench.camera_poses(24, 4321, dev, moving=True)
target = torch.rand(H, W, 3, device=dev)
stages = {}


def mark(name, t0, sync):
    t1 = time.perf_counter()
    if sync:
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    h, d = stages.setdefault(name, [0.0, 0.0])
    stages[name] = [h + (t1 - t0), d + (t2 - t0)]
    return time.perf_counter()


def iteration(i, sync):
    t = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    bg = torch.full((3,), float(i % 255) / 255.0, device=dev)
    t = mark("zero_grad+bg", t, sync)
    img, culled, uv = fused.rasterize(g, poses[i % 24], cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
    uv.retain_grad()
    t = mark("rasterize", t, sync)
    loss = ssim_l1_loss(img, target, 0.2)
    t = mark("loss", t, sync)
    loss.backward()
    t = mark("backward", t, sync)
    opt.step()
    t = mark("adam", t, sync)
    ctrl.accumulate(uv.grad, culled, cam)
    t = mark("statistics", t, sync)


for i in range(20):
    iteration(i, False)
torch.cuda.synchronize()
stages.clear()
t0 = time.perf_counter()
for i in range(100):
    iteration(i, False)
torch.cuda.synchronize()
free = (time.perf_counter() - t0) * 10
host = {k: v[0] * 10 for k, v in stages.items()}
stages.clear()
for i in range(100):
    iteration(i, True)
print(f"N={N}: {free:.3f} ms/iteration free-running; host enqueue ms: " + ", ".join(f"{k} {v:.3f}" for k, v in host.items()))
print("with a sync after every stage (host + GPU) ms: " + ", ".join(f"{k} {v[1] * 10:.3f}" for k, v in stages.items()))
