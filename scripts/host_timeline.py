"""Host-side timeline of one owner-mode band frame (where does the host block, how long does it
take to enqueue each stage).  usage: python scripts/host_timeline.py [--world 8] [--defer 0|1]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip, fused, sharded
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--defer", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--native", type=int, default=1, help="0: the Python orchestration of the sharded frame")
ap.add_argument("--compact", type=int, default=1, help="0: replicated per-Gaussian stage (native path)")
ap.add_argument("--fused", type=int, default=1, help="0: the three-call band pipeline of rounds 3-5 (A/B)")
a = ap.parse_args()
if fused.native() is not None:
    fused.native().set_band_fused(bool(a.fused))
sharded.DEFER_HOST_READ = bool(a.defer)
sharded.NATIVE = bool(a.native)
sharded.BAND_COMPACT = bool(a.compact)
N, W, H, deg = WORKLOADS["D"]
g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
gi = make_grad_image(W, H, seed=1, device="cuda")
bg = torch.zeros(3, device="cuda")
rank = a.world // 2
owned = sharded.owned_slice(g, a.world, rank)
rast = sharded.ShardedRasterizer(H, a.world, rank, grad_mode="owner", all_to_all=lambda r, s, rs, ss: r.zero_())

marks = []
orig_call = _hip.call


def timed_call(name, *args):
    t0 = time.perf_counter()
    orig_call(name, *args)
    marks.append((name, t0, time.perf_counter()))


_hip.call = timed_call
fused._hip.call = timed_call
orig_sync = torch.cuda.Event.synchronize


def timed_sync(self):
    t0 = time.perf_counter()
    orig_sync(self)
    marks.append(("EVENT_SYNC", t0, time.perf_counter()))


torch.cuda.Event.synchronize = timed_sync


def step():
    for p in (owned.xyz, owned.rgb, owned.opacity, owned.scale, owned.quaternion, owned.sh):
        p.grad = None
    img, _, _ = rast.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, owned=owned, **DEFAULTS)
    marks.append(("FORWARD_RETURNED", time.perf_counter(), time.perf_counter()))
    img.backward(gi)


for _ in range(5):
    step()
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
print(f"native={a.native} compact={a.compact} fused={a.fused} defer={a.defer} world={a.world}: {(time.perf_counter() - t_all) / a.steps * 1e3:.3f} ms/step")
marks.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
step()
t_host_done = time.perf_counter()
torch.cuda.synchronize()
t_end = time.perf_counter()
for name, s, e in marks:
    print(f"{(s - t0) * 1e3:8.3f} ms  +{(e - s) * 1e3:6.3f}  {name}")
print(f"host done at {(t_host_done - t0) * 1e3:.3f} ms, GPU done at {(t_end - t0) * 1e3:.3f} ms")
