"""Per-visit latency of ONE workgroup of the render kernels (timeline build of the library).

    GSPLAT_HIP_LIB=gaussian_splatting_amd/libgsplat_hip_timeline.so python scripts/visit_latency.py

A 16x16 image (one tile) covered by N faint Gaussians: every wave visits every splat, every lane passes the
alpha test and no pixel saturates, so wave time / visits is the latency of one visit with nothing else on
the SIMD -- the regime the drain phase of a full frame runs in.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSPLAT_HIP_LIB", os.path.join(ROOT, "gaussian_splatting_amd", "libgsplat_hip_timeline.so"))

from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.splat_py.structs import Camera, Gaussians  # noqa: E402

CAP = 1 << 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1200)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fused.NATIVE = False
    lib = _hip.lib()
    N, W, H = args.n, 16, 16
    gen = torch.Generator().manual_seed(0)
    xyz = torch.zeros(N, 3)
    xyz[:, :2] = (torch.rand(N, 2, generator=gen) - 0.5) * 0.02
    xyz[:, 2] = 2.0 + torch.rand(N, generator=gen)
    g = Gaussians(xyz=xyz.to(dev), rgb=torch.rand(N, 3, generator=gen).to(dev),
                  opacity=torch.full((N, 1), -5.0).to(dev),            # sigmoid -> 0.0067
                  scale=torch.full((N, 3), 1.0).to(dev),               # exp -> 2.7: sigma ~ 21 px, covers the tile
                  quaternion=torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1).to(dev), sh=None)
    K = torch.tensor([[20.0, 0, 8.0], [0, 20.0, 8.0], [0, 0, 1.0]])
    cam = Camera(width=W, height=H, K=K.to(dev))
    T = torch.eye(4, device=dev)
    for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion):
        p.requires_grad_(True)
    gi = torch.ones(H, W, 3, device=dev)
    bg = torch.zeros(3, device=dev)
    buf = (ctypes.c_ulonglong * (2 * CAP * 10))()

    def frame():
        img, _, _ = fused.rasterize(g, T, cam, 0.3, 500.0, 100, 3.0, True, bg)
        img.backward(gi)
        return img

    for _ in range(3):
        frame()
    _hip.check(lib.gs_debug_render_timeline(buf, CAP))
    img = frame()
    _hip.check(lib.gs_debug_render_timeline(buf, CAP))
    rec = np.ctypeslib.as_array(buf).reshape(2, CAP, 10)
    out = {"tag": args.tag, "N": N, "image_mean": float(img.mean())}
    for name, r in (("forward", rec[0]), ("backward", rec[1])):
        r = r[r[:, 1] != 0]
        dur = (r[:, 1] - r[:, 0]).astype(np.float64) / 100.0
        cyc = (r[:, 4] & 0xffffffffffff).astype(np.float64)
        out[name] = {"waves": int(len(r)), "visits_per_wave": float(r[:, 3].mean()), "chunks_per_wave": float((r[:, 4] >> 48).mean()),
                     "shader_clock_mhz": float(cyc.sum() / dur.sum()),
                     "wave_us": float(dur.mean()), "ns_per_visit": float(1000 * dur.sum() / max(r[:, 3].sum(), 1))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
