"""Experiment: does walking the visible Gaussians in a spatially coherent order (sorted by the coarse screen
cell of their centre) make the binning's key scatter cheaper?  Times gs_tile_count + gs_tile_emit_sort over
all visible Gaussians in index order and through a permutation passed as the `subset` list; the resulting
tile lists must be identical.  usage: python scripts/exp_spatial_order.py [--workload D] [--cell 8]"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="D")
ap.add_argument("--cell", type=int, default=8, help="cell edge in tiles")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, W, H, deg = WORKLOADS[a.workload]
g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
d = DEFAULTS
f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H, d["near_thresh"],
                             d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None, _hip.GS_SORT_PREFIX)
V, S = f.V, f.S
ntx, nty, Tn = f.ntx, f.nty, f.T
uv = f.uv[:V]
tx = (uv[:, 0] / 16).floor().clamp(0, ntx - 1).long() // a.cell
ty = (uv[:, 1] / 16).floor().clamp(0, nty - 1).long() // a.cell
cell = ty * ((ntx + a.cell - 1) // a.cell) + tx
perm = torch.argsort(cell, stable=True).to(torch.int32).contiguous()
nperm = torch.tensor([V], dtype=torch.int32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
mh = ctypes.c_float(d["mh_dist"])


def run(subset, subset_n, prefix):
    ws = torch.empty(_hip.lib().gs_tile_workspace_ints(Tn), dtype=torch.int32, device=dev)
    ranges = torch.empty(Tn + 2, dtype=torch.int32, device=dev)
    keys = torch.empty(S, dtype=torch.int64, device=dev)
    out = torch.empty(S, dtype=torch.int32, device=dev)
    _hip.call("gs_tile_count", p(f.uv), p(f.conic), N, p(f.count), p(subset), p(subset_n), ntx, nty, mh, 0, nty, p(ws),
              p(ranges), None, stream)
    _hip.call("gs_tile_emit_sort", p(f.uv), p(f.xyz_cam), p(f.conic), N, p(f.count), p(subset), p(subset_n), ntx, nty, mh,
              0, nty, p(ranges), p(ws), p(keys), ctypes.c_int64(S), p(out), prefix, stream)
    return ranges, out


res = {}
for label, sub, n in (("index order", None, None), (f"cell order ({a.cell}x{a.cell} tiles)", perm, nperm)):
    ref = run(sub, n, 0)
    for _ in range(3):
        run(sub, n, _hip.GS_SORT_PREFIX)
    _hip.reserve_events(8 * a.reps)
    _hip.enable_timing(True)
    for _ in range(a.reps):
        run(sub, n, _hip.GS_SORT_PREFIX)
    tm = _hip.collect_timing()
    _hip.enable_timing(False)
    res[label] = (ref, {k: round(statistics.median(v), 4) for k, v in tm.items()})
    print(label, res[label][1])
(r0, s0), (r1, s1) = [v[0] for v in res.values()]
print("ranges equal:", torch.equal(r0[:Tn + 1], r1[:Tn + 1]), " full-sort lists equal:", torch.equal(s0, s1))
