"""What a depth-split forward of a band would have to walk (costing record, no kernel): from the forward's own
num_splats_per_pixel at workload D, per 8x8 patch (= one wave of the render kernels):
  * serial walk today: list entries up to the patch's largest num_splats (the wave walks until its last lane saturates)
  * a K-way split by depth with segments of SEG entries composited from T = 1 and combined needs, per pixel, a second
    sequential walk of the segment in which the pixel saturates (the stopping index and the colour up to it depend on
    the transmittance in front of the segment); a wave walks the UNION of its 64 lanes' stopping segments
usage: python scripts/experiments/depth_split_stats.py [--rows R0 R1]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs=2, default=[28, 35])
a = ap.parse_args()
N, W, H, deg = WORKLOADS["D"]
dev = torch.device("cuda", 0)
g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
d = DEFAULTS
f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H, d["near_thresh"],
                             d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None, 0)
bg = torch.zeros(3, device=dev)
image, nsp, fw, cost, seg = fused.render_forward(f.packed, f.rgb_render[:f.V], f.ranges, f.sorted_g, f.keys, bg, H, W,
                                                 None, 0, segments=False)
r0, r1 = a.rows
ntx = (W + 15) // 16
nsp = nsp[16 * r0:min(H, 16 * r1)].float()
Hb = nsp.shape[0] // 8 * 8
Wb = W // 8 * 8
p = nsp[:Hb, :Wb].reshape(Hb // 8, 8, Wb // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)   # [patches, 64 lanes]
lens = (f.ranges[1:] - f.ranges[:-1])[r0 * ntx:r1 * ntx].float()
out = {"band_rows": [r0, r1], "patches": int(p.shape[0]), "mean_list_len": float(lens.mean()),
       "num_splats_mean": float(p.mean()), "num_splats_p10_p50_p90": [float(x) for x in torch.quantile(p.flatten(), torch.tensor([0.1, 0.5, 0.9], device=dev))],
       "patch_max_mean": float(p.max(1).values.mean()), "patch_min_mean": float(p.min(1).values.mean()),
       "patch_spread_mean": float((p.max(1).values - p.min(1).values).mean())}
for SEG in (64, 128, 256):
    lo = torch.div(p.min(1).values, SEG, rounding_mode="floor")
    hi = torch.div((p.max(1).values - 1).clamp(min=0), SEG, rounding_mode="floor")
    # distinct stopping segments among the lanes (what a wave has to re-walk), and the contiguous span
    distinct = torch.stack([(torch.div((p - 1).clamp(min=0), SEG, rounding_mode="floor") == k).any(1) for k in range(int(hi.max()) + 1)], 1).sum(1).float()
    out[f"seg{SEG}"] = {"fixup_entries_distinct_mean": float(distinct.mean() * SEG),
                        "fixup_entries_span_mean": float(((hi - lo + 1) * SEG).mean()),
                        "phase_a_entries_longest_segment": SEG,
                        "today_serial_entries_mean": float(p.max(1).values.mean())}
print(json.dumps(out, indent=1))
