"""Event counts of the render kernels on one frame (instrumented build: make -C gaussian_splatting_amd/csrc stats).

    GSPLAT_HIP_LIB=gaussian_splatting_amd/libgsplat_hip_stats.so python scripts/render_stats.py --workload D

Prints, for forward and backward: waves, chunks, visits (touch-mask bits walked by a wave), visits with a
lane inside the cutoff circle, visits with a contributing lane, contributing (pixel, splat) pairs, and the
pixel-splat evaluation count E of SURVEY.md 8(d).
"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSPLAT_HIP_LIB", os.path.join(ROOT, "gaussian_splatting_amd", "libgsplat_hip_stats.so"))

from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene  # noqa: E402

FWD = ["waves", "chunks", "visits", "visits_in_cutoff", "visits_hit", "pairs_hit", "live_lanes_at_visit",
       "list_entries", "entries_staged", "half_wave_lockstep_visits", "upper_half_visits_in_cutoff",
       "lower_half_visits_in_cutoff"]
BWD = ["waves", "chunks", "visits", "visits_in_cutoff", "visits_hit", "pairs_hit", "reaching_lanes_at_visit",
       "list_entries", "entries_used", "visits_reached", "visits_few_path", "rows_flushed",
       "half_wave_lockstep_visits", "upper_half_visits_in_cutoff", "lower_half_visits_in_cutoff"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="D")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fused.NATIVE = False   # the native frame module is linked against the product library, not the instrumented one
    lib = _hip.lib()
    N, W, H, deg = WORKLOADS[args.workload]
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
    for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh):
        if p is not None:
            p.requires_grad_(True)
    gi = make_grad_image(W, H, seed=1, device=dev)
    bg = torch.zeros(3, device=dev)
    buf = (ctypes.c_ulonglong * 32)()

    def frame():
        img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
        img.backward(gi)

    frame()   # warm-up: the second frame takes the speculative path like the bench
    _hip.check(lib.gs_debug_render_stats(buf, 1))
    frame()
    _hip.check(lib.gs_debug_render_stats(buf, 1))
    v = list(buf)
    out = {"workload": args.workload, "forward": dict(zip(FWD, v[:len(FWD)])),
           "backward": dict(zip(BWD, v[16:16 + len(BWD)]))}
    f, b = out["forward"], out["backward"]
    out["derived"] = {
        "E_forward_pixel_splat_evaluations": 64 * f["visits"],
        "E_backward_pixel_splat_evaluations": 64 * b["visits_reached"],
        "fwd_visits_per_wave": f["visits"] / max(f["waves"], 1),
        "fwd_hit_visit_fraction": f["visits_hit"] / max(f["visits"], 1),
        "fwd_lane_utilisation_of_hit_visits": f["pairs_hit"] / max(64 * f["visits_hit"], 1),
        "fwd_fraction_of_lists_staged": f["entries_staged"] / max(f["list_entries"], 1),
        "bwd_visits_per_wave": b["visits"] / max(b["waves"], 1),
        "bwd_reached_fraction": b["visits_reached"] / max(b["visits"], 1),
        "bwd_hit_visit_fraction": b["visits_hit"] / max(b["visits"], 1),
        "bwd_lane_utilisation_of_hit_visits": b["pairs_hit"] / max(64 * b["visits_hit"], 1),
        "bwd_few_path_fraction_of_hits": b["visits_few_path"] / max(b["visits_hit"], 1),
        "bwd_fraction_of_lists_used": b["entries_used"] / max(b["list_entries"], 1),
        # two 8x4 half-waves in lockstep: steps per chunk = the larger of the halves' visit counts (cutoff test per
        # lane: the lower bound of a rectangle test), against the visits the kernels walk today
        "fwd_half_wave_lockstep_over_visits": f["half_wave_lockstep_visits"] / max(f["visits"], 1),
        "fwd_visits_in_cutoff_over_visits": f["visits_in_cutoff"] / max(f["visits"], 1),
        "bwd_half_wave_lockstep_over_visits_reached": b["half_wave_lockstep_visits"] / max(b["visits_reached"], 1),
        "bwd_visits_in_cutoff_over_visits_reached": b["visits_in_cutoff"] / max(b["visits_reached"], 1),
    }
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text + "\n")


if __name__ == "__main__":
    main()
