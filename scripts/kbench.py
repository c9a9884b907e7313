"""Kernel-level A/B harness: times the entry points of ONE build of the library on one frame.

    GSPLAT_HIP_LIB=<path to a libgsplat_hip*.so> python scripts/kbench.py --workload D [--save ref.pt | --check ref.pt]

Runs the per-Gaussian stage, binning and sort once, then `--reps` x (render forward, render backward,
[--full: the whole stage chain]) with events around every C-ABI call; prints the median GPU time per entry
point.  --save keeps the image and the render-gradient slab of this build; --check compares this build's
against a saved one (image: bit equality; slab: error relative to the tensor's scale), so that several
variant builds can be compared in one GPU call.
"""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gaussian_splatting_amd import _hip, fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="D")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--save", default="")
    ap.add_argument("--check", default="")
    ap.add_argument("--full", action="store_true", help="also time preprocess / binning / per-Gaussian backward")
    ap.add_argument("--tag", default="")
    ap.add_argument("--natural-order", action="store_true", help="backward without the longest-first tile order")
    ap.add_argument("--binning-only", action="store_true")
    ap.add_argument("--segments", type=int, default=0, help="1: depth-segmented backward (forward leaves the state)")
    ap.add_argument("--depth-cut", type=int, default=0, help="1: depth-bucketed binning (csrc/binning.hip 'depth cut')")
    ap.add_argument("--rows", type=int, nargs=2, default=None, metavar=("R0", "R1"),
                    help="render forward / backward only the tile rows [R0, R1) (a multi-GPU rank's band)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    _hip.lib()
    N, W, H, deg = WORKLOADS[args.workload]
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
    gi = make_grad_image(W, H, seed=1, device=dev)
    bg = torch.zeros(3, device=dev)
    d = DEFAULTS

    def stage1():
        return fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H,
                                        d["near_thresh"], d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None,
                                        _hip.GS_SORT_PREFIX, depth_cut=bool(args.depth_cut))

    rows = tuple(args.rows) if args.rows else None
    if args.binning_only:
        # experiment builds of binning.hip whose emit writes elsewhere (timing only): the lists are garbage, so
        # nothing downstream of the sort may run
        for _ in range(3):
            stage1()
        torch.cuda.synchronize()
        _hip.reserve_events(2 * 16 * args.reps)
        _hip.enable_timing(True)
        for _ in range(args.reps):
            stage1()
        timing = _hip.collect_timing()
        _hip.enable_timing(False)
        print(json.dumps({"lib": os.path.basename(_hip.LIB_PATH), "tag": args.tag, "workload": args.workload,
                          "median_ms": {k: round(statistics.median(v), 4) for k, v in sorted(timing.items())}}))
        return
    f = stage1()
    V = f.V
    rgb_v = f.rgb_render[:V]

    def fwd():
        return fused.render_forward(f.packed, rgb_v, f.ranges, f.sorted_g, f.keys, bg, H, W, rows, _hip.GS_SORT_PREFIX,
                                    segments=args.segments, cut=f.cut)

    def bwd(nsp, fw, cost=None, seg=None):
        return fused.render_backward(f.packed, rgb_v, f.ranges, f.sorted_g, bg, nsp, fw, gi, H, W, rows, V, cost,
                                     seg_state=seg, cut=f.cut)

    image, nsp, fw, cost, seg = fwd()
    slab = bwd(nsp, fw, None if args.natural_order else cost, seg)
    torch.cuda.synchronize()
    for _ in range(3):
        image, nsp, fw, cost, seg = fwd()
        slab = bwd(nsp, fw, None if args.natural_order else cost, seg)
    torch.cuda.synchronize()
    _hip.reserve_events(2 * 16 * args.reps)
    _hip.enable_timing(True)
    for _ in range(args.reps):
        if args.full:
            f2 = stage1()
            if args.depth_cut:
                f = f2   # (the render passes read the frame's own cut record)
        image, nsp, fw, cost, seg = fwd()
        slab = bwd(nsp, fw, None if args.natural_order else cost, seg)
        if args.full:
            fused.preprocess_backward(g.xyz, g.quaternion, g.scale, T, cam.K, f, slab)
    timing = _hip.collect_timing()
    _hip.enable_timing(False)
    out = {"lib": os.path.basename(_hip.LIB_PATH), "tag": args.tag, "workload": args.workload, "V": V, "S": f.S,
           "median_ms": {k: round(statistics.median(v), 4) for k, v in sorted(timing.items())},
           "min_ms": {k: round(min(v), 4) for k, v in sorted(timing.items())}}
    if args.save:
        torch.save({"image": image.cpu(), "slab": slab.cpu(), "nsp": nsp.cpu()}, args.save)
    if args.check:
        ref = torch.load(args.check)
        out["image_equal"] = bool(torch.equal(image.cpu(), ref["image"]))
        out["image_max_abs_diff"] = float((image.cpu() - ref["image"]).abs().max())
        out["nsp_mismatches"] = int((nsp.cpu() != ref["nsp"]).sum())
        out["nsp_equal"] = bool(torch.equal(nsp.cpu(), ref["nsp"]))
        a, b = slab.cpu().double(), ref["slab"].double()
        out["slab_scaled_err_per_column"] = [float(((a[:, j] - b[:, j]).abs().max() / b[:, j].abs().max().clamp(min=1e-300)))
                                             for j in range(a.shape[1])]
        # SURVEY 8(d)'s measure with the 1 % floor, per gradient tensor (colour 0-2, opacity 3, uv 4-5, conic 6-8)
        def floor_err(c0, c1):
            top = b[:, c0:c1].abs().max().clamp(min=1e-300)
            return float(((a[:, c0:c1] - b[:, c0:c1]).abs() / b[:, c0:c1].abs().clamp(min=1e-2 * top)).max())
        out["slab_rel_err_floor_1e-2"] = {"rgb": floor_err(0, 3), "opacity": floor_err(3, 4), "uv": floor_err(4, 6),
                                          "conic": floor_err(6, 9)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
