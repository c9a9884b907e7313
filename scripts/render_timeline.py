"""Wave timeline of the render kernels on one frame (instrumented build: make -C gaussian_splatting_amd/csrc stats ->
libgsplat_hip_timeline.so).

    python scripts/render_timeline.py --workload D [--out gpurun_out/timeline_D.json]

Every wave of k_render_fwd / k_render_bwd records its begin and end time (100 MHz constant clock), the
compute unit it ran on and its visit count.  Printed per kernel: span, mean concurrency (waves in flight),
the concurrency profile over 20 equal time slices, the share of the span during which fewer than half the
peak number of waves were in flight (the tail), and the spread of per-wave durations and per-CU busy time.
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
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene  # noqa: E402

CAP = 1 << 16   # GS_TIMELINE_CAP of csrc/render.hip


def analyse(rec, slices=20):
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    start, end = t0.min(), t1.max()
    span = max(int(end - start), 1)
    dur = (t1 - t0).astype(np.float64)
    hw = rec[:, 2]
    hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
    cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
    edges = np.linspace(start, end, slices + 1)
    conc = []
    for a, b in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None)
        conc.append(float(overlap.sum() / max(b - a, 1)))
    # concurrency at fine resolution for the tail measure
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])
    dt = np.diff(np.append(ev[:, 0], end))
    peak = level.max()
    tail = float(dt[level < 0.5 * peak].sum() / span)
    busy = {}
    for c, d in zip(cu, dur):
        busy[c] = busy.get(c, 0.0) + d
    b = np.array(list(busy.values()))
    visits = rec[:, 3].astype(np.float64)
    return {
        "waves": int(len(rec)), "span_us": span / 100.0, "wave_time_sum_us": float(dur.sum() / 100.0),
        "mean_waves_in_flight": float(dur.sum() / span), "peak_waves_in_flight": int(peak),
        "share_of_span_below_half_peak": tail,
        "waves_in_flight_by_slice": [round(c, 1) for c in conc],
        "wave_duration_us": {"mean": float(dur.mean() / 100), "p50": float(np.percentile(dur, 50) / 100),
                             "p90": float(np.percentile(dur, 90) / 100), "max": float(dur.max() / 100)},
        "compute_units_seen": int(len(b)),
        "cu_busy_wave_us": {"min": float(b.min() / 100), "mean": float(b.mean() / 100), "max": float(b.max() / 100)},
        "visits_per_wave": {"mean": float(visits.mean()), "p90": float(np.percentile(visits, 90)),
                            "max": float(visits.max())},
        "shader_clock_mhz": float((rec[:, 4] & 0xffffffffffff).astype(np.float64).sum() / (dur.sum() / 100.0)),
        "phase_share_of_wave_cycles": dict(zip(["stage", "touch_masks", "walk", "barrier_after_walk", "flush"],
                                               [round(float(x), 4) for x in rec[:, 5:10].astype(np.float64).sum(0) /
                                                max((rec[:, 4] & 0xffffffffffff).astype(np.float64).sum(), 1)])),
        "ns_per_visit_per_wave": float(dur.sum() * 10.0 / max(visits.sum(), 1)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="D")
    ap.add_argument("--out", default="")
    ap.add_argument("--dump", default="", help="npy file for the raw records")
    ap.add_argument("--rows", type=int, nargs=2, default=None, help="tile rows [R0, R1): a multi-GPU rank's band")
    ap.add_argument("--segments", default="auto", choices=["auto", "on", "off"], help="depth-segmented backward")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fused.NATIVE = False   # the native frame module is linked against the product library
    fused.SEGMENTS = {"auto": "auto", "on": True, "off": False}[args.segments]
    rows = tuple(args.rows) if args.rows else None
    lib = _hip.lib()
    N, W, H, deg = WORKLOADS[args.workload]
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
    for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh):
        if p is not None:
            p.requires_grad_(True)
    gi = make_grad_image(W, H, seed=1, device=dev)
    bg = torch.zeros(3, device=dev)
    buf = (ctypes.c_ulonglong * (2 * CAP * 10))()

    def frame():
        img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows, **DEFAULTS)
        img.backward(gi)

    for _ in range(3):
        frame()
    _hip.check(lib.gs_debug_render_timeline(buf, CAP))
    frame()
    _hip.check(lib.gs_debug_render_timeline(buf, CAP))
    rec = np.ctypeslib.as_array(buf).reshape(2, CAP, 10).copy()
    fwd, bwd = (r[r[:, 1] != 0] for r in rec)
    out = {"workload": args.workload, "tile_rows": list(rows) if rows else None, "segments": args.segments,
           "forward": analyse(fwd), "backward": analyse(bwd)}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text + "\n")
    if args.dump:
        np.save(args.dump, rec)


if __name__ == "__main__":
    main()
