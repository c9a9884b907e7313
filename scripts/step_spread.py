"""Where does the step-time spread of `python bench.py` come from?  (Review r05 weak #8: 1.379 / 1.421 / 1.562 ms min /
median / max over the driver's 20 steps.)

N steps of the headline frame (workload D, fused.rasterize forward + backward) with, per step: the GPU time between
events on the launch stream, the host's enqueue timeline (step start, forward returned -- i.e. after the frame's one
host read --, backward returned), and -- sampled by a side thread every few ms -- the shader clock the driver reports
(/sys/class/drm/card*/device/pp_dpm_sclk, the line marked '*'; rocm-smi reads the same file).  Prints p10 / p50 / p90 /
p99 / max, the slow steps (> 1.05 x median) with their host timeline next to the typical one, and the clock levels seen.

usage: python scripts/step_spread.py [--steps 400] [--out gpurun_out/step_spread.json]"""
import argparse
import glob
import json
import os
import statistics
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import fused  # noqa: E402
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--workload", default="D")
ap.add_argument("--out", default="")
a = ap.parse_args()

N, W, H, deg = WORKLOADS[a.workload]
dev = torch.device("cuda", 0)
g, cam, T = make_scene(N, W, H, deg, seed=0, device=dev)
params = [p for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh) if p is not None]
for p in params:
    p.requires_grad_(True)
gi = make_grad_image(W, H, seed=1, device=dev)
bg = torch.zeros(3, device=dev)


def step(rec=None):
    for p in params:
        p.grad = None
    t0 = time.perf_counter()
    img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
    t1 = time.perf_counter()
    img.backward(gi)
    t2 = time.perf_counter()
    if rec is not None:
        rec.append((t0, t1, t2))


for _ in range(120):
    step()
torch.cuda.synchronize()

# ---- clock sampler ----------------------------------------------------------------------------------------------
sclk_files = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
samples, stop = [], threading.Event()


def read_sclk():
    out = []
    for f in sclk_files:
        try:
            for line in open(f):
                if "*" in line:
                    out.append(line.strip())
        except OSError:
            pass
    return out


def sampler():
    while not stop.is_set():
        samples.append((time.perf_counter(), read_sclk()))
        time.sleep(0.004)


th = threading.Thread(target=sampler, daemon=True)
if sclk_files:
    th.start()

import gc
gc.collect()
gc.disable()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
host = []
marks[0].record()
for k in range(a.steps):
    step(host)
    marks[k + 1].record()
torch.cuda.synchronize()
gc.enable()
stop.set()
gpu = [marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps)]
srt = sorted(gpu)
pct = lambda q: srt[min(len(srt) - 1, int(q * len(srt)))]
med = statistics.median(gpu)
fwd_host = [(t1 - t0) * 1e3 for t0, t1, t2 in host]
bwd_host = [(t2 - t1) * 1e3 for t0, t1, t2 in host]
gap_host = [(host[k + 1][0] - host[k][2]) * 1e3 for k in range(a.steps - 1)] + [0.0]
slow = [k for k in range(a.steps) if gpu[k] > 1.05 * med]
levels = {}
for _, s in samples:
    for line in s:
        levels[line] = levels.get(line, 0) + 1
out = {
    "workload": a.workload, "steps": a.steps,
    "gpu_ms": {"min": round(srt[0], 4), "p10": round(pct(0.10), 4), "p50": round(med, 4), "p90": round(pct(0.90), 4),
               "p99": round(pct(0.99), 4), "max": round(srt[-1], 4), "mean": round(sum(gpu) / len(gpu), 4)},
    "host_ms_typical": {"forward_call": round(statistics.median(fwd_host), 4), "backward_call": round(statistics.median(bwd_host), 4),
                        "between_steps": round(statistics.median(gap_host), 4)},
    "slow_steps_over_1.05_median": len(slow),
    "slow_steps": [{"step": k, "gpu_ms": round(gpu[k], 4), "host_forward_call_ms": round(fwd_host[k], 4),
                    "host_backward_call_ms": round(bwd_host[k], 4), "host_gap_before_ms": round(gap_host[k - 1] if k else 0.0, 4)}
                   for k in slow[:40]],
    "slow_step_runs": [],   # consecutive runs of slow steps (a clock dip shows as a run, a host hiccup as a single step)
    "sclk_levels_seen": levels, "sclk_samples": len(samples),
    "frame_counters": fused.counters(),
}
run = []
for k in slow:
    if run and k == run[-1] + 1:
        run.append(k)
    else:
        if run:
            out["slow_step_runs"].append([run[0], len(run)])
        run = [k]
if run:
    out["slow_step_runs"].append([run[0], len(run)])
# does a slow GPU step coincide with a slow host call of the same step?
if slow:
    out["slow_steps_with_host_forward_over_1.5x"] = sum(1 for k in slow if fwd_host[k] > 1.5 * out["host_ms_typical"]["forward_call"])
    out["slow_steps_with_host_backward_over_1.5x"] = sum(1 for k in slow if bwd_host[k] > 1.5 * out["host_ms_typical"]["backward_call"])
print(json.dumps(out, indent=1))
if a.out:
    json.dump(dict(out, gpu_ms_trace=[round(x, 4) for x in gpu]), open(a.out, "w"), indent=1)
