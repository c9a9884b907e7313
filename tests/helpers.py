"""Shared test helpers: fixture loading and scene construction."""
import os

import numpy as np
import torch

from gaussian_splatting_amd.splat_py.structs import Camera, Gaussians

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SH_0 = 0.28209479177387814


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def t(a, device="cpu", dtype=None):
    x = torch.from_numpy(np.asarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(device).contiguous()


def scene_from_fixture(fx, device="cpu", requires_grad=False, opacity_key="in_opacity"):
    def p(key):
        x = t(fx[key], device)
        return x.requires_grad_(True) if requires_grad else x

    sh = p("in_sh") if "in_sh" in fx.files else None
    g = Gaussians(p("in_xyz"), p("in_rgb"), p(opacity_key), p("in_scale"), p("in_quaternion"), sh)
    cam = Camera(int(fx["in_width"]), int(fx["in_height"]), t(fx["in_K"], device))
    return g, cam, t(fx["in_camera_T_world"], device)


def scene6(device="cpu"):
    """The reference's 6-Gaussian test scene (test/gaussian_test_data.py:6-86), from the fixture;
    opacity already passed through inverse_sigmoid as test_rasterize.py:18 does."""
    fx = load("ref_host_scene6.npz")
    return scene_from_fixture(fx, device, opacity_key="in_opacity_logit") + (fx,)


def rel_err(g, ref, floor_frac=1e-2):
    """Element-wise relative error max |g - ref| / max(|ref|, floor_frac * max|ref|).

    The floor is needed because per-Gaussian gradients are fp32 sums over pixels in an unspecified
    order (warp reduce + atomicAdd in the reference, wave reduce + atomics here): an element that
    is a near-cancelling sum carries the rounding noise of its largest terms (measured: 5e-7 of the
    tensor's max), so its relative error is unbounded as it approaches zero.  With the floor at 1 %
    of the tensor's max the measured error is 3e-5; at 1e-6 (the floor SURVEY.md 8(d) proposed) it
    is 1e-2 for the same data -- reorder noise, not a kernel difference."""
    g = g.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if ref.numel() == 0:
        return 0.0
    floor = floor_frac * ref.abs().max().item()
    den = torch.clamp(ref.abs(), min=max(floor, 1e-300))
    return ((g - ref).abs() / den).max().item()


def scaled_err(g, ref):
    """max |g - ref| / max|ref| -- error relative to the tensor's scale"""
    g = g.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if ref.numel() == 0:
        return 0.0
    return ((g - ref).abs().max() / ref.abs().max().clamp(min=1e-300)).item()


def noise_normalised_err(g, ref, abs_sum):
    """max over ALL elements (no floor) of |g - ref| / abs_sum.

    `abs_sum` is the oracle's sum, over the pixels that contribute to a gradient element, of the
    magnitudes of the LEAF terms of its formula -- every product that enters a sum or a difference
    (oracle.gs_oracle.render_tiles_backward_abs).  It is the scale of the unavoidable fp32 noise of that
    element: a different but equally valid fp32 evaluation (other factoring, reciprocal instead of
    division, another summation order -- the reference's own warp reduce + atomicAdd has that freedom)
    deviates from the oracle's value by a few 2^-24 of it, whatever cancels afterwards.  A correct kernel
    therefore stays within ~1e-5 of every element's own scale, including the small, cancelling elements
    that `rel_err`'s floor lets through."""
    g = g.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    a = abs_sum.detach().double().cpu().reshape(-1)
    if ref.numel() == 0:
        return 0.0
    touched = a > 0
    assert not g[~touched].any(), "gradient on an element that no pixel contributes to"
    if not touched.any():
        return 0.0
    return ((g - ref).abs()[touched] / a[touched]).max().item()


# ---- parity report: numbers the GPU tests measured, printed at the end of the run and written to
# gpurun_out/parity_report.json (tests/conftest.py) -------------------------------------------------------
REPORT = []


def report(test, **values):
    REPORT.append(dict(test=test, **{k: (float(v) if isinstance(v, (int, float)) else v) for k, v in values.items()}))


def grad_errors(g, ref, abs_sum=None):
    """the three error measures of one gradient tensor: SURVEY.md 8(d)'s (floor 1e-6 of the max), the
    1 %-floor form the assertions use, the error relative to the tensor's scale, and -- when the oracle's
    abs-sums are given -- the floor-free noise-normalised error"""
    out = {"rel_floor_1e-6": rel_err(g, ref, 1e-6), "rel_floor_1e-2": rel_err(g, ref, 1e-2), "scaled": scaled_err(g, ref)}
    if abs_sum is not None:
        out["noise_normalised"] = noise_normalised_err(g, ref, abs_sum)
    return out
