"""Parity at BASELINE.json's FULL sizes (configs B, C, D) against the CPU oracle run on the WHOLE scene.

tests/test_gpu_scale.py checks the full-size frames by properties and lets the oracle render a few tile rows from
the GPU's own per-splat inputs and the GPU's own tile lists -- a tile instance the HIP binning dropped at 2.86 M
Gaussians would pass there.  Here the oracle does the per-Gaussian stage (projection.cu:9-257, the cull of
rasterize.py:33-75) and get_sorted_gaussian_list (tile_culling.cu:124-340) for every Gaussian of the scene
(about 3 s of host time at D) and the fused HIP path must reproduce, bit for bit: culling mask, uv,
xyz_camera_frame, conic, opacity, the visible index, tile_ranges and sorted_gaussians.  The band checks then feed the
ORACLE's per-splat values and the ORACLE's lists to the oracle's renderer and compare the GPU frame -- the
product's default path (native orchestration, prefix sort) -- with that."""
import pytest
import torch

from gaussian_splatting_amd import fused
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

from .helpers import grad_errors, rel_err, report
from .test_gpu_fused import cpu_expected_stages
from .test_gpu_scale import RENDER_GRADS, check_band_backward, oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"
BAND = {"B": (30, 33), "C": (25, 27), "D": (26, 28)}


def expected(workload):
    N, W, H, deg = WORKLOADS[workload]
    g, cam, T = make_scene(N, W, H, deg, seed=0)
    d = DEFAULTS
    return cpu_expected_stages(g, cam, T, d["near_thresh"], d["far_thresh"], d["cull_mask_padding"], d["mh_dist"])


def oracle_band(exp, rgb, W, H, rows, bg, grad_image=None, sum_mode=0):
    """the oracle's render (+ backward) of tile rows from the ORACLE's own per-splat values and lists"""
    orc = oracle()
    img = torch.zeros(H, W, 3)
    nsp = torch.zeros(H, W, dtype=torch.int32)
    fw = torch.zeros(H, W)
    rays = torch.zeros(1, 1, 1)
    args = (exp["uv"], exp["opacity"].reshape(-1, 1).contiguous(), rgb, exp["conic"], rays, exp["ranges"], exp["sorted"], bg)
    orc.render_tiles_cuda(*args, nsp, fw, img, tile_rows=rows)
    out = dict(image=img, nsp=nsp)
    if grad_image is not None:
        V = exp["V"]
        shapes = ((V, 3), (V, 1), (V, 2), (V, 3))
        g = [torch.zeros(*s) for s in shapes]
        orc.set_backward_sum(sum_mode)
        try:
            orc.render_tiles_backward_cuda(*args, nsp, fw, grad_image, *g, tile_rows=rows)
        finally:
            orc.set_backward_sum(0)
        out.update(g_rgb=g[0], g_opa=g[1], g_uv=g[2], g_conic=g[3])
        if sum_mode == 0:
            a = [torch.zeros(*s) for s in shapes]
            orc.render_tiles_backward_abs(*args, nsp, fw, grad_image, *a, tile_rows=rows)
            out.update(a_rgb=a[0], a_opa=a[1], a_uv=a[2], a_conic=a[3])
    return out


@pytest.mark.parametrize("workload", ["B", "C", "D"])
def test_full_scene_stages_and_tile_lists_equal_the_oracle(workload):
    N, W, H, deg = WORKLOADS[workload]
    exp = expected(workload)
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
    bg = torch.full((3,), 0.5, device=DEV)
    image, mask, uv, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, return_aux=True,
                                           **DEFAULTS)
    V = exp["V"]
    S = int(exp["sorted"].numel())
    report(f"full_scene_parity[{workload}]", N=N, V=V, S=S)
    assert 0 < V < N
    assert torch.equal(mask.cpu(), exp["culled"])
    assert torch.equal(uv.detach().cpu(), exp["uv"])
    assert torch.equal(aux["xyz_camera_frame"].cpu(), exp["xyz_c"])
    assert torch.equal(aux["conic"].detach().cpu(), exp["conic"])
    assert torch.equal(aux["opacity"].detach().cpu().reshape(-1), exp["opacity"].reshape(-1))
    assert torch.equal(aux["vis_idx"].cpu().long(), torch.nonzero(~exp["culled"]).flatten())
    # every tile instance of the frame, in the oracle's order (tile_culling.cu:124-340)
    assert torch.equal(aux["tile_ranges"].cpu(), exp["ranges"])
    assert torch.equal(aux["sorted_gaussians"].cpu(), exp["sorted"])
    # SH colour: identity camera, so the camera centre is (0, 0, 0) on both sides
    rgb_gpu = aux["rgb"].detach().cpu().contiguous()
    assert (rgb_gpu - exp["rgb"]).abs().max() < 2e-6
    rgb = exp["rgb"] if torch.equal(rgb_gpu, exp["rgb"]) else rgb_gpu
    report(f"full_scene_parity[{workload}]", sh_colour_bit_equal=float(rgb is exp["rgb"]))
    # the image of the band from the oracle's values and lists
    rows = BAND[workload]
    ref = oracle_band(exp, rgb, W, H, rows, bg.cpu())
    y0, y1 = rows[0] * 16, rows[1] * 16
    assert torch.equal(image.detach().cpu()[y0:y1], ref["image"][y0:y1])
    # the product's default path (native orchestration, prefix sort, no aux) renders the same frame
    image2, mask2, uv2 = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
    assert torch.equal(image2, image) and torch.equal(mask2, mask) and torch.equal(uv2, uv)


@pytest.mark.parametrize("workload", ["B", "C", "D"])
def test_band_backward_from_the_oracles_own_inputs(workload):
    """render backward of a band: GPU (its own pipeline end to end) vs the oracle fed with the oracle's
    per-splat values and the oracle's tile lists"""
    N, W, H, deg = WORKLOADS[workload]
    rows = BAND[workload]
    exp = expected(workload)
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        getattr(g, k).requires_grad_(True)
    bg = torch.full((3,), 0.5, device=DEV)
    img, mask, uv, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows,
                                         return_aux=True, **DEFAULTS)
    for k in ("conic", "opacity", "rgb"):
        aux[k].retain_grad()
    uv.retain_grad()
    gi = make_grad_image(W, H, seed=1)
    img.backward(gi.to(DEV))
    grads = dict(uv=uv.grad, conic=aux["conic"].grad, opacity_act=aux["opacity"].grad, rgb_render=aux["rgb"].grad)
    rgb_gpu = aux["rgb"].detach().cpu().contiguous()
    rgb = exp["rgb"] if torch.equal(rgb_gpu, exp["rgb"]) else rgb_gpu
    ref = oracle_band(exp, rgb, W, H, rows, bg.cpu(), gi)
    assert torch.equal(img.detach().cpu(), ref["image"])
    check_band_backward(f"band_backward_from_oracle_inputs[{workload}] rows {rows[0]}-{rows[1]}", grads, ref)


# how far above the pure summation-order spread the kernel may read at the 1e-6 floor, per tensor.  Measured at
# workload D (rows 26-28): colour 1.3x, uv 1.1x, conic 1.0x, opacity 4x.  The colour gradient's terms
# (alpha weight Y0 grad_image) are formed as in the oracle, so only the order differs: 2x.  The opacity / uv / conic
# terms hang on grad_alpha = sum (c weight - colour_accum / (1 - alpha)) grad_image, a cancelling difference that
# the kernel evaluates with fused multiply-adds and in a different factoring than the oracle's literal, uncontracted
# form (as an nvcc build of the reference would: nvcc contracts by default): per-term differences of a few ulp on
# top of the order noise, <= 3e-7 of the element's own leaf-term magnitude (the floor-free criterion asserted in
# check_band_backward).  NOT the cause: the hardware reciprocal in the transmittance walk -- a build with a Newton
# step on it (the BWD_RCP_REFINE build of scripts/experiments/render_macro_experiments.patch) reads the same 0.9e-3 .. 1.0e-3 for the opacity.
# Round 5: the factors are what was MEASURED plus a quarter, per kernel -- the unsegmented walk (band rows 26-28 /
# the whole frame at D, profiles/r04/parity_report.json): colour 1.35 / 1.48, opacity 4.06 / 3.22, uv 1.31 / 2.75,
# conic 0.70 / 1.14 x the spread; the depth-segmented walk (band): 0.99, 7.55, 4.81, 0.64 (one more rounding per
# segment boundary on every weight behind it).  Until round 4 one table allowed 8x for three tensors
# of BOTH kernels: a kernel error worth 5x the noise floor on small elements would have passed.  Across the four
# tensors together the unsegmented kernel must stay within 2x of the largest spread (measured 1.77 on the whole
# frame: 0.0145 against 0.0082).
REORDER_FACTOR = {"rgb_render": 2.0, "opacity_act": 5.0, "uv": 3.5, "conic": 2.0}
REORDER_FACTOR_SEGMENTED = {"rgb_render": 2.0, "opacity_act": 9.5, "uv": 6.0, "conic": 2.0}
REORDER_FACTOR_ALL_TENSORS = 2.0


@pytest.mark.parametrize("segments", [False, True], ids=["unsegmented", "depth-segmented"])
def test_gradient_error_is_within_the_fp32_reorder_spread(segments):
    """SURVEY.md 8(d) writes the gradient criterion with a floor of 1e-6 of the tensor's maximum; the HIP kernel
    reads ~1e-3 .. 3e-3 there at workload D (the first parity number of bench.py's line), target 1e-4.  Evidence
    that this is fp32 rounding of cancelling sums and not a kernel error: the oracle sums ITS OWN per-pixel terms
    (bit-identical terms) in fp32 in two fixed orders (tests/test_grad_noise_floor.py); the criterion between
    either of those and the double sum is the figure an fp32 implementation with the oracle's exact per-term
    arithmetic gets -- already 2e-4 .. 1.3e-3, above the target.  The kernel must stay within REORDER_FACTOR of
    it, per tensor, at the 1e-6 floor, on a D band -- both backward kernels a band can take: the unsegmented walk
    (what the full frame runs; tests/test_gpu_wholeframe_parity.py repeats this over all rows) and the
    depth-segmented one (the default below 1500 tiles: a multi-GPU rank's band).  The segmented walk resumes a
    pixel from products of stored factors instead of carrying one running weight through the whole list: one more
    rounding per segment boundary on every weight behind it, on top of the order noise (uv read 4.8x the spread in
    round 3, inside the factor asserted here)."""
    workload = "D"
    N, W, H, deg = WORKLOADS[workload]
    rows = BAND[workload]
    exp = expected(workload)
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        getattr(g, k).requires_grad_(True)
    bg = torch.full((3,), 0.5, device=DEV)
    prev = fused.SEGMENTS
    fused.SEGMENTS = segments
    try:
        img, mask, uv, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows,
                                             return_aux=True, **DEFAULTS)
    finally:
        fused.SEGMENTS = prev
    for k in ("conic", "opacity", "rgb"):
        aux[k].retain_grad()
    uv.retain_grad()
    gi = make_grad_image(W, H, seed=1)
    img.backward(gi.to(DEV))
    grads = dict(uv=uv.grad, conic=aux["conic"].grad, opacity_act=aux["opacity"].grad, rgb_render=aux["rgb"].grad)
    rgb_gpu = aux["rgb"].detach().cpu().contiguous()
    rgb = exp["rgb"] if torch.equal(rgb_gpu, exp["rgb"]) else rgb_gpu
    ref64 = oracle_band(exp, rgb, W, H, rows, bg.cpu(), gi, sum_mode=0)
    ref_a = oracle_band(exp, rgb, W, H, rows, bg.cpu(), gi, sum_mode=1)
    ref_b = oracle_band(exp, rgb, W, H, rows, bg.cpu(), gi, sum_mode=2)
    rows_out = []
    for name, key, _ in RENDER_GRADS:
        spread = max(rel_err(ref_a[key], ref64[key], 1e-6), rel_err(ref_b[key], ref64[key], 1e-6))
        between = rel_err(ref_a[key], ref_b[key], 1e-6)
        kernel = rel_err(grads[name], ref64[key], 1e-6)
        report(f"fp32_reorder_spread[D rows 26-28, {'depth-segmented' if segments else 'unsegmented'}]", tensor=name,
               kernel_vs_double_floor_1e6=kernel,
               fp32_order_vs_double_floor_1e6=spread, fp32_order_a_vs_b_floor_1e6=between,
               kernel_floor_1e2=rel_err(grads[name], ref64[key], 1e-2),
               fp32_order_floor_1e2=max(rel_err(ref_a[key], ref64[key], 1e-2), rel_err(ref_b[key], ref64[key], 1e-2)))
        rows_out.append((name, kernel, spread))
        # the pure order spread alone is already above the 1e-4 target at this floor
        assert spread > 1e-4, (name, spread)
    factor = REORDER_FACTOR_SEGMENTED if segments else REORDER_FACTOR
    for name, kernel, spread in rows_out:
        assert kernel <= factor[name] * spread, (name, kernel, spread)
    if not segments:
        assert max(k for _, k, _ in rows_out) <= REORDER_FACTOR_ALL_TENSORS * max(sp for _, _, sp in rows_out), rows_out
