"""Whole-frame parity at BASELINE.json's full sizes (B, C, D) on the kernels the bench times.

tests/test_gpu_fullsize_parity.py compares the per-Gaussian stage and all tile lists of the whole scene with the
oracle but renders only 2-3 tile rows of the frame on the CPU -- and a band of fewer than 1500 tiles takes the
depth-segmented backward, not the kernel `bench.py` times on the full frame.  Here the oracle renders EVERY tile row
of the frame forward and backward (render.cu:8-189, render_backward.cu:12-285; ~1.2 s on the GPU box's cores at D)
from its own per-splat values and its own lists, and chains the per-Gaussian backward (projection_backward.cu:9-471,
precompute_sh.cu:61-111) at full size, against the fused HIP frame WITHOUT tile_rows (4 346 tiles at D: the
unsegmented k_render_bwd, the kernel the headline times):

  * image and num_splats_per_pixel of the whole frame: torch.equal
  * the four render gradients over all rows: tests/test_gpu_scale.py::check_band_backward's three criteria
  * the dense xyz / quaternion / scale / opacity / rgb / sh gradients (k_preprocess_bwd) at full N against the
    oracle's chained per-stage backward kernels, fed the same render gradients
  * the product's default path (native orchestration, prefix sort, longest-first backward) gives the same image
    bit for bit and the same dense gradients up to summation order
One case runs a tilted, translated camera with another seed (the identity camera of synthetic.make_scene exercises
the world->camera transform and the camera centre of the SH colour trivially)."""
import numpy as np
import pytest
import torch

from gaussian_splatting_amd import fused, splat_cuda
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

from .helpers import rel_err, report, scaled_err
from .test_gpu_fullsize_parity import REORDER_FACTOR, REORDER_FACTOR_ALL_TENSORS
from .test_gpu_fused import cpu_expected_stages
from .test_gpu_scale import PARAMS, RENDER_GRADS, check_band_backward, oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"

TILT = torch.tensor([[0.9999, 0.0089, 0.0073, 0.05], [-0.0106, 0.9568, 0.2905, -0.1], [-0.0044, -0.2906, 0.9568, 0.3],
                     [0.0, 0.0, 0.0, 1.0]])


def oracle_frame(exp, rgb, W, H, bg, grad_image, sum_mode=0, with_abs=True):
    """the oracle's render forward + backward of ALL tile rows from its own per-splat values and lists"""
    orc = oracle()
    img = torch.zeros(H, W, 3)
    nsp = torch.zeros(H, W, dtype=torch.int32)
    fw = torch.zeros(H, W)
    rays = torch.zeros(1, 1, 1)
    args = (exp["uv"], exp["opacity"].reshape(-1, 1).contiguous(), rgb, exp["conic"], rays, exp["ranges"], exp["sorted"], bg)
    orc.render_tiles_cuda(*args, nsp, fw, img)
    V = exp["V"]
    shapes = ((V, 3), (V, 1), (V, 2), (V, 3))
    g = [torch.zeros(*s) for s in shapes]
    orc.set_backward_sum(sum_mode)
    try:
        orc.render_tiles_backward_cuda(*args, nsp, fw, grad_image, *g)
    finally:
        orc.set_backward_sum(0)
    out = dict(image=img, nsp=nsp, fw=fw, g_rgb=g[0], g_opa=g[1], g_uv=g[2], g_conic=g[3])
    if with_abs:
        a = [torch.zeros(*s) for s in shapes]
        orc.render_tiles_backward_abs(*args, nsp, fw, grad_image, *a)
        out.update(a_rgb=a[0], a_opa=a[1], a_uv=a[2], a_conic=a[3])
    return out


def oracle_chain(g, cam, T, exp, g_uv, g_conic, g_opa, g_rgb):
    """dense parameter gradients from the render gradients: the oracle's per-stage backward kernels chained as the
    reference's autograd graph chains them (cuda_autograd_functions.py:19-219 + the glue of rasterize.py:29-99)"""
    orc = oracle()
    N = g.xyz.shape[0]
    V = exp["V"]
    keep = ~exp["culled"]
    q, s = g.quaternion[keep].contiguous(), g.scale[keep].contiguous()
    sigma = torch.zeros(V, 3, 3)
    orc.compute_sigma_world_cuda(q, s, sigma)
    J = torch.zeros(V, 2, 3)
    orc.compute_projection_jacobian_cuda(exp["xyz_c"], cam.K, J)
    g_sigma, g_J = torch.zeros(V, 3, 3), torch.zeros(V, 2, 3)
    orc.compute_conic_backward_cuda(sigma, J, T, g_conic, g_sigma, g_J)
    g_q, g_s = torch.zeros(V, 4), torch.zeros(V, 3)
    orc.compute_sigma_world_backward_cuda(q, s, g_sigma, g_q, g_s)
    gx1, gx2 = torch.zeros(V, 3), torch.zeros(V, 3)
    orc.compute_projection_jacobian_backward_cuda(exp["xyz_c"], cam.K, g_J, gx1)
    orc.camera_projection_backward_cuda(exp["xyz_c"], cam.K, g_uv, gx2)
    g_xyz_v = (gx1 + gx2) @ T[:3, :3]   # rows: R^T g
    A = T[:3, :3].double().numpy()
    center = torch.from_numpy((-np.linalg.inv(A) @ T[:3, 3].double().numpy()).astype(np.float32))
    Minv = torch.eye(4)
    Minv[:3, 3] = center
    n_coeff = 1 if g.sh is None else g.sh.shape[2] + 1
    g_coeff = torch.zeros(V, 3, n_coeff)
    orc.precompute_rgb_from_sh_backward_cuda(g.xyz[keep].contiguous(), Minv, g_rgb, g_coeff)
    y = exp["opacity"].reshape(-1, 1)
    g_logit = g_opa * (1 - y) * y

    def dense(v, shape):
        out = torch.zeros(shape)
        out[keep] = v
        return out

    expect = dict(xyz=dense(g_xyz_v, (N, 3)), quaternion=dense(g_q, (N, 4)), scale=dense(g_s, (N, 3)),
                  opacity=dense(g_logit, (N, 1)), rgb=dense(g_coeff[:, :, 0], (N, 3)))
    if g.sh is not None:
        expect["sh"] = dense(g_coeff[:, :, 1:], (N, 3, n_coeff - 1))
    return expect


def gpu_frame(workload, seed, T, aux):
    N, W, H, deg = WORKLOADS[workload]
    g, cam, _ = make_scene(N, W, H, deg, seed=seed, device=DEV)
    for k in PARAMS:
        getattr(g, k).requires_grad_(True)
    bg = torch.full((3,), 0.5, device=DEV)
    out = fused.rasterize(g, T.to(DEV), cam, use_sh_precompute=True, background_rgb=bg, return_aux=aux, **DEFAULTS)
    return g, cam, out


@pytest.mark.parametrize("workload,seed,tilt,reorder", [("B", 0, False, False), ("C", 0, False, False),
                                                        ("D", 0, False, True), ("D", 1, True, False)])
def test_whole_frame_forward_and_backward_equal_the_oracle(workload, seed, tilt, reorder):
    N, W, H, deg = WORKLOADS[workload]
    tag = f"whole_frame[{workload} seed {seed}{' tilted' if tilt else ''}]"
    g, cam, T = make_scene(N, W, H, deg, seed=seed)
    T = TILT.clone() if tilt else T
    d = DEFAULTS
    exp = cpu_expected_stages(g, cam, T, d["near_thresh"], d["far_thresh"], d["cull_mask_padding"], d["mh_dist"])
    V, S = exp["V"], int(exp["sorted"].numel())
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    assert fused.want_segments(S, ntx * nty) is False   # the full frame: the unsegmented kernel the bench times

    # ---- the fused frame with its intermediates (Python orchestration, complete lists) ----------------------
    gd, camd, (image, mask, uv, aux) = gpu_frame(workload, seed, T, True)
    for k in ("conic", "opacity", "rgb"):
        aux[k].retain_grad()
    uv.retain_grad()
    gi = make_grad_image(W, H, seed=1)
    image.backward(gi.to(DEV))
    assert torch.equal(mask.cpu(), exp["culled"])
    assert torch.equal(uv.detach().cpu(), exp["uv"])
    assert torch.equal(aux["conic"].detach().cpu(), exp["conic"])
    assert torch.equal(aux["tile_ranges"].cpu(), exp["ranges"])
    assert torch.equal(aux["sorted_gaussians"].cpu(), exp["sorted"])
    rgb_gpu = aux["rgb"].detach().cpu().contiguous()
    assert (rgb_gpu - exp["rgb"]).abs().max() < 2e-6
    # (the SH colour hangs on the camera centre, -R^-1 t: the kernel's fp32 expression and the checker's double
    # inverse can differ in the last ulp for a general pose; the render check then uses the GPU's colours)
    rgb = exp["rgb"] if torch.equal(rgb_gpu, exp["rgb"]) else rgb_gpu
    bg = torch.full((3,), 0.5)

    # ---- render: every tile row on the oracle, from the oracle's values and lists --------------------------------
    ref = oracle_frame(exp, rgb, W, H, bg, gi)
    assert torch.equal(image.detach().cpu(), ref["image"])
    # num_splats_per_pixel / final_weight through the reference-signature entry point on the same inputs
    nsp = torch.zeros(H, W, dtype=torch.int32, device=DEV)
    fw = torch.zeros(H, W, device=DEV)
    img2 = torch.zeros(H, W, 3, device=DEV)
    splat_cuda.render_tiles_cuda(uv.detach(), aux["opacity"].detach(), aux["rgb"].detach(), aux["conic"].detach(),
                                 torch.zeros(1, 1, 1, device=DEV), aux["tile_ranges"], aux["sorted_gaussians"],
                                 bg.to(DEV), nsp, fw, img2)
    assert torch.equal(nsp.cpu(), ref["nsp"]) and torch.equal(fw.cpu(), ref["fw"]) and torch.equal(img2.cpu(), ref["image"])
    grads = dict(uv=uv.grad, conic=aux["conic"].grad, opacity_act=aux["opacity"].grad, rgb_render=aux["rgb"].grad)
    report(tag, N=N, V=V, S=S, tiles=ntx * nty, sh_colour_bit_equal=float(rgb is exp["rgb"]))
    check_band_backward(tag + " render backward, all rows", grads, ref)

    # ---- per-Gaussian backward at full N: oracle chain fed the SAME render gradients -------------------------
    c = lambda t: t.detach().cpu().contiguous()
    expect = oracle_chain(g, cam, T, exp, c(uv.grad), c(aux["conic"].grad), c(aux["opacity"].grad), c(aux["rgb"].grad))
    dense = {}
    for k, e in expect.items():
        got = getattr(gd, k).grad.cpu()
        assert not got[exp["culled"]].any(), k
        dense[k] = scaled_err(got, e)
        report(tag + " dense gradients vs oracle chain", tensor=k, scaled=dense[k], rel_floor_1e2=rel_err(got, e, 1e-2))
        assert dense[k] < 2e-5, (k, dense[k])
    # ... and end to end: oracle chain fed the ORACLE's render gradients (double-summed) against the GPU's dense ones
    expect2 = oracle_chain(g, cam, T, exp, ref["g_uv"], ref["g_conic"], ref["g_opa"], ref["g_rgb"])
    for k, e in expect2.items():
        err = scaled_err(getattr(gd, k).grad.cpu(), e)
        report(tag + " dense gradients, oracle end to end", tensor=k, scaled=err)
        assert err < 2e-5, (k, err)

    # ---- the product's default path: native orchestration, prefix sort, longest-first backward -------------------
    g2, _, (image2, mask2, uv2) = gpu_frame(workload, seed, T, False)
    image2.backward(gi.to(DEV))
    assert torch.equal(image2.detach(), image.detach()) and torch.equal(mask2, mask) and torch.equal(uv2.detach(), uv.detach())
    for k in expect:
        err = scaled_err(getattr(g2, k).grad, getattr(gd, k).grad)
        report(tag + " default path vs python orchestration", tensor=k, scaled=err)
        assert err < 1e-5, (k, err)
        assert scaled_err(getattr(g2, k).grad.cpu(), expect2[k]) < 2e-5, k

    if reorder:
        # SURVEY 8(d)'s criterion (floor 1e-6) on the whole frame next to two fp32 summation orders of the oracle's
        # own terms: the kernel must stay within REORDER_FACTOR of pure order noise, per tensor
        ref_a = oracle_frame(exp, rgb, W, H, bg, gi, sum_mode=1, with_abs=False)
        ref_b = oracle_frame(exp, rgb, W, H, bg, gi, sum_mode=2, with_abs=False)
        worst_kernel = worst_spread = 0.0
        for name, key, _ in RENDER_GRADS:
            spread = max(rel_err(ref_a[key], ref[key], 1e-6), rel_err(ref_b[key], ref[key], 1e-6))
            kernel = rel_err(grads[name], ref[key], 1e-6)
            report(tag + " fp32 reorder spread, all rows", tensor=name, kernel_vs_double_floor_1e6=kernel,
                   fp32_order_vs_double_floor_1e6=spread, kernel_floor_1e2=rel_err(grads[name], ref[key], 1e-2))
            assert kernel <= REORDER_FACTOR[name] * spread, (name, kernel, spread)
            worst_kernel, worst_spread = max(worst_kernel, kernel), max(worst_spread, spread)
        # ... and over the four tensors together within 2x of the largest spread (the bench line's headline pair)
        assert worst_kernel <= REORDER_FACTOR_ALL_TENSORS * worst_spread, (worst_kernel, worst_spread)
