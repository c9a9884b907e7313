"""What SURVEY.md 8(d)'s gradient criterion can resolve.

The criterion: max |g - g_ref| / max(|g_ref|, 1e-6 max|g_ref|) <= 1e-4 per gradient tensor.  A per-Gaussian render
gradient is a sum of fp32 per-pixel terms; every GPU implementation (the reference's warp reduce + atomicAdd as
much as this build's wave reduction + atomics) adds them in fp32 in SOME order, while the oracle adds them in
double.  Here the oracle adds its own per-pixel terms -- bit-identical terms -- in fp32 in two different fixed
orders (oracle.gs_oracle.set_backward_sum 1 / 2) and the criterion is evaluated between the two results and
against the double sum: that spread is what the criterion reads for an implementation WITHOUT any error of its
own.  It exceeds 1e-4 by an order of magnitude at the 1e-6 floor (elements whose terms cancel), and drops
below 1e-4 with the 1 % floor the assertions in tests/ use; the GPU counterpart
(tests/test_gpu_fullsize_parity.py::test_gradient_error_is_within_the_fp32_reorder_spread) asserts that the HIP kernel's
figure stays within 2x of this spread."""
import torch

from gaussian_splatting_amd.synthetic import make_grad_image

from .helpers import rel_err


def projected_scene(V, W, H, seed):
    """per-splat render inputs (uv, opacity, rgb, conic) + exact tile lists of a dense little frame"""
    from oracle import gs_oracle as orc
    gen = torch.Generator().manual_seed(seed)
    uv = (torch.rand(V, 2, generator=gen) * torch.tensor([W + 20.0, H + 20.0]) - 10.0).contiguous()
    sig = 1.0 + 5.0 * torch.rand(V, 2, generator=gen)
    rho = (torch.rand(V, generator=gen) * 1.6 - 0.8)
    conic = torch.stack([sig[:, 0] ** 2, 2 * rho * sig[:, 0] * sig[:, 1], sig[:, 1] ** 2], dim=1).contiguous()
    z = (1.5 + 28.5 * torch.rand(V, 1, generator=gen))
    xyz_c = torch.cat([torch.zeros(V, 2), z], dim=1).contiguous()
    opacity = torch.sigmoid(2 * torch.randn(V, 1, generator=gen)).contiguous()
    rgb = torch.rand(V, 3, generator=gen).contiguous()
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    sorted_g, ranges = orc.get_sorted_gaussian_list(1024, uv, xyz_c, conic, ntx, nty, 3.0)
    return uv, opacity, rgb, conic, sorted_g, ranges


def backward_in_mode(mode, scene, W, H, gi, bg):
    from oracle import gs_oracle as orc
    uv, opacity, rgb, conic, sorted_g, ranges = scene
    V = uv.shape[0]
    img = torch.zeros(H, W, 3)
    nsp = torch.zeros(H, W, dtype=torch.int32)
    fw = torch.zeros(H, W)
    rays = torch.zeros(1, 1, 1)
    orc.render_tiles_cuda(uv, opacity, rgb, conic, rays, ranges, sorted_g, bg, nsp, fw, img)
    g = [torch.zeros(V, 3), torch.zeros(V, 1), torch.zeros(V, 2), torch.zeros(V, 3)]
    orc.set_backward_sum(mode)
    try:
        orc.render_tiles_backward_cuda(uv, opacity, rgb, conic, rays, ranges, sorted_g, bg, nsp, fw, gi, *g)
    finally:
        orc.set_backward_sum(0)
    return g


def reorder_spread(scene, W, H, gi, bg):
    """-> ({floor: worst criterion value between / against the fp32 orders}, the double-sum gradients)"""
    g64 = backward_in_mode(0, scene, W, H, gi, bg)
    ga = backward_in_mode(1, scene, W, H, gi, bg)
    gb = backward_in_mode(2, scene, W, H, gi, bg)
    out = {}
    for floor in (1e-6, 1e-2):
        out[floor] = dict(
            between_orders=max(rel_err(a, b, floor) for a, b in zip(ga, gb)),
            vs_double=max(max(rel_err(a, r, floor), rel_err(b, r, floor)) for a, b, r in zip(ga, gb, g64)))
    return out, g64


def test_two_fp32_summation_orders_of_the_same_terms_exceed_the_1e6_floor_criterion():
    from oracle import gs_oracle as orc
    orc.set_modes(0, 0)
    W, H, V = 160, 96, 6000
    scene = projected_scene(V, W, H, seed=3)
    counts = scene[5][1:] - scene[5][:-1]
    assert int(counts.max()) > 300   # dense: hundreds of splats per tile, many pixels per Gaussian
    gi = make_grad_image(W, H, seed=1)
    spread, g64 = reorder_spread(scene, W, H, gi, torch.full((3,), 0.5))
    # deterministic: the same order twice gives the same bits
    again = backward_in_mode(1, scene, W, H, gi, torch.full((3,), 0.5))
    first = backward_in_mode(1, scene, W, H, gi, torch.full((3,), 0.5))
    assert all(torch.equal(a, b) for a, b in zip(again, first))
    # the criterion as SURVEY.md 8(d) writes it (floor 1e-6) reads MORE than its own 1e-4 target between two
    # exact-term fp32 sums: it measures summation order, not kernel error ...
    assert spread[1e-6]["between_orders"] > 1e-4, spread
    assert spread[1e-6]["vs_double"] > 1e-4, spread
    # ... while with the 1 % floor the same two sums agree to 1e-4 (what tests/ and bench.py assert)
    assert spread[1e-2]["between_orders"] < 1e-4, spread
    assert spread[1e-2]["vs_double"] < 1e-4, spread
