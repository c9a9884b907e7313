"""fp64 torch.autograd.gradcheck of the six autograd Functions over the CPU oracle -- the
reference's own gradient-test recipes (test/test_cuda_autograd_functions.py:68-236,
test/test_rasterize_autograd.py:73-341).  They validate the oracle's backward restatements
(fp64 / no alpha-skip / single chunk, exactly the regime the reference tests)."""
import pytest
import torch

from gaussian_splatting_amd.splat_py.cuda_autograd_functions import (
    CameraPointProjection, ComputeConic, ComputeProjectionJacobian, ComputeSigmaWorld, PrecomputeRGBFromSH,
    RenderImage)
from gaussian_splatting_amd.splat_py.structs import Camera, Tiles
from gaussian_splatting_amd.splat_py.tile_culling import get_splats
from gaussian_splatting_amd.splat_py.utils import compute_rays_in_world_frame

D = torch.float64


def per_gaussian_inputs(device="cpu"):
    xyz = torch.tensor([[1.0, 2.0, 15.0], [2.5, -1.0, 4.0], [-1.0, -2.0, 10.0]], dtype=D, device=device,
                       requires_grad=True)
    K = torch.tensor([[430.0, 0.0, 320.0], [0.0, 410.0, 240.0], [0.0, 0.0, 1.0]], dtype=D, device=device)
    q = torch.tensor([[0.8, 0.2, 0.2, 0.2], [0.714, -0.002, -0.664, 0.221], [0.0, 0.0, 1.0, 0.0]], dtype=D,
                     device=device, requires_grad=True)
    s = torch.tensor([[0.02, 0.03, 0.04], [0.09, 0.03, 0.01], [2.0, 1.0, 0.1]], dtype=D, device=device,
                     requires_grad=True)
    T = torch.tensor([[0.9999, 0.0089, 0.0073, -0.3283], [-0.0106, 0.9568, 0.2905, -1.9260],
                      [-0.0044, -0.2906, 0.9568, 2.9581], [0.0, 0.0, 0.0, 1.0]], dtype=D, device=device)
    return xyz, K, q, s, T


def run_per_gaussian_gradchecks(device="cpu"):
    xyz, K, q, s, T = per_gaussian_inputs(device)
    assert torch.autograd.gradcheck(CameraPointProjection.apply, (xyz, K), raise_exception=True)
    assert torch.autograd.gradcheck(ComputeProjectionJacobian.apply, (xyz, K), raise_exception=True)
    assert torch.autograd.gradcheck(ComputeSigmaWorld.apply, (q, s), raise_exception=True)
    g = torch.Generator().manual_seed(0)
    sigma = torch.rand(1, 3, 3, dtype=D, generator=g).to(device).requires_grad_(True)
    J = torch.rand(1, 2, 3, dtype=D, generator=g).to(device).requires_grad_(True)
    assert torch.autograd.gradcheck(ComputeConic.apply, (sigma, J, T), raise_exception=True)


def run_sh_gradcheck(n_sh, device="cpu"):
    g = torch.Generator().manual_seed(n_sh)
    N = 100
    sh = torch.ones(N, 3, n_sh, dtype=D, device=device, requires_grad=True)
    xyz = torch.rand(N, 3, dtype=D, generator=g).to(device)
    M = torch.zeros(4, 4, dtype=D, device=device)
    assert torch.autograd.gradcheck(PrecomputeRGBFromSH.apply, (sh, xyz, M), raise_exception=True)


def render_inputs(device="cpu"):
    """test/test_rasterize_autograd.py:15-71"""
    K = torch.tensor([[43.0, 0.0, 30.0], [0.0, 41.0, 20.0], [0.0, 0.0, 1.0]], dtype=D, device=device)
    camera = Camera(60, 40, K)
    T = torch.eye(4, dtype=D, device=device)
    rays = compute_rays_in_world_frame(camera, T)
    xyz_c = torch.tensor([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]], dtype=D, device=device)
    uv = torch.tensor([[32.8523, 24.88553], [25.0, 25.0], [45.339926, 13.85983]], dtype=D, device=device,
                      requires_grad=True)
    conic = torch.tensor([[1.3287e03, 9.7362e02 * 2, 7.3605e02], [90.0, 20.0 * 2, 60.0],
                          [776.215, -2464.463 * 2, 8276.755]], dtype=D, device=device, requires_grad=True)
    tiles = Tiles(40, 60, device)
    sorted_idx, ranges = get_splats(uv.detach().float(), tiles, conic.detach().float(), xyz_c.float(), 3.0)
    opacity = torch.ones(3, 1, dtype=D, device=device, requires_grad=True)
    return rays, uv, conic, sorted_idx, ranges, opacity


def run_render_gradcheck(n_sh, bgval, device="cpu"):
    rays, uv, conic, sorted_idx, ranges, opacity = render_inputs(device)
    if n_sh == 1:
        rgb = torch.ones(3, 3, dtype=D, device=device) * 0.5
        rgb[0, 0] = 0.0
        rgb[1, 1] = 0.0
        rgb.requires_grad_(True)
    else:
        rgb = (torch.ones(3, 3, n_sh, dtype=D, device=device) * 0.5).requires_grad_(True)
    bg = torch.ones(3, dtype=D, device=device) * bgval
    size = torch.tensor([40, 60], dtype=torch.int)
    kw = dict(atol=3e-5) if n_sh == 16 else {}
    assert torch.autograd.gradcheck(
        RenderImage.apply, (rgb, opacity, uv, conic, rays, ranges, sorted_idx, size, bg), raise_exception=True, **kw)


def test_per_gaussian_gradchecks(oracle_backend):
    run_per_gaussian_gradchecks()


@pytest.mark.parametrize("n_sh", [1, 4, 9, 16])
def test_sh_precompute_gradcheck(oracle_backend, n_sh):
    run_sh_gradcheck(n_sh)


@pytest.mark.parametrize("n_sh", [1, 4, 9, 16])
@pytest.mark.parametrize("bgval", [0.5, 0.0])
def test_render_image_gradcheck(oracle_backend, n_sh, bgval):
    run_render_gradcheck(n_sh, bgval)
