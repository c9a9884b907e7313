"""Randomised parity sweep: odd image sizes, degenerate Gaussians (huge / tiny scales, saturated and
vanishing opacities, coincident centres, depth ties), all SH degrees -- the fused HIP path against
the CPU oracle on the kernel's own inputs: tile lists, num_splats, image bit-exact; gradients 1e-4."""
import numpy as np
import pytest
import torch

from gaussian_splatting_amd import fused
from gaussian_splatting_amd.splat_py.structs import Camera, Gaussians
from gaussian_splatting_amd.synthetic import make_grad_image, make_scene

from .helpers import rel_err, scaled_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def oracle():
    from oracle import gs_oracle
    gs_oracle.set_modes(0, 0)
    gs_oracle.set_sh_band1_mode(0)
    return gs_oracle


def degenerate_scene(rng, N, W, H, deg):
    g, cam, T = make_scene(N, W, H, deg, seed=int(rng.integers(1 << 30)))
    n = N // 10
    idx = torch.from_numpy(rng.permutation(N))
    g.scale[idx[:n]] += 3.0                         # huge: cover many tiles
    g.scale[idx[n:2 * n]] -= 4.0                    # sub-pixel
    g.opacity[idx[2 * n:3 * n]] = 20.0              # alpha -> 0.9999 clamp in backward
    g.opacity[idx[3 * n:4 * n]] = -8.0              # below 1/255 everywhere
    g.xyz[idx[4 * n:5 * n]] = g.xyz[idx[4 * n]].clone()     # coincident centres and depth ties
    g.scale[idx[5 * n:6 * n], 0] += 2.5             # needles
    g.xyz[idx[6 * n:7 * n], 2] = 0.31               # just beyond the near plane
    return g, cam, T


@pytest.mark.parametrize("case", range(10))
def test_fuzz_fused_vs_oracle(case):
    orc = oracle()
    rng = np.random.default_rng(1000 + case)
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 220))
    N = int(rng.integers(1, 2500))
    deg = int(rng.integers(0, 4))
    g, cam, T = degenerate_scene(rng, N, W, H, deg)
    bgv = float(rng.choice([0.0, 0.5, 1.0]))
    bg = torch.full((3,), bgv)
    gd = Gaussians(*(t.to(DEV) if t is not None else None
                     for t in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh)))
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        if getattr(gd, k) is not None:
            getattr(gd, k).requires_grad_(True)
    camd = Camera(W, H, cam.K.to(DEV))
    img, mask, uv, aux = fused.rasterize(gd, T.to(DEV), camd, 0.3, 500.0, 50, 3.0, True, bg.to(DEV), return_aux=True)
    for k in ("conic", "opacity", "rgb"):
        aux[k].retain_grad()
    uv.retain_grad()
    gi = make_grad_image(W, H, seed=case)
    img.backward(gi.to(DEV))
    assert torch.isfinite(img).all()

    c = lambda t: t.detach().cpu().contiguous()
    uvc, conic, opa, rgb, xyz_c = c(uv), c(aux["conic"]), c(aux["opacity"]), c(aux["rgb"]), c(aux["xyz_camera_frame"])
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    ref_sorted, ref_ranges = orc.get_sorted_gaussian_list(1024, uvc, xyz_c, conic, ntx, nty, 3.0)
    assert torch.equal(c(aux["tile_ranges"]), ref_ranges)
    assert torch.equal(c(aux["sorted_gaussians"]), ref_sorted)
    ref_img = torch.zeros(H, W, 3)
    nsp = torch.zeros(H, W, dtype=torch.int32)
    fw = torch.zeros(H, W)
    orc.render_tiles_cuda(uvc, opa, rgb, conic, torch.zeros(1, 1, 1), ref_ranges, ref_sorted, bg, nsp, fw, ref_img)
    assert torch.equal(img.detach().cpu(), ref_img)
    V = uvc.shape[0]
    gr = [torch.zeros(V, 3), torch.zeros(V, 1), torch.zeros(V, 2), torch.zeros(V, 3)]
    orc.render_tiles_backward_cuda(uvc, opa, rgb, conic, torch.zeros(1, 1, 1), ref_ranges, ref_sorted, bg, nsp, fw, gi,
                                   *gr)
    for name, ref in (("rgb", gr[0]), ("opacity", gr[1]), ("conic", gr[3])):
        got = aux[name].grad.cpu()
        if ref.abs().max() > 0:
            assert scaled_err(got, ref) < 1e-5 and rel_err(got, ref) < 1e-4, name
    if gr[2].abs().max() > 0:
        assert scaled_err(uv.grad.cpu(), gr[2]) < 1e-5
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion"):
        assert torch.isfinite(getattr(gd, k).grad).all(), k


@pytest.mark.parametrize("script,args", [("stress_fused.py", ["--n", "16", "--seed", "5"]),
                                         ("stress_sharded.py", ["--n", "6", "--seed", "5"])])
def test_randomised_stress_scripts(script, args):
    """a short run of the randomised checks under scripts/ (prefix-sort vs full-sort frames; owner-mode
    frames with simulated ranks) so that every GPU test run covers fresh shapes of both"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", script)] + args, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("ok"), out.stdout[-2000:] + out.stderr[-2000:]
