"""Seeded synthetic scenes for parity tests and bench.py (recipe: SURVEY.md 8(d)).

Camera at the origin looking down +z, fx = fy = 0.9 W, principal point at the image centre;
z ~ U(1.5, 30); x, y = z * U(-1, 1) * 0.6 * {W, H} / f (about 1/6 of the centres per axis fall
outside the image, so the frustum cull is exercised); quaternion ~ N(0,1)^4;
scale = log(z * U(0.5, 6) / fx) per axis (0.5-6 px sigma); opacity logits ~ N(0, 2);
rgb ~ U(0,1) / SH_0; sh ~ N(0, 0.05) [N,3,15] for degree 3.
Generated on the CPU with a fixed torch.Generator so CPU and GPU runs see identical inputs.
"""
import torch

from .splat_py.structs import Camera, Gaussians

SH_0 = 0.28209479177387814

WORKLOADS = {
    # name: (N, W, H, sh_degree)
    "A": (1_000, 256, 256, 0),
    "B": (100_000, 1920, 1080, 3),
    "C": (1_500_000, 1297, 840, 3),
    "D": (2_860_000, 1297, 840, 3),
}

DEFAULTS = dict(near_thresh=0.3, far_thresh=500.0, cull_mask_padding=100, mh_dist=3.0)


def make_scene(N, W, H, sh_degree=0, seed=0, device="cpu", dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)

    def u(*shape):
        return torch.rand(*shape, generator=g, dtype=torch.float64)

    def n(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float64)

    fx = fy = 0.9 * W
    z = 1.5 + 28.5 * u(N)
    x = z * (2 * u(N) - 1) * 0.6 * W / fx
    y = z * (2 * u(N) - 1) * 0.6 * H / fy
    xyz = torch.stack([x, y, z], dim=1)
    quaternion = n(N, 4)
    scale = torch.log(z[:, None] * (0.5 + 5.5 * u(N, 3)) / fx)
    opacity = 2.0 * n(N, 1)
    rgb = u(N, 3) / SH_0
    sh = None
    if sh_degree > 0:
        n_extra = (sh_degree + 1) ** 2 - 1
        sh = 0.05 * n(N, 3, n_extra)
    K = torch.tensor([[fx, 0.0, W / 2.0], [0.0, fy, H / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    camera_T_world = torch.eye(4, dtype=torch.float64)

    def cvt(t):
        return None if t is None else t.to(dtype).to(device).contiguous()

    gaussians = Gaussians(cvt(xyz), cvt(rgb), cvt(opacity), cvt(scale), cvt(quaternion), cvt(sh))
    camera = Camera(W, H, cvt(K))
    return gaussians, camera, cvt(camera_T_world)


def make_grad_image(W, H, seed=1, device="cpu", dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    gi = torch.randn(H, W, 3, generator=g, dtype=torch.float64) / (H * W)
    return gi.to(dtype).to(device).contiguous()
