"""Hot-path helpers (reference: splat_py/utils.py:7-16,60-72,75-123)."""
import torch


def inverse_sigmoid_torch(x):
    clipped = torch.clip(x, 1e-4, 1 - 1e-4)
    return torch.log(clipped / (1.0 - clipped))


def transform_points_torch(pts, transform):
    """xyz' = (T [xyz, 1])[:3]  (utils.py:60-72).  Evaluated as the same homogeneous batched
    matmul as the reference so that results agree bitwise with it on the same device."""
    ones = torch.ones(pts.shape[0], 1, dtype=pts.dtype, device=pts.device)
    hom = torch.cat([pts, ones], dim=1)
    out = torch.matmul(transform, hom.unsqueeze(-1)).squeeze(-1)[:, :3]
    return out.contiguous()


def compute_rays(camera):
    """Unit ray per pixel in the camera frame, row-major (v, u) order (utils.py:75-109)."""
    K = camera.K
    u = torch.linspace(0, camera.width - 1, camera.width, dtype=K.dtype, device=K.device)
    v = torch.linspace(0, camera.height - 1, camera.height, dtype=K.dtype, device=K.device)
    v, u = torch.meshgrid(v, u, indexing="ij")
    v = v.flatten()
    u = u.flatten()
    ray_dir = torch.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], torch.ones_like(u)], dim=-1)
    return ray_dir / torch.norm(ray_dir, dim=1, keepdim=True)


def compute_rays_in_world_frame(camera, camera_T_world):
    """Unit ray per pixel in the world frame, [H, W, 3] (utils.py:112-123)."""
    rays = compute_rays(camera)
    world_T_camera = torch.inverse(camera_T_world)
    rays = (world_T_camera[:3, :3] @ rays.T).T
    rays = rays / torch.norm(rays, dim=1, keepdim=True)
    return rays.reshape(camera.height, camera.width, 3).contiguous()
