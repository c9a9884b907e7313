"""Hot-path helpers (reference: splat_py/utils.py:7-16,60-72,75-123)."""
import torch


def inverse_sigmoid_torch(x):
    clipped = torch.clip(x, 1e-4, 1 - 1e-4)
    return torch.log(clipped / (1.0 - clipped))


def inverse_sigmoid(x):
    """numpy form of the above (utils.py:6-11): logit of x clipped to [1e-4, 1 - 1e-4]"""
    import numpy as np
    c = np.clip(x, 1e-4, 1 - 1e-4)
    return np.log(c / (1.0 - c))


def quaternion_to_rotation_torch(q):
    """[N, 4] normalised quaternions (w, x, y, z) -> rotation matrices [N, 3, 3] (utils.py:40-57; the
    device kernels use the same convention, pg_math.h quat_to_rot)"""
    w, x, y, z = q.unbind(dim=1)
    xx, yy, zz = x * x, y * y, z * z
    xy, xz, yz, wx, wy, wz = x * y, x * z, y * z, w * x, w * y, w * z
    rows = (1 - 2 * yy - 2 * zz, 2 * xy - 2 * wz, 2 * xz + 2 * wy,
            2 * xy + 2 * wz, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * yz + 2 * wx, 1 - 2 * xx - 2 * yy)
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def compute_initial_scale_from_sparse_points(points, num_neighbors, neighbor_dist_to_scale_factor,
                                             max_initial_scale):
    """log-scale initialisation from the mean distance to the nearest points (utils.py:19-37; the
    k nearest INCLUDE the point itself at distance 0, as in the reference) -> [N, 3] float32.
    One batched KD-tree query instead of one query per point."""
    import numpy as np
    from scipy.spatial import KDTree
    pts = points.detach().cpu().numpy()
    dist, _ = KDTree(pts).query(pts, k=num_neighbors, workers=-1)
    dist = np.asarray(dist, dtype=np.float64).reshape(pts.shape[0], -1)
    initial = np.minimum(dist.mean(axis=1), max_initial_scale)
    scale = np.log(initial * neighbor_dist_to_scale_factor).astype(np.float32)
    return torch.from_numpy(scale).unsqueeze(1).repeat(1, 3)


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
