"""Per-frame pipeline (reference: splat_py/rasterize.py:18-112), same signature and return value.

This is the reference-shaped path: six autograd nodes and PyTorch glue between them, every
`splat_cuda` call a HIP kernel.  gaussian_splatting_amd.fused.rasterize is the fast path with the
same contract.
"""
import torch

from .cuda_autograd_functions import (
    CameraPointProjection,
    ComputeConic,
    ComputeProjectionJacobian,
    ComputeSigmaWorld,
    PrecomputeRGBFromSH,
    RenderImage,
)
from .structs import Gaussians, Tiles
from .tile_culling import get_splats
from .utils import compute_rays_in_world_frame, transform_points_torch


def frustum_culling_mask(xyz_camera_frame, uv, camera, near_thresh, far_thresh, cull_mask_padding):
    """True = culled (rasterize.py:33-49)."""
    z = xyz_camera_frame[:, 2]
    mask = (z < near_thresh) | (z > far_thresh)
    mask = mask | (uv[:, 0] < -1 * cull_mask_padding) | (uv[:, 0] > camera.width + cull_mask_padding)
    mask = mask | (uv[:, 1] < -1 * cull_mask_padding) | (uv[:, 1] > camera.height + cull_mask_padding)
    return mask


def rasterize(gaussians, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding, mh_dist,
              use_sh_precompute, background_rgb, tile_rows=None, grad_sync=None):
    """-> (image[H,W,3], culling_mask bool[N], uv[V,2]); uv is the post-cull autograd intermediate
    whose .grad (if retained) is the render-backward grad_uv (trainer.py:360,379).

    tile_rows / grad_sync are the multi-GPU hooks of gaussian_splatting_amd.sharded (not part of the
    reference signature): render only the tile rows [row0, row1) and pass the render inputs
    through grad_sync so that their gradients can be summed across ranks."""
    xyz_camera_frame = transform_points_torch(gaussians.xyz, camera_T_world)
    uv = CameraPointProjection.apply(xyz_camera_frame, camera.K)
    culling_mask = frustum_culling_mask(xyz_camera_frame, uv, camera, near_thresh, far_thresh, cull_mask_padding)
    keep = ~culling_mask

    uv = uv[keep, :]
    xyz_camera_frame = xyz_camera_frame[keep, :]
    culled = Gaussians(
        xyz=gaussians.xyz[keep, :],
        quaternion=gaussians.quaternion[keep, :],
        scale=gaussians.scale[keep, :],
        opacity=torch.sigmoid(gaussians.opacity[keep]),
        rgb=gaussians.rgb[keep, :],
        sh=gaussians.sh[keep, :] if gaussians.sh is not None else None,
    )

    sigma_world = ComputeSigmaWorld.apply(culled.quaternion, culled.scale)
    J = ComputeProjectionJacobian.apply(xyz_camera_frame, camera.K)
    conic = ComputeConic.apply(sigma_world, J, camera_T_world)

    tiles = Tiles(camera.height, camera.width, uv.device)
    sorted_gaussian_idx_by_splat_idx, splat_start_end_idx_by_tile_idx = get_splats(
        uv, tiles, conic, xyz_camera_frame, mh_dist)

    rays = torch.zeros(1, 1, 1, dtype=gaussians.xyz.dtype, device=gaussians.xyz.device)
    if culled.sh is not None:
        sh_coeffs = torch.cat((culled.rgb.unsqueeze(dim=2), culled.sh), dim=2)
        if use_sh_precompute:
            # the third argument is world_T_camera: the kernel reads its translation column as the
            # camera centre (Q8)
            render_rgb = PrecomputeRGBFromSH.apply(sh_coeffs, culled.xyz, torch.inverse(camera_T_world).contiguous())
        else:
            render_rgb = sh_coeffs
            rays = compute_rays_in_world_frame(camera, camera_T_world)
    else:
        render_rgb = culled.rgb

    r_rgb, r_opacity, r_uv, r_conic = (render_rgb, culled.opacity, uv, conic) if grad_sync is None else grad_sync(
        render_rgb, culled.opacity, uv, conic)
    image = RenderImage.apply(
        r_rgb, r_opacity, r_uv, r_conic, rays, splat_start_end_idx_by_tile_idx,
        sorted_gaussian_idx_by_splat_idx, (camera.height, camera.width), background_rgb, tile_rows)
    return image, culling_mask, uv
