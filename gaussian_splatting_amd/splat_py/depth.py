"""Depth rendering (reference: splat_py/depth.py:17-88): distance of the first Gaussian at which
the accumulated alpha passes `alpha_threshold`; -1 where it never does.  No gradient."""
import torch

from .. import backend
from .cuda_autograd_functions import (
    CameraPointProjection,
    ComputeConic,
    ComputeProjectionJacobian,
    ComputeSigmaWorld,
)
from .structs import Tiles
from .tile_culling import get_splats
from .utils import transform_points_torch


def render_depth(gaussians, alpha_threshold, camera_T_world, camera, near_thresh, cull_mask_padding, mh_dist):
    with torch.no_grad():
        xyz_camera_frame = transform_points_torch(gaussians.xyz, camera_T_world)
        uv = CameraPointProjection.apply(xyz_camera_frame, camera.K)
        mask = xyz_camera_frame[:, 2] < near_thresh
        mask = mask | (uv[:, 0] < -1 * cull_mask_padding) | (uv[:, 0] > camera.width + cull_mask_padding)
        mask = mask | (uv[:, 1] < -1 * cull_mask_padding) | (uv[:, 1] > camera.height + cull_mask_padding)
        keep = ~mask
        uv = uv[keep, :]
        xyz_camera_frame = xyz_camera_frame[keep, :]
        opacity = torch.sigmoid(gaussians.opacity[keep])
        sigma_world = ComputeSigmaWorld.apply(gaussians.quaternion[keep, :], gaussians.scale[keep, :])
        J = ComputeProjectionJacobian.apply(xyz_camera_frame, camera.K)
        conic = ComputeConic.apply(sigma_world, J, camera_T_world)
        tiles = Tiles(camera.height, camera.width, uv.device)
        sorted_idx, tile_ranges = get_splats(uv, tiles, conic, xyz_camera_frame, mh_dist)
        depth_image = torch.full((camera.height, camera.width, 1), -1.0, dtype=torch.float32, device=uv.device)
        backend.get().render_depth_cuda(
            xyz_camera_frame, uv, opacity, conic, tile_ranges, sorted_idx, alpha_threshold, depth_image)
        return depth_image
