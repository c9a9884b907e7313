"""Tile binning entry point (reference: splat_py/tile_culling.py:8-27)."""
import torch

from .. import backend


def get_splats(uvs, tiles, conic, xyz_camera_frame, mh_dist):
    """-> (sorted_gaussian_idx_by_splat_idx int32[S], splat_start_end_idx_by_tile_idx int32[T+1]).

    Non-finite camera-frame coordinates would corrupt the depth order; the reference prints and
    exit()s (tile_culling.py:15-18), here the same guard raises."""
    if torch.any(~torch.isfinite(xyz_camera_frame)):
        raise FloatingPointError("xyz_camera_frame has NaN")
    return backend.get().get_sorted_gaussian_list(
        1024, uvs, xyz_camera_frame, conic, tiles.x_tiles_count, tiles.y_tiles_count, mh_dist)
