"""Data contracts of the hot path (reference: splat_py/structs.py:4,30-43,46-114,117-138)."""
import math

import torch

TILE_EDGE_LENGTH_PX = 16


class Camera:
    """Pinhole camera: image size and K[3,3] (structs.py:30-43)."""

    def __init__(self, width, height, K):
        self.width = width
        self.height = height
        self.K = K


class Image:
    """Image, camera id and world->camera pose (structs.py:14-27)."""

    def __init__(self, image, camera_id, camera_T_world):
        self.image = image
        self.camera_id = camera_id
        self.camera_T_world = camera_T_world


class Gaussians(torch.nn.Module):
    """All mutable Gaussian parameters (structs.py:46-114).

    xyz[N,3]; rgb[N,3] (scaled by 1/SH_0); opacity[N,1] logits; scale[N,3] log; quaternion[N,4]
    (w,x,y,z); sh None or [N,3,{3,8,15}].
    """

    def __init__(self, xyz, rgb, opacity, scale, quaternion, sh=None):
        super().__init__()
        self.xyz = xyz
        self.rgb = rgb
        self.opacity = opacity
        self.scale = scale
        self.quaternion = quaternion
        self.sh = sh
        self.verify_sizes()

    def __len__(self):
        return self.xyz.shape[0]

    def verify_sizes(self):
        n = self.xyz.shape[0]
        for name, width in (("xyz", 3), ("rgb", 3), ("opacity", 1), ("scale", 3), ("quaternion", 4)):
            t = getattr(self, name)
            assert t.shape[0] == n, f"{name} has {t.shape[0]} rows, expected {n}"
            assert t.dim() >= 2 and t.shape[1] == width, f"{name} must be Nx{width}"
        if self.sh is not None:
            assert self.sh.shape[0] == n and self.sh.shape[1] == 3

    def _names(self):
        return ["xyz", "rgb", "opacity", "scale", "quaternion"] + (["sh"] if self.sh is not None else [])

    def filter_in_place(self, keep_mask):
        for name in self._names():
            setattr(self, name, torch.nn.Parameter(getattr(self, name).detach()[keep_mask]))
        self.verify_sizes()

    def append(self, xyz, rgb, opacity, scale, quaternion, sh=None):
        new = dict(xyz=xyz, rgb=rgb, opacity=opacity, scale=scale, quaternion=quaternion)
        if sh is not None:
            new["sh"] = sh
        for name, t in new.items():
            cur = getattr(self, name)
            setattr(self, name, torch.nn.Parameter(torch.cat((cur.detach(), t.detach()), dim=0)))
        self.verify_sizes()


class Tiles:
    """16x16 tile grid covering the image (structs.py:117-138)."""

    def __init__(self, image_height, image_width, device):
        self.image_height = image_height
        self.image_width = image_width
        self.device = device
        self.tile_edge_size = TILE_EDGE_LENGTH_PX
        self.y_tiles_count = math.ceil(image_height / self.tile_edge_size)
        self.x_tiles_count = math.ceil(image_width / self.tile_edge_size)
        self.image_height_padded = self.y_tiles_count * self.tile_edge_size
        self.image_width_padded = self.x_tiles_count * self.tile_edge_size
        self.tile_count = self.y_tiles_count * self.x_tiles_count
