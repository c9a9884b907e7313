"""Single-node multi-GPU rasterization: one frame sharded by tile rows (BASELINE.json north_star,
SURVEY.md 8(e)).  One process per GPU, torch.distributed over RCCL/xGMI (backend "nccl"); the same
code runs over gloo on CPU tensors for tests.

Per frame, on every rank (Gaussian parameters are replicated, as in data-parallel training):
  forward   per-Gaussian stage for all N (cheap, replicated) -> binning, sort and render ONLY for the
            rank's band of tile rows -> all-reduce(SUM) of the image: every other rank contributes
            zeros outside its band, so the sum is the gather (12 B/px, ~13 MB at 1 MP).
  backward  every rank holds the same grad_image (the loss is evaluated on the gathered image on
            every rank); render backward over the rank's band gives PARTIAL per-Gaussian gradients
            (uv 2, conic 3, opacity 1, colour 3 floats: a Gaussian can straddle bands) ->
            all-reduce(SUM) of that [V, 9] slab (36 B per visible Gaussian, ~100 MB at 2.86 M) ->
            the per-Gaussian backward runs replicated and yields identical dense parameter
            gradients on every rank.
Tile lists of a band equal the single-GPU lists restricted to the band (tested bit-exactly), so
the sharded image equals the single-GPU image bit for bit and gradients differ only by fp32
summation order.
"""
import torch
import torch.distributed as dist


def band_of(n_tile_rows, world_size, rank):
    """Contiguous, near-equal split of the tile rows: [row0, row1) of `rank`."""
    base, rem = divmod(n_tile_rows, world_size)
    row0 = rank * base + min(rank, rem)
    return row0, row0 + base + (1 if rank < rem else 0)


class _GatherImage(torch.autograd.Function):
    """forward: sum over ranks of band images (disjoint support == gather); backward: identity,
    because every rank evaluates the same loss on the same gathered image."""

    @staticmethod
    def forward(ctx, image, group):
        out = image.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        return grad, None


class _SumGradsAcrossRanks(torch.autograd.Function):
    """identity in forward; in backward the partial per-Gaussian render gradients of all bands are
    summed with ONE all-reduce over a flat slab."""

    @staticmethod
    def forward(ctx, group, *tensors):
        ctx.group = group
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        shapes = [g.shape for g in grads]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        out, off = [], 0
        for s in shapes:
            n = s.numel()
            out.append(flat[off:off + n].view(s))
            off += n
        return (None,) + tuple(out)


class ShardedRasterizer:
    def __init__(self, image_height, world_size=None, rank=None, group=None, fused=True):
        self.group = group
        self.world_size = world_size if world_size is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.fused = fused
        n_tile_rows = (image_height + 15) // 16
        self.tile_rows = band_of(n_tile_rows, self.world_size, self.rank)

    def _grad_sync(self, *tensors):
        return _SumGradsAcrossRanks.apply(self.group, *tensors)

    def _slab_sync(self, flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def rasterize(self, gaussians, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding, mh_dist,
                  use_sh_precompute, background_rgb):
        """Same contract as splat_py.rasterize.rasterize; the returned image is the full frame on
        every rank."""
        if self.fused:
            from . import fused
            impl = fused.rasterize
        else:
            from .splat_py.rasterize import rasterize as impl
        image, culling_mask, uv = impl(
            gaussians, camera_T_world, camera, near_thresh, far_thresh, cull_mask_padding, mh_dist,
            use_sh_precompute, background_rgb, tile_rows=self.tile_rows, grad_sync=self._grad_sync,
            **({"slab_sync": self._slab_sync} if self.fused else {}))
        image = _GatherImage.apply(image, self.group)
        return image, culling_mask, uv
