"""The six torch.autograd.Function wrappers of the hot path, same names and argument orders as
the reference (splat_py/cuda_autograd_functions.py:19-219).  Each allocates its outputs, calls
the `splat_cuda` provider selected in gaussian_splatting_amd.backend and saves what its backward
needs."""
import torch

from .. import backend


def _zeros_like_shape(shape, ref):
    return torch.zeros(shape, dtype=ref.dtype, device=ref.device)


class CameraPointProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_camera, K):
        uv = _zeros_like_shape((xyz_camera.shape[0], 2), xyz_camera)
        backend.get().camera_projection_cuda(xyz_camera, K, uv)
        ctx.save_for_backward(xyz_camera, K)
        return uv

    @staticmethod
    def backward(ctx, grad_uv):
        xyz_camera, K = ctx.saved_tensors
        grad_xyz_camera = torch.zeros_like(xyz_camera)
        backend.get().camera_projection_backward_cuda(xyz_camera, K, grad_uv.contiguous(), grad_xyz_camera)
        return grad_xyz_camera, None


class ComputeSigmaWorld(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quaternion, scale):
        sigma_world = _zeros_like_shape((quaternion.shape[0], 3, 3), quaternion)
        backend.get().compute_sigma_world_cuda(quaternion, scale, sigma_world)
        ctx.save_for_backward(quaternion, scale)
        return sigma_world

    @staticmethod
    def backward(ctx, grad_sigma_world):
        quaternion, scale = ctx.saved_tensors
        grad_quaternion = torch.zeros_like(quaternion)
        grad_scale = torch.zeros_like(scale)
        backend.get().compute_sigma_world_backward_cuda(
            quaternion, scale, grad_sigma_world.contiguous(), grad_quaternion, grad_scale)
        return grad_quaternion, grad_scale


class ComputeProjectionJacobian(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz_camera, K):
        jacobian = _zeros_like_shape((xyz_camera.shape[0], 2, 3), xyz_camera)
        backend.get().compute_projection_jacobian_cuda(xyz_camera, K, jacobian)
        ctx.save_for_backward(xyz_camera, K)
        return jacobian

    @staticmethod
    def backward(ctx, grad_jacobian):
        xyz_camera, K = ctx.saved_tensors
        grad_xyz_camera = torch.zeros_like(xyz_camera)
        backend.get().compute_projection_jacobian_backward_cuda(
            xyz_camera, K, grad_jacobian.contiguous(), grad_xyz_camera)
        return grad_xyz_camera, None


class ComputeConic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigma_world, J, camera_T_world):
        conic = _zeros_like_shape((J.shape[0], 3), sigma_world)
        backend.get().compute_conic_cuda(sigma_world, J, camera_T_world, conic)
        ctx.save_for_backward(sigma_world, camera_T_world, J)
        return conic

    @staticmethod
    def backward(ctx, grad_conic):
        sigma_world, camera_T_world, J = ctx.saved_tensors
        grad_sigma_world = torch.zeros_like(sigma_world)
        grad_J = torch.zeros_like(J)
        backend.get().compute_conic_backward_cuda(
            sigma_world, J, camera_T_world, grad_conic.contiguous(), grad_sigma_world, grad_J)
        return grad_sigma_world, grad_J, None


class PrecomputeRGBFromSH(torch.autograd.Function):
    """No gradient flows to xyz through the view direction (Q7; cuda_autograd_functions.py:127)."""

    @staticmethod
    def forward(ctx, sh_coeffs, xyz, camera_T_world):
        rgb = torch.zeros(xyz.shape[0], 3, dtype=sh_coeffs.dtype, device=sh_coeffs.device)
        backend.get().precompute_rgb_from_sh_cuda(xyz, sh_coeffs, camera_T_world, rgb)
        ctx.save_for_backward(xyz, camera_T_world)
        ctx.sh_shape = tuple(sh_coeffs.shape)   # kept on the host: no .item() sync in backward
        return rgb

    @staticmethod
    def backward(ctx, grad_rgb):
        xyz, camera_T_world = ctx.saved_tensors
        n_sh = ctx.sh_shape[2] if len(ctx.sh_shape) == 3 else 1
        grad_sh_coeffs = torch.zeros(xyz.shape[0], 3, n_sh, dtype=xyz.dtype, device=xyz.device)
        backend.get().precompute_rgb_from_sh_backward_cuda(xyz, camera_T_world, grad_rgb.contiguous(), grad_sh_coeffs)
        return grad_sh_coeffs.reshape(ctx.sh_shape), None, None


def _hw(image_size):
    # the reference passes a device tensor and indexes it (implicit D2H); ints avoid the sync
    if torch.is_tensor(image_size):
        h, w = image_size.tolist()
        return int(h), int(w)
    return int(image_size[0]), int(image_size[1])


class RenderImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, opacity, uvs, conic, rays, splat_start_end_idx_by_tile_idx,
                sorted_gaussian_idx_by_splat_idx, image_size, background_rgb, tile_rows=None):
        """tile_rows (optional, beyond the reference signature): render only these tile rows."""
        H, W = _hw(image_size)
        kw = {} if tile_rows is None else dict(tile_rows=tile_rows)
        ctx.kw = kw
        rendered_image = torch.zeros(H, W, 3, dtype=rgb.dtype, device=rgb.device)
        num_splats_per_pixel = torch.zeros(H, W, dtype=torch.int, device=rgb.device)
        final_weight_per_pixel = torch.zeros(H, W, dtype=rgb.dtype, device=rgb.device)
        backend.get().render_tiles_cuda(
            uvs, opacity, rgb, conic, rays, splat_start_end_idx_by_tile_idx, sorted_gaussian_idx_by_splat_idx,
            background_rgb, num_splats_per_pixel, final_weight_per_pixel, rendered_image, **kw)
        ctx.save_for_backward(
            uvs, opacity, rgb, conic, rays, splat_start_end_idx_by_tile_idx, sorted_gaussian_idx_by_splat_idx,
            background_rgb, num_splats_per_pixel, final_weight_per_pixel)
        return rendered_image

    @staticmethod
    def backward(ctx, grad_rendered_image):
        (uvs, opacity, rgb, conic, rays, splat_start_end_idx_by_tile_idx, sorted_gaussian_idx_by_splat_idx,
         background_rgb, num_splats_per_pixel, final_weight_per_pixel) = ctx.saved_tensors
        grad_rgb = torch.zeros_like(rgb)
        grad_opacity = torch.zeros_like(opacity)
        grad_uv = torch.zeros_like(uvs)
        grad_conic = torch.zeros_like(conic)
        backend.get().render_tiles_backward_cuda(
            uvs, opacity, rgb, conic, rays, splat_start_end_idx_by_tile_idx, sorted_gaussian_idx_by_splat_idx,
            background_rgb, num_splats_per_pixel, final_weight_per_pixel, grad_rendered_image.contiguous(),
            grad_rgb, grad_opacity, grad_uv, grad_conic, **ctx.kw)
        return grad_rgb, grad_opacity, grad_uv, grad_conic, None, None, None, None, None, None
