"""`splat_cuda` for MI355X: the 14 functions of the reference extension (src/bindings.cpp:118-159)
on top of the C ABI in include/gsplat_hip.h.

Same names, positional signatures, in-place output convention and error behaviour (RuntimeError
for a non-device / non-contiguous tensor, wrong dtype or shape, unsupported SH count;
src/checks.cuh:5-14).  Differences, all deliberate: kernels are enqueued on torch's current HIP
stream and nothing synchronises the device (the reference device-syncs after almost every
kernel); `get_sorted_gaussian_list` performs exactly one 4-byte device-to-host
read (the instance count that sizes its result).

`install()` registers this module as `splat_cuda` in sys.modules so that the reference's own
`splat_py` package imports it unchanged.
"""
import ctypes
import sys

import torch

from . import _hip
from ._hip import GS_F32, GS_F64


def install(name="splat_cuda"):
    sys.modules[name] = sys.modules[__name__]


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return _hip.current_stream()


def _valid(**tensors):
    # CHECK_VALID_INPUT (checks.cuh:5-9); plus: one device per call, and it is the current one -- the kernels
    # are launched on the calling thread's current device and its current stream (the compiled module
    # csrc/bindings_hip.cpp switches devices itself; ctypes has no guard to offer)
    first = None
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} is not a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} is not a contiguous tensor")
        if first is None:
            first = t.device
            if first.index != torch.cuda.current_device():
                raise RuntimeError(f"{name} is on {first} but the current device is cuda:{torch.cuda.current_device()}: "
                                   "call under torch.cuda.device(...) of the tensors' device")
        elif t.device != first:
            raise RuntimeError(f"{name} is on {t.device} but the call's first tensor is on {first}")


def _dtype(first, **others):
    if first.dtype == torch.float32:
        code = GS_F32
    elif first.dtype == torch.float64:
        code = GS_F64
    else:
        raise RuntimeError("Inputs must be float32 or float64")
    for name, t in others.items():
        if t.dtype != first.dtype:
            kind = "float" if code == GS_F32 else "double"
            raise RuntimeError(f"{name} is not a {kind} tensor")
    return code


def _int(**tensors):
    for name, t in tensors.items():
        if t.dtype != torch.int32:
            raise RuntimeError(f"{name} is not an int tensor")


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _n_sh(t):
    n = t.shape[2] if t.dim() == 3 else 1
    _require(n in (1, 4, 9, 16), "Unsupported number of SH coefficients")
    return n


# ---- projection.cu / projection_backward.cu ---------------------------------------------------------
def camera_projection_cuda(xyz, K, uv):
    _valid(xyz=xyz, K=K, uv=uv)
    N = xyz.shape[0]
    _require(xyz.shape[1] == 3, "xyz must have shape Nx3")
    _require(tuple(K.shape) == (3, 3), "K must have shape 3x3")
    _require(tuple(uv.shape) == (N, 2), "uv must have shape Nx2")
    dt = _dtype(xyz, K=K, uv=uv)
    _hip.call("gs_camera_projection", _p(xyz), _p(K), N, _p(uv), dt, _stream())


def camera_projection_backward_cuda(xyz, K, uv_grad_out, xyz_grad_in):
    _valid(xyz=xyz, K=K, uv_grad_out=uv_grad_out, xyz_grad_in=xyz_grad_in)
    N = xyz.shape[0]
    _require(xyz.shape[1] == 3, "xyz must be of shape Nx3")
    _require(tuple(K.shape) == (3, 3), "K must be of shape 3x3")
    _require(tuple(uv_grad_out.shape) == (N, 2), "uv_grad_out must be of shape Nx2")
    _require(tuple(xyz_grad_in.shape) == (N, 3), "xyz_grad_in must be of shape Nx3")
    dt = _dtype(xyz, K=K, uv_grad_out=uv_grad_out, xyz_grad_in=xyz_grad_in)
    _hip.call("gs_camera_projection_backward", _p(xyz), _p(K), _p(uv_grad_out), N, _p(xyz_grad_in), dt,
                                                   _stream())


def compute_sigma_world_cuda(quaternion, scale, sigma_world):
    _valid(quaternion=quaternion, scale=scale, sigma_world=sigma_world)
    N = quaternion.shape[0]
    _require(quaternion.shape[1] == 4, "quaternion must have shape Nx4")
    _require(scale.shape[0] == N, "scale must have shape Nx1")
    _require(tuple(sigma_world.shape) == (N, 3, 3), "sigma_world must have shape Nx3x3")
    dt = _dtype(quaternion, scale=scale, sigma_world=sigma_world)
    _hip.call("gs_compute_sigma_world", _p(quaternion), _p(scale), N, _p(sigma_world), dt, _stream())


def compute_sigma_world_backward_cuda(quaternion, scale, sigma_world_grad_out, quaternion_grad_in, scale_grad_in):
    _valid(quaternion=quaternion, scale=scale, sigma_world_grad_out=sigma_world_grad_out,
           quaternion_grad_in=quaternion_grad_in, scale_grad_in=scale_grad_in)
    N = quaternion.shape[0]
    _require(quaternion.shape[1] == 4, "quaternion must have shape Nx4")
    _require(tuple(scale.shape) == (N, 3), "scale must have shape Nx3")
    _require(tuple(sigma_world_grad_out.shape) == (N, 3, 3), "sigma_world_grad_out must have shape Nx3x3")
    _require(tuple(quaternion_grad_in.shape) == (N, 4), "quaternion_grad_in must have shape Nx4")
    _require(tuple(scale_grad_in.shape) == (N, 3), "scale_grad_in must have shape Nx3")
    dt = _dtype(quaternion, scale=scale, sigma_world_grad_out=sigma_world_grad_out,
                quaternion_grad_in=quaternion_grad_in, scale_grad_in=scale_grad_in)
    _hip.call("gs_compute_sigma_world_backward", _p(quaternion), _p(scale), _p(sigma_world_grad_out), N,
                                                     _p(quaternion_grad_in), _p(scale_grad_in), dt, _stream())


def compute_projection_jacobian_cuda(xyz, K, J):
    _valid(xyz=xyz, K=K, J=J)
    N = xyz.shape[0]
    _require(xyz.shape[1] == 3, "xyz must have shape Nx3")
    _require(tuple(K.shape) == (3, 3), "K must have shape 3x3")
    _require(tuple(J.shape) == (N, 2, 3), "J must have shape Nx2x3")
    dt = _dtype(xyz, K=K, J=J)
    _hip.call("gs_compute_projection_jacobian", _p(xyz), _p(K), N, _p(J), dt, _stream())


def compute_projection_jacobian_backward_cuda(xyz, K, jac_grad_out, xyz_grad_in):
    _valid(xyz=xyz, K=K, jac_grad_out=jac_grad_out, xyz_grad_in=xyz_grad_in)
    N = xyz.shape[0]
    _require(tuple(jac_grad_out.shape) == (N, 2, 3), "jac_grad_out must have shape Nx2x3")
    _require(tuple(xyz_grad_in.shape) == (N, 3), "xyz_grad_in must have shape Nx3")
    dt = _dtype(xyz, K=K, jac_grad_out=jac_grad_out, xyz_grad_in=xyz_grad_in)
    _hip.call("gs_compute_projection_jacobian_backward", _p(xyz), _p(K), _p(jac_grad_out), N,
                                                             _p(xyz_grad_in), dt, _stream())


def compute_conic_cuda(sigma_world, J, camera_T_world, conic):
    _valid(sigma_world=sigma_world, J=J, camera_T_world=camera_T_world, conic=conic)
    N = sigma_world.shape[0]
    _require(tuple(sigma_world.shape[1:]) == (3, 3), "sigma_world must have shape Nx3x3")
    _require(tuple(J.shape) == (N, 2, 3), "J must have shape Nx2x3")
    _require(tuple(camera_T_world.shape) == (4, 4), "camera_T_world must have shape 4x4")
    _require(tuple(conic.shape) == (N, 3), "conic must have shape Nx3")
    dt = _dtype(sigma_world, J=J, camera_T_world=camera_T_world, conic=conic)
    _hip.call("gs_compute_conic", _p(sigma_world), _p(J), _p(camera_T_world), N, _p(conic), dt, _stream())


def compute_conic_backward_cuda(sigma_world, J, camera_T_world, conic_grad_out, sigma_world_grad_in, J_grad_in):
    _valid(sigma_world=sigma_world, J=J, camera_T_world=camera_T_world, conic_grad_out=conic_grad_out,
           sigma_world_grad_in=sigma_world_grad_in, J_grad_in=J_grad_in)
    N = sigma_world.shape[0]
    _require(tuple(sigma_world.shape[1:]) == (3, 3), "sigma_world must have shape Nx3x3")
    _require(tuple(J.shape) == (N, 2, 3), "J must have shape Nx2x3")
    _require(tuple(camera_T_world.shape) == (4, 4), "camera_T_world must have shape 4x4")
    _require(tuple(conic_grad_out.shape) == (N, 3), "conic_grad_out must have shape Nx3")
    _require(tuple(sigma_world_grad_in.shape) == (N, 3, 3), "sigma_world_grad_in must have shape Nx3x3")
    _require(tuple(J_grad_in.shape) == (N, 2, 3), "J_grad_in must have shape Nx2x3")
    dt = _dtype(sigma_world, J=J, camera_T_world=camera_T_world, conic_grad_out=conic_grad_out,
                sigma_world_grad_in=sigma_world_grad_in, J_grad_in=J_grad_in)
    _hip.call("gs_compute_conic_backward", _p(sigma_world), _p(J), _p(camera_T_world), _p(conic_grad_out), N,
                                               _p(sigma_world_grad_in), _p(J_grad_in), dt, _stream())


# ---- precompute_sh.cu -------------------------------------------------------------------------------------
def precompute_rgb_from_sh_cuda(xyz, sh_coeff, camera_T_world, rgb):
    _valid(xyz=xyz, sh_coeff=sh_coeff, camera_T_world=camera_T_world, rgb=rgb)
    N = xyz.shape[0]
    _require(xyz.shape[1] == 3, "Input xyz should have 3 channels")
    _require(sh_coeff.shape[0] == N, "N xyz and sh_coeff should match")
    _require(sh_coeff.shape[1] == 3, "SH coefficients should have 3 channels")
    n_sh = _n_sh(sh_coeff)
    _require(tuple(camera_T_world.shape) == (4, 4), "camera_T_world should be 4x4 transformation matrix")
    _require(tuple(rgb.shape) == (N, 3), "Output rgb should have 3 channels")
    dt = _dtype(xyz, sh_coeff=sh_coeff, camera_T_world=camera_T_world, rgb=rgb)
    _hip.call("gs_precompute_rgb_from_sh", _p(xyz), _p(sh_coeff), _p(camera_T_world), N, n_sh, _p(rgb), dt,
                                               _stream())


def precompute_rgb_from_sh_backward_cuda(xyz, camera_T_world, grad_rgb, grad_sh):
    _valid(xyz=xyz, camera_T_world=camera_T_world, grad_rgb=grad_rgb, grad_sh=grad_sh)
    N = xyz.shape[0]
    _require(xyz.shape[1] == 3, "Input xyz should have 3 channels")
    _require(tuple(camera_T_world.shape) == (4, 4), "camera_T_world should be 4x4 transformation matrix")
    _require(tuple(grad_rgb.shape) == (N, 3), "Input grad_rgb should have 3 channels")
    _require(grad_sh.shape[0] == N and grad_sh.shape[1] == 3, "Output grad_sh should have 3 channels")
    n_sh = _n_sh(grad_sh)
    dt = _dtype(xyz, camera_T_world=camera_T_world, grad_rgb=grad_rgb, grad_sh=grad_sh)
    _hip.call("gs_precompute_rgb_from_sh_backward", _p(xyz), _p(camera_T_world), _p(grad_rgb), N, n_sh,
                                                        _p(grad_sh), dt, _stream())


# ---- tile_culling.cu ----------------------------------------------------------------------------------------
def get_sorted_gaussian_list(max_tiles_per_gaussian, uvs, xyz_camera_frame, conic, n_tiles_x, n_tiles_y, mh_dist,
                             tile_rows=None):
    """-> (sorted_gaussians int32[S], splat_start_end_idx_by_tile_idx int32[T+1]).
    max_tiles_per_gaussian is accepted and unused, as in the reference (tile_culling.cu:245).
    tile_rows=(row0, row1) restricts binning to those tile rows (extension used for sharding)."""
    _valid(uvs=uvs, xyz_camera_frame=xyz_camera_frame, conic=conic)
    for name, t in (("uvs", uvs), ("xyz_camera_frame", xyz_camera_frame), ("conic", conic)):
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} is not a float tensor")
    V = uvs.shape[0]
    T = int(n_tiles_x) * int(n_tiles_y)
    row0, row1 = tile_rows if tile_rows is not None else (0, int(n_tiles_y))
    dev = uvs.device
    counts = torch.empty(_hip.lib().gs_tile_workspace_ints(T), dtype=torch.int32, device=dev)
    ranges = torch.empty(T + 1, dtype=torch.int32, device=dev)
    mh = ctypes.c_float(mh_dist)
    _hip.call("gs_tile_count", _p(uvs), _p(conic), V, None, None, None, int(n_tiles_x), int(n_tiles_y), mh, row0, row1, _p(counts),
                          _p(ranges), None, _stream())
    S = int(ranges[T].item())   # the one host read: sizes the result
    sorted_g = torch.empty(S, dtype=torch.int32, device=dev)
    if S > 0:
        keys = torch.empty(S, dtype=torch.int64, device=dev)
        _hip.call("gs_tile_emit_sort", _p(uvs), _p(xyz_camera_frame), _p(conic), V, None, None, None, int(n_tiles_x), int(n_tiles_y), mh,
                                  row0, row1, _p(ranges), _p(counts), _p(keys), ctypes.c_int64(S), _p(sorted_g),
                                  0, _stream())
    return sorted_g, ranges


def band_mask(uvs, conic, n_tiles_x, n_tiles_y, mh_dist, band_rows):
    """Multi-GPU bookkeeping (no reference counterpart; gaussian_splatting_amd.sharded): bit s of
    mask[g] is set when the candidate tile window of Gaussian g reaches tile rows
    [band_rows[s], band_rows[s+1]).  int32 tensor [V]."""
    _valid(uvs=uvs, conic=conic)
    V, G = uvs.shape[0], len(band_rows) - 1
    i32 = dict(dtype=torch.int32, device=uvs.device)
    if V == 0:
        return torch.empty(0, **i32)
    nblk = (V + 255) // 256
    count = torch.full((1,), V, **i32)
    pre_ws = torch.zeros(2 * nblk, **i32)
    mask = torch.empty(V, **i32)
    ws = torch.empty(_hip.lib().gs_halo_workspace_ints(V, G), **i32)
    send_index = torch.empty(V, **i32)
    plan = torch.empty(4 + 2 * G, **i32)
    rows = (ctypes.c_int32 * (G + 1))(*[int(r) for r in band_rows])
    blks = (ctypes.c_int32 * (G + 1))(*([0] * G + [nblk]))
    _hip.call("gs_halo_plan", _p(uvs), _p(conic), V, _p(count), _p(pre_ws), int(n_tiles_x), int(n_tiles_y),
              ctypes.c_float(float(mh_dist)), rows, blks, G, 0, _p(mask), _p(ws), _p(send_index), _p(plan),
              _stream())
    return mask


# ---- render.cu / render_backward.cu / depth.cu --------------------------------------------------------------
def _pack(uvs, opacity, conic, dt, rgb=None):
    V = uvs.shape[0]
    packed = torch.empty(V, 12, dtype=uvs.dtype, device=uvs.device)
    col = _p(rgb) if (rgb is not None and _n_sh(rgb) == 1) else None
    _hip.call("gs_pack_splats", _p(uvs), _p(opacity), _p(conic), col, V, _p(packed), dt, _stream())
    return packed


def _render_checks(uvs, opacity, rgb, conic):
    N = uvs.shape[0]
    _require(uvs.dim() == 2 and uvs.shape[1] == 2, "uvs must be Nx2 (u, v)")
    _require(opacity.shape[0] == N, "Opacity must have the same number of elements as uvs")
    _require(opacity.dim() == 2 and opacity.shape[1] == 1, "Opacity must be Nx1")
    _require(rgb.shape[0] == N, "RGB must have the same number of elements as uvs")
    _require(rgb.shape[1] == 3, "RGB must be Nx3")
    _require(conic.shape[0] == N, "Conic must have the same number of elements as uvs")
    _require(conic.shape[1] == 3, "Conic must be Nx3")
    return N


def render_tiles_cuda(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
                      gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel, final_weight_per_pixel,
                      rendered_image, tile_rows=None):
    _valid(uvs=uvs, opacity=opacity, rgb=rgb, conic=conic, view_dir_by_pixel=view_dir_by_pixel,
           splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx, background_rgb=background_rgb,
           num_splats_per_pixel=num_splats_per_pixel, final_weight_per_pixel=final_weight_per_pixel,
           rendered_image=rendered_image)
    _render_checks(uvs, opacity, rgb, conic)
    _require(rendered_image.dim() == 3 and rendered_image.shape[2] == 3, "Image must be HxWx3")
    _require(background_rgb.dim() == 1, "Background RGB must be 1D")
    _require(background_rgb.shape[0] == 3, "Background RGB must have 3 elements")
    H, W = rendered_image.shape[0], rendered_image.shape[1]
    n_sh = _n_sh(rgb)
    if n_sh > 1:
        _require(tuple(view_dir_by_pixel.shape) == (H, W, 3), "view_dir_by_pixel must have the same size as the image")
    dt = _dtype(uvs, opacity=opacity, rgb=rgb, conic=conic, view_dir_by_pixel=view_dir_by_pixel,
                background_rgb=background_rgb, final_weight_per_pixel=final_weight_per_pixel,
                rendered_image=rendered_image)
    _int(splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
         gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx, num_splats_per_pixel=num_splats_per_pixel)
    nty = (H + 15) // 16
    _require(splat_start_end_idx_by_tile_idx.shape[0] == ((W + 15) // 16) * nty + 1,
             "splat_start_end_idx_by_tile_idx must have n_tiles + 1 entries")
    row0, row1 = tile_rows if tile_rows is not None else (0, nty)
    _hip.call("gs_render_tiles", _p(uvs), _p(opacity), _p(rgb), _p(conic), _p(view_dir_by_pixel),
              _p(splat_start_end_idx_by_tile_idx), _p(gaussian_idx_by_splat_idx), _p(background_rgb),
              _p(num_splats_per_pixel), _p(final_weight_per_pixel), _p(rendered_image), W, H, n_sh, row0, row1, dt,
              _stream())


def render_tiles_backward_cuda(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
                               gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel,
                               final_weight_per_pixel, grad_image, grad_rgb, grad_opacity, grad_uv, grad_conic,
                               tile_rows=None):
    _valid(uvs=uvs, opacity=opacity, rgb=rgb, conic=conic, view_dir_by_pixel=view_dir_by_pixel,
           splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx, background_rgb=background_rgb,
           num_splats_per_pixel=num_splats_per_pixel, final_weight_per_pixel=final_weight_per_pixel,
           grad_image=grad_image, grad_rgb=grad_rgb, grad_opacity=grad_opacity, grad_uv=grad_uv,
           grad_conic=grad_conic)
    _render_checks(uvs, opacity, rgb, conic)
    _require(background_rgb.dim() == 1 and background_rgb.shape[0] == 3, "Background RGB must have 3 elements")
    H, W = num_splats_per_pixel.shape[0], num_splats_per_pixel.shape[1]
    n_sh = _n_sh(rgb)
    if n_sh > 1:
        _require(tuple(view_dir_by_pixel.shape) == (H, W, 3), "view_dir_by_pixel must have the same size as the image")
    nty = (H + 15) // 16
    _require(splat_start_end_idx_by_tile_idx.shape[0] == ((W + 15) // 16) * nty + 1,
             "splat_start_end_idx_by_tile_idx ")
    _require(tuple(final_weight_per_pixel.shape) == (H, W), "final_weight_per_pixel must have the same size as the image")
    _require(tuple(grad_image.shape) == (H, W, 3), "grad_image must have the same size as the image")
    _require(grad_rgb.shape == rgb.shape and grad_opacity.shape == opacity.shape and grad_uv.shape == uvs.shape
             and grad_conic.shape == conic.shape, "gradient outputs must match their inputs")
    dt = _dtype(uvs, opacity=opacity, rgb=rgb, conic=conic, view_dir_by_pixel=view_dir_by_pixel,
                background_rgb=background_rgb, final_weight_per_pixel=final_weight_per_pixel, grad_image=grad_image,
                grad_rgb=grad_rgb, grad_opacity=grad_opacity, grad_uv=grad_uv, grad_conic=grad_conic)
    _int(splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
         gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx, num_splats_per_pixel=num_splats_per_pixel)
    row0, row1 = tile_rows if tile_rows is not None else (0, nty)
    _hip.call("gs_render_tiles_backward", _p(uvs), _p(opacity), _p(rgb), _p(conic), _p(view_dir_by_pixel),
              _p(splat_start_end_idx_by_tile_idx), _p(gaussian_idx_by_splat_idx), _p(background_rgb),
              _p(num_splats_per_pixel), _p(final_weight_per_pixel), _p(grad_image), _p(grad_rgb), _p(grad_opacity),
              _p(grad_uv), _p(grad_conic), W, H, n_sh, row0, row1, dt, _hip.GS_BACKWARD_DEFAULT, _stream())


def render_depth_cuda(xyz_camera_frame, uvs, opacity, conic, splat_start_end_idx_by_tile_idx,
                      gaussian_idx_by_splat_idx, alpha_threshold, depth_image):
    _valid(xyz_camera_frame=xyz_camera_frame, uvs=uvs, opacity=opacity, conic=conic,
           splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx, depth_image=depth_image)
    for name, t in (("xyz_camera_frame", xyz_camera_frame), ("uvs", uvs), ("opacity", opacity), ("conic", conic),
                    ("depth_image", depth_image)):
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} is not a float tensor")
    _int(splat_start_end_idx_by_tile_idx=splat_start_end_idx_by_tile_idx,
         gaussian_idx_by_splat_idx=gaussian_idx_by_splat_idx)
    H, W = depth_image.shape[0], depth_image.shape[1]
    packed = _pack(uvs, opacity, conic, GS_F32)
    _hip.call("gs_render_depth", _p(packed), _p(xyz_camera_frame), _p(splat_start_end_idx_by_tile_idx),
                                     _p(gaussian_idx_by_splat_idx), W, H, ctypes.c_float(alpha_threshold),
                                     _p(depth_image), _stream())
