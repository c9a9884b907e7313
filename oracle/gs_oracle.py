"""ctypes front-end of the CPU ORACLE (test infrastructure -- see gs_oracle.cpp header).

Exposes the 14 functions of the reference's `splat_cuda` module (src/bindings.cpp:118-159)
on CPU torch tensors, with the reference's argument order and in-place output convention, so
the host-side mirror (gaussian_splatting_amd.splat_py) and the reference's own Python host
code can be driven end-to-end on CPU.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gs_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_tile_count.restype = ctypes.c_int64
        _lib.orc_det_expf.restype = ctypes.c_float
        _lib.orc_det_expf.argtypes = [ctypes.c_float]
    return _lib


def set_modes(exp_mode=0, trig_mode=0):
    """exp_mode 0 = deterministic polynomial, 1 = libm; trig_mode 0 = algebraic, 1 = libm."""
    lib().orc_set_modes(int(exp_mode), int(trig_mode))


def set_sh_band1_mode(mode):
    """0 = shipped spherical_harmonics.cuh (default); 1 = notebook convention."""
    lib().orc_set_sh_band1_mode(int(mode))


def sigmoid_det(x):
    """the IEEE-only sigmoid of the fused preprocess kernel (fp32 CPU tensor -> tensor)"""
    x = x.contiguous()
    y = torch.empty_like(x)
    lib().orc_sigmoid_f32(_p(x), x.numel(), _p(y))
    return y


def num_threads():
    return lib().orc_num_threads()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _sfx(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise RuntimeError("Inputs must be float32 or float64")


def _check(*ts):
    for t in ts:
        if t.is_cuda:
            raise RuntimeError("oracle takes CPU tensors")
        if not t.is_contiguous():
            raise RuntimeError("tensor is not contiguous")


def _fn(name, t):
    return getattr(lib(), f"orc_{name}_{_sfx(t)}")


# ---- the 14 splat_cuda entry points (bindings.cpp:118-159) ---------------------------------
def camera_projection_cuda(xyz, K, uv):
    _check(xyz, K, uv)
    _fn("camera_projection", xyz)(_p(xyz), _p(K), xyz.shape[0], _p(uv))


def camera_projection_backward_cuda(xyz, K, uv_grad_out, xyz_grad_in):
    _check(xyz, K, uv_grad_out, xyz_grad_in)
    _fn("camera_projection_backward", xyz)(_p(xyz), _p(K), _p(uv_grad_out), xyz.shape[0], _p(xyz_grad_in))


def compute_sigma_world_cuda(quaternion, scale, sigma_world):
    _check(quaternion, scale, sigma_world)
    _fn("compute_sigma_world", quaternion)(_p(quaternion), _p(scale), quaternion.shape[0], _p(sigma_world))


def compute_sigma_world_backward_cuda(quaternion, scale, sigma_world_grad_out, quaternion_grad_in, scale_grad_in):
    _check(quaternion, scale, sigma_world_grad_out, quaternion_grad_in, scale_grad_in)
    _fn("compute_sigma_world_backward", quaternion)(
        _p(quaternion), _p(scale), _p(sigma_world_grad_out), quaternion.shape[0],
        _p(quaternion_grad_in), _p(scale_grad_in))


def compute_projection_jacobian_cuda(xyz, K, J):
    _check(xyz, K, J)
    _fn("compute_projection_jacobian", xyz)(_p(xyz), _p(K), xyz.shape[0], _p(J))


def compute_projection_jacobian_backward_cuda(xyz, K, jac_grad_out, xyz_grad_in):
    _check(xyz, K, jac_grad_out, xyz_grad_in)
    _fn("compute_projection_jacobian_backward", xyz)(
        _p(xyz), _p(K), _p(jac_grad_out), xyz.shape[0], _p(xyz_grad_in))


def compute_conic_cuda(sigma_world, J, camera_T_world, conic):
    _check(sigma_world, J, camera_T_world, conic)
    _fn("compute_conic", sigma_world)(_p(sigma_world), _p(J), _p(camera_T_world), sigma_world.shape[0], _p(conic))


def compute_conic_backward_cuda(sigma_world, J, camera_T_world, conic_grad_out, sigma_world_grad_in, J_grad_in):
    _check(sigma_world, J, camera_T_world, conic_grad_out, sigma_world_grad_in, J_grad_in)
    _fn("compute_conic_backward", sigma_world)(
        _p(sigma_world), _p(J), _p(camera_T_world), _p(conic_grad_out), sigma_world.shape[0],
        _p(sigma_world_grad_in), _p(J_grad_in))


def _n_sh(t):
    return t.shape[2] if t.dim() == 3 else 1


def precompute_rgb_from_sh_cuda(xyz, sh_coeff, camera_T_world, rgb):
    _check(xyz, sh_coeff, camera_T_world, rgb)
    n_sh = _n_sh(sh_coeff)
    if n_sh not in (1, 4, 9, 16):
        raise RuntimeError("Unsupported number of SH coefficients")
    cam = camera_T_world[:3, 3].contiguous()   # precompute_sh.cu:149-151
    _fn("precompute_rgb_from_sh", xyz)(_p(xyz), _p(sh_coeff), _p(cam), xyz.shape[0], n_sh, _p(rgb))


def precompute_rgb_from_sh_backward_cuda(xyz, camera_T_world, grad_rgb, grad_sh):
    _check(xyz, camera_T_world, grad_rgb, grad_sh)
    n_sh = _n_sh(grad_sh)
    if n_sh not in (1, 4, 9, 16):
        raise RuntimeError("Unsupported number of SH coefficients")
    cam = camera_T_world[:3, 3].contiguous()
    _fn("precompute_rgb_from_sh_backward", xyz)(_p(xyz), _p(cam), _p(grad_rgb), xyz.shape[0], n_sh, _p(grad_sh))


def get_sorted_gaussian_list(max_tiles_per_gaussian, uvs, xyz_camera_frame, conic, n_tiles_x, n_tiles_y,
                             mh_dist, return_keys=False):
    _check(uvs, xyz_camera_frame, conic)
    for t in (uvs, xyz_camera_frame, conic):
        if t.dtype != torch.float32:
            raise RuntimeError("tile culling takes float32 tensors")
    N = uvs.shape[0]
    T = n_tiles_x * n_tiles_y
    per_g = torch.zeros(N, dtype=torch.int32)
    per_t = torch.zeros(T, dtype=torch.int32)
    S = lib().orc_tile_count(_p(uvs), _p(conic), int(n_tiles_x), int(n_tiles_y), ctypes.c_float(mh_dist), N,
                             _p(per_g), _p(per_t))
    sorted_g = torch.zeros(S, dtype=torch.int32)
    ranges = torch.zeros(T + 1, dtype=torch.int32)
    keys = torch.zeros(S, dtype=torch.int64) if return_keys else None
    lib().orc_tile_emit_sort(_p(uvs), _p(xyz_camera_frame), _p(conic), int(n_tiles_x), int(n_tiles_y),
                             ctypes.c_float(mh_dist), N, _p(per_g), _p(per_t), ctypes.c_int64(S),
                             _p(sorted_g), _p(ranges), _p(keys) if return_keys else None)
    if return_keys:
        return sorted_g, ranges, keys
    return sorted_g, ranges


def band_mask(uvs, conic, n_tiles_x, n_tiles_y, mh_dist, band_rows):
    """bit s of mask[g]: the candidate tile window of Gaussian g reaches tile rows
    [band_rows[s], band_rows[s+1]) (checker for gs_halo_plan; int32 tensor, G <= 31)"""
    _check(uvs, conic)
    G = len(band_rows) - 1
    rows = (ctypes.c_int * (G + 1))(*[int(r) for r in band_rows])
    mask = torch.zeros(uvs.shape[0], dtype=torch.int32)
    lib().orc_band_mask(_p(uvs), _p(conic), int(n_tiles_x), int(n_tiles_y), ctypes.c_float(mh_dist),
                        uvs.shape[0], rows, G, _p(mask))
    return mask


def render_tiles_cuda(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
                      gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel,
                      final_weight_per_pixel, rendered_image, tile_rows=None):
    _check(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel, final_weight_per_pixel,
           rendered_image)
    n_sh = _n_sh(rgb)
    if n_sh not in (1, 4, 9, 16):
        raise RuntimeError("Unsupported number of SH coefficients")
    H, W = rendered_image.shape[0], rendered_image.shape[1]
    y0, y1 = tile_rows if tile_rows is not None else (0, (H + 15) // 16)
    _fn("render_tiles", uvs)(
        _p(uvs), _p(opacity), _p(rgb), _p(conic), _p(view_dir_by_pixel),
        _p(splat_start_end_idx_by_tile_idx), _p(gaussian_idx_by_splat_idx), _p(background_rgb),
        W, H, n_sh, _p(num_splats_per_pixel), _p(final_weight_per_pixel), _p(rendered_image), y0, y1)


def render_tiles_backward_cuda(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
                               gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel,
                               final_weight_per_pixel, grad_image, grad_rgb, grad_opacity, grad_uv,
                               grad_conic, tile_rows=None):
    _check(uvs, opacity, rgb, conic, view_dir_by_pixel, splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx, background_rgb, num_splats_per_pixel, final_weight_per_pixel,
           grad_image, grad_rgb, grad_opacity, grad_uv, grad_conic)
    n_sh = _n_sh(rgb)
    H, W = num_splats_per_pixel.shape[0], num_splats_per_pixel.shape[1]
    y0, y1 = tile_rows if tile_rows is not None else (0, (H + 15) // 16)
    _fn("render_tiles_backward", uvs)(
        _p(uvs), _p(opacity), _p(rgb), _p(conic), _p(view_dir_by_pixel),
        _p(splat_start_end_idx_by_tile_idx), _p(gaussian_idx_by_splat_idx), _p(background_rgb),
        _p(num_splats_per_pixel), _p(final_weight_per_pixel), _p(grad_image), W, H, n_sh, uvs.shape[0],
        _p(grad_rgb), _p(grad_opacity), _p(grad_uv), _p(grad_conic), y0, y1)


def set_backward_exact(on):
    """1: exact gradient (global splat index in the transmittance update) instead of the reference's
    render_backward.cu:185 -- the checker of the library's GS_BACKWARD_EXACT mode"""
    lib().orc_set_backward_exact(int(bool(on)))


def set_backward_sum(mode):
    """how render_tiles_backward_cuda sums a gradient element's per-pixel fp32 terms: 0 = in double, rounded
    once (default); 1 = in fp32, pixels and tiles ascending; 2 = in fp32, both descending (two equally valid
    fp32 summation orders of the same terms: their difference is pure summation-order noise)"""
    lib().orc_set_backward_sum(int(mode))


def render_tiles_backward_abs(*args, **kw):
    """render_tiles_backward_cuda with |term| accumulated instead of term: per gradient element the sum
    of the magnitudes of its per-pixel terms (checker aid, no reference counterpart)."""
    lib().orc_set_backward_abs(1)
    try:
        render_tiles_backward_cuda(*args, **kw)
    finally:
        lib().orc_set_backward_abs(0)


def render_tiles_with_contrib_count(*args, **kw):
    """render_tiles_cuda that also returns int32[H, W]: splats that passed the alpha >= 1/255 test at
    each pixel before it saturated (checker aid: decision fingerprint of a pixel)."""
    import torch
    img = args[10] if len(args) > 10 else kw["rendered_image"]
    count = torch.zeros(img.shape[0], img.shape[1], dtype=torch.int32)
    lib().orc_set_contrib_count(_p(count))
    try:
        render_tiles_cuda(*args, **kw)
    finally:
        lib().orc_set_contrib_count(None)
    return count


def render_depth_cuda(xyz_camera_frame, uvs, opacity, conic, splat_start_end_idx_by_tile_idx,
                      gaussian_idx_by_splat_idx, alpha_threshold, depth_image):
    _check(xyz_camera_frame, uvs, opacity, conic, splat_start_end_idx_by_tile_idx,
           gaussian_idx_by_splat_idx, depth_image)
    H, W = depth_image.shape[0], depth_image.shape[1]
    lib().orc_render_depth_f32(_p(xyz_camera_frame), _p(uvs), _p(opacity), _p(conic),
                               _p(splat_start_end_idx_by_tile_idx), _p(gaussian_idx_by_splat_idx), W, H,
                               ctypes.c_float(alpha_threshold), _p(depth_image))


SPLAT_CUDA_API = [
    "render_tiles_cuda", "render_tiles_backward_cuda", "camera_projection_cuda",
    "camera_projection_backward_cuda", "compute_sigma_world_cuda", "compute_sigma_world_backward_cuda",
    "compute_projection_jacobian_cuda", "compute_projection_jacobian_backward_cuda", "compute_conic_cuda",
    "compute_conic_backward_cuda", "get_sorted_gaussian_list", "precompute_rgb_from_sh_cuda",
    "precompute_rgb_from_sh_backward_cuda", "render_depth_cuda",
]
