// bindings_hip.cpp -- the native `splat_cuda` Python module for MI355X.
//
// Same 14 functions, positional signatures on torch::Tensor, in-place output convention and
// TORCH_CHECK error behaviour as the reference's pybind module (src/bindings.cpp:3-159,
// src/checks.cuh:5-14), implemented on the C ABI of libgsplat_hip.so (include/gsplat_hip.h).  The one
// translation unit of this build that sees torch headers: everything below the binding is plain
// pointers + sizes + a hipStream_t.  Differences from the reference, all deliberate: kernels are
// enqueued on torch's CURRENT HIP stream and nothing synchronises the device (the reference launches
// on the legacy stream and device-syncs after almost every kernel);
// get_sorted_gaussian_list performs exactly one 4-byte device-to-host read (the instance count that
// sizes its result).  The GIL stays held, as in the reference (no gil_scoped_release).
#include <ATen/hip/HIPContext.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <tuple>

#include "../../include/gsplat_hip.h"

namespace {

// The device of a call = the device of its first tensor: the guard makes it current for the launches (a
// stream of device 1 under current device 0 is an invalid handle), every further tensor must live there
// too, and the kernels are enqueued on THAT device's current stream.
struct DeviceOf {
    c10::OptionalDeviceGuard guard;
    c10::Device device{c10::kCPU};
    bool set = false;
    void see(const torch::Tensor& t, const char* name) {
        if (!set) {
            device = t.device();
            guard.reset_device(device);
            set = true;
        } else {
            TORCH_CHECK(t.device() == device, name, " is on ", t.device(), " but the call's first tensor is on ", device);
        }
    }
    void* stream() const { return (void*)c10::hip::getCurrentHIPStream(set ? device.index() : -1).stream(); }
};

void ok(int status) { TORCH_CHECK(status == GS_OK, gs_last_error()); }

// CHECK_VALID_INPUT (checks.cuh:5-9)
#define VALID(x)                                                                                   \
    TORCH_CHECK((x).is_cuda(), #x " is not a CUDA tensor");                                        \
    TORCH_CHECK((x).is_contiguous(), #x " is not a contiguous tensor");                            \
    dev_.see((x), #x)
#define IS_INT(x) TORCH_CHECK((x).scalar_type() == torch::kInt32, #x " is not an int tensor")

int float_type(const torch::Tensor& first) {
    if (first.scalar_type() == torch::kFloat32) return GS_F32;
    if (first.scalar_type() == torch::kFloat64) return GS_F64;
    AT_ERROR("Inputs must be float32 or float64");
}
#define SAME_TYPE(dt, x)                                                                           \
    TORCH_CHECK((x).scalar_type() == ((dt) == GS_F32 ? torch::kFloat32 : torch::kFloat64),        \
                #x " is not a ", (dt) == GS_F32 ? "float" : "double", " tensor")

int n_sh_of(const torch::Tensor& t) {
    const int64_t n = t.dim() == 3 ? t.size(2) : 1;
    TORCH_CHECK(n == 1 || n == 4 || n == 9 || n == 16, "Unsupported number of SH coefficients");
    return (int)n;
}

bool has_shape(const torch::Tensor& t, std::initializer_list<int64_t> shape) {
    if (t.dim() != (int64_t)shape.size()) return false;
    int64_t d = 0;
    for (int64_t s : shape)
        if (t.size(d++) != s) return false;
    return true;
}

torch::Tensor pack(const torch::Tensor& uvs, const torch::Tensor& opacity, const torch::Tensor& conic,
                   const torch::Tensor* rgb, int dt, void* stream) {
    const int64_t V = uvs.size(0);
    torch::Tensor packed = torch::empty({V, GS_PACKED_WIDTH}, uvs.options());
    const void* col = (rgb != nullptr && n_sh_of(*rgb) == 1) ? rgb->data_ptr() : nullptr;
    ok(gs_pack_splats(uvs.data_ptr(), opacity.data_ptr(), conic.data_ptr(), col, (int)V, packed.data_ptr(), dt,
                      stream));
    return packed;
}

int64_t render_checks(const torch::Tensor& uvs, const torch::Tensor& opacity, const torch::Tensor& rgb,
                      const torch::Tensor& conic) {
    const int64_t N = uvs.size(0);
    TORCH_CHECK(uvs.dim() == 2 && uvs.size(1) == 2, "uvs must be Nx2 (u, v)");
    TORCH_CHECK(opacity.size(0) == N, "Opacity must have the same number of elements as uvs");
    TORCH_CHECK(opacity.dim() == 2 && opacity.size(1) == 1, "Opacity must be Nx1");
    TORCH_CHECK(rgb.size(0) == N, "RGB must have the same number of elements as uvs");
    TORCH_CHECK(rgb.size(1) == 3, "RGB must be Nx3");
    TORCH_CHECK(conic.size(0) == N, "Conic must have the same number of elements as uvs");
    TORCH_CHECK(conic.size(1) == 3, "Conic must be Nx3");
    return N;
}

}  // namespace

// ---- render.cu / render_backward.cu / depth.cu ---------------------------------------------------------
void render_tiles_cuda(torch::Tensor uvs, torch::Tensor opacity, torch::Tensor rgb, torch::Tensor conic,
                       torch::Tensor view_dir_by_pixel, torch::Tensor splat_start_end_idx_by_tile_idx,
                       torch::Tensor gaussian_idx_by_splat_idx, torch::Tensor background_rgb,
                       torch::Tensor num_splats_per_pixel, torch::Tensor final_weight_per_pixel,
                       torch::Tensor rendered_image) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(uvs); VALID(opacity); VALID(rgb); VALID(conic); VALID(view_dir_by_pixel);
    VALID(splat_start_end_idx_by_tile_idx); VALID(gaussian_idx_by_splat_idx); VALID(background_rgb);
    VALID(num_splats_per_pixel); VALID(final_weight_per_pixel); VALID(rendered_image);
    render_checks(uvs, opacity, rgb, conic);
    TORCH_CHECK(rendered_image.dim() == 3 && rendered_image.size(2) == 3, "Image must be HxWx3");
    TORCH_CHECK(background_rgb.dim() == 1, "Background RGB must be 1D");
    TORCH_CHECK(background_rgb.size(0) == 3, "Background RGB must have 3 elements");
    const int H = (int)rendered_image.size(0), W = (int)rendered_image.size(1);
    const int n_sh = n_sh_of(rgb);
    if (n_sh > 1)
        TORCH_CHECK(has_shape(view_dir_by_pixel, {H, W, 3}), "view_dir_by_pixel must have the same size as the image");
    const int dt = float_type(uvs);
    SAME_TYPE(dt, opacity); SAME_TYPE(dt, rgb); SAME_TYPE(dt, conic); SAME_TYPE(dt, view_dir_by_pixel);
    SAME_TYPE(dt, background_rgb); SAME_TYPE(dt, final_weight_per_pixel); SAME_TYPE(dt, rendered_image);
    IS_INT(splat_start_end_idx_by_tile_idx); IS_INT(gaussian_idx_by_splat_idx); IS_INT(num_splats_per_pixel);
    const int nty = (H + 15) / 16;
    TORCH_CHECK(splat_start_end_idx_by_tile_idx.size(0) == (int64_t)((W + 15) / 16) * nty + 1,
                "splat_start_end_idx_by_tile_idx must have n_tiles + 1 entries");
    ok(gs_render_tiles(uvs.data_ptr(), opacity.data_ptr(), rgb.data_ptr(), conic.data_ptr(), view_dir_by_pixel.data_ptr(),
                       splat_start_end_idx_by_tile_idx.data_ptr<int32_t>(),
                       gaussian_idx_by_splat_idx.data_ptr<int32_t>(), background_rgb.data_ptr(),
                       num_splats_per_pixel.data_ptr<int32_t>(), final_weight_per_pixel.data_ptr(),
                       rendered_image.data_ptr(), W, H, n_sh, 0, nty, dt, dev_.stream()));
}

void render_tiles_backward_cuda(torch::Tensor uvs, torch::Tensor opacity, torch::Tensor rgb, torch::Tensor conic,
                                torch::Tensor view_dir_by_pixel, torch::Tensor splat_start_end_idx_by_tile_idx,
                                torch::Tensor gaussian_idx_by_splat_idx, torch::Tensor background_rgb,
                                torch::Tensor num_splats_per_pixel, torch::Tensor final_weight_per_pixel,
                                torch::Tensor grad_image, torch::Tensor grad_rgb, torch::Tensor grad_opacity,
                                torch::Tensor grad_uvs, torch::Tensor grad_conic) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(uvs); VALID(opacity); VALID(rgb); VALID(conic); VALID(view_dir_by_pixel);
    VALID(splat_start_end_idx_by_tile_idx); VALID(gaussian_idx_by_splat_idx); VALID(background_rgb);
    VALID(num_splats_per_pixel); VALID(final_weight_per_pixel); VALID(grad_image); VALID(grad_rgb);
    VALID(grad_opacity); VALID(grad_uvs); VALID(grad_conic);
    render_checks(uvs, opacity, rgb, conic);
    TORCH_CHECK(background_rgb.dim() == 1 && background_rgb.size(0) == 3, "Background RGB must have 3 elements");
    TORCH_CHECK(num_splats_per_pixel.dim() == 2, "num_splats_per_pixel must be HxW");
    const int H = (int)num_splats_per_pixel.size(0), W = (int)num_splats_per_pixel.size(1);
    const int n_sh = n_sh_of(rgb);
    if (n_sh > 1)
        TORCH_CHECK(has_shape(view_dir_by_pixel, {H, W, 3}), "view_dir_by_pixel must have the same size as the image");
    const int nty = (H + 15) / 16;
    TORCH_CHECK(splat_start_end_idx_by_tile_idx.size(0) == (int64_t)((W + 15) / 16) * nty + 1,
                "splat_start_end_idx_by_tile_idx must have n_tiles + 1 entries");
    TORCH_CHECK(has_shape(final_weight_per_pixel, {H, W}), "final_weight_per_pixel must have the same size as the image");
    TORCH_CHECK(has_shape(grad_image, {H, W, 3}), "grad_image must have the same size as the image");
    TORCH_CHECK(grad_rgb.sizes() == rgb.sizes() && grad_opacity.sizes() == opacity.sizes() &&
                    grad_uvs.sizes() == uvs.sizes() && grad_conic.sizes() == conic.sizes(),
                "gradient outputs must match their inputs");
    const int dt = float_type(uvs);
    SAME_TYPE(dt, opacity); SAME_TYPE(dt, rgb); SAME_TYPE(dt, conic); SAME_TYPE(dt, view_dir_by_pixel);
    SAME_TYPE(dt, background_rgb); SAME_TYPE(dt, final_weight_per_pixel); SAME_TYPE(dt, grad_image);
    SAME_TYPE(dt, grad_rgb); SAME_TYPE(dt, grad_opacity); SAME_TYPE(dt, grad_uvs); SAME_TYPE(dt, grad_conic);
    IS_INT(splat_start_end_idx_by_tile_idx); IS_INT(gaussian_idx_by_splat_idx); IS_INT(num_splats_per_pixel);
    ok(gs_render_tiles_backward(uvs.data_ptr(), opacity.data_ptr(), rgb.data_ptr(), conic.data_ptr(),
                                view_dir_by_pixel.data_ptr(), splat_start_end_idx_by_tile_idx.data_ptr<int32_t>(),
                                gaussian_idx_by_splat_idx.data_ptr<int32_t>(), background_rgb.data_ptr(),
                                num_splats_per_pixel.data_ptr<int32_t>(), final_weight_per_pixel.data_ptr(),
                                grad_image.data_ptr(), grad_rgb.data_ptr(), grad_opacity.data_ptr(), grad_uvs.data_ptr(),
                                grad_conic.data_ptr(), W, H, n_sh, 0, nty, dt, GS_BACKWARD_DEFAULT, dev_.stream()));
}

void render_depth_cuda(torch::Tensor xyz_camera_frame, torch::Tensor uvs, torch::Tensor opacity, torch::Tensor conic,
                       torch::Tensor splat_start_end_idx_by_tile_idx, torch::Tensor gaussian_idx_by_splat_idx,
                       const float alpha_threshold, torch::Tensor depth_image) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz_camera_frame); VALID(uvs); VALID(opacity); VALID(conic); VALID(splat_start_end_idx_by_tile_idx);
    VALID(gaussian_idx_by_splat_idx); VALID(depth_image);
    SAME_TYPE(GS_F32, xyz_camera_frame); SAME_TYPE(GS_F32, uvs); SAME_TYPE(GS_F32, opacity);
    SAME_TYPE(GS_F32, conic); SAME_TYPE(GS_F32, depth_image);
    IS_INT(splat_start_end_idx_by_tile_idx); IS_INT(gaussian_idx_by_splat_idx);
    TORCH_CHECK(depth_image.dim() == 3 && depth_image.size(2) == 1, "Depth Image must be HxWx1");   // depth.cu:148
    const int H = (int)depth_image.size(0), W = (int)depth_image.size(1);
    torch::Tensor packed = pack(uvs, opacity, conic, nullptr, GS_F32, dev_.stream());
    ok(gs_render_depth(packed.data_ptr(), xyz_camera_frame.data_ptr(),
                       splat_start_end_idx_by_tile_idx.data_ptr<int32_t>(),
                       gaussian_idx_by_splat_idx.data_ptr<int32_t>(), W, H, alpha_threshold, depth_image.data_ptr(),
                       dev_.stream()));
}

// ---- projection.cu / projection_backward.cu --------------------------------------------------------------
void camera_projection_cuda(torch::Tensor xyz, torch::Tensor K, torch::Tensor uv) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(K); VALID(uv);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "xyz must have shape Nx3");
    TORCH_CHECK(has_shape(K, {3, 3}), "K must have shape 3x3");
    TORCH_CHECK(has_shape(uv, {N, 2}), "uv must have shape Nx2");
    const int dt = float_type(xyz);
    SAME_TYPE(dt, K); SAME_TYPE(dt, uv);
    ok(gs_camera_projection(xyz.data_ptr(), K.data_ptr(), (int)N, uv.data_ptr(), dt, dev_.stream()));
}

void camera_projection_backward_cuda(torch::Tensor xyz, torch::Tensor K, torch::Tensor uv_grad_out,
                                     torch::Tensor xyz_grad_in) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(K); VALID(uv_grad_out); VALID(xyz_grad_in);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "xyz must be of shape Nx3");
    TORCH_CHECK(has_shape(K, {3, 3}), "K must be of shape 3x3");
    TORCH_CHECK(has_shape(uv_grad_out, {N, 2}), "uv_grad_out must be of shape Nx2");
    TORCH_CHECK(has_shape(xyz_grad_in, {N, 3}), "xyz_grad_in must be of shape Nx3");
    const int dt = float_type(xyz);
    SAME_TYPE(dt, K); SAME_TYPE(dt, uv_grad_out); SAME_TYPE(dt, xyz_grad_in);
    ok(gs_camera_projection_backward(xyz.data_ptr(), K.data_ptr(), uv_grad_out.data_ptr(), (int)N,
                                     xyz_grad_in.data_ptr(), dt, dev_.stream()));
}

void compute_sigma_world_cuda(torch::Tensor quaternion, torch::Tensor scale, torch::Tensor sigma_world) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(quaternion); VALID(scale); VALID(sigma_world);
    const int64_t N = quaternion.size(0);
    TORCH_CHECK(quaternion.dim() == 2 && quaternion.size(1) == 4, "quaternion must have shape Nx4");
    TORCH_CHECK(scale.size(0) == N, "scale must have shape Nx1");
    TORCH_CHECK(has_shape(sigma_world, {N, 3, 3}), "sigma_world must have shape Nx3x3");
    const int dt = float_type(quaternion);
    SAME_TYPE(dt, scale); SAME_TYPE(dt, sigma_world);
    ok(gs_compute_sigma_world(quaternion.data_ptr(), scale.data_ptr(), (int)N, sigma_world.data_ptr(), dt, dev_.stream()));
}

void compute_sigma_world_backward_cuda(torch::Tensor quaternion, torch::Tensor scale, torch::Tensor sigma_world_grad_out,
                                       torch::Tensor quaternion_grad_in, torch::Tensor scale_grad_in) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(quaternion); VALID(scale); VALID(sigma_world_grad_out); VALID(quaternion_grad_in); VALID(scale_grad_in);
    const int64_t N = quaternion.size(0);
    TORCH_CHECK(quaternion.dim() == 2 && quaternion.size(1) == 4, "quaternion must have shape Nx4");
    TORCH_CHECK(has_shape(scale, {N, 3}), "scale must have shape Nx3");
    TORCH_CHECK(has_shape(sigma_world_grad_out, {N, 3, 3}), "sigma_world_grad_out must have shape Nx3x3");
    TORCH_CHECK(has_shape(quaternion_grad_in, {N, 4}), "quaternion_grad_in must have shape Nx4");
    TORCH_CHECK(has_shape(scale_grad_in, {N, 3}), "scale_grad_in must have shape Nx3");
    const int dt = float_type(quaternion);
    SAME_TYPE(dt, scale); SAME_TYPE(dt, sigma_world_grad_out); SAME_TYPE(dt, quaternion_grad_in); SAME_TYPE(dt, scale_grad_in);
    ok(gs_compute_sigma_world_backward(quaternion.data_ptr(), scale.data_ptr(), sigma_world_grad_out.data_ptr(), (int)N,
                                       quaternion_grad_in.data_ptr(), scale_grad_in.data_ptr(), dt, dev_.stream()));
}

void compute_projection_jacobian_cuda(torch::Tensor xyz, torch::Tensor K, torch::Tensor J) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(K); VALID(J);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "xyz must have shape Nx3");
    TORCH_CHECK(has_shape(K, {3, 3}), "K must have shape 3x3");
    TORCH_CHECK(has_shape(J, {N, 2, 3}), "J must have shape Nx2x3");
    const int dt = float_type(xyz);
    SAME_TYPE(dt, K); SAME_TYPE(dt, J);
    ok(gs_compute_projection_jacobian(xyz.data_ptr(), K.data_ptr(), (int)N, J.data_ptr(), dt, dev_.stream()));
}

void compute_projection_jacobian_backward_cuda(torch::Tensor xyz, torch::Tensor K, torch::Tensor jac_grad_out,
                                               torch::Tensor xyz_grad_in) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(K); VALID(jac_grad_out); VALID(xyz_grad_in);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(has_shape(jac_grad_out, {N, 2, 3}), "jac_grad_out must have shape Nx2x3");
    TORCH_CHECK(has_shape(xyz_grad_in, {N, 3}), "xyz_grad_in must have shape Nx3");
    const int dt = float_type(xyz);
    SAME_TYPE(dt, K); SAME_TYPE(dt, jac_grad_out); SAME_TYPE(dt, xyz_grad_in);
    ok(gs_compute_projection_jacobian_backward(xyz.data_ptr(), K.data_ptr(), jac_grad_out.data_ptr(), (int)N,
                                               xyz_grad_in.data_ptr(), dt, dev_.stream()));
}

void compute_conic_cuda(torch::Tensor sigma_world, torch::Tensor J, torch::Tensor camera_T_world, torch::Tensor conic) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(sigma_world); VALID(J); VALID(camera_T_world); VALID(conic);
    const int64_t N = sigma_world.size(0);
    TORCH_CHECK(has_shape(sigma_world, {N, 3, 3}), "sigma_world must have shape Nx3x3");
    TORCH_CHECK(has_shape(J, {N, 2, 3}), "J must have shape Nx2x3");
    TORCH_CHECK(has_shape(camera_T_world, {4, 4}), "camera_T_world must have shape 4x4");
    TORCH_CHECK(has_shape(conic, {N, 3}), "conic must have shape Nx3");
    const int dt = float_type(sigma_world);
    SAME_TYPE(dt, J); SAME_TYPE(dt, camera_T_world); SAME_TYPE(dt, conic);
    ok(gs_compute_conic(sigma_world.data_ptr(), J.data_ptr(), camera_T_world.data_ptr(), (int)N, conic.data_ptr(), dt,
                        dev_.stream()));
}

void compute_conic_backward_cuda(torch::Tensor sigma_world, torch::Tensor J, torch::Tensor camera_T_world,
                                 torch::Tensor conic_grad_out, torch::Tensor sigma_world_grad_in, torch::Tensor J_grad_in) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(sigma_world); VALID(J); VALID(camera_T_world); VALID(conic_grad_out); VALID(sigma_world_grad_in); VALID(J_grad_in);
    const int64_t N = sigma_world.size(0);
    TORCH_CHECK(has_shape(sigma_world, {N, 3, 3}), "sigma_world must have shape Nx3x3");
    TORCH_CHECK(has_shape(J, {N, 2, 3}), "J must have shape Nx2x3");
    TORCH_CHECK(has_shape(camera_T_world, {4, 4}), "camera_T_world must have shape 4x4");
    TORCH_CHECK(has_shape(conic_grad_out, {N, 3}), "conic_grad_out must have shape Nx3");
    TORCH_CHECK(has_shape(sigma_world_grad_in, {N, 3, 3}), "sigma_world_grad_in must have shape Nx3x3");
    TORCH_CHECK(has_shape(J_grad_in, {N, 2, 3}), "J_grad_in must have shape Nx2x3");
    const int dt = float_type(sigma_world);
    SAME_TYPE(dt, J); SAME_TYPE(dt, camera_T_world); SAME_TYPE(dt, conic_grad_out); SAME_TYPE(dt, sigma_world_grad_in);
    SAME_TYPE(dt, J_grad_in);
    ok(gs_compute_conic_backward(sigma_world.data_ptr(), J.data_ptr(), camera_T_world.data_ptr(), conic_grad_out.data_ptr(),
                                 (int)N, sigma_world_grad_in.data_ptr(), J_grad_in.data_ptr(), dt, dev_.stream()));
}

// ---- tile_culling.cu -----------------------------------------------------------------------------------
std::tuple<torch::Tensor, torch::Tensor> get_sorted_gaussian_list(const int max_tiles_per_gaussian, torch::Tensor uvs,
                                                                  torch::Tensor xyz_camera_frame, torch::Tensor conic,
                                                                  const int n_tiles_x, const int n_tiles_y,
                                                                  const float mh_dist) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    (void)max_tiles_per_gaussian;   // accepted and unused, as in the reference (tile_culling.cu:245)
    VALID(uvs); VALID(xyz_camera_frame); VALID(conic);
    SAME_TYPE(GS_F32, uvs); SAME_TYPE(GS_F32, xyz_camera_frame); SAME_TYPE(GS_F32, conic);
    const int V = (int)uvs.size(0);
    const int64_t T = (int64_t)n_tiles_x * n_tiles_y;
    auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(uvs.device());
    torch::Tensor workspace = torch::empty({(int64_t)gs_tile_workspace_ints((int)T)}, i32);
    torch::Tensor ranges = torch::empty({T + 1}, i32);
    void* stream = dev_.stream();
    ok(gs_tile_count(uvs.data_ptr(), conic.data_ptr(), V, nullptr, nullptr, nullptr, n_tiles_x, n_tiles_y, mh_dist, 0,
                     n_tiles_y, workspace.data_ptr<int32_t>(), ranges.data_ptr<int32_t>(), nullptr, stream));
    const int64_t S = ranges[T].item<int32_t>();   // the one host read: sizes the result
    torch::Tensor sorted = torch::empty({S}, i32);
    if (S > 0) {
        torch::Tensor keys = torch::empty({S}, i32.dtype(torch::kInt64));
        ok(gs_tile_emit_sort(uvs.data_ptr(), xyz_camera_frame.data_ptr(), conic.data_ptr(), V, nullptr, nullptr, nullptr,
                             n_tiles_x, n_tiles_y, mh_dist, 0, n_tiles_y, ranges.data_ptr<int32_t>(),
                             workspace.data_ptr<int32_t>(), (uint64_t*)keys.data_ptr<int64_t>(), S,
                             sorted.data_ptr<int32_t>(), 0, stream));
    }
    return std::make_tuple(sorted, ranges);
}

// ---- precompute_sh.cu ------------------------------------------------------------------------------------
void precompute_rgb_from_sh_cuda(const torch::Tensor xyz, const torch::Tensor sh_coeff, const torch::Tensor camera_T_world,
                                 torch::Tensor rgb) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(sh_coeff); VALID(camera_T_world); VALID(rgb);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "Input xyz should have 3 channels");
    TORCH_CHECK(sh_coeff.size(0) == N, "N xyz and sh_coeff should match");
    TORCH_CHECK(sh_coeff.size(1) == 3, "SH coefficients should have 3 channels");
    const int n_sh = n_sh_of(sh_coeff);
    TORCH_CHECK(has_shape(camera_T_world, {4, 4}), "camera_T_world should be 4x4 transformation matrix");
    TORCH_CHECK(has_shape(rgb, {N, 3}), "Output rgb should have 3 channels");
    const int dt = float_type(xyz);
    SAME_TYPE(dt, sh_coeff); SAME_TYPE(dt, camera_T_world); SAME_TYPE(dt, rgb);
    ok(gs_precompute_rgb_from_sh(xyz.data_ptr(), sh_coeff.data_ptr(), camera_T_world.data_ptr(), (int)N, n_sh,
                                 rgb.data_ptr(), dt, dev_.stream()));
}

void precompute_rgb_from_sh_backward_cuda(const torch::Tensor xyz, const torch::Tensor camera_T_world,
                                          const torch::Tensor grad_rgb, torch::Tensor grad_sh) {
    DeviceOf dev_;   // kernels go to the tensors' device and its current stream, whatever the caller's current device
    VALID(xyz); VALID(camera_T_world); VALID(grad_rgb); VALID(grad_sh);
    const int64_t N = xyz.size(0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "Input xyz should have 3 channels");
    TORCH_CHECK(has_shape(camera_T_world, {4, 4}), "camera_T_world should be 4x4 transformation matrix");
    TORCH_CHECK(has_shape(grad_rgb, {N, 3}), "Input grad_rgb should have 3 channels");
    TORCH_CHECK(grad_sh.size(0) == N && grad_sh.size(1) == 3, "Output grad_sh should have 3 channels");
    const int n_sh = n_sh_of(grad_sh);
    const int dt = float_type(xyz);
    SAME_TYPE(dt, camera_T_world); SAME_TYPE(dt, grad_rgb); SAME_TYPE(dt, grad_sh);
    ok(gs_precompute_rgb_from_sh_backward(xyz.data_ptr(), camera_T_world.data_ptr(), grad_rgb.data_ptr(), (int)N, n_sh,
                                          grad_sh.data_ptr(), dt, dev_.stream()));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "splat_cuda for MI355X (gfx950): the reference extension's 14 functions over libgsplat_hip.so";
    m.def("render_tiles_cuda", &render_tiles_cuda, "Render tiles");
    m.def("render_tiles_backward_cuda", &render_tiles_backward_cuda, "Render tiles backward");
    m.def("camera_projection_cuda", &camera_projection_cuda, "project point into image");
    m.def("camera_projection_backward_cuda", &camera_projection_backward_cuda, "project point into image backward");
    m.def("compute_sigma_world_cuda", &compute_sigma_world_cuda, "compute sigma world");
    m.def("compute_sigma_world_backward_cuda", &compute_sigma_world_backward_cuda, "compute sigma world backward");
    m.def("compute_projection_jacobian_cuda", &compute_projection_jacobian_cuda, "compute projection jacobian");
    m.def("compute_projection_jacobian_backward_cuda", &compute_projection_jacobian_backward_cuda,
          "compute projection jacobian backward");
    m.def("compute_conic_cuda", &compute_conic_cuda, "compute conic");
    m.def("compute_conic_backward_cuda", &compute_conic_backward_cuda, "compute conic backward");
    m.def("get_sorted_gaussian_list", &get_sorted_gaussian_list, "get sorted gaussian list");
    m.def("precompute_rgb_from_sh_cuda", &precompute_rgb_from_sh_cuda, "precompute rgb from sh per gaussian");
    m.def("precompute_rgb_from_sh_backward_cuda", &precompute_rgb_from_sh_backward_cuda,
          "precompute rgb from sh per gaussian backward");
    m.def("render_depth_cuda", &render_depth_cuda, "Render depth");
}
