// per_gaussian.hip -- one-thread-per-Gaussian kernels behind the eight projection entry points,
// the two SH-precompute entry points and gs_pack_splats.  HBM-bound streaming kernels: 256-thread
// workgroups, no LDS, no host synchronisation, launched on the caller's stream.
//
// Forward kernels keep the operation order and operand precisions of the reference so that their
// fp32 outputs are bit-identical to the CPU restatement used by the tests; backward kernels
// compute in T and are compared with a tolerance.
#include "pg_math.h"

namespace gs {

constexpr int PG_BLOCK = 256;

// projection.cu:9-19
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_camera_projection(const T* __restrict__ xyz,
                                                                const T* __restrict__ K, int N,
                                                                T* __restrict__ uv) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    uv[i * 2 + 0] = K[0] * x / z + K[2];
    uv[i * 2 + 1] = K[4] * y / z + K[5];
}

// projection_backward.cu:9-36
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_camera_projection_bwd(const T* __restrict__ xyz,
                                                                    const T* __restrict__ K,
                                                                    const T* __restrict__ g_uv,
                                                                    int N, T* __restrict__ g_xyz) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    if (z <= T(0)) return;   // Q10: output keeps the caller's zeros
    const T du_dx = K[0] / z;
    const T dv_dy = K[4] / z;
    const T du_dz = -K[0] * x / (z * z);
    const T dv_dz = -K[4] * y / (z * z);
    const T gu = g_uv[i * 2 + 0], gv = g_uv[i * 2 + 1];
    g_xyz[i * 3 + 0] = gu * du_dx;
    g_xyz[i * 3 + 1] = gv * dv_dy;
    g_xyz[i * 3 + 2] = gu * du_dz + gv * dv_dz;
}

template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_sigma_world(const T* __restrict__ q,
                                                          const T* __restrict__ scale, int N,
                                                          T* __restrict__ sigma) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T q4[4] = {q[i * 4 + 0], q[i * 4 + 1], q[i * 4 + 2], q[i * 4 + 3]};
    const T s3[3] = {scale[i * 3 + 0], scale[i * 3 + 1], scale[i * 3 + 2]};
    T S[9];
    sigma_world_of(q4, s3, S);
#pragma unroll
    for (int k = 0; k < 9; k++) sigma[(size_t)i * 9 + k] = S[k];
}

// projection_backward.cu:174-315
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_sigma_world_bwd(const T* __restrict__ quat,
                                                              const T* __restrict__ scale,
                                                              const T* __restrict__ gSig, int N,
                                                              T* __restrict__ g_q,
                                                              T* __restrict__ g_scale) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T q4[4] = {quat[i * 4 + 0], quat[i * 4 + 1], quat[i * 4 + 2], quat[i * 4 + 3]};
    const T s3[3] = {scale[i * 3 + 0], scale[i * 3 + 1], scale[i * 3 + 2]};
    T G[9], gq[4], gs[3];
#pragma unroll
    for (int k = 0; k < 9; k++) G[k] = gSig[(size_t)i * 9 + k];
    sigma_world_bwd_of(q4, s3, G, gq, gs);
#pragma unroll
    for (int k = 0; k < 4; k++) g_q[i * 4 + k] = gq[k];
#pragma unroll
    for (int k = 0; k < 3; k++) g_scale[i * 3 + k] = gs[k];
}

// projection.cu:155-175
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_jacobian(const T* __restrict__ xyz,
                                                       const T* __restrict__ K, int N,
                                                       T* __restrict__ J) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    J[i * 6 + 0] = K[0] / z;
    J[i * 6 + 1] = 0;
    J[i * 6 + 2] = -K[0] * x / (z * z);
    J[i * 6 + 3] = 0;
    J[i * 6 + 4] = K[4] / z;
    J[i * 6 + 5] = -K[4] * y / (z * z);
}

// projection_backward.cu:93-120
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_jacobian_bwd(const T* __restrict__ xyz,
                                                           const T* __restrict__ K,
                                                           const T* __restrict__ gJ, int N,
                                                           T* __restrict__ g_xyz) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const T fx = K[0], fy = K[4];
    const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    const T z2 = z * z, z3 = z * z * z;
    g_xyz[i * 3 + 0] = gJ[i * 6 + 2] * -fx / z2;
    g_xyz[i * 3 + 1] = gJ[i * 6 + 5] * -fy / z2;
    g_xyz[i * 3 + 2] = gJ[i * 6 + 0] * -fx / z2 + gJ[i * 6 + 4] * -fy / z2 +
                       gJ[i * 6 + 2] * 2 * x * fx / z3 + gJ[i * 6 + 5] * 2 * y * fy / z3;
}

template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_conic(const T* __restrict__ sigma,
                                                    const T* __restrict__ J,
                                                    const T* __restrict__ M, int N,
                                                    T* __restrict__ conic) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    T W[9], J6[6], S9[9], c3[3];
    load_rotation(M, W);
#pragma unroll
    for (int k = 0; k < 6; k++) J6[k] = J[(size_t)i * 6 + k];
#pragma unroll
    for (int k = 0; k < 9; k++) S9[k] = sigma[(size_t)i * 9 + k];
    conic_of(J6, W, S9, c3);
    conic[i * 3 + 0] = c3[0];
    conic[i * 3 + 1] = c3[1];
    conic[i * 3 + 2] = c3[2];
}

template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_conic_bwd(const T* __restrict__ sigma,
                                                        const T* __restrict__ J,
                                                        const T* __restrict__ M,
                                                        const T* __restrict__ g_conic, int N,
                                                        T* __restrict__ g_sigma,
                                                        T* __restrict__ g_J) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= N) return;
    T W[9], J6[6], S9[9], gc3[3], gS9[9], gJ6[6];
    load_rotation(M, W);
#pragma unroll
    for (int k = 0; k < 6; k++) J6[k] = J[(size_t)i * 6 + k];
#pragma unroll
    for (int k = 0; k < 9; k++) S9[k] = sigma[(size_t)i * 9 + k];
    gc3[0] = g_conic[i * 3 + 0]; gc3[1] = g_conic[i * 3 + 1]; gc3[2] = g_conic[i * 3 + 2];
    conic_bwd_of(J6, W, S9, gc3, gS9, gJ6);
#pragma unroll
    for (int k = 0; k < 9; k++) g_sigma[(size_t)i * 9 + k] = gS9[k];
#pragma unroll
    for (int k = 0; k < 6; k++) g_J[(size_t)i * 6 + k] = gJ6[k];
}

// precompute_sh.cu:8-58
template <typename T, int N_SH>
__global__ __launch_bounds__(PG_BLOCK) void k_sh_rgb(const T* __restrict__ xyz,
                                                     const T* __restrict__ sh,
                                                     const T* __restrict__ M, int N,
                                                     T* __restrict__ rgb) {
    const int g = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (g >= N) return;
    if constexpr (N_SH == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) rgb[g * 3 + c] = sh[g * 3 + c];
    } else {
        T d[3], Y[N_SH];
        view_dir_of(xyz, M, g, d);
        sh_basis<T, N_SH>(d, Y);
        const T* co = sh + (size_t)g * N_SH * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            T t = 0;
#pragma unroll
            for (int s = 0; s < N_SH; s++) t += Y[s] * co[N_SH * c + s];
            t *= T(GS_R_SH_0);
            rgb[g * 3 + c] = t;
        }
    }
}

// precompute_sh.cu:61-111
template <typename T, int N_SH>
__global__ __launch_bounds__(PG_BLOCK) void k_sh_rgb_bwd(const T* __restrict__ xyz,
                                                         const T* __restrict__ M,
                                                         const T* __restrict__ g_rgb, int N,
                                                         T* __restrict__ g_sh) {
    const int g = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (g >= N) return;
    if constexpr (N_SH == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) g_sh[g * 3 + c] = g_rgb[g * 3 + c];
    } else {
        T d[3], Y[N_SH];
        view_dir_of(xyz, M, g, d);
        sh_basis<T, N_SH>(d, Y);
        T* out = g_sh + (size_t)g * N_SH * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const T gl = g_rgb[g * 3 + c] * T(GS_R_SH_0);
#pragma unroll
            for (int s = 0; s < N_SH; s++) out[N_SH * c + s] = gl * Y[s];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_pack(const T* __restrict__ uvs,
                                                   const T* __restrict__ opacity,
                                                   const T* __restrict__ conic,
                                                   const T* __restrict__ rgb, int V,
                                                   T* __restrict__ packed) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= V) return;
    const T c3[3] = {conic[i * 3 + 0], conic[i * 3 + 1], conic[i * 3 + 2]};
    T col[3] = {0, 0, 0};
    if (rgb) { col[0] = rgb[i * 3 + 0]; col[1] = rgb[i * 3 + 1]; col[2] = rgb[i * 3 + 2]; }
    T p[GS_PACKED_WIDTH];
    pack_record<T>(uvs[i * 2 + 0], uvs[i * 2 + 1], c3, opacity[i], col, p);
#pragma unroll
    for (int k = 0; k < GS_PACKED_WIDTH; k++) packed[(size_t)i * GS_PACKED_WIDTH + k] = p[k];
}

template <typename F>
inline int launch1d(int N, const char* what, F f) {
    if (N <= 0) return GS_OK;
    f(dim3(div_up(N, PG_BLOCK)), dim3(PG_BLOCK));
    return check_launch(what);
}

}  // namespace gs

using namespace gs;

#define DISPATCH_T(dtype, CALL)                                                                    \
    if ((dtype) == GS_F32) {                                                                       \
        using T = float;                                                                           \
        CALL;                                                                                      \
    } else if ((dtype) == GS_F64) {                                                                \
        using T = double;                                                                          \
        CALL;                                                                                      \
    } else {                                                                                       \
        gs::set_error("Inputs must be float32 or float64");                                        \
        return GS_EINVAL;                                                                          \
    }

#define DISPATCH_SH(n_sh, CALL)                                                                    \
    switch (n_sh) {                                                                                \
        case 1: { constexpr int N_SH = 1; CALL; } break;                                           \
        case 4: { constexpr int N_SH = 4; CALL; } break;                                           \
        case 9: { constexpr int N_SH = 9; CALL; } break;                                           \
        case 16: { constexpr int N_SH = 16; CALL; } break;                                         \
        default:                                                                                   \
            gs::set_error("Unsupported number of SH coefficients: %d", n_sh);                      \
            return GS_EINVAL;                                                                      \
    }

extern "C" {

int gs_camera_projection(const void* xyz, const void* K, int N, void* uv, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "camera_projection", [&](dim3 g, dim3 b) {
                   k_camera_projection<T><<<g, b, 0, s>>>((const T*)xyz, (const T*)K, N, (T*)uv);
               }));
}

int gs_camera_projection_backward(const void* xyz, const void* K, const void* uv_grad_out, int N,
                                  void* xyz_grad_in, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "camera_projection_backward", [&](dim3 g, dim3 b) {
                   k_camera_projection_bwd<T><<<g, b, 0, s>>>((const T*)xyz, (const T*)K,
                                                              (const T*)uv_grad_out, N,
                                                              (T*)xyz_grad_in);
               }));
}

int gs_compute_sigma_world(const void* quaternion, const void* scale, int N, void* sigma_world,
                           int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "compute_sigma_world", [&](dim3 g, dim3 b) {
                   k_sigma_world<T><<<g, b, 0, s>>>((const T*)quaternion, (const T*)scale, N,
                                                    (T*)sigma_world);
               }));
}

int gs_compute_sigma_world_backward(const void* quaternion, const void* scale,
                                    const void* sigma_world_grad_out, int N,
                                    void* quaternion_grad_in, void* scale_grad_in, int dtype,
                                    void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "compute_sigma_world_backward", [&](dim3 g, dim3 b) {
                   k_sigma_world_bwd<T><<<g, b, 0, s>>>((const T*)quaternion, (const T*)scale,
                                                        (const T*)sigma_world_grad_out, N,
                                                        (T*)quaternion_grad_in, (T*)scale_grad_in);
               }));
}

int gs_compute_projection_jacobian(const void* xyz, const void* K, int N, void* J, int dtype,
                                   void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "compute_projection_jacobian", [&](dim3 g, dim3 b) {
                   k_jacobian<T><<<g, b, 0, s>>>((const T*)xyz, (const T*)K, N, (T*)J);
               }));
}

int gs_compute_projection_jacobian_backward(const void* xyz, const void* K,
                                            const void* jac_grad_out, int N, void* xyz_grad_in,
                                            int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
               return launch1d(N, "compute_projection_jacobian_backward", [&](dim3 g, dim3 b) {
                   k_jacobian_bwd<T><<<g, b, 0, s>>>((const T*)xyz, (const T*)K,
                                                     (const T*)jac_grad_out, N, (T*)xyz_grad_in);
               }));
}

int gs_compute_conic(const void* sigma_world, const void* J, const void* camera_T_world, int N,
                     void* conic, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "compute_conic", [&](dim3 g, dim3 b) {
                   k_conic<T><<<g, b, 0, s>>>((const T*)sigma_world, (const T*)J,
                                              (const T*)camera_T_world, N, (T*)conic);
               }));
}

int gs_compute_conic_backward(const void* sigma_world, const void* J, const void* camera_T_world,
                              const void* conic_grad_out, int N, void* sigma_world_grad_in,
                              void* J_grad_in, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(N, "compute_conic_backward", [&](dim3 g, dim3 b) {
                   k_conic_bwd<T><<<g, b, 0, s>>>((const T*)sigma_world, (const T*)J,
                                                  (const T*)camera_T_world,
                                                  (const T*)conic_grad_out, N,
                                                  (T*)sigma_world_grad_in, (T*)J_grad_in);
               }));
}

int gs_precompute_rgb_from_sh(const void* xyz, const void* sh_coeff, const void* matrix, int N,
                              int n_sh, void* rgb, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_SH(n_sh, return launch1d(N, "precompute_rgb_from_sh",
                                                        [&](dim3 g, dim3 b) {
                                                            k_sh_rgb<T, N_SH><<<g, b, 0, s>>>(
                                                                (const T*)xyz, (const T*)sh_coeff,
                                                                (const T*)matrix, N, (T*)rgb);
                                                        })));
    return GS_OK;
}

int gs_precompute_rgb_from_sh_backward(const void* xyz, const void* matrix, const void* grad_rgb,
                                       int N, int n_sh, void* grad_sh, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_SH(n_sh, return launch1d(N, "precompute_rgb_from_sh_backward",
                                                        [&](dim3 g, dim3 b) {
                                                            k_sh_rgb_bwd<T, N_SH><<<g, b, 0, s>>>(
                                                                (const T*)xyz, (const T*)matrix,
                                                                (const T*)grad_rgb, N,
                                                                (T*)grad_sh);
                                                        })));
    return GS_OK;
}

int gs_pack_splats(const void* uvs, const void* opacity, const void* conic, const void* rgb, int V,
                   void* packed, int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(V, "pack_splats", [&](dim3 g, dim3 b) {
                   k_pack<T><<<g, b, 0, s>>>((const T*)uvs, (const T*)opacity, (const T*)conic,
                                             (const T*)rgb, V, (T*)packed);
               }));
}

}  // extern "C"
