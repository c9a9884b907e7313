// per_gaussian.hip -- one-thread-per-Gaussian kernels behind the eight projection entry points,
// the two SH-precompute entry points and gs_pack_splats.  HBM-bound streaming kernels: 256-thread
// workgroups, no LDS, no host synchronisation, launched on the caller's stream.
//
// Forward kernels keep the operation order and operand precisions of the reference so that their
// fp32 outputs are bit-identical to the CPU restatement (oracle/gs_oracle.cpp); backward kernels
// compute in T and are compared with a tolerance.
#include "gs_common.h"

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
__device__ inline void quat_to_rot(T qw, T qx, T qy, T qz, T* R) {   // projection.cu:72-80
    R[0] = 1 - 2 * qy * qy - 2 * qz * qz;
    R[1] = 2 * qx * qy - 2 * qz * qw;
    R[2] = 2 * qx * qz + 2 * qy * qw;
    R[3] = 2 * qx * qy + 2 * qz * qw;
    R[4] = 1 - 2 * qx * qx - 2 * qz * qz;
    R[5] = 2 * qy * qz - 2 * qx * qw;
    R[6] = 2 * qx * qz - 2 * qy * qw;
    R[7] = 2 * qy * qz + 2 * qx * qw;
    R[8] = 1 - 2 * qx * qx - 2 * qy * qy;
}

// projection.cu:57-109
template <typename T>
__device__ inline void sigma_world_of(const T* q4, const T* s3, T* S) {
    T qw = q4[0], qx = q4[1], qy = q4[2], qz = q4[3];
    const T norm = gsqrt<T>(qx * qx + qy * qy + qz * qz + qw * qw);
    qx /= norm; qy /= norm; qz /= norm; qw /= norm;
    T r[9];
    quat_to_rot(qw, qx, qy, qz, r);
    const T sx = gexp<T>(s3[0]), sy = gexp<T>(s3[1]), sz = gexp<T>(s3[2]);
    const T sx2 = sx * sx, sy2 = sy * sy, sz2 = sz * sz;
    S[0] = r[0] * r[0] * sx2 + r[1] * r[1] * sy2 + r[2] * r[2] * sz2;
    S[1] = r[0] * r[3] * sx2 + r[1] * r[4] * sy2 + r[2] * r[5] * sz2;
    S[2] = r[0] * r[6] * sx2 + r[1] * r[7] * sy2 + r[2] * r[8] * sz2;
    S[3] = S[1];
    S[4] = r[3] * r[3] * sx2 + r[4] * r[4] * sy2 + r[5] * r[5] * sz2;
    S[5] = r[3] * r[6] * sx2 + r[4] * r[7] * sy2 + r[5] * r[8] * sz2;
    S[6] = S[2];
    S[7] = S[5];
    S[8] = r[6] * r[6] * sx2 + r[7] * r[7] * sy2 + r[8] * r[8] * sz2;
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
    const T e0 = gexp<T>(scale[i * 3 + 0]), e1 = gexp<T>(scale[i * 3 + 1]),
            e2 = gexp<T>(scale[i * 3 + 2]);
    const T S[9] = {e0, 0, 0, 0, e1, 0, 0, 0, e2};
    const T qw = quat[i * 4 + 0], qx = quat[i * 4 + 1], qy = quat[i * 4 + 2], qz = quat[i * 4 + 3];
    const T norm_q = gsqrt<T>(qw * qw + qx * qx + qy * qy + qz * qz);
    const T w = qw / norm_q, x = qx / norm_q, y = qy / norm_q, z = qz / norm_q;
    T R[9];
    quat_to_rot(w, x, y, z, R);
    T G[9];
#pragma unroll
    for (int k = 0; k < 9; k++) G[k] = gSig[(size_t)i * 9 + k];
    T RS[9], gradRS[9], RSt[9], gradSR[9], gradR[9], SgradSR[9], Rt[9], gradS[9], gradSRR[9];
    matmul<T, 3, 3, 3>(R, S, RS);
    matmul<T, 3, 3, 3>(G, RS, gradRS);
    transp<T, 3, 3>(RS, RSt);
    matmul<T, 3, 3, 3>(RSt, G, gradSR);
    matmul<T, 3, 3, 3>(gradRS, S, gradR);
    matmul<T, 3, 3, 3>(S, gradSR, SgradSR);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) gradR[r * 3 + c] += SgradSR[c * 3 + r];
    transp<T, 3, 3>(R, Rt);
    matmul<T, 3, 3, 3>(Rt, gradRS, gradS);
    matmul<T, 3, 3, 3>(gradSR, R, gradSRR);
    g_scale[i * 3 + 0] = (gradS[0] + gradSRR[0]) * e0;
    g_scale[i * 3 + 1] = (gradS[4] + gradSRR[4]) * e1;
    g_scale[i * 3 + 2] = (gradS[8] + gradSRR[8]) * e2;
    T gq[4];
    gq[0] = -2 * z * gradR[1] + 2 * y * gradR[2] + 2 * z * gradR[3] - 2 * x * gradR[5] -
            2 * y * gradR[6] + 2 * x * gradR[7];
    gq[1] = 2 * y * gradR[1] + 2 * z * gradR[2] + 2 * y * gradR[3] - 4 * x * gradR[4] -
            2 * w * gradR[5] + 2 * z * gradR[6] + 2 * w * gradR[7] - 4 * x * gradR[8];
    gq[2] = -4 * y * gradR[0] + 2 * x * gradR[1] + 2 * w * gradR[2] + 2 * x * gradR[3] +
            2 * z * gradR[5] - 2 * w * gradR[6] + 2 * z * gradR[7] - 4 * y * gradR[8];
    gq[3] = -4 * z * gradR[0] - 2 * w * gradR[1] + 2 * x * gradR[2] + 2 * w * gradR[3] -
            4 * z * gradR[4] + 2 * y * gradR[5] + 2 * x * gradR[6] + 2 * y * gradR[7];
    const T n3 = norm_q * norm_q * norm_q;
    const T inv = T(1) / norm_q;
    g_q[i * 4 + 0] = (inv - qw * qw / n3) * gq[0] - qw * qx / n3 * gq[1] - qw * qy / n3 * gq[2] -
                     qw * qz / n3 * gq[3];
    g_q[i * 4 + 1] = -qw * qx / n3 * gq[0] + (inv - qx * qx / n3) * gq[1] - qx * qy / n3 * gq[2] -
                     qx * qz / n3 * gq[3];
    g_q[i * 4 + 2] = -qw * qy / n3 * gq[0] - qx * qy / n3 * gq[1] + (inv - qy * qy / n3) * gq[2] -
                     qy * qz / n3 * gq[3];
    g_q[i * 4 + 3] = -qw * qz / n3 * gq[0] - qx * qz / n3 * gq[1] - qy * qz / n3 * gq[2] +
                     (inv - qz * qz / n3) * gq[3];
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
__device__ inline void load_rotation(const T* __restrict__ M, T* W) {   // projection.cu:226-235
    W[0] = M[0]; W[1] = M[1]; W[2] = M[2];
    W[3] = M[4]; W[4] = M[5]; W[5] = M[6];
    W[6] = M[8]; W[7] = M[9]; W[8] = M[10];
}

// projection.cu:214-257
template <typename T>
__device__ inline void conic_of(const T* J6, const T* W, const T* S9, T* conic3) {
    T JW[6], JWS[6], JWt[6], S2[4];
    matmul<T, 2, 3, 3>(J6, W, JW);
    matmul<T, 2, 3, 3>(JW, S9, JWS);
    transp<T, 2, 3>(JW, JWt);
    matmul<T, 2, 3, 2>(JWS, JWt, S2);
    conic3[0] = S2[0];
    conic3[1] = S2[1] + S2[2];
    conic3[2] = S2[3];
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

// projection_backward.cu:385-471
template <typename T>
__device__ inline void conic_bwd_of(const T* J6, const T* W, const T* S9, const T* gc3, T* gS9,
                                    T* gJ6) {
    T JW[6], JWt[6], G2[4], A[6], SJ[6], L[6], St[9], StJ[6], Rr[6], gJWt[6], gJt[6];
    matmul<T, 2, 3, 3>(J6, W, JW);
    transp<T, 2, 3>(JW, JWt);
    G2[0] = gc3[0]; G2[1] = gc3[1]; G2[2] = gc3[1]; G2[3] = gc3[2];
    matmul<T, 3, 2, 2>(JWt, G2, A);
    matmul<T, 3, 2, 3>(A, JW, gS9);
    matmul<T, 3, 3, 2>(S9, JWt, SJ);
    matmul<T, 3, 2, 2>(SJ, G2, L);   // G2 is symmetric: its transpose is itself
    transp<T, 3, 3>(S9, St);
    matmul<T, 3, 3, 2>(St, JWt, StJ);
    matmul<T, 3, 2, 2>(StJ, G2, Rr);
#pragma unroll
    for (int k = 0; k < 6; k++) gJWt[k] = L[k] + Rr[k];
    matmul<T, 3, 3, 2>(W, gJWt, gJt);
    transp<T, 3, 2>(gJt, gJ6);
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

// precompute_sh.cu:28-39 ; rsqrt -> 1/sqrt (IEEE)
template <typename T>
__device__ inline void view_dir_of(const T* __restrict__ xyz, const T* __restrict__ M, int g,
                                   T* d) {
    d[0] = xyz[g * 3 + 0] - M[3];
    d[1] = xyz[g * 3 + 1] - M[7];
    d[2] = xyz[g * 3 + 2] - M[11];
    const T r = T(1) / gsqrt<T>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] *= r; d[1] *= r; d[2] *= r;
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

// render.cu:117-129 / render_backward.cu:141-152: the per-splat part of the per-pixel loop,
// hoisted (identical values: it depends on the splat only)
template <typename T>
__global__ __launch_bounds__(PG_BLOCK) void k_pack(const T* __restrict__ uvs,
                                                   const T* __restrict__ opacity,
                                                   const T* __restrict__ conic, int V,
                                                   T* __restrict__ packed) {
    const int i = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (i >= V) return;
    constexpr bool fast = sizeof(T) == 4;
    T a, c;
    const T b = conic[i * 3 + 1] * 0.5;
    if (fast) {
        a = conic[i * 3 + 0] + 0.25;
        c = conic[i * 3 + 2] + 0.25;
    } else {
        a = conic[i * 3 + 0];
        c = conic[i * 3 + 2];
    }
    const T det = a * c - b * b;
    const T rdet = 1.0 / det;
    T* p = packed + (size_t)i * GS_PACKED_WIDTH;
    p[0] = uvs[i * 2 + 0];
    p[1] = uvs[i * 2 + 1];
    p[2] = a;
    p[3] = b;
    p[4] = c;
    p[5] = det;
    p[6] = rdet;
    p[7] = opacity[i];
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

int gs_pack_splats(const void* uvs, const void* opacity, const void* conic, int V, void* packed,
                   int dtype, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, return launch1d(V, "pack_splats", [&](dim3 g, dim3 b) {
                   k_pack<T><<<g, b, 0, s>>>((const T*)uvs, (const T*)opacity, (const T*)conic, V,
                                             (T*)packed);
               }));
}

}  // extern "C"
