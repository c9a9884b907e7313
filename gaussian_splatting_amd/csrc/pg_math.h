// pg_math.h -- per-Gaussian device math shared by the stand-alone kernels (per_gaussian.hip) and the
// fused preprocess kernels (preprocess.hip).  Forward functions keep the reference's operation order
// (bit-identical fp32 results); see gs_common.h for the arithmetic contract.
#pragma once
#include "gs_common.h"

namespace gs {

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


// Conservative squared cutoff radius of a splat for the fp32 alpha threshold (render.cu:145):
// alpha >= 1/255  <=>  mh^2 <= tau = 2 ln(255 opacity), and mh^2 = d' S^-1 d >= |d|^2 / lmax(S),
// so |d|^2 > tau * lmax  =>  alpha < 1/255.  The margin covers the rounding of the fp32
// quadratic form (relative error ~ condition number * 2^-24) and of log/exp; parity tests against
// the oracle (which evaluates every pixel) are bit-exact, so any non-conservative case would show.
__device__ inline float cutoff_r2(float a, float b, float c, float det, float opa) {
    if (!(det > 0.0f) || !(a > 0.0f) || !(c > 0.0f) || !(opa == opa)) return __builtin_inff();
    const float s = opa * 255.0f * 1.001f;   // margin: opacity within rounding of 1/255
    if (!(s > 1.0f)) return -1.0f;           // alpha <= opacity < 1/255 everywhere
    const float tau = 2.0f * __builtin_logf(s);
    const float half = 0.5f * (a + c);
    const float lmax = half + __builtin_sqrtf(0.25f * (a - c) * (a - c) + b * b);
    const float kappa = lmax * lmax / det;   // lmax / lmin
    return tau * lmax * (1.05f + 8e-6f * kappa) + 1e-3f;
}
template <typename T> __device__ inline T cutoff_r2_t(T a, T b, T c, T det, T opa);
template <> __device__ inline float cutoff_r2_t<float>(float a, float b, float c, float det, float opa) {
    return cutoff_r2(a, b, c, det, opa);
}
template <> __device__ inline double cutoff_r2_t<double>(double, double, double, double, double) {
    return (double)__builtin_inff();   // fp64 has no alpha threshold (render.cu:145: use_fast_exp)
}

// the packed render record (gs_pack_splats): the per-splat part of the per-pixel loop of
// render.cu:117-129 / render_backward.cu:141-152, hoisted (identical values)
template <typename T>
__device__ inline void pack_record(T u, T v, const T* conic3, T opa, const T* col3, T* p) {
    constexpr bool fast = sizeof(T) == 4;
    T a, c;
    const T b = conic3[1] * 0.5;
    if (fast) {
        a = conic3[0] + 0.25;
        c = conic3[2] + 0.25;
    } else {
        a = conic3[0];
        c = conic3[2];
    }
    const T det = a * c - b * b;
    const T rdet = 1.0 / det;
    p[0] = u; p[1] = v; p[2] = cutoff_r2_t<T>(a, b, c, det, opa); p[3] = opa;
    p[4] = a; p[5] = b; p[6] = c; p[7] = det;
    p[8] = rdet;
    // the splat's colour as the render loops use it with one coefficient per channel: sh_to_rgb's
    // Y0 * coefficient (spherical_harmonics.cuh:83), the same product formed once per splat
    p[9] = col3 ? T(GS_SH_0) * col3[0] : T(0);
    p[10] = col3 ? T(GS_SH_0) * col3[1] : T(0);
    p[11] = col3 ? T(GS_SH_0) * col3[2] : T(0);
}

// projection_backward.cu:174-315
template <typename T>
__device__ inline void sigma_world_bwd_of(const T* q4, const T* s3, const T* G, T* g_q4, T* g_s3) {
    const T e0 = gexp<T>(s3[0]), e1 = gexp<T>(s3[1]),
            e2 = gexp<T>(s3[2]);
    const T S[9] = {e0, 0, 0, 0, e1, 0, 0, 0, e2};
    const T qw = q4[0], qx = q4[1], qy = q4[2], qz = q4[3];
    const T norm_q = gsqrt<T>(qw * qw + qx * qx + qy * qy + qz * qz);
    const T w = qw / norm_q, x = qx / norm_q, y = qy / norm_q, z = qz / norm_q;
    T R[9];
    quat_to_rot(w, x, y, z, R);
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
    g_s3[0] = (gradS[0] + gradSRR[0]) * e0;
    g_s3[1] = (gradS[4] + gradSRR[4]) * e1;
    g_s3[2] = (gradS[8] + gradSRR[8]) * e2;
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
    g_q4[0] = (inv - qw * qw / n3) * gq[0] - qw * qx / n3 * gq[1] - qw * qy / n3 * gq[2] -
                     qw * qz / n3 * gq[3];
    g_q4[1] = -qw * qx / n3 * gq[0] + (inv - qx * qx / n3) * gq[1] - qx * qy / n3 * gq[2] -
                     qx * qz / n3 * gq[3];
    g_q4[2] = -qw * qy / n3 * gq[0] - qx * qy / n3 * gq[1] + (inv - qy * qy / n3) * gq[2] -
                     qy * qz / n3 * gq[3];
    g_q4[3] = -qw * qz / n3 * gq[0] - qx * qz / n3 * gq[1] - qy * qz / n3 * gq[2] +
                     (inv - qz * qz / n3) * gq[3];
}

}  // namespace gs
