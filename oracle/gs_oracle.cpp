// gs_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// A literal CPU restatement of the reference's per-frame rasterization hot path
// (joeyan/gaussian_splatting, src/*.cu).  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library; the product path
// (gaussian_splatting_amd/) never does and fails loudly without its HIP library.
//
// Parity status: PINNED against the reference's own known-answer tests
// (test/test_projection.py, test/test_tile_culling.py, test/test_rasterize.py,
// test/test_depth.py) -- see tests/test_oracle_golden.py.  fp32 gradients, multi-chunk
// tiles and alpha-skip behaviour are not pinned by any reference test.
//
// Arithmetic contract (what "literal" means here):
//  * Expressions are written with the same operand types and literal types as the
//    reference so that C++ promotion rules give the same intermediate precisions
//    (e.g. `alpha * (1.0 - alpha_accum)` is evaluated in double and narrowed, exactly as
//    render.cu:150 does).  Compile with -ffp-contract=off; no fast-math.
//  * Functions that are not IEEE-defined are substituted by deterministic IEEE-only
//    restatements so that CPU and GPU agree bit-for-bit:
//      __expf(x), exp(float)  -> det_expf (6th-order polynomial 2^f, <= 2 ulp), mode 0
//                                (mode 1 = libm expf, for cross-checking only)
//      rsqrt(x)               -> 1 / sqrt(x)
//      atan2f/cosf/sinf (OBB) -> algebraic cos/sin of the eigenvector angle, mode 0
//                                (mode 1 = libm atan2f/cosf/sinf: the literal form)
//    fp64 instantiations use libm exp/sqrt.
//  * float->int conversions follow CUDA semantics (saturating, NaN -> 0).
//
// Each function cites the reference file:line it follows.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

int g_exp_mode = 0;   // 0 deterministic polynomial, 1 libm
int g_trig_mode = 0;  // 0 algebraic, 1 libm (literal)
// 0 = band-1 axes as shipped in spherical_harmonics.cuh:39-42 (y, z, x) -- what the CUDA path
// computes; 1 = (x, y, z) as in analytic_diff.ipynb, the convention the constants in
// test/test_rasterize.py:85-92,124-131 were generated with.  Mode 1 exists only
// to pin the rest of the SH machinery against those constants.
int g_sh_band1_mode = 0;
// Checker aids (no reference counterpart):
// g_bwd_abs: render_tiles_backward returns, per gradient element, the sum over pixels of the magnitudes
//   of the LEAF terms of its formula (every product entering a sum or difference) -- the scale against
//   which an fp32 evaluation's rounding + summation-order noise is measured
//   (tests/helpers.py: noise_normalised_err).
// g_contrib_count: render_tiles additionally writes, per pixel, how many splats passed the
//   alpha >= 1/255 test before the pixel saturated (render.cu:145-163) -- two runs whose inputs differ
//   in the last ulp took the same decisions at a pixel iff this count and num_splats agree.
int g_bwd_abs = 0;
// g_bwd_sum: how render_tiles_backward sums the per-pixel fp32 terms of a gradient element.  0 (the checker's
// default): in double, rounded once.  1 / 2: in fp32, as every GPU implementation must (the reference: warp
// reduce + atomicAdd, order unspecified) -- 1 = pixels of a tile in ascending row-major order, tiles ascending;
// 2 = both descending.  Two equally valid fp32 summation orders of the SAME terms: their difference is the
// summation-order noise of the criterion SURVEY.md 8(d) writes down (tests/test_grad_noise_floor.py).
int g_bwd_sum = 0;
int g_q1_exact = 0;   // 1: the transmittance update uses the global splat index (exact gradient) instead of
                      // render_backward.cu:185's chunk-local one -- the checker of GS_BACKWARD_EXACT
int* g_contrib_count = nullptr;

// ---------------------------------------------------------------------------------------
// deterministic exp for fp32 (IEEE ops only; explicit fma)
// ---------------------------------------------------------------------------------------
inline float det_expf(float x) {
    const float t = x * 1.44269504088896341f;
    if (!(t > -125.0f)) return (t != t) ? t : 0.0f;   // underflow -> 0, NaN propagates
    if (t > 127.0f) return INFINITY;
    const float n = nearbyintf(t);
    const float f = t - n;
    float p = 1.54035303933816e-4f;
    p = fmaf(p, f, 1.33335581464284e-3f);
    p = fmaf(p, f, 9.61812910762848e-3f);
    p = fmaf(p, f, 5.55041086648216e-2f);
    p = fmaf(p, f, 2.40226506959101e-1f);
    p = fmaf(p, f, 6.93147180559945e-1f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

template <typename T> inline T exp_accurate(T x);   // stands for exp() on T
template <> inline float exp_accurate<float>(float x) { return g_exp_mode ? expf(x) : det_expf(x); }
template <> inline double exp_accurate<double>(double x) { return exp(x); }

// stands for __expf (fp32 fast mode) / exp (fp64) inside the render kernels
inline float exp_fast(float x) { return g_exp_mode ? expf(x) : det_expf(x); }

template <typename T> inline T rsqrt_t(T x) { return T(1) / std::sqrt(x); }

// CUDA float -> int conversion (cvt.rzi.s32.f32): saturating, NaN -> 0
inline int f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}

// matrix.cuh:15-30  C = A(ra x ca) * B(ca x cb), accumulate from 0 in index order
template <typename T>
inline void matmul(const T* A, const T* B, T* C, int ra, int ca, int cb) {
    for (int r = 0; r < ra; r++)
        for (int c = 0; c < cb; c++) {
            T sum = 0;
            for (int k = 0; k < ca; k++) sum += A[r * ca + k] * B[k * cb + c];
            C[r * cb + c] = sum;
        }
}
// matrix.cuh:4-13
template <typename T>
inline void transp(const T* A, T* At, int rows, int cols) {
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) At[c * rows + r] = A[r * cols + c];
}

// spherical_harmonics.cuh:4-24 (float constants)
const float SH_0 = 0.28209479177387814;
const float r_SH_0 = 3.544907701811032;
const float SH_1[3] = {-0.4886025119029199, 0.4886025119029199, -0.4886025119029199};
const float SH_2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                       -1.0925484305920792, 0.5462742152960396};
const float SH_3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                       0.263875515352797,   -0.4570457994644658, 1.445305721320277,
                       -0.5900435899266435};

// spherical_harmonics.cuh:26-72  basis at a view direction, n_sh in {1,4,9,16}
template <typename T>
inline void sh_basis(const T* d, int n_sh, T* Y) {
    Y[0] = T(SH_0);
    if (n_sh < 4) return;
    const T x = d[0], y = d[1], z = d[2];
    if (g_sh_band1_mode == 0) {
        Y[1] = T(SH_1[0]) * y;
        Y[2] = T(SH_1[1]) * z;
        Y[3] = T(SH_1[2]) * x;
    } else {
        Y[1] = T(SH_1[0]) * x;
        Y[2] = T(SH_1[1]) * y;
        Y[3] = T(SH_1[2]) * z;
    }
    if (n_sh < 9) return;
    const T xy = x * y, yz = y * z, xz = x * z, xx = x * x, yy = y * y, zz = z * z;
    Y[4] = T(SH_2[0]) * xy;
    Y[5] = T(SH_2[1]) * yz;
    Y[6] = T(SH_2[2]) * (3 * zz - 1.0);   // double literal: evaluated in double, narrowed
    Y[7] = T(SH_2[3]) * xz;
    Y[8] = T(SH_2[4]) * (xx - yy);
    if (n_sh < 16) return;
    Y[9] = T(SH_3[0]) * y * (3 * xx - yy);
    Y[10] = T(SH_3[1]) * xy * z;
    Y[11] = T(SH_3[2]) * y * (5 * zz - 1.0);
    Y[12] = T(SH_3[3]) * z * (5 * zz - 3.0);
    Y[13] = T(SH_3[4]) * x * (5 * zz - 1.0);
    Y[14] = T(SH_3[5]) * z * (xx - yy);
    Y[15] = T(SH_3[6]) * x * (xx - 3 * yy);
}

// spherical_harmonics.cuh:74-96
template <typename T>
inline void sh_to_rgb(const T* coeff, const T* Y, int n_sh, T* rgb) {
    for (int c = 0; c < 3; c++) rgb[c] = Y[0] * coeff[n_sh * c];
    if (n_sh < 4) return;
    for (int s = 1; s < n_sh; s++)
        for (int c = 0; c < 3; c++) rgb[c] += Y[s] * coeff[n_sh * c + s];
}

// ---------------------------------------------------------------------------------------
// per-Gaussian forward kernels  (projection.cu)
// ---------------------------------------------------------------------------------------
// projection.cu:9-19
template <typename T>
void camera_projection(const T* xyz, const T* K, int N, T* uv) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        uv[i * 2 + 0] = K[0] * xyz[i * 3 + 0] / xyz[i * 3 + 2] + K[2];
        uv[i * 2 + 1] = K[4] * xyz[i * 3 + 1] / xyz[i * 3 + 2] + K[5];
    }
}

// projection.cu:57-109
template <typename T>
void compute_sigma_world(const T* q, const T* scale, int N, T* sigma) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        T qw = q[i * 4 + 0], qx = q[i * 4 + 1], qy = q[i * 4 + 2], qz = q[i * 4 + 3];
        const T norm = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
        qx /= norm; qy /= norm; qz /= norm; qw /= norm;
        const T r00 = 1 - 2 * qy * qy - 2 * qz * qz;
        const T r01 = 2 * qx * qy - 2 * qz * qw;
        const T r02 = 2 * qx * qz + 2 * qy * qw;
        const T r10 = 2 * qx * qy + 2 * qz * qw;
        const T r11 = 1 - 2 * qx * qx - 2 * qz * qz;
        const T r12 = 2 * qy * qz - 2 * qx * qw;
        const T r20 = 2 * qx * qz - 2 * qy * qw;
        const T r21 = 2 * qy * qz + 2 * qx * qw;
        const T r22 = 1 - 2 * qx * qx - 2 * qy * qy;
        const T sx = exp_accurate<T>(scale[i * 3 + 0]);
        const T sy = exp_accurate<T>(scale[i * 3 + 1]);
        const T sz = exp_accurate<T>(scale[i * 3 + 2]);
        const T sx2 = sx * sx, sy2 = sy * sy, sz2 = sz * sz;
        T* S = sigma + (size_t)i * 9;
        S[0] = r00 * r00 * sx2 + r01 * r01 * sy2 + r02 * r02 * sz2;
        S[1] = r00 * r10 * sx2 + r01 * r11 * sy2 + r02 * r12 * sz2;
        S[2] = r00 * r20 * sx2 + r01 * r21 * sy2 + r02 * r22 * sz2;
        S[3] = r00 * r10 * sx2 + r01 * r11 * sy2 + r02 * r12 * sz2;
        S[4] = r10 * r10 * sx2 + r11 * r11 * sy2 + r12 * r12 * sz2;
        S[5] = r10 * r20 * sx2 + r11 * r21 * sy2 + r12 * r22 * sz2;
        S[6] = r00 * r20 * sx2 + r01 * r21 * sy2 + r02 * r22 * sz2;
        S[7] = r10 * r20 * sx2 + r11 * r21 * sy2 + r12 * r22 * sz2;
        S[8] = r20 * r20 * sx2 + r21 * r21 * sy2 + r22 * r22 * sz2;
    }
}

// projection.cu:155-175
template <typename T>
void compute_projection_jacobian(const T* xyz, const T* K, int N, T* J) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        J[i * 6 + 0] = K[0] / z;
        J[i * 6 + 1] = 0;
        J[i * 6 + 2] = -K[0] * x / (z * z);
        J[i * 6 + 3] = 0;
        J[i * 6 + 4] = K[4] / z;
        J[i * 6 + 5] = -K[4] * y / (z * z);
    }
}

template <typename T>
inline void rotation_of(const T* cam_T_world, T* W) {   // projection.cu:226-235
    W[0] = cam_T_world[0]; W[1] = cam_T_world[1]; W[2] = cam_T_world[2];
    W[3] = cam_T_world[4]; W[4] = cam_T_world[5]; W[5] = cam_T_world[6];
    W[6] = cam_T_world[8]; W[7] = cam_T_world[9]; W[8] = cam_T_world[10];
}

// projection.cu:214-257
template <typename T>
void compute_conic(const T* sigma, const T* J, const T* cam_T_world, int N, T* conic) {
    T W[9];
    rotation_of(cam_T_world, W);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        T JW[6], JWS[6], JWt[6], S2[4];
        matmul(J + (size_t)i * 6, W, JW, 2, 3, 3);
        matmul(JW, sigma + (size_t)i * 9, JWS, 2, 3, 3);
        transp(JW, JWt, 2, 3);
        matmul(JWS, JWt, S2, 2, 3, 2);
        conic[i * 3 + 0] = S2[0];
        conic[i * 3 + 1] = S2[1] + S2[2];
        conic[i * 3 + 2] = S2[3];
    }
}

// ---------------------------------------------------------------------------------------
// per-Gaussian backward kernels  (projection_backward.cu)
// ---------------------------------------------------------------------------------------
// projection_backward.cu:9-36
template <typename T>
void camera_projection_backward(const T* xyz, const T* K, const T* g_uv, int N, T* g_xyz) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        if (xyz[i * 3 + 2] <= 0.0) continue;   // leaves the (zero-initialised) output untouched
        const T du_dx = K[0] / xyz[i * 3 + 2];
        const T dv_dy = K[4] / xyz[i * 3 + 2];
        const T du_dz = -K[0] * xyz[i * 3 + 0] / (xyz[i * 3 + 2] * xyz[i * 3 + 2]);
        const T dv_dz = -K[4] * xyz[i * 3 + 1] / (xyz[i * 3 + 2] * xyz[i * 3 + 2]);
        g_xyz[i * 3 + 0] = g_uv[i * 2 + 0] * du_dx;
        g_xyz[i * 3 + 1] = g_uv[i * 2 + 1] * dv_dy;
        g_xyz[i * 3 + 2] = g_uv[i * 2 + 0] * du_dz + g_uv[i * 2 + 1] * dv_dz;
    }
}

// projection_backward.cu:93-120
template <typename T>
void compute_projection_jacobian_backward(const T* xyz, const T* K, const T* gJ, int N, T* g_xyz) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const T fx = K[0], fy = K[4];
        const T x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
        g_xyz[i * 3 + 0] = gJ[i * 6 + 2] * -fx / (z * z);
        g_xyz[i * 3 + 1] = gJ[i * 6 + 5] * -fy / (z * z);
        g_xyz[i * 3 + 2] = gJ[i * 6 + 0] * -fx / (z * z) + gJ[i * 6 + 4] * -fy / (z * z) +
                           gJ[i * 6 + 2] * 2 * x * fx / (z * z * z) +
                           gJ[i * 6 + 5] * 2 * y * fy / (z * z * z);
    }
}

// projection_backward.cu:174-315
template <typename T>
void compute_sigma_world_backward(const T* quat, const T* scale, const T* gSig, int N, T* g_q,
                                  T* g_scale) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        T S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        S[0] = exp_accurate<T>(scale[i * 3 + 0]);
        S[4] = exp_accurate<T>(scale[i * 3 + 1]);
        S[8] = exp_accurate<T>(scale[i * 3 + 2]);
        const T qw = quat[i * 4 + 0], qx = quat[i * 4 + 1], qy = quat[i * 4 + 2],
                qz = quat[i * 4 + 3];
        const T norm_q = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
        const T w = qw / norm_q, x = qx / norm_q, y = qy / norm_q, z = qz / norm_q;
        T R[9];
        R[0] = 1.0 - 2.0 * y * y - 2.0 * z * z;
        R[1] = 2.0 * x * y - 2.0 * z * w;
        R[2] = 2.0 * x * z + 2 * y * w;
        R[3] = 2.0 * x * y + 2 * z * w;
        R[4] = 1.0 - 2.0 * x * x - 2.0 * z * z;
        R[5] = 2.0 * y * z - 2.0 * x * w;
        R[6] = 2.0 * x * z - 2.0 * y * w;
        R[7] = 2.0 * y * z + 2.0 * x * w;
        R[8] = 1.0 - 2.0 * x * x - 2.0 * y * y;
        const T* G = gSig + (size_t)i * 9;
        T RS[9], gradRS[9], RSt[9], gradSR[9], gradR[9], SgradSR[9], Rt[9], gradS[9], gradSRR[9];
        matmul(R, S, RS, 3, 3, 3);
        matmul(G, RS, gradRS, 3, 3, 3);
        transp(RS, RSt, 3, 3);
        matmul(RSt, G, gradSR, 3, 3, 3);
        matmul(gradRS, S, gradR, 3, 3, 3);
        matmul(S, gradSR, SgradSR, 3, 3, 3);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) gradR[r * 3 + c] += SgradSR[c * 3 + r];
        transp(R, Rt, 3, 3);
        matmul(Rt, gradRS, gradS, 3, 3, 3);
        matmul(gradSR, R, gradSRR, 3, 3, 3);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) gradS[r * 3 + c] += gradSRR[c * 3 + r];
        g_scale[i * 3 + 0] = gradS[0] * exp_accurate<T>(scale[i * 3 + 0]);
        g_scale[i * 3 + 1] = gradS[4] * exp_accurate<T>(scale[i * 3 + 1]);
        g_scale[i * 3 + 2] = gradS[8] * exp_accurate<T>(scale[i * 3 + 2]);
        T gq[4];
        gq[0] = -2.0 * z * gradR[1] + 2.0 * y * gradR[2] + 2.0 * z * gradR[3] - 2.0 * x * gradR[5] -
                2.0 * y * gradR[6] + 2.0 * x * gradR[7];
        gq[1] = 2.0 * y * gradR[1] + 2.0 * z * gradR[2] + 2.0 * y * gradR[3] - 4.0 * x * gradR[4] -
                2.0 * w * gradR[5] + 2.0 * z * gradR[6] + 2.0 * w * gradR[7] - 4.0 * x * gradR[8];
        gq[2] = -4.0 * y * gradR[0] + 2.0 * x * gradR[1] + 2.0 * w * gradR[2] + 2.0 * x * gradR[3] +
                2.0 * z * gradR[5] - 2.0 * w * gradR[6] + 2.0 * z * gradR[7] - 4.0 * y * gradR[8];
        gq[3] = -4.0 * z * gradR[0] - 2.0 * w * gradR[1] + 2.0 * x * gradR[2] + 2.0 * w * gradR[3] -
                4.0 * z * gradR[4] + 2.0 * y * gradR[5] + 2.0 * x * gradR[6] + 2.0 * y * gradR[7];
        const T n3 = norm_q * norm_q * norm_q;
        g_q[i * 4 + 0] = (1.0 / norm_q - qw * qw / n3) * gq[0] - qw * qx / n3 * gq[1] -
                         qw * qy / n3 * gq[2] - qw * qz / n3 * gq[3];
        g_q[i * 4 + 1] = -qw * qx / n3 * gq[0] + (1.0 / norm_q - qx * qx / n3) * gq[1] -
                         qx * qy / n3 * gq[2] - qx * qz / n3 * gq[3];
        g_q[i * 4 + 2] = -qw * qy / n3 * gq[0] - qx * qy / n3 * gq[1] +
                         (1.0 / norm_q - qy * qy / n3) * gq[2] - qy * qz / n3 * gq[3];
        g_q[i * 4 + 3] = -qw * qz / n3 * gq[0] - qx * qz / n3 * gq[1] - qy * qz / n3 * gq[2] +
                         (1.0 / norm_q - qz * qz / n3) * gq[3];
    }
}

// projection_backward.cu:385-471
template <typename T>
void compute_conic_backward(const T* sigma, const T* J, const T* cam_T_world, const T* g_conic,
                            int N, T* g_sigma, T* g_J) {
    T W[9];
    rotation_of(cam_T_world, W);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        T JW[6], JWt[6], G2[4], A[6], SJ[6], G2t[4], L[6], St[9], StJ[6], Rr[6], gJWt[6], gJt[6];
        matmul(J + (size_t)i * 6, W, JW, 2, 3, 3);
        transp(JW, JWt, 2, 3);
        G2[0] = g_conic[i * 3 + 0];
        G2[1] = g_conic[i * 3 + 1];
        G2[2] = g_conic[i * 3 + 1];
        G2[3] = g_conic[i * 3 + 2];
        matmul(JWt, G2, A, 3, 2, 2);
        matmul(A, JW, g_sigma + (size_t)i * 9, 3, 2, 3);
        matmul(sigma + (size_t)i * 9, JWt, SJ, 3, 3, 2);
        transp(G2, G2t, 2, 2);
        matmul(SJ, G2t, L, 3, 2, 2);
        transp(sigma + (size_t)i * 9, St, 3, 3);
        matmul(St, JWt, StJ, 3, 3, 2);
        matmul(StJ, G2, Rr, 3, 2, 2);
        for (int k = 0; k < 6; k++) gJWt[k] = L[k] + Rr[k];
        matmul(W, gJWt, gJt, 3, 3, 2);
        transp(gJt, g_J + (size_t)i * 6, 3, 2);
    }
}

// ---------------------------------------------------------------------------------------
// SH precompute  (precompute_sh.cu)
// ---------------------------------------------------------------------------------------
template <typename T>
inline void gaussian_view_dir(const T* xyz, const T* cam, int g, T* d) {   // precompute_sh.cu:28-39
    d[0] = xyz[g * 3 + 0] - cam[0];
    d[1] = xyz[g * 3 + 1] - cam[1];
    d[2] = xyz[g * 3 + 2] - cam[2];
    const T r = rsqrt_t<T>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; k++) d[k] *= r;
}

// precompute_sh.cu:8-58 ; cam = translation column of the matrix argument (:149-151)
template <typename T>
void precompute_rgb_from_sh(const T* xyz, const T* sh, const T* cam, int N, int n_sh, T* rgb) {
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; g++) {
        if (n_sh == 1) {
            for (int c = 0; c < 3; c++) rgb[g * 3 + c] = sh[g * 3 + c];
            continue;
        }
        T d[3], Y[16];
        gaussian_view_dir(xyz, cam, g, d);
        sh_basis(d, n_sh, Y);
        for (int c = 0; c < 3; c++) {
            T t = 0.0;
            for (int s = 0; s < n_sh; s++) t += Y[s] * sh[(size_t)g * n_sh * 3 + n_sh * c + s];
            t *= r_SH_0;
            rgb[g * 3 + c] = t;
        }
    }
}

// precompute_sh.cu:61-111
template <typename T>
void precompute_rgb_from_sh_backward(const T* xyz, const T* cam, const T* g_rgb, int N, int n_sh,
                                     T* g_sh) {
#pragma omp parallel for schedule(static)
    for (int g = 0; g < N; g++) {
        if (n_sh == 1) {
            for (int c = 0; c < 3; c++) g_sh[g * 3 + c] = g_rgb[g * 3 + c];
            continue;
        }
        T d[3], Y[16];
        gaussian_view_dir(xyz, cam, g, d);
        sh_basis(d, n_sh, Y);
        const T gl[3] = {g_rgb[g * 3 + 0] * r_SH_0, g_rgb[g * 3 + 1] * r_SH_0,
                         g_rgb[g * 3 + 2] * r_SH_0};
        for (int c = 0; c < 3; c++)
            for (int s = 0; s < n_sh; s++)
                g_sh[(size_t)g * n_sh * 3 + n_sh * c + s] = gl[c] * Y[s];
    }
}

// ---------------------------------------------------------------------------------------
// tile culling  (tile_culling.cu) -- fp32 only
// ---------------------------------------------------------------------------------------
// tile_culling.cu:8-66
inline bool split_axis_test(const float* obb, const float* tb) {
    const float mnx = fminf(fminf(obb[0], obb[2]), fminf(obb[4], obb[6]));
    const float mxx = fmaxf(fmaxf(obb[0], obb[2]), fmaxf(obb[4], obb[6]));
    if (mnx > tb[1] || mxx < tb[0]) return false;
    const float mny = fminf(fminf(obb[1], obb[3]), fminf(obb[5], obb[7]));
    const float mxy = fmaxf(fmaxf(obb[1], obb[3]), fmaxf(obb[5], obb[7]));
    if (mny > tb[3] || mxy < tb[2]) return false;
    for (int axis = 0; axis < 2; axis++) {
        // axis 0: major (top-right - top-left); axis 1: minor (top-right - bottom-right)
        const int o = axis == 0 ? 0 : 6;
        const float ax = obb[2] - obb[o];
        const float ay = obb[3] - obb[o + 1];
        const float tl = ax * tb[0] + ay * tb[2];
        const float tr = ax * tb[1] + ay * tb[2];
        const float bl = ax * tb[0] + ay * tb[3];
        const float br = ax * tb[1] + ay * tb[3];
        const float mn_t = fminf(fminf(tl, tr), fminf(bl, br));
        const float mx_t = fmaxf(fmaxf(tl, tr), fmaxf(bl, br));
        const float p0 = ax * obb[2] + ay * obb[3];
        const float p1 = ax * obb[o] + ay * obb[o + 1];
        const float mn_o = fminf(p0, p1);
        const float mx_o = fmaxf(p0, p1);
        if (mn_t > mx_o || mx_t < mn_o) return false;
    }
    return true;
}

// tile_culling.cu:69-122
inline int compute_obb(float u, float v, float a, float b, float c, float mh, float* obb) {
    const float left = (a + c) / 2;
    const float right = sqrtf((a - c) * (a - c) / 4.0f + b * b);
    const float l1 = left + right;
    const float l2 = left - right;
    const float r_major = mh * sqrtf(l1);
    const float r_minor = mh * sqrtf(l2);
    float ct, st;
    if (g_trig_mode == 1) {
        float theta;
        if (fabsf(b) < 1e-16) theta = (a >= c) ? 0.0f : (float)(M_PI / 2);
        else theta = atan2f(l1 - a, b);
        ct = cosf(theta);
        st = sinf(theta);
    } else {
        if (fabsf(b) < 1e-16) {
            if (a >= c) { ct = 1.0f; st = 0.0f; }
            else { ct = -4.37113883e-8f; st = 1.0f; }   // cosf/sinf of float(pi/2), correctly rounded
        } else {
            const float y = l1 - a;
            const float h = sqrtf(b * b + y * y);
            ct = b / h;
            st = y / h;
        }
    }
    obb[0] = -1 * r_major * ct + r_minor * st + u;
    obb[1] = -1 * r_major * st - r_minor * ct + v;
    obb[2] = r_major * ct + r_minor * st + u;
    obb[3] = r_major * st - r_minor * ct + v;
    obb[4] = -1 * r_major * ct - r_minor * st + u;
    obb[5] = -1 * r_major * st + r_minor * ct + v;
    obb[6] = r_major * ct - r_minor * st + u;
    obb[7] = r_major * st + r_minor * ct + v;
    return f2i(ceilf(r_major / 16.0f) + 1);
}

// shared candidate loop of compute_num_splats_kernel / compute_tiles_kernel
// (tile_culling.cu:138-176, 197-241). Calls emit(tile_idx) for every intersecting tile in
// x-outer / y-inner order.
template <typename F>
inline void for_each_tile(const float* uvs, const float* conic, int g, int ntx, int nty, float mh,
                          F emit) {
    const float u = uvs[g * 2], v = uvs[g * 2 + 1];
    const float a = conic[g * 3] + 0.25f;
    const float b = conic[g * 3 + 1] / 2.0f;
    const float c = conic[g * 3 + 2] + 0.25f;
    float obb[8];
    const int r = compute_obb(u, v, a, b, c, mh, obb);
    // int add/sub written with unsigned wrap so that CPU and GPU agree for absurd radii
    const int px = f2i(floorf(u / 16.0f));
    const int sx = f2i(fmaxf(0.0f, (float)(int)((unsigned)px - (unsigned)r)));
    const int ex = f2i(fminf((float)ntx, (float)(int)((unsigned)px + (unsigned)r)));
    const int py = f2i(floorf(v / 16.0f));
    const int sy = f2i(fmaxf(0.0f, (float)(int)((unsigned)py - (unsigned)r)));
    const int ey = f2i(fminf((float)nty, (float)(int)((unsigned)py + (unsigned)r)));
    for (int tx = sx; tx < ex; tx++)
        for (int ty = sy; ty < ey; ty++) {
            float tb[4] = {(float)tx * 16.0f, (float)(tx + 1) * 16.0f, (float)ty * 16.0f,
                           (float)(ty + 1) * 16.0f};
            if (split_axis_test(obb, tb)) emit(ty * ntx + tx);
        }
}

}  // namespace

// =========================================================================================
// C entry points
// =========================================================================================
extern "C" {

void orc_set_modes(int exp_mode, int trig_mode) {
    g_exp_mode = exp_mode;
    g_trig_mode = trig_mode;
}
void orc_set_sh_band1_mode(int m) { g_sh_band1_mode = m; }
void orc_set_backward_abs(int on) { g_bwd_abs = on; }
void orc_set_backward_exact(int on) { g_q1_exact = on; }
void orc_set_backward_sum(int mode) { g_bwd_sum = mode; }
void orc_set_contrib_count(int* per_pixel) { g_contrib_count = per_pixel; }
int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
float orc_det_expf(float x) { return det_expf(x); }
// IEEE-only sigmoid used by the fused preprocess kernel in place of torch.sigmoid
// (splat_py/rasterize.py:60-62); within a few ulp of it
void orc_sigmoid_f32(const float* x, int N, float* y) {
    for (int i = 0; i < N; i++) y[i] = 1.0f / (1.0f + det_expf(-x[i]));
}

#define INST(T, SFX)                                                                               \
    void orc_camera_projection_##SFX(const T* xyz, const T* K, int N, T* uv) {                     \
        camera_projection<T>(xyz, K, N, uv);                                                       \
    }                                                                                              \
    void orc_camera_projection_backward_##SFX(const T* xyz, const T* K, const T* g, int N,         \
                                              T* out) {                                            \
        camera_projection_backward<T>(xyz, K, g, N, out);                                          \
    }                                                                                              \
    void orc_compute_sigma_world_##SFX(const T* q, const T* s, int N, T* out) {                    \
        compute_sigma_world<T>(q, s, N, out);                                                      \
    }                                                                                              \
    void orc_compute_sigma_world_backward_##SFX(const T* q, const T* s, const T* g, int N, T* gq,  \
                                                T* gs) {                                           \
        compute_sigma_world_backward<T>(q, s, g, N, gq, gs);                                       \
    }                                                                                              \
    void orc_compute_projection_jacobian_##SFX(const T* xyz, const T* K, int N, T* J) {            \
        compute_projection_jacobian<T>(xyz, K, N, J);                                              \
    }                                                                                              \
    void orc_compute_projection_jacobian_backward_##SFX(const T* xyz, const T* K, const T* g,      \
                                                        int N, T* out) {                           \
        compute_projection_jacobian_backward<T>(xyz, K, g, N, out);                                \
    }                                                                                              \
    void orc_compute_conic_##SFX(const T* sig, const T* J, const T* M, int N, T* conic) {          \
        compute_conic<T>(sig, J, M, N, conic);                                                     \
    }                                                                                              \
    void orc_compute_conic_backward_##SFX(const T* sig, const T* J, const T* M, const T* g, int N, \
                                          T* gs, T* gJ) {                                          \
        compute_conic_backward<T>(sig, J, M, g, N, gs, gJ);                                        \
    }                                                                                              \
    void orc_precompute_rgb_from_sh_##SFX(const T* xyz, const T* sh, const T* cam, int N,          \
                                          int n_sh, T* rgb) {                                      \
        precompute_rgb_from_sh<T>(xyz, sh, cam, N, n_sh, rgb);                                     \
    }                                                                                              \
    void orc_precompute_rgb_from_sh_backward_##SFX(const T* xyz, const T* cam, const T* g, int N,  \
                                                   int n_sh, T* gsh) {                             \
        precompute_rgb_from_sh_backward<T>(xyz, cam, g, N, n_sh, gsh);                             \
    }
INST(float, f32)
INST(double, f64)
#undef INST

// Multi-GPU bookkeeping (no reference counterpart; checker for gs_halo_plan): bit s of mask[g] is
// set when the candidate tile window of Gaussian g (tile_culling.cu:138-156) reaches the tile rows
// [band_rows[s], band_rows[s+1]) -- a superset of the rows in which the separating-axis test
// accepts a tile.
void orc_band_mask(const float* uvs, const float* conic, int ntx, int nty, float mh, int N,
                   const int* band_rows, int G, uint32_t* mask) {
    for (int g = 0; g < N; g++) {
        const float u = uvs[g * 2], v = uvs[g * 2 + 1];
        const float a = conic[g * 3] + 0.25f;
        const float b = conic[g * 3 + 1] / 2.0f;
        const float c = conic[g * 3 + 2] + 0.25f;
        float obb[8];
        const int r = compute_obb(u, v, a, b, c, mh, obb);
        const int px = f2i(floorf(u / 16.0f));
        const int sx = f2i(fmaxf(0.0f, (float)(int)((unsigned)px - (unsigned)r)));
        const int ex = f2i(fminf((float)ntx, (float)(int)((unsigned)px + (unsigned)r)));
        const int py = f2i(floorf(v / 16.0f));
        const int sy = f2i(fmaxf(0.0f, (float)(int)((unsigned)py - (unsigned)r)));
        const int ey = f2i(fminf((float)nty, (float)(int)((unsigned)py + (unsigned)r)));
        uint32_t m = 0;
        if (sx < ex && sy < ey)
            for (int s = 0; s < G; s++)
                if (sy < band_rows[s + 1] && ey > band_rows[s]) m |= 1u << s;
        mask[g] = m;
    }
}

// tile_culling.cu:124-177 + host :269-296.  Returns the number of Gaussian-tile instances S.
// num_tiles_per_gaussian[N], num_gaussians_per_tile[T] are overwritten.
int64_t orc_tile_count(const float* uvs, const float* conic, int ntx, int nty, float mh, int N,
                       int* num_tiles_per_gaussian, int* num_gaussians_per_tile) {
    std::fill(num_gaussians_per_tile, num_gaussians_per_tile + (size_t)ntx * nty, 0);
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : total)
    for (int g = 0; g < N; g++) {
        int n = 0;
        for_each_tile(uvs, conic, g, ntx, nty, mh, [&](int t) {
#pragma omp atomic
            num_gaussians_per_tile[t]++;
            n++;
        });
        num_tiles_per_gaussian[g] = n;
        total += n;
    }
    return total;
}

// tile_culling.cu:179-242 + host :298-337.  Emits (fp64 key, gaussian) per instance in Gaussian
// order, stable-sorts by the fp64 key z + (max_z+1)*tile (:236-237,:307-309) and builds the
// per-tile ranges.  sorted_gaussians[S], tile_ranges[T+1].  If key_out != NULL it receives the
// sorted integer keys (tile<<32 | zbits) of the same instances for the key-equivalence test.
void orc_tile_emit_sort(const float* uvs, const float* xyz_cam, const float* conic, int ntx,
                        int nty, float mh, int N, const int* num_tiles_per_gaussian,
                        const int* num_gaussians_per_tile, int64_t S, int* sorted_gaussians,
                        int* tile_ranges, uint64_t* key_out) {
    std::vector<int64_t> start(N + 1, 0);
    for (int g = 0; g < N; g++) start[g + 1] = start[g] + num_tiles_per_gaussian[g];
    std::vector<double> keys((size_t)S);
    std::vector<int> gidx((size_t)S);
    std::vector<int> tidx((size_t)S);
    float max_z = -INFINITY;
    for (int g = 0; g < N; g++) max_z = std::max(max_z, xyz_cam[g * 3 + 2]);
    const double mult = (double)(max_z + 1.0f);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int g = 0; g < N; g++) {
        const double z = (double)xyz_cam[g * 3 + 2];
        int64_t n = 0;
        const int64_t s0 = start[g], s1 = start[g + 1];
        for_each_tile(uvs, conic, g, ntx, nty, mh, [&](int t) {
            if (s0 + n < s1) {
                gidx[s0 + n] = g;
                tidx[s0 + n] = t;
                keys[s0 + n] = z + mult * (double)t;
                n++;
            }
        });
    }
    std::vector<int64_t> order((size_t)S);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int64_t a, int64_t b) { return keys[a] < keys[b]; });
    for (int64_t i = 0; i < S; i++) {
        sorted_gaussians[i] = gidx[order[i]];
        if (key_out) {
            uint32_t zb;
            float zf = xyz_cam[gidx[order[i]] * 3 + 2];
            memcpy(&zb, &zf, 4);
            zb = (zb & 0x80000000u) ? ~zb : (zb | 0x80000000u);
            key_out[i] = ((uint64_t)(uint32_t)tidx[order[i]] << 32) | zb;
        }
    }
    tile_ranges[0] = 0;
    const int T = ntx * nty;
    for (int t = 0; t < T; t++) tile_ranges[t + 1] = tile_ranges[t] + num_gaussians_per_tile[t];
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// render forward / backward  (render.cu, render_backward.cu)
// ---------------------------------------------------------------------------------------
namespace {

template <typename T> struct RenderMode;
template <> struct RenderMode<float> { static constexpr bool fast = true; };    // render.cu:266-333
template <> struct RenderMode<double> { static constexpr bool fast = false; };  // render.cu:347-414

template <typename T> inline T render_exp(T x);
template <> inline float render_exp<float>(float x) { return exp_fast(x); }
template <> inline double render_exp<double>(double x) { return exp(x); }

// render.cu:8-189
template <typename T>
void render_tiles(const T* uvs, const T* opacity, const T* rgb, const T* conic, const T* view_dir,
                  const int* tile_ranges, const int* sorted, const T* background, int W, int H,
                  int n_sh, int* num_splats_px, T* final_weight_px, T* image, int tile_y0,
                  int tile_y1) {
    const bool fast = RenderMode<T>::fast;
    const int ntx = (W + 15) / 16;
    const int nty = (H + 15) / 16;
    if (tile_y1 > nty) tile_y1 = nty;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = tile_y0; ty < tile_y1; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            const int tile = ty * ntx + tx;
            const int s0 = tile_ranges[tile], s1 = tile_ranges[tile + 1];
            const int n_tile = s1 - s0;
            for (int py = 0; py < 16; py++)
                for (int px = 0; px < 16; px++) {
                    const int u_px = tx * 16 + px, v_px = ty * 16 + py;
                    if (u_px >= W || v_px >= H) continue;
                    T Y[16];
                    if (n_sh > 1) sh_basis(view_dir + ((size_t)v_px * W + u_px) * 3, n_sh, Y);
                    else Y[0] = T(SH_0);
                    T alpha_accum = 0.0, alpha_weight = 0.0;
                    int num_splats = 0, n_contrib = 0;
                    T img[3] = {0.0, 0.0, 0.0};
                    for (int k = 0; k < n_tile; k++) {
                        if (alpha_accum > 0.9999) break;
                        const int g = sorted[s0 + k];
                        const T u_diff = T(u_px) - uvs[g * 2 + 0];
                        const T v_diff = T(v_px) - uvs[g * 2 + 1];
                        T a, c;
                        const T b = conic[g * 3 + 1] * 0.5;
                        if (fast) {
                            a = conic[g * 3 + 0] + 0.25;
                            c = conic[g * 3 + 2] + 0.25;
                        } else {
                            a = conic[g * 3 + 0];
                            c = conic[g * 3 + 2];
                        }
                        const T det = a * c - b * b;
                        T alpha = 0.0;
                        const T mh_sq =
                            (c * u_diff * u_diff - (b + b) * u_diff * v_diff + a * v_diff * v_diff) /
                            det;
                        if (mh_sq > 0.0) {
                            T norm_prob = render_exp<T>(-0.5 * mh_sq);
                            alpha = opacity[g] * norm_prob;
                        }
                        if (alpha < 0.00392156862 && fast) {
                            num_splats++;
                            continue;
                        }
                        alpha_weight = 1.0 - alpha_accum;
                        const T weight = alpha * (1.0 - alpha_accum);
                        T col[3];
                        sh_to_rgb(rgb + (size_t)g * 3 * n_sh, Y, n_sh, col);
                        for (int ch = 0; ch < 3; ch++) img[ch] += col[ch] * weight;
                        alpha_accum += weight;
                        num_splats++;
                        n_contrib++;
                    }
                    if (alpha_accum < 0.999)
                        for (int ch = 0; ch < 3; ch++) img[ch] += background[ch] * (1.0 - alpha_accum);
                    const size_t p = (size_t)v_px * W + u_px;
                    num_splats_px[p] = num_splats;
                    if (g_contrib_count) g_contrib_count[p] = n_contrib;
                    final_weight_px[p] = alpha_weight;
                    for (int ch = 0; ch < 3; ch++) image[p * 3 + ch] = img[ch];
                }
        }
}

// reference chunk sizes (render_backward.cu:402,423,445,466,505,526,547,568)
template <typename T> inline int ref_chunk(int n_sh);
template <> inline int ref_chunk<float>(int n_sh) {
    return n_sh == 1 ? 960 : n_sh == 4 ? 576 : n_sh == 9 ? 320 : 160;
}
template <> inline int ref_chunk<double>(int n_sh) {
    return n_sh == 1 ? 320 : n_sh == 4 ? 160 : n_sh == 9 ? 128 : 64;
}

// render_backward.cu:12-285.  Per-pixel contributions are formed in T exactly as the kernel
// does; the sum over pixels (warp reduce + atomicAdd in the reference, order unspecified) is
// accumulated in double and rounded once.
template <typename T>
void render_tiles_backward(const T* uvs, const T* opacity, const T* rgb, const T* conic,
                           const T* view_dir, const int* tile_ranges, const int* sorted,
                           const T* background, const int* num_splats_px, const T* final_weight_px,
                           const T* grad_image, int W, int H, int n_sh, int V, T* g_rgb,
                           T* g_opacity, T* g_uv, T* g_conic, int tile_y0, int tile_y1) {
    const bool fast = RenderMode<T>::fast;
    const bool abs_mode = g_bwd_abs != 0;
    const int sum_mode = g_bwd_sum;   // 0: double; 1, 2: fp32 in a fixed order (ascending / descending)
    // acc += term in the selected arithmetic (fp32: one IEEE float addition per term)
    auto add = [sum_mode](double& acc, double term) {
        if (sum_mode == 0) acc += term;
        else { float t = (float)acc; t += (float)term; acc = (double)t; }
    };
    const int CH = ref_chunk<T>(n_sh);
    const int ntx = (W + 15) / 16;
    const int nty = (H + 15) / 16;
    if (tile_y1 > nty) tile_y1 = nty;
    const int C = 3 * n_sh;
    std::vector<double> acc_rgb((size_t)V * C, 0.0), acc_opa((size_t)V, 0.0),
        acc_uv((size_t)V * 2, 0.0), acc_con((size_t)V * 3, 0.0);
    // fixed-order fp32 modes: the tiles' sums are kept per instance and combined serially afterwards
    const int band_s0 = tile_ranges[std::min(tile_y0, nty) * ntx], band_s1 = tile_ranges[tile_y1 * ntx];
    std::vector<float> inst;
    if (sum_mode != 0) inst.assign((size_t)std::max(0, band_s1 - band_s0) * (C + 6), 0.0f);
#pragma omp parallel
    {
        std::vector<double> loc;   // per-tile per-splat accumulators: [n_tile][C + 6]
#pragma omp for schedule(dynamic, 1) collapse(2)
        for (int ty = tile_y0; ty < tile_y1; ty++)
            for (int tx = 0; tx < ntx; tx++) {
                const int tile = ty * ntx + tx;
                const int s0 = tile_ranges[tile], s1 = tile_ranges[tile + 1];
                const int n_tile = s1 - s0;
                if (n_tile <= 0) continue;
                const int RW = C + 6;
                loc.assign((size_t)n_tile * RW, 0.0);
                int max_used = 0;
                for (int pp = 0; pp < 256; pp++) {
                    {
                        const int pi = sum_mode == 2 ? 255 - pp : pp;
                        const int py = pi >> 4, px = pi & 15;
                        const int u_px = tx * 16 + px, v_px = ty * 16 + py;
                        if (u_px >= W || v_px >= H) continue;
                        const size_t p = (size_t)v_px * W + u_px;
                        const int nsp = num_splats_px[p];
                        if (nsp > max_used) max_used = nsp;
                        T gi[3] = {grad_image[p * 3 + 0], grad_image[p * 3 + 1],
                                   grad_image[p * 3 + 2]};
                        T Y[16];
                        if (n_sh > 1) sh_basis(view_dir + p * 3, n_sh, Y);
                        else Y[0] = T(SH_0);
                        T weight = final_weight_px[p];
                        T color_accum[3] = {0.0, 0.0, 0.0};
                        bool bg_init = false;
                        for (int k = std::min(nsp, n_tile) - 1; k >= 0; k--) {
                            const int i = g_q1_exact ? k : k % CH;   // chunk-local index (render_backward.cu:120)
                            const int g = sorted[s0 + k];
                            const T u_diff = T(u_px) - uvs[g * 2 + 0];
                            const T v_diff = T(v_px) - uvs[g * 2 + 1];
                            T a, c;
                            const T b = conic[g * 3 + 1] * 0.5;
                            if (fast) {
                                a = conic[g * 3 + 0] + 0.25;
                                c = conic[g * 3 + 2] + 0.25;
                            } else {
                                a = conic[g * 3 + 0];
                                c = conic[g * 3 + 2];
                            }
                            const T det = a * c - b * b;
                            T norm_prob = 0.0;
                            T rdet = 1.0 / det;
                            const T mh_sq = (c * u_diff * u_diff - (b + b) * u_diff * v_diff +
                                             a * v_diff * v_diff) *
                                            rdet;
                            if (mh_sq > 0.0) norm_prob = render_exp<T>(-0.5 * mh_sq);
                            // min(0.9999, T) -> double overload, narrowed (render_backward.cu:167)
                            T alpha = (T)std::min(0.9999, (double)(opacity[g] * norm_prob));
                            if (!(alpha >= 0.00392156862 || !fast)) continue;
                            if (!bg_init) {
                                const T bw = 1.0 - (alpha * weight + 1.0 - weight);
                                if (bw > 0.001)
                                    for (int ch = 0; ch < 3; ch++) color_accum[ch] += background[ch] * bw;
                                bg_init = true;
                            }
                            const T r1ma = 1.0 / (1.0 - alpha);
                            if (i < nsp - 1) weight = weight * r1ma;   // Q1: chunk-local i
                            T grl[3];
                            for (int ch = 0; ch < 3; ch++) grl[ch] = alpha * weight * gi[ch];
                            T col[3];
                            sh_to_rgb(rgb + (size_t)g * C, Y, n_sh, col);
                            double* L = &loc[(size_t)k * RW];
                            if (abs_mode) {
                                // checker aid: per element the sum of the magnitudes of the LEAF terms
                                // of the formulas below (every product that enters a sum or difference),
                                // the scale of an fp32 evaluation's rounding and summation-order noise
                                double ga = 0.0;   // leaf magnitude of grad_alpha
                                for (int ch = 0; ch < 3; ch++)
                                    ga += (std::fabs((double)col[ch] * weight) +
                                           std::fabs((double)color_accum[ch] * r1ma)) *
                                          std::fabs((double)gi[ch]);
                                for (int s = 0; s < n_sh; s++)
                                    for (int ch = 0; ch < 3; ch++)
                                        L[n_sh * ch + s] += std::fabs((double)Y[s] * grl[ch]);
                                const double gm = 0.5 * (double)norm_prob * std::fabs((double)opacity[g]) * ga;
                                const double u = u_diff, v = v_diff, rd = rdet;
                                const double cfm = (std::fabs((double)a) * v * v + 2 * std::fabs((double)b * u * v) +
                                                    std::fabs((double)c) * u * u) * rd * rd;
                                L[C + 0] += (double)norm_prob * ga;
                                L[C + 1] += (2 * std::fabs((double)b * v) + 2 * std::fabs((double)c * u)) * rd * gm;
                                L[C + 2] += (2 * std::fabs((double)a * v) + 2 * std::fabs((double)b * u)) * rd * gm;
                                L[C + 3] += (std::fabs((double)c) * cfm + v * v * rd) * gm;
                                L[C + 4] += (std::fabs((double)b) * cfm + std::fabs(u * v) * rd) * gm;
                                L[C + 5] += (std::fabs((double)a) * cfm + u * u * rd) * gm;
                                for (int ch = 0; ch < 3; ch++) color_accum[ch] += col[ch] * alpha * weight;
                                continue;
                            }
                            for (int s = 0; s < n_sh; s++)
                                for (int ch = 0; ch < 3; ch++)
                                    add(L[n_sh * ch + s], (double)(T)(Y[s] * grl[ch]));
                            T grad_alpha = 0.0;
                            for (int ch = 0; ch < 3; ch++)
                                grad_alpha += (col[ch] * weight - color_accum[ch] * r1ma) * gi[ch];
                            const T grad_opa = norm_prob * grad_alpha;
                            const T grad_prob = opacity[g] * grad_alpha;
                            const T grad_mh = -0.5 * norm_prob * grad_prob;
                            const T grad_u =
                                -(-b * v_diff - b * v_diff + 2 * c * u_diff) * rdet * grad_mh;
                            const T grad_v =
                                -(2 * a * v_diff - b * u_diff - b * u_diff) * rdet * grad_mh;
                            const T cf = (a * v_diff * v_diff - b * u_diff * v_diff -
                                          b * u_diff * v_diff + c * u_diff * u_diff) *
                                         rdet * rdet;
                            const T gc0 = (-c * cf + v_diff * v_diff * rdet) * grad_mh;
                            const T gc1 = (b * cf - u_diff * v_diff * rdet) * grad_mh;
                            const T gc2 = (-a * cf + u_diff * u_diff * rdet) * grad_mh;
                            add(L[C + 0], (double)grad_opa);
                            add(L[C + 1], (double)grad_u);
                            add(L[C + 2], (double)grad_v);
                            add(L[C + 3], (double)gc0);
                            add(L[C + 4], (double)gc1);
                            add(L[C + 5], (double)gc2);
                            for (int ch = 0; ch < 3; ch++) color_accum[ch] += col[ch] * alpha * weight;
                        }
                    }
                }
                max_used = std::min(max_used, n_tile);
                if (sum_mode != 0) {
                    for (int k = 0; k < max_used; k++)
                        for (int j = 0; j < RW; j++)
                            inst[(size_t)(s0 - band_s0 + k) * RW + j] = (float)loc[(size_t)k * RW + j];
                    continue;
                }
                for (int k = 0; k < max_used; k++) {
                    const int g = sorted[s0 + k];
                    const double* L = &loc[(size_t)k * RW];
                    for (int j = 0; j < C; j++) {
#pragma omp atomic
                        acc_rgb[(size_t)g * C + j] += L[j];
                    }
#pragma omp atomic
                    acc_opa[g] += L[C];
#pragma omp atomic
                    acc_uv[g * 2 + 0] += L[C + 1];
#pragma omp atomic
                    acc_uv[g * 2 + 1] += L[C + 2];
                    for (int j = 0; j < 3; j++) {
#pragma omp atomic
                        acc_con[g * 3 + j] += L[C + 3 + j];
                    }
                }
            }
    }
    if (sum_mode != 0) {   // the tiles' fp32 sums, added in fp32 in ascending / descending tile order
        const int RW = C + 6;
        const int t_lo = tile_y0 * ntx, t_hi = tile_y1 * ntx;
        for (int q = t_lo; q < t_hi; q++) {
            const int tile = sum_mode == 2 ? t_hi - 1 - (q - t_lo) : q;
            for (int s = tile_ranges[tile]; s < tile_ranges[tile + 1]; s++) {
                const int g = sorted[s];
                const float* L = &inst[(size_t)(s - band_s0) * RW];
                for (int j = 0; j < C; j++) add(acc_rgb[(size_t)g * C + j], L[j]);
                add(acc_opa[g], L[C]);
                add(acc_uv[g * 2 + 0], L[C + 1]);
                add(acc_uv[g * 2 + 1], L[C + 2]);
                for (int j = 0; j < 3; j++) add(acc_con[g * 3 + j], L[C + 3 + j]);
            }
        }
    }
    // accumulate into the caller's buffers (atomicAdd semantics, render_backward.cu:269-281)
    for (size_t j = 0; j < (size_t)V * C; j++) g_rgb[j] += (T)acc_rgb[j];
    for (size_t j = 0; j < (size_t)V; j++) g_opacity[j] += (T)acc_opa[j];
    for (size_t j = 0; j < (size_t)V * 2; j++) g_uv[j] += (T)acc_uv[j];
    for (size_t j = 0; j < (size_t)V * 3; j++) g_conic[j] += (T)acc_con[j];
}

}  // namespace

extern "C" {

#define INST_R(T, SFX)                                                                             \
    void orc_render_tiles_##SFX(const T* uvs, const T* opacity, const T* rgb, const T* conic,      \
                                const T* view_dir, const int* tile_ranges, const int* sorted,      \
                                const T* background, int W, int H, int n_sh, int* nsp, T* fw,      \
                                T* image, int tile_y0, int tile_y1) {                              \
        render_tiles<T>(uvs, opacity, rgb, conic, view_dir, tile_ranges, sorted, background, W, H, \
                        n_sh, nsp, fw, image, tile_y0, tile_y1);                                   \
    }                                                                                              \
    void orc_render_tiles_backward_##SFX(                                                          \
        const T* uvs, const T* opacity, const T* rgb, const T* conic, const T* view_dir,           \
        const int* tile_ranges, const int* sorted, const T* background, const int* nsp,            \
        const T* fw, const T* grad_image, int W, int H, int n_sh, int V, T* g_rgb, T* g_opa,       \
        T* g_uv, T* g_conic, int tile_y0, int tile_y1) {                                           \
        render_tiles_backward<T>(uvs, opacity, rgb, conic, view_dir, tile_ranges, sorted,          \
                                 background, nsp, fw, grad_image, W, H, n_sh, V, g_rgb, g_opa,     \
                                 g_uv, g_conic, tile_y0, tile_y1);                                 \
    }
INST_R(float, f32)
INST_R(double, f64)
#undef INST_R

// depth.cu:7-115 (fp32 only).  depth_image is pre-filled by the caller (-1) and only written
// where the accumulated alpha passes the threshold.
void orc_render_depth_f32(const float* xyz_cam, const float* uvs, const float* opacity,
                          const float* conic, const int* tile_ranges, const int* sorted, int W,
                          int H, float alpha_threshold, float* depth_image) {
    const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            const int tile = ty * ntx + tx;
            const int s0 = tile_ranges[tile], s1 = tile_ranges[tile + 1];
            for (int py = 0; py < 16; py++)
                for (int px = 0; px < 16; px++) {
                    const int u_px = tx * 16 + px, v_px = ty * 16 + py;
                    if (u_px >= W || v_px >= H) continue;
                    float alpha_accum = 0.0;
                    for (int k = s0; k < s1; k++) {
                        const int g = sorted[k];
                        const float u_diff = (float)u_px - uvs[g * 2 + 0];
                        const float v_diff = (float)v_px - uvs[g * 2 + 1];
                        const float a = conic[g * 3 + 0] + 0.25;
                        const float b = conic[g * 3 + 1] * 0.5;
                        const float c = conic[g * 3 + 2] + 0.25;
                        const float det = a * c - b * b;
                        float alpha = 0.0;
                        const float mh_sq =
                            (c * u_diff * u_diff - (b + b) * u_diff * v_diff + a * v_diff * v_diff) /
                            det;
                        if (mh_sq > 0.0) {
                            const float norm_prob = exp_fast(-0.5 * mh_sq);
                            alpha = opacity[g] * norm_prob;
                        }
                        const float weight = alpha * (1.0 - alpha_accum);
                        alpha_accum += weight;
                        if (alpha_accum > alpha_threshold) {
                            const float x = xyz_cam[g * 3 + 0], y = xyz_cam[g * 3 + 1],
                                        z = xyz_cam[g * 3 + 2];
                            depth_image[(size_t)v_px * W + u_px] = sqrtf(x * x + y * y + z * z);
                            break;
                        }
                    }
                }
        }
}

}  // extern "C"
