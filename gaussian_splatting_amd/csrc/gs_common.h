// gs_common.h -- shared device helpers for the gfx950 kernels.
//
// Arithmetic contract (see DESIGN.md "Arithmetic contract"): all translation units are built with
// -ffp-contract=off, so every rounding is the one written in the source; fused multiply-adds
// appear only as explicit fmaf().  fp32 division and sqrt are the correctly rounded forms
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  Non-IEEE library functions of the
// reference (__expf, exp(float), rsqrt, atan2f/cosf/sinf) are replaced by IEEE-only
// restatements so that the fp32 forward path is bit-reproducible on any IEEE machine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsplat_hip.h"

#define GS_WAVE 64

namespace gs {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define GS_REQUIRE(cond, ...)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            gs::set_error(__VA_ARGS__);                                                            \
            return GS_EINVAL;                                                                      \
        }                                                                                          \
    } while (0)

__host__ __device__ inline float bits_f(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}

// ---- thresholds -------------------------------------------------------------------------------
// The reference compares T values with double literals (render.cu:106,145,169;
// render_backward.cu:167,170,174; tile_culling.cu:89).  For T=float the comparison
// (double)x OP lit is equivalent to a float comparison against the neighbouring float of lit:
//   x >  lit  <=>  x >  RD(lit)        x <  lit  <=>  x <  RU(lit)        x >= lit <=> x >= RU(lit)
template <typename T> struct Thr;
template <> struct Thr<float> {
    __device__ static float sat_gt() { return bits_f(0x3f7ff972u); }     // x > 0.9999
    __device__ static float alpha_min() { return bits_f(0x3b808081u); }  // x < / >= 0.00392156862
    __device__ static float bg_lt() { return bits_f(0x3f7fbe77u); }      // x < 0.999
    __device__ static float bgw_gt() { return bits_f(0x3a83126eu); }     // x > 0.001
    __device__ static float alpha_cap() { return bits_f(0x3f7ff972u); }  // (float)0.9999
};
template <> struct Thr<double> {
    __device__ static double sat_gt() { return 0.9999; }
    __device__ static double alpha_min() { return 0.00392156862; }
    __device__ static double bg_lt() { return 0.999; }
    __device__ static double bgw_gt() { return 0.001; }
    __device__ static double alpha_cap() { return 0.9999; }
};

// ---- deterministic exp (fp32): 2^f by a 6th-order polynomial, IEEE ops only -----------------------
// Branch-free: the range cases are selects, so a render loop keeps one exec region per splat.
// Same value as the CPU restatement for every input (the polynomial is evaluated on the clamped
// argument, which equals the argument wherever the polynomial result is the one selected).
__device__ inline float det_expf(float x) {
    const float t = x * 1.44269504088896341f;
    const float tc = fminf(fmaxf(t, -126.0f), 127.0f);
    const float n = __builtin_rintf(tc);
    const float f = tc - n;
    float p = 1.54035303933816e-4f;
    p = __builtin_fmaf(p, f, 1.33335581464284e-3f);
    p = __builtin_fmaf(p, f, 9.61812910762848e-3f);
    p = __builtin_fmaf(p, f, 5.55041086648216e-2f);
    p = __builtin_fmaf(p, f, 2.40226506959101e-1f);
    p = __builtin_fmaf(p, f, 6.93147180559945e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    float r = __builtin_ldexpf(p, (int)n);
    r = (t > 127.0f) ? __builtin_inff() : r;
    r = (t > -125.0f) ? r : 0.0f;   // underflow -> 0
    r = (t != t) ? t : r;           // NaN propagates
    return r;
}

// exp() of the reference (projection.cu:91-93) and __expf/exp of the render kernels
template <typename T> __device__ inline T gexp(T x);
template <> __device__ inline float gexp<float>(float x) { return det_expf(x); }
template <> __device__ inline double gexp<double>(double x) { return exp(x); }

template <typename T> __device__ inline T gsqrt(T x);
template <> __device__ inline float gsqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ inline double gsqrt<double>(double x) { return __builtin_sqrt(x); }

// CUDA-style float -> int (saturating, NaN -> 0), tile_culling.cu:119,149-156
__device__ inline int f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}

// matrix.cuh:15-30
template <typename T, int RA, int CA, int CB>
__device__ inline void matmul(const T* A, const T* B, T* C) {
#pragma unroll
    for (int r = 0; r < RA; r++)
#pragma unroll
        for (int c = 0; c < CB; c++) {
            T sum = 0;
#pragma unroll
            for (int k = 0; k < CA; k++) sum += A[r * CA + k] * B[k * CB + c];
            C[r * CB + c] = sum;
        }
}
template <typename T, int R, int C>
__device__ inline void transp(const T* A, T* At) {
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < C; c++) At[c * R + r] = A[r * C + c];
}

// ---- spherical harmonics (spherical_harmonics.cuh:4-72) -------------------------------------------
// constants are the float-rounded literals of the reference
#define GS_SH_0 0.28209479177387814f
#define GS_R_SH_0 3.544907701811032f
#define GS_SH_1 0.4886025119029199f
#define GS_SH_2_0 1.0925484305920792f
#define GS_SH_2_2 0.31539156525252005f
#define GS_SH_2_4 0.5462742152960396f
#define GS_SH_3_0 0.5900435899266435f
#define GS_SH_3_1 2.890611442640554f
#define GS_SH_3_2 0.4570457994644658f
#define GS_SH_3_3 0.263875515352797f
#define GS_SH_3_5 1.445305721320277f

template <typename T, int N_SH>
__device__ inline void sh_basis(const T* d, T* Y) {
    Y[0] = T(GS_SH_0);
    if constexpr (N_SH >= 4) {
        const T x = d[0], y = d[1], z = d[2];
        Y[1] = T(-GS_SH_1) * y;
        Y[2] = T(GS_SH_1) * z;
        Y[3] = T(-GS_SH_1) * x;
        if constexpr (N_SH >= 9) {
            const T xy = x * y, yz = y * z, xz = x * z, xx = x * x, yy = y * y, zz = z * z;
            Y[4] = T(GS_SH_2_0) * xy;
            Y[5] = T(-GS_SH_2_0) * yz;
            Y[6] = T(GS_SH_2_2) * (3 * zz - 1.0);   // double literal: promoted, then narrowed
            Y[7] = T(-GS_SH_2_0) * xz;
            Y[8] = T(GS_SH_2_4) * (xx - yy);
            if constexpr (N_SH >= 16) {
                Y[9] = T(-GS_SH_3_0) * y * (3 * xx - yy);
                Y[10] = T(GS_SH_3_1) * xy * z;
                Y[11] = T(-GS_SH_3_2) * y * (5 * zz - 1.0);
                Y[12] = T(GS_SH_3_3) * z * (5 * zz - 3.0);
                Y[13] = T(-GS_SH_3_2) * x * (5 * zz - 1.0);
                Y[14] = T(GS_SH_3_5) * z * (xx - yy);
                Y[15] = T(-GS_SH_3_0) * x * (xx - 3 * yy);
            }
        }
    }
}

// spherical_harmonics.cuh:74-96 ; coeff layout [3][N_SH]
template <typename T, int N_SH>
__device__ inline void sh_to_rgb(const T* coeff, const T* Y, T* rgb) {
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[c] = Y[0] * coeff[N_SH * c];
    if constexpr (N_SH >= 4) {
#pragma unroll
        for (int s = 1; s < N_SH; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) rgb[c] += Y[s] * coeff[N_SH * c + s];
    }
}

// The same sum with fused multiply-adds: for gradient VALUES only (the render backward's recomputed colour, checked
// at 1e-4); every forward quantity uses sh_to_rgb above, whose separate roundings the oracle reproduces.
template <typename T, int N_SH>
__device__ inline void sh_to_rgb_contracted(const T* coeff, const T* Y, T* rgb) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int c = 0; c < 3; c++) rgb[c] = Y[0] * coeff[N_SH * c];
    if constexpr (N_SH >= 4) {
#pragma unroll
        for (int s = 1; s < N_SH; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) rgb[c] += Y[s] * coeff[N_SH * c + s];
    }
}

// ---- prefix-sorted tile lists (binning.hip "prefix sort") --------------------------------------------
// Largest tile list the LDS sort kernels take; longer ones are sorted in full in global memory.
constexpr int SORT_MAX_LDS_KEYS = 8192;
// In prefix mode only the first sort_prefix entries of such a tile's segment are ordered (and
// valid).  The sort and the forward render kernel share this predicate.
__host__ __device__ inline bool prefix_sorted_tile(int n, int sort_prefix) {
    return sort_prefix > 0 && n > sort_prefix && n <= SORT_MAX_LDS_KEYS;
}

// binning.hip: full sort of the tiles with flags[t] != 0 (repair pass of the prefix mode)
int sort_flagged_tiles(const int* ranges, const uint64_t* keys, int* sorted, int tile0, int nt,
                       int64_t S, const int* flags, hipStream_t s);

// ---- depth-bucketed binning (binning.hip "depth cut") ------------------------------------------------------
constexpr int DC_PART = 256;            // workgroups of the partition passes
struct CutState {                       // views into the caller's cut workspace (gs_cut_workspace_ints)
    int* ctrl;                          // [0] deepest bucket any tile wants, [1] flagged tiles of this frame
    int* totals;                        // [T]  n(t)
    int* bstar;                         // [T]  b*(t), -1: nothing kept
    int* boff;                          // [BUCKETS + 1] scratch: bucket totals at [1 ..]
    int* boff2;                         // [BUCKETS + 1] first list position of each bucket, [BUCKETS] = V
    uint32_t* bounds;                   // [BUCKETS] inclusive upper depth bound (sortable bits) of each bucket
    int* phist;                         // [DC_PART][BUCKETS]
    int* list;                          // [N]
    uint16_t* bucket_of;                // [N]
};
__host__ __device__ inline size_t cut_ws_ints(int N, int T) {
    const size_t n = (size_t)(N > 0 ? N : 1), t = (size_t)(T > 0 ? T : 1);
    return 8 + 2 * t + 2 * (GS_CUT_BUCKETS + 8) + GS_CUT_BUCKETS + (size_t)DC_PART * GS_CUT_BUCKETS + n + (n + 1) / 2 + 16;
}
__host__ __device__ inline CutState cut_state_of(int32_t* ws, int N, int T) {
    const size_t n = (size_t)(N > 0 ? N : 1), t = (size_t)(T > 0 ? T : 1);
    CutState c;
    int* p = ws;
    c.ctrl = p; p += 8;
    c.totals = p; p += t;
    c.bstar = p; p += t;
    c.boff = p; p += GS_CUT_BUCKETS + 8;
    c.boff2 = p; p += GS_CUT_BUCKETS + 8;
    c.bounds = (uint32_t*)p; p += GS_CUT_BUCKETS;
    c.phist = p; p += (size_t)DC_PART * GS_CUT_BUCKETS;
    c.list = p; p += n;
    c.bucket_of = (uint16_t*)p;
    return c;
}

// binning.hip "depth cut": repair pass (complete lists of the flagged tiles -> overflow buffers, sorted) and the
// frame's flagged-tile counter inside the cut workspace
int depth_cut_repair(const float* bin_records, int N, int ntx, int nty, float mh, int row0, int row1,
                     const int* full_ranges, int32_t* workspace, int32_t* cut_ws, uint64_t* okeys, int64_t ocap,
                     int* osorted, const int* flags, hipStream_t s, bool sort_too = true);
int* depth_cut_flag_counter(int32_t* cut_ws, int N, int T);

inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace gs
