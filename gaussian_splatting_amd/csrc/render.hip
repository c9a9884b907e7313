// render.hip -- tile renderer for gfx950: forward compositing, backward, depth.
//
// Replaces render_tiles_kernel (render.cu:8-189), render_tiles_backward_kernel
// (render_backward.cu:12-285) and render_depth_kernel (depth.cu:7-115).
//
// Mapping.  One 256-thread workgroup (4 waves) per 16x16 tile; wave w owns the 8x8 pixel
// quadrant (w&1, w>>1), lane l the pixel (l&7, l>>3) inside it, so that a wave covers a compact
// patch: a splat that misses the patch is skipped by the whole wave with one ballot.  Workgroups
// are laid out so that each XCD renders a contiguous band of tiles (block b runs on XCD b%8):
// neighbouring tiles share Gaussians and hit the same 4 MiB L2.
//
// Data flow.  Per chunk of splats (256 in the fp32 forward, 64 in its backward) the workgroup gathers the
// 48-byte packed record (gs_pack_splats / gs_preprocess_forward: u v r2 opacity | a b c det | 1/det colour --
// three 16-byte loads; or forms it from the reference's separate arrays, stage_chunk) and, for per-pixel SH,
// the colour coefficients of the depth-sorted Gaussians into LDS; every thread then tests the record it
// staged against the tile's four 8x8 patches (build_touch_masks) and a wave walks only the splats whose
// cutoff ellipse reaches its patch, with wave-uniform (broadcast) LDS reads -- in the fp32 forward the next
// visit's record is requested while the current one is composited (lds_record_fetch).  Barriers are uniform
// (Q2 of SURVEY.md is not replicated).  Forward leaves the chunk loop as soon as every pixel of the tile is
// saturated (result-preserving: a saturated pixel ignores all later splats, render.cu:106).
//
// Prefix mode (binning.hip "prefix sort"): a long tile list may have only its 1024 nearest entries
// ordered; the forward reads no further, raises tile_flags[t] if a pixel is still unsaturated there,
// and k_render_fwd_flagged renders such tiles again after their full sort -- results are exact.
//
// Backward starts at the tile's largest num_splats_per_pixel instead of the end of the list and skips the
// reduction for waves none of whose lanes the splat reaches.  The fused renderer's kernel (fp32, one colour
// coefficient: SLOTS) sums the nine per-splat values over the wave with a transposing DPP / permlane-swap
// reduction into a wave-private LDS slot (plain store, no LDS atomics), the flush adds the four waves' slots
// and issues the global atomics nine lanes per 36-byte row of the [V, 9] slab; the other instantiations
// (per-pixel SH, fp64) keep DPP row sums + LDS float atomics and one global atomic per value per
// (splat, tile) -- the reference issues eight (one per warp), unconditionally.  The fused backward starts its
// tiles longest-first from durations the forward measured (k_bwd_prologue: tile_order_body).
//
// Numerics.  The fp32 forward is bit-identical to the CPU restatement: same operation order and
// operand precisions as render.cu (including its double-literal promotions), the IEEE quotient (formed
// from the stored reciprocal, div_by_reciprocal), det_expf in place of __expf.  Backward forms alpha and
// the skip decisions bit-identically (render_backward.cu:141-170) and evaluates the gradient formulas in T.
#include "pg_math.h"
#include "tile_sort.h"
#include <atomic>

namespace gs {

// Instrumented build only (make stats: -DGS_STATS -> libgsplat_hip_stats.so, scripts/render_stats.py):
// per-wave event counts of the render kernels, summed into g_render_stats.  Expands to nothing in
// the product library.
#ifdef GS_STATS
__device__ unsigned long long g_render_stats[32];
// wave timeline (scripts/render_timeline.py, -DGS_STATS -DGS_TIMELINE): one record {begin, end, hw id |
// xcc id, visits, chunks} per wave at slot (kernel, block, wave), times from the 100 MHz constant clock.
// The timeline build does not add to the shared counters (their atomics serialise the waves' exits).
constexpr int GS_TIMELINE_CAP = 1 << 16;   // waves per kernel
constexpr int GS_TIMELINE_W = 10;
__device__ unsigned long long g_timeline[2 * GS_TIMELINE_CAP * GS_TIMELINE_W];
#define GS_STAT_DECL                                                                               \
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                  \
    const unsigned long long st_t0_ = wall_clock64();                                             \
    const unsigned long long st_c0_ = __builtin_readcyclecounter();                                \
    unsigned long long st_ph_ = st_c0_;                                                            \
    (void)st_t0_;                                                                                  \
    (void)st_c0_;                                                                                  \
    (void)st_ph_
#define GS_STAT_FLAG(name) bool name = false
#define GS_STAT_SET(name) name = true
#ifdef GS_TIMELINE
#define GS_STAT(i, v)                                                                              \
    if ((i) == 1 || (i) == 2) st_[i] += (unsigned long long)(v)
// cycles since the previous mark go to phase k (k = 0..4)
#define GS_PHASE(k)                                                                                \
    {                                                                                              \
        const unsigned long long c_ = __builtin_readcyclecounter();                                \
        st_[10 + (k)] += c_ - st_ph_;                                                              \
        st_ph_ = c_;                                                                               \
    }
#define GS_STAT_FLUSH(base)                                                                        \
    if ((threadIdx.x & 63) == 0) {                                                                 \
        const unsigned r_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                   \
        if (r_ < (unsigned)GS_TIMELINE_CAP) {                                                      \
            unsigned long long* t_ = g_timeline + ((size_t)((base) ? GS_TIMELINE_CAP : 0) + r_) * GS_TIMELINE_W; \
            for (int q_ = 0; q_ < 5; q_++) t_[5 + q_] = st_[10 + q_];                              \
            t_[0] = st_t0_;                                                                        \
            t_[1] = wall_clock64();                                                                \
            t_[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32) | \
                    (unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));              \
            t_[3] = st_[2];                                                                        \
            t_[4] = (st_[1] << 48) | ((__builtin_readcyclecounter() - st_c0_) & 0xffffffffffffull);   \
        }                                                                                          \
    }
#else
#define GS_PHASE(k)
#define GS_STAT(i, v) st_[i] += (unsigned long long)(v)
// what two independent 8x4 half-waves in lockstep would walk: per chunk, the visits in which the upper
// / the lower half of the patch has a lane inside the cutoff circle; the lockstep walk takes the larger of the two
#define GS_HALF_DECL unsigned st_ha_ = 0, st_hb_ = 0
#define GS_HALF_VISIT(bal)                                                                         \
    {                                                                                              \
        const unsigned long long b_ = (bal);                                                       \
        st_ha_ += (b_ & 0xffffffffull) != 0;                                                       \
        st_hb_ += (b_ >> 32) != 0;                                                                 \
    }
#define GS_HALF_CHUNK(i)                                                                           \
    {                                                                                              \
        st_[i] += st_ha_ > st_hb_ ? st_ha_ : st_hb_;                                               \
        st_[(i) + 1] += st_ha_;                                                                    \
        st_[(i) + 2] += st_hb_;                                                                    \
        st_ha_ = st_hb_ = 0;                                                                       \
    }
#define GS_STAT_FLUSH(base)                                                                        \
    if ((threadIdx.x & 63) == 0) {                                                                 \
        for (int q_ = 0; q_ < 16; q_++)                                                            \
            if (st_[q_]) atomicAdd(&g_render_stats[(base) + q_], st_[q_]);                         \
    }
#endif
#else
#define GS_STAT_DECL
#define GS_STAT(i, v)
#define GS_STAT_FLAG(name)
#define GS_STAT_SET(name)
#define GS_STAT_FLUSH(base)
#define GS_PHASE(k)
#endif
#ifndef GS_HALF_DECL
#define GS_HALF_DECL
#define GS_HALF_VISIT(bal)
#define GS_HALF_CHUNK(i)
#endif

#ifndef GS_BWD_GROUP
#define GS_BWD_GROUP 16   // lanes summed with DPP before the LDS atomic (measured: 16 -> 0.90 ms, 64 -> 1.03, 8 -> 1.52)
#endif
#ifndef GS_BWD_CHUNK
// splats staged per step by the fused renderer's backward.  64: 12.5 KB of LDS and 71 VGPRs = 7 waves per
// SIMD (128: 25 KB and 78 VGPRs = 6; 0.514 -> 0.505 ms at D, 0.155 -> 0.149 at B; with amdgpu_waves_per_eu(8)
// the compiler reaches 64 VGPRs without spilling but the kernel is no faster)
#define GS_BWD_CHUNK 64
#endif
constexpr int RB = 256;      // workgroup size = pixels per tile
// (experiment builds: bytes of unused dynamic LDS per workgroup of the fused frame's render kernels -- caps the
// workgroups a CU holds, i.e. trades waves in flight for shorter-lived ones; profiles/r04/kbench_lds_pad.txt)
#ifndef GS_FWD_LDS_PAD
#define GS_FWD_LDS_PAD 0
#endif
#ifndef GS_BWD_LDS_PAD
#define GS_BWD_LDS_PAD 0
#endif

// splats staged in LDS per step (at most one per thread); smaller for the wide test-only
// instantiations so that every kernel stays below 64 KiB of static LDS
#ifndef GS_FWD_CHUNK
#define GS_FWD_CHUNK 256   // the fp32 one-coefficient forward (the fused renderer's); 128: +3 %, 64: +9 % at D
#endif
template <typename T, int N_SH> struct Chunk {
    static constexpr int value = (sizeof(T) * N_SH <= 4) ? GS_FWD_CHUNK : (sizeof(T) * N_SH <= 8) ? 256 : (sizeof(T) * N_SH <= 36) ? 128 : 64;
};

template <typename T> struct alignas(16) Vec4 { T x, y, z, w; };

template <int N_SH> struct ColW { static constexpr int value = (3 * N_SH + 3) & ~3; };

// reference chunk sizes, needed only for bug-compatibility with render_backward.cu:185 (Q1)
template <typename T> __host__ __device__ constexpr int ref_chunk(int n_sh);
template <> __host__ __device__ constexpr int ref_chunk<float>(int n_sh) {
    return n_sh == 1 ? 960 : n_sh == 4 ? 576 : n_sh == 9 ? 320 : 160;
}
template <> __host__ __device__ constexpr int ref_chunk<double>(int n_sh) {
    return n_sh == 1 ? 320 : n_sh == 4 ? 160 : n_sh == 9 ? 128 : 64;
}

// XCD-aware tile order: block b -> tile index inside [0, nt) (or >= nt: nothing to do).  The hardware
// hands block b to XCD b % 8, a fixed eighth of the grid each.  Default: one contiguous eighth of the
// frame per XCD (its L2 holds the records neighbouring tiles share).  Dealing runs of 8 / 16 / 32 / 82 consecutive
// tiles to the XCDs in turn instead was measured (scripts/experiments/render_macro_experiments.patch, its XCD_SEG option;
// profiles/r02/kbench_xcd_segments.log): kernel times within 1.5 %, not kept.
__host__ __device__ inline int render_grid(int nt) {
    return ((nt + 7) / 8) * 8;
}
__device__ inline int tile_of_block(int b, int nt) {
    const int per = (nt + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

struct PixelMap {
    int u, v;
};
__device__ inline PixelMap pixel_of_thread(int tile_x, int tile_y, int tid) {
    const int w = tid >> 6, l = tid & 63;
    PixelMap p;
    p.u = tile_x * 16 + ((w & 1) << 3) + (l & 7);
    p.v = tile_y * 16 + ((w >> 1) << 3) + (l >> 3);
    return p;
}

// gather one chunk of splats into LDS: the 12-scalar packed record (three 16-byte loads) and, for
// N_SH > 1, the [3, N_SH] colour coefficients.  s_idx (optional) keeps the Gaussian indices.
// src_opacity != nullptr: the splats come as the reference's separate arrays (render_tiles_cuda's
// uvs / opacity / conic / rgb: `packed` is then uvs[V,2]) and the record is formed here, with the
// function gs_pack_splats uses -- the same values, no packing pass and no [V,12] buffer for the caller.
// TWO_B: the cross term's factor b + b (render.cu:129, render_backward.cu:155) is the same for every pixel -- one add per
// staged record where it was one per visit.  1 (the backward, which never reads det): in word 7 of the staged record;
// 2 (the pipelined forward, which needs det): in word 5, INSTEAD of b -- its only other reader, the touch-mask test of
// the same thread, halves it again (exact).
template <typename T, int N_SH, int TWO_B = 0>
__device__ inline void stage_chunk(const T* __restrict__ packed, const T* __restrict__ rgb,
                                   const int* __restrict__ sorted, int first, int count, int tid,
                                   T* s_geom, T* s_col, int* s_idx, const T* __restrict__ src_opacity = nullptr,
                                   const T* __restrict__ src_conic = nullptr) {
    constexpr int CW = ColW<N_SH>::value;
    if (tid < count) {
        const int g = sorted[first + tid];
        if (src_opacity != nullptr) {
            const T c3[3] = {src_conic[(size_t)g * 3 + 0], src_conic[(size_t)g * 3 + 1], src_conic[(size_t)g * 3 + 2]};
            T col[3] = {0, 0, 0};
            if constexpr (N_SH == 1) {
                col[0] = rgb[(size_t)g * 3 + 0]; col[1] = rgb[(size_t)g * 3 + 1]; col[2] = rgb[(size_t)g * 3 + 2];
            }
            T p[GS_PACKED_WIDTH];
            pack_record<T>(packed[(size_t)g * 2 + 0], packed[(size_t)g * 2 + 1], c3, src_opacity[g], col, p);
#pragma unroll
            for (int k = 0; k < GS_PACKED_WIDTH; k++) s_geom[tid * GS_PACKED_WIDTH + k] = p[k];
        } else {
            const Vec4<T>* src = reinterpret_cast<const Vec4<T>*>(packed + (size_t)g * GS_PACKED_WIDTH);
            Vec4<T>* dst = reinterpret_cast<Vec4<T>*>(s_geom + tid * GS_PACKED_WIDTH);
            dst[0] = src[0];
            dst[1] = src[1];
            dst[2] = src[2];
        }
        if constexpr (TWO_B != 0) {
            const T b = s_geom[tid * GS_PACKED_WIDTH + 5];
            s_geom[tid * GS_PACKED_WIDTH + (TWO_B == 1 ? 7 : 5)] = b + b;
        }
        if (s_idx) s_idx[tid] = g;
    }
    if constexpr (N_SH > 1) {
        // the [3, N_SH] coefficient rows, by the whole workgroup: consecutive threads read consecutive words of
        // a row (one thread per row with 3 N_SH dependent-latency loads was 0.2 ms of the 0.4 ms skeleton of the
        // N_SH = 16 backward at workload B)
        constexpr int C = 3 * N_SH;
        for (int k = tid; k < count * C; k += RB) {
            const int r = k / C, c = k - r * C;
            s_col[r * CW + c] = rgb[(size_t)sorted[first + r] * C + c];
        }
    }
}

template <typename T> __device__ constexpr bool fast_mode() { return sizeof(T) == 4; }

// wave ballot straight from the comparison's lane mask (HIP's __ballot first materialises the
// predicate as an integer: v_cndmask + v_cmp per call)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// exp(-0.5 * mh) of the render loops (render.cu:134, render_backward.cu:160), fp32: det_expf without
// its range selects.  -0.5 * mh is exact, so t below is the same single rounding as det_expf's; for
// t > -125 the remaining operations are det_expf's own, bit for bit.  Below that both values are
// < 2^-100, and in fp32 the callers only use such a value through `opacity * value < 1/255`
// (render.cu:145, render_backward.cu:170), which it fails either way.  mh > 0 and not NaN here.
__device__ __forceinline__ float exp_neg_half(float mh) {
    const float t = mh * (-0.5f * 1.44269504088896341f);
    const float n = __builtin_rintf(t);
    const float f = t - n;
    float p = 1.54035303933816e-4f;
    p = __builtin_fmaf(p, f, 1.33335581464284e-3f);
    p = __builtin_fmaf(p, f, 9.61812910762848e-3f);
    p = __builtin_fmaf(p, f, 5.55041086648216e-2f);
    p = __builtin_fmaf(p, f, 2.40226506959101e-1f);
    p = __builtin_fmaf(p, f, 6.93147180559945e-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}
__device__ __forceinline__ double exp_neg_half(double mh) { return exp(-0.5 * mh); }

// q / det, correctly rounded, from the correctly rounded reciprocal r = RN(1 / det) of the packed
// record ((float)(1.0 / (double)det) equals 1.0f / det for every float): y = RN(q r),
// e = q - det y (exact in one fma), RN(y + e r) is the IEEE quotient (Markstein) for every normal
// q, det with a normal quotient -- 3 instructions instead of the 10 of the division expansion.
// Checked against `/` on 3.2e9 random and edge-mantissa pairs.  Where the quotient
// overflows or det is 0 the result is NaN instead of +-inf; `mh > 0` is then false where the
// reference finds exp(-inf) = 0: alpha = 0 either way.
__device__ __forceinline__ float div_by_reciprocal(float q, float det, float r) {
    const float y = q * r;
    const float e = __builtin_fmaf(-det, y, q);
    return __builtin_fmaf(e, r, y);
}
__device__ __forceinline__ double div_by_reciprocal(double q, double det, double) { return q / det; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }
// background weight of the backward's first contributor, render_backward.cu:175:
//     const T background_weight = 1.0 - (alpha * weight + 1.0 - weight);
// fp32: x = alpha * weight is a float product; the double literals promote the rest -- (x + 1.0) - weight, then 1.0 - that
// -- and the result is narrowed to float.  With alpha in [1/255, 0.9999] (the branch it sits in) and weight = the
// forward's final weight 1 - acc in [1e-4, 1], x and weight are multiples of 2^-45 below 2, so every one of the three
// double operations is EXACT (46 bits of 53) and the value narrowed is weight - x itself: RN_float(weight - x), which is
// what one IEEE float subtraction returns.  One v_sub_f32 for 2 conversions + 3 fp64 adds + the conversion back (fp64-rate
// instructions, ~4 cycles each, executed in every visit in which some pixel of the wave meets its first contributor);
// bit-identical (4e8 random and edge-mantissa pairs on the host: tests/test_host_logic.py holds a sample).
template <typename T> __device__ __forceinline__ T background_weight(T alpha, T weight);
template <> __device__ __forceinline__ float background_weight<float>(float alpha, float weight) {
    const float x = alpha * weight;
    return weight - x;
}
template <> __device__ __forceinline__ double background_weight<double>(double alpha, double weight) {
    return 1.0 - (alpha * weight + 1.0 - weight);
}

// a register the compiler must treat as defined without an instruction that defines it (its content is whatever the
// lane held): for values only SOME lanes compute and the others are masked out of afterwards
template <typename T> __device__ __forceinline__ T unset() {
    T x;
    asm volatile("" : "=v"(x));
    return x;
}
// a * b with 0 * x = 0 for EVERY x (v_mul_legacy_f32: NaN and infinity included): the masked-out factor of a lane may be
// anything.  Equal to a * b whenever both are finite
__device__ __forceinline__ float mul_zero_wins(float a, float b) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double mul_zero_wins(double a, double b) { return b == 0.0 ? 0.0 : a * b; }

template <typename T> __device__ inline T tmin(T a, T b) { return b < a ? b : a; }
template <typename T> __device__ inline T tmax(T a, T b) { return b > a ? b : a; }

// moves a wave-uniform 64-bit value into scalar registers
__device__ inline unsigned long long wave_uniform(unsigned long long m) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m >> 32));
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}

// Per-chunk wave-level touch masks.  Thread t tests the cutoff circle (centre u,v, radius^2 r2) of
// staged splat t against the four 8x8 pixel patches of the tile; ballots turn that into one 64-bit
// mask per (patch, 64 splats).  A wave then visits only the splats whose mask bit is set.  Exactly
// result-preserving: the patch distance (dx, dy) is the per-pixel (du, dv) of the nearest pixel or
// 0, float multiply/add are monotone, so dx*dx + dy*dy > r2 implies du*du + dv*dv > r2 for every
// pixel of the patch, i.e. every lane would have taken the "alpha < 1/255" path.  NaNs compare
// false and keep the splat.
template <typename T, int CHUNK, bool B_DOUBLED = false>
__device__ inline void build_touch_masks(const T* s_geom, int cnt, int tid, int tile_x, int tile_y,
                                         unsigned long long (*s_mask)[CHUNK / 64 > 0 ? CHUNK / 64 : 1]) {
    if (tid < CHUNK) {   // CHUNK <= 256 == workgroup size
        unsigned touch = 0;
        if (tid < cnt) {
            const T* rec = s_geom + tid * GS_PACKED_WIDTH;
            const T u = rec[0], v = rec[1], r2 = rec[2];
            // Exact test of the cutoff ellipse {d : d' S^-1 d <= tau_m} against each patch rectangle:
            // the minimum of q(d) = (c dx^2 - 2 b dx dy + a dy^2) / det over the rectangle of pixel
            // offsets is 0 if the centre lies inside, else it is attained on an edge, where q is a
            // 1-D parabola.  tau_m = r2 / lmax >= 1.05 tau (r2 carries the margin of cutoff_r2), and
            // the continuous minimum bounds every pixel's q from below, so q_min > tau_m means no
            // pixel of the patch can reach alpha >= 1/255.  Cheaper bounds first (circle).
            // (b/a, b/c, 1/lmax from the hardware reciprocal, 1 ulp: where the clamped minimiser lands
            // 1e-7 off, q exceeds the true edge minimum by a second-order 1e-14 -- the 1.001 on tau_m and
            // the 5 % margin inside r2 cover that a million times over.  17 IEEE divisions per splat and
            // chunk were 14 % of the forward kernel's wave time.)
            bool use_q = false;
            T a = 0, b = 0, c = 0, rdet = 0, tau_m = 0, b_over_a = 0, b_over_c = 0;
            if (fast_mode<T>() && r2 > T(0) && r2 < T(1e30)) {
                a = rec[4]; b = B_DOUBLED ? T(0.5) * rec[5] : rec[5]; c = rec[6]; rdet = rec[8];
                const T half = T(0.5) * (a + c);
                const T lmax = half + fast_sqrt(T(0.25) * (a - c) * (a - c) + b * b);
                tau_m = (r2 * fast_rcp(lmax)) * T(1.001);
                use_q = a > T(0) && c > T(0) && rdet > T(0);
                b_over_a = b * fast_rcp(a);
                b_over_c = b * fast_rcp(c);
            }
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const T x0 = T(tile_x * 16 + ((p & 1) << 3)), y0 = T(tile_y * 16 + ((p >> 1) << 3));
                const T x1 = x0 + T(7), y1 = y0 + T(7);
                T dx = T(0), dy = T(0);
                if (u < x0) dx = x0 - u;
                if (u > x1) dx = x1 - u;
                if (v < y0) dy = y0 - v;
                if (v > y1) dy = y1 - v;
                bool hit = !(dx * dx + dy * dy > r2);
                if (hit && use_q && (dx != T(0) || dy != T(0))) {
                    // rectangle of offsets [X0, X1] x [Y0, Y1]; minimise q on its four edges
                    const T X0 = x0 - u, X1 = x1 - u, Y0 = y0 - v, Y1 = y1 - v;
                    T qmin = T(3.0e38);
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const T X = e ? X1 : X0;            // vertical edges: dy* = b X / a
                        T yy = b_over_a * X;
                        yy = tmin<T>(tmax<T>(yy, Y0), Y1);
                        qmin = tmin<T>(qmin, (c * X * X - T(2) * b * X * yy + a * yy * yy) * rdet);
                        const T Y = e ? Y1 : Y0;            // horizontal edges: dx* = b Y / c
                        T xx = b_over_c * Y;
                        xx = tmin<T>(tmax<T>(xx, X0), X1);
                        qmin = tmin<T>(qmin, (c * xx * xx - T(2) * b * xx * Y + a * Y * Y) * rdet);
                    }
                    hit = !(qmin > tau_m);
                }
                if (hit) touch |= 1u << p;
            }
        }
        const int w = tid >> 6;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const unsigned long long m = __ballot((touch >> p) & 1u);
            if ((tid & 63) == 0) s_mask[p][w] = m;
        }
    }
}

// The same masks for a 64-entry chunk (the fused backward's), by the whole workgroup: wave p tests the 64 staged
// records against patch p -- one patch test per thread instead of four on wave 0 while the other three waves wait at
// the barrier (the masks were 9 % of the backward's wave time, profiles/r04/wave_timeline_D.json).  Every
// (record, patch) pair is tested with the very expressions of build_touch_masks: the same bits.  The records must be
// visible to all waves (barrier after staging).
template <typename T>
__device__ inline void build_touch_masks_by_patch(const T* s_geom, int cnt, int tid, int tile_x, int tile_y,
                                                  unsigned long long (*s_mask)[1]) {
    const int r = tid & 63, p = tid >> 6;
    bool hit = false;
    if (r < cnt) {
        const T* rec = s_geom + r * GS_PACKED_WIDTH;
        const T u = rec[0], v = rec[1], r2 = rec[2];
        bool use_q = false;
        T a = 0, b = 0, c = 0, rdet = 0, tau_m = 0, b_over_a = 0, b_over_c = 0;
        if (fast_mode<T>() && r2 > T(0) && r2 < T(1e30)) {
            a = rec[4]; b = rec[5]; c = rec[6]; rdet = rec[8];
            const T half = T(0.5) * (a + c);
            const T lmax = half + fast_sqrt(T(0.25) * (a - c) * (a - c) + b * b);
            tau_m = (r2 * fast_rcp(lmax)) * T(1.001);
            use_q = a > T(0) && c > T(0) && rdet > T(0);
            b_over_a = b * fast_rcp(a);
            b_over_c = b * fast_rcp(c);
        }
        const T x0 = T(tile_x * 16 + ((p & 1) << 3)), y0 = T(tile_y * 16 + ((p >> 1) << 3));
        const T x1 = x0 + T(7), y1 = y0 + T(7);
        T dx = T(0), dy = T(0);
        if (u < x0) dx = x0 - u;
        if (u > x1) dx = x1 - u;
        if (v < y0) dy = y0 - v;
        if (v > y1) dy = y1 - v;
        hit = !(dx * dx + dy * dy > r2);
        if (hit && use_q && (dx != T(0) || dy != T(0))) {
            const T X0 = x0 - u, X1 = x1 - u, Y0 = y0 - v, Y1 = y1 - v;
            T qmin = T(3.0e38);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const T X = e ? X1 : X0;
                T yy = b_over_a * X;
                yy = tmin<T>(tmax<T>(yy, Y0), Y1);
                qmin = tmin<T>(qmin, (c * X * X - T(2) * b * X * yy + a * yy * yy) * rdet);
                const T Y = e ? Y1 : Y0;
                T xx = b_over_c * Y;
                xx = tmin<T>(tmax<T>(xx, X0), X1);
                qmin = tmin<T>(qmin, (c * xx * xx - T(2) * b * xx * Y + a * Y * Y) * rdet);
            }
            hit = !(qmin > tau_m);
        }
    }
    const unsigned long long m = __ballot(hit);
    if (r == 0) s_mask[p][0] = m;
}

// Touch masks handed from the forward to the backward (round 6).  The backward walks the same lists as the forward and
// needs the same (64 entries x 4 patches) touch masks; the forward, which builds them for every chunk it stages, leaves
// the first GS_MASK_WORDS words of every tile behind: touch_masks[(tile * GS_MASK_WORDS + word) * 4 + patch], 512 bytes
// per tile of the GRID (the words beyond -- complete lists walked deeper than 1024 entries -- the backward builds itself).
// Whatever list a tile's final forward pass rendered from (ordered prefix, kept depth prefix, or the complete list of a
// flagged tile's repair pass) is the list its backward walks, and the last pass to render the tile wrote the words.
constexpr int GS_MASK_WORDS = GS_SORT_PREFIX / 64;

// colour of splat i of the staged chunk at this pixel's view direction
template <typename T, int N_SH>
__device__ inline void splat_colour(const T* s_geom, const T* s_col, int i, const T* Y, T* col) {
    if constexpr (N_SH == 1) {
        // sh_to_rgb with one coefficient per channel (spherical_harmonics.cuh:83): the record holds
        // Y0 * coefficient
        const Vec4<T> g2 = *reinterpret_cast<const Vec4<T>*>(s_geom + i * GS_PACKED_WIDTH + 8);
        col[0] = g2.y;
        col[1] = g2.z;
        col[2] = g2.w;
    } else {
        sh_to_rgb<T, N_SH>(s_col + i * ColW<N_SH>::value, Y, col);
    }
}

// ---- software-pipelined LDS record reads (fp32 kernels of the fused renderer) -------------------------
// A wave walking its touch mask is latency-bound, not issue-bound: the wave timeline
// (scripts/render_timeline.py) shows the same ~1200 cycles per visit per wave whether 8 waves or 1 share
// the SIMD, and three dependent LDS round trips (u v r2 | conic, opacity | colour) are a third of it.  The
// whole 48-byte record of the NEXT visit is therefore requested (three ds_read_b128, uniform address)
// before the current visit is worked on, into the other of two register sets.  Volatile loads, so that
// the compiler neither sinks them to their first use nor merges them, pinned by a scheduling barrier; it
// still tracks them, i.e. waits for the current record while the next one is in flight.  Costs 18 VGPRs
// (54 -> 72, 8 -> 7 waves per SIMD) and still wins: forward 0.274 -> 0.265 ms at workload D; prefetching
// only u v r2 opacity | conic (61 VGPRs, 8 waves) and the colour quad at the visit is slower (0.256 vs
// 0.248 ms with the division-free touch masks in both).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) f32x4* LdsVec4Ptr;
struct LdsRecord {
    f32x4 g0, g1, g2;   // u v r2 opacity | a b c det | 1/det colour
};
__device__ __forceinline__ void lds_record_fetch(LdsRecord& r, const float* rec) {
    LdsVec4Ptr p = (LdsVec4Ptr)(const __attribute__((address_space(3))) float*)rec;
    r.g0 = p[0];
    r.g1 = p[1];
    r.g2 = p[2];
    __builtin_amdgcn_sched_barrier(0);
}

// ---- longest-first tile order for the backward -------------------------------------------------------
// The hardware starts workgroups in grid order and the kernel ends when the last one does; a workgroup
// lives ~150 us of the 520 us backward, so the tail in which the chip drains is a fifth of the kernel
// (scripts/render_timeline.py).  Started longest-first, the tiles that finish last are the short ones.
// The forward measures each tile's own duration (shader clock; the backward of a tile is slow where its
// forward was: same lists, same hit pattern, correlation 0.7) and tile_order_body turns the costs into a
// launch order.  Every XCD keeps its contiguous eighth of the frame (tile_of_block: neighbouring tiles
// add to the same Gaussians' rows, and with one L2 per XCD a frame-wide order doubled the kernel's HBM
// traffic, 194 -> 405 MB): the order is longest-first WITHIN each eighth -- a counting sort on
// 8 x 128 (eighth, cost class) bins -- and block b = 8 j + x starts the j-th tile of eighth x.
// Only the start order changes, no result does.
constexpr int GS_LPT_MIN_TILES = 2048;   // smaller grids: the 6 us of the order kernel exceed the gain
__device__ __forceinline__ void tile_order_body(const int* __restrict__ cost, int tile0, int nt, int n_grid,
                                                int* __restrict__ order) {
    __shared__ int s_hist[1024];
    __shared__ int s_wave[16];
    __shared__ int s_max;
    const int tid = threadIdx.x;
    const int per = n_grid >> 3;   // tiles per XCD (render_grid / tile_of_block)
    s_hist[tid] = 0;
    if (tid == 0) s_max = 1;
    for (int b = tid; b < n_grid; b += 1024) order[b] = -1;
    __syncthreads();
    int mx = 0;
    for (int t = tid; t < nt; t += 1024) mx = max(mx, cost[tile0 + t]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
    if ((tid & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const float scale = 127.0f / (float)s_max;
    auto bin_of = [&](int t) {   // class 0 = most expensive
        const int c = (int)((float)max(cost[tile0 + t], 0) * scale);
        return (t / per) * 128 + 127 - min(c, 127);
    };
    for (int t = tid; t < nt; t += 1024) atomicAdd(&s_hist[bin_of(t)], 1);
    __syncthreads();
    // exclusive scan of the bin counts: wave scans + one pass over the 16 wave totals
    const int v = s_hist[tid];
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < (tid >> 6); w++) off += s_wave[w];
    s_hist[tid] = off + incl - v;
    __syncthreads();
    for (int t = tid; t < nt; t += 1024) {
        const int x = t / per;
        const int rank = atomicAdd(&s_hist[bin_of(t)], 1) - x * per;   // the eighths before x are full
        order[rank * 8 + x] = t;
    }
}

// The backward's prologue, one launch: workgroup 0 makes the launch order (if the tiles are to be started
// longest-first), the others clear the gradient slab the backward accumulates into -- two independent jobs that
// were a 7 us single-workgroup kernel and a 14 us fill one after the other.
constexpr int GS_PROLOGUE_FILL_BLOCKS = 2048;
__global__ __launch_bounds__(1024) void k_bwd_prologue(const int* __restrict__ cost, int tile0, int nt, int n_grid,
                                                       int* __restrict__ order, int ordered, float* __restrict__ slab,
                                                       size_t n_floats) {
    int fill_block = blockIdx.x, fill_blocks = gridDim.x;
    if (order != nullptr) {
        if (blockIdx.x == 0) {
            if (ordered) {
                tile_order_body(cost, tile0, nt, n_grid, order);
            } else {   // (small grids: the natural order, spelled out)
                for (int b = threadIdx.x; b < n_grid; b += 1024) {
                    const int t = tile_of_block(b, nt);
                    order[b] = t < nt ? t : -1;
                }
            }
            return;
        }
        fill_block -= 1;
        fill_blocks -= 1;
    }
    const size_t n4 = n_floats >> 2;   // (the slab is 16-byte aligned: a torch allocation)
    float4* s4 = reinterpret_cast<float4*>(slab);
    const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (size_t i = (size_t)fill_block * 1024 + threadIdx.x; i < n4; i += (size_t)fill_blocks * 1024) s4[i] = z;
    if (fill_block == 0 && threadIdx.x < (n_floats & 3)) slab[(n4 << 2) + threadIdx.x] = 0.0f;
}

// ---- depth segments of the fused backward ------------------------------------------------------------
// The backward of a tile is a serial walk over its list, back to front; one workgroup per tile lives
// ~150 us, the chip drains for the last fifth of the kernel (4b of DESIGN.md) and a multi-GPU rank's band of
// ~540 tiles is two workgroups per CU.  Splitting the walk by DEPTH gives 3-4x more, 3-4x shorter work
// items -- but a segment that starts in the middle of a pixel's list needs the state the walk would have
// there: the transmittance in front of it and the colour accumulated behind it.  The forward knows both
// and leaves them behind (CK instantiation of the fp32 one-coefficient kernel):
//   per (tile, segment s, pixel)  rec = (P_s, E_s): P_s = product of (1 - alpha) over the segment's contributors,
//                                 E_s = sum of colour * alpha * (product of (1 - alpha) over the segment's EARLIER
//                                 contributors) -- the segment's transmittance and colour as seen from its near
//                                 boundary; 16 bytes, written when the forward crosses into the next segment
//   per pixel                     kend = index after its last contributing splat, and -- evaluated in the
//                                 epilogue for that one splat with the BACKWARD's arithmetic -- 1 - alpha
//                                 and the background weight the backward's first contributor adds
// The backward's walk multiplies weight by 1 / (1 - alpha) per contributor, starting from final_weight (x the
// factor q that render_backward.cu:185 applies -- or not -- at the first contributor, SURVEY.md Q1, which
// then stays in every weight of the walk).  Its weight at the near boundary of segment s' is therefore
//   w(s') = w(s' + 1) / P_s'      from      w(e) = final_weight q (1 - alpha_last) / P_e     (e: the segment
// of the last contributor, whose own factor is in P_e), the colour the walk has accumulated behind segment s is
// bg_weight bg + sum_{s' > s} w(s') E_s', and a workgroup (tile, s) starts every pixel whose list goes deeper
// than its segment from those two; pixels that end inside the segment start as before, pixels that ended in
// front of it do nothing.  (Why products of the (1 - alpha) and not the forward's own transmittance 1 - acc:
// the reference's walk starts from final_weight = 1 - acc of a nearly saturated pixel, whose cancellation error
// -- up to 1e-3 relative -- it carries into every weight of the pixel; a segment that started from the
// forward's accurate 1 - acc at its boundary would be closer to the true derivative and 3e-4 away from the
// reference.  Measured.)
// "Contributor" means: in the BACKWARD's arithmetic -- its alpha differs from the forward's in the last ulp
// and the two can disagree at the 1/255 threshold; the forward evaluates the backward's form where they could
// (see its visit), so the state it leaves describes exactly the walk the unsegmented kernel would do.
#ifndef GS_SEG_LEN
#define GS_SEG_LEN 128
#endif
#ifndef GS_SEG_MAX
#define GS_SEG_MAX 8
#endif
constexpr int SEG_LEN = GS_SEG_LEN;   // entries per segment (a multiple of the 64-entry mask words)
constexpr int SEG_MAX = GS_SEG_MAX;   // segments per tile; the last one is open-ended
static_assert(SEG_LEN % 64 == 0 && SEG_MAX >= 2, "segment geometry");
// the segmented backward walks whole chunks from seg_lo / GS_BWD_CHUNK with no `k >= seg_lo` guard: a segment must
// start on a chunk boundary or its first chunk would re-visit the previous segment's entries
static_assert(SEG_LEN % GS_BWD_CHUNK == 0, "a depth segment must be a whole number of backward chunks");
struct SegState {              // views into the caller's workspace (gs_render_segment_workspace_bytes)
    int* kend;                 // [band pixels], by pixel index - pix0
    float* oma_last;           // same
    float* bgw;                // same
    Vec4<float>* rec;          // [band tiles][SEG_MAX][256], by tile index - tile0
    int tile0;                 // first tile of the band
    int pix0;                  // first pixel of the band
};
constexpr SegState SEG_NONE{nullptr, nullptr, nullptr, nullptr, 0, 0};
// The workspace covers the tile rows [row0, row1) only -- segments are on for bands (a multi-GPU rank's eighth of
// the frame): 32 KB per tile of the BAND, not of the grid (round-3 advisor finding: ~1 GB per frame at 4K to use
// an eighth of it).
__host__ __device__ inline size_t seg_band_pixels(int W, int H, int row0, int row1) {
    const int y0 = row0 * 16 < H ? row0 * 16 : H, y1 = row1 * 16 < H ? row1 * 16 : H;
    return (size_t)W * (size_t)(y1 > y0 ? y1 - y0 : 0);
}
__host__ __device__ inline SegState seg_state_of(void* ws, int W, int H, int row0, int row1) {
    SegState st = SEG_NONE;
    if (ws == nullptr) return st;
    const size_t P = seg_band_pixels(W, H, row0, row1);
    const size_t ntx = (size_t)((W + 15) / 16);
    st.tile0 = (int)ntx * row0;
    st.pix0 = W * (row0 * 16 < H ? row0 * 16 : H);
    char* p = (char*)ws;
    st.rec = (Vec4<float>*)p;   // first: 16-byte aligned
    p += ntx * (size_t)(row1 - row0) * SEG_MAX * 256 * sizeof(Vec4<float>);
    st.kend = (int*)p;
    st.oma_last = (float*)(p + 4 * P);
    st.bgw = (float*)(p + 8 * P);
    return st;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// One tile by one workgroup.  sort_prefix > 0: prefix mode (binning.hip "prefix sort"), only the
// first sort_prefix entries of a prefix-sorted tile's segment are there; tile_flags[tile] is set to
// whether the tile ran out of them.  flagged_only: the repair call, full list, flags untouched.
template <typename T, int N_SH, bool CK = false>
__device__ __forceinline__ void render_tile_fwd(
    const int tile, const T* __restrict__ packed, const T* __restrict__ rgb,
    const T* __restrict__ view_dir, const int* __restrict__ ranges, const int* __restrict__ sorted,
    const T* __restrict__ bg, int W, int H, int ntx, int* __restrict__ nsp_out,
    T* __restrict__ fw_out, T* __restrict__ image, int sort_prefix, int* __restrict__ tile_flags,
    bool flagged_only, int64_t cap, int* __restrict__ tile_cost = nullptr,
    const T* __restrict__ src_opacity = nullptr, const T* __restrict__ src_conic = nullptr,
    const SegState seg = SEG_NONE, const int* __restrict__ full_ranges = nullptr,
    int* __restrict__ flag_counter = nullptr, unsigned long long* __restrict__ touch_masks = nullptr) {
    static_assert(!CK || (sizeof(T) == 4 && N_SH == 1), "segment checkpoints: the fused renderer's kernel only");
    constexpr bool fast = sizeof(T) == 4;
    // the tile's own duration in 16-cycle units: the launch-order key of the backward (tile_order_body)
    const unsigned long long cost_c0 = tile_cost ? __builtin_readcyclecounter() : 0ull;
    constexpr int CW = ColW<N_SH>::value;
    constexpr int RCHUNK = Chunk<T, N_SH>::value;
    __shared__ alignas(16) T s_geom[RCHUNK * GS_PACKED_WIDTH];
    __shared__ alignas(16) T s_col[N_SH > 1 ? RCHUNK * CW : 4];

    const int tid = threadIdx.x;
    const PixelMap px = pixel_of_thread(tile % ntx, tile / ntx, tid);
    const bool valid = px.u < W && px.v < H;
    const int s0 = ranges[tile];
    const int n_tile = ranges[tile + 1] - s0;
    // lists enqueued with a speculative capacity: a segment beyond it was never written (the caller
    // repeats the frame's binning and this render with the exact size)
    if ((int64_t)s0 + n_tile > cap) {
        if (tile_cost != nullptr && tid == 0) tile_cost[tile] = 0;
        return;
    }
    // prefix mode (binning.hip "prefix sort"): only the first sort_prefix entries are there
    const bool prefix_only = !flagged_only && prefix_sorted_tile(n_tile, sort_prefix);
    const int n_list = prefix_only ? sort_prefix : n_tile;
    // depth cut (binning.hip "depth cut"): `ranges` describe the kept depth prefix of every list, full_ranges the
    // complete ones; a tile whose list was cut short and that reaches its end unsaturated is flagged and redone
    const bool truncated = !flagged_only && full_ranges != nullptr && full_ranges[tile + 1] - full_ranges[tile] > n_tile;

    T Y[N_SH];
    if constexpr (N_SH > 1) {
        T d[3] = {0, 0, 0};
        if (valid) {
            const T* vd = view_dir + ((size_t)px.v * W + px.u) * 3;
            d[0] = vd[0]; d[1] = vd[1]; d[2] = vd[2];
        }
        sh_basis<T, N_SH>(d, Y);
    } else {
        Y[0] = T(GS_SH_0);
    }

    // A pixel is done when its accumulated alpha exceeds 0.9999 (render.cu:146,162) -- or when it lies outside the
    // image: those lanes start saturated.  "done" is READ OFF acc (one compare per visit) instead of being carried as a
    // flag: the compiler kept the flag as a 0/1 VGPR across the pipelined visits and spent an and, a compare and five
    // moves of every visit's ~52 vector instructions on it.  acc of an outside lane is never used (nothing is stored).
    T acc = valid ? T(0) : T(2), fw = 0;
    auto is_done = [&]() { return acc > Thr<T>::sat_gt(); };
    T img[3] = {0, 0, 0};
    // num_splats_per_pixel == index of the first splat at whose turn the pixel is saturated
    // (render.cu:106,146,162), or the list length: set when the pixel saturates
    int nsp = n_tile;
    const T pu = T(px.u), pv = T(px.v);
    const int wave = tid >> 6;
    constexpr int NW = RCHUNK / 64 > 0 ? RCHUNK / 64 : 1;
    __shared__ unsigned long long s_mask[4][NW];

    // CK: state for the depth-segmented backward (see "depth segments" above)
    T segL = 1;                    // product of (1 - alpha) over the current segment's contributors so far
    T sg0 = 0, sg1 = 0, sg2 = 0;   // sum of colour * alpha * (that product before the splat): the segment's colour
    int kend = 0;                  // index after the last contributing splat
    int b_cur = 0;                 // current segment (wave-uniform)
    bool wave_fin = false;         // the wave's last record is written (every pixel of the patch saturated)
    auto write_record = [&]() {
        if constexpr (CK) {
            Vec4<float> r4;
            r4.x = segL; r4.y = sg0; r4.z = sg1; r4.w = sg2;
            seg.rec[((size_t)(tile - seg.tile0) * SEG_MAX + b_cur) * RB + tid] = r4;
            segL = 1; sg0 = 0; sg1 = 0; sg2 = 0;
        }
    };

    bool all_done = false;
    GS_STAT_DECL;
    GS_HALF_DECL;
    GS_STAT(0, 1);          // waves
    GS_STAT(7, n_tile);     // list entries of the wave's tile
    for (int base = 0; base < n_list; base += RCHUNK) {
        const int cnt = min(RCHUNK, n_list - base);
        GS_STAT(1, 1);      // chunks
        GS_STAT(8, cnt);    // list entries staged
        constexpr bool B2 = fast && N_SH == 1;   // the pipelined walk below reads b + b from the record
        stage_chunk<T, N_SH, B2 ? 2 : 0>(packed, rgb, sorted, s0 + base, cnt, tid, s_geom, s_col, nullptr, src_opacity, src_conic);
        GS_PHASE(0);
        // (no barrier in between: thread t tests the record thread t staged)
        build_touch_masks<T, RCHUNK, B2>(s_geom, cnt, tid, tile % ntx, tile / ntx, s_mask);
        __syncthreads();
        if constexpr (fast && N_SH == 1) {
            // the chunk's masks for the backward (see GS_MASK_WORDS): thread (word, patch) stores one 8-byte word
            if (touch_masks != nullptr && tid < 4 * NW) {
                const int word = base / 64 + (tid >> 2);
                if (word < GS_MASK_WORDS) touch_masks[((size_t)tile * GS_MASK_WORDS + word) * 4 + (tid & 3)] = s_mask[tid & 3][tid >> 2];
            }
        }
        GS_PHASE(1);
        if constexpr (fast && N_SH == 1) {
            // pipelined walk: the record of the next visit is in flight while this one is composited
            auto visit = [&](const LdsRecord& r, int i) {
                GS_STAT(2, 1);
                GS_STAT(6, __popcll(ballot(!is_done())));
                GS_STAT_FLAG(st_in);
                GS_STAT_FLAG(st_hit);
                if (!is_done()) {
                    const T du = pu - r.g0.x, dv = pv - r.g0.y;
                    if (!(du * du + dv * dv > r.g0.z)) {
                        GS_STAT_SET(st_in);
                        const T a = r.g1.x, tb = r.g1.y, c = r.g1.z, det = r.g1.w;   // tb = b + b (stage_chunk)
                        const T mh_num = c * du * du - tb * du * dv + a * dv * dv;
                        const T mh = div_by_reciprocal(mh_num, det, r.g2.x);
                        T alpha = r.g0.w * exp_neg_half(mh);
                        // render.cu:133: alpha = 0 unless mh > 0 -- and a zero alpha fails the 1/255 test below.  Without
                        // the segment state only that test reads alpha before it is known to pass: `mh > 0` joins the
                        // test's lane mask (one scalar and) instead of a select in front of it
                        const bool mh_pos = mh > T(0);
                        if constexpr (CK) alpha = mh_pos ? alpha : T(0);
                        if constexpr (CK) {
                            // What the BACKWARD's walk will see at this entry.  It forms alpha from mh * (1 / det)
                            // (render_backward.cu:153-157; the forward divides), a last-ulp difference that matters
                            // in two places: next to the 1/255 threshold, where the two forms can disagree on whether
                            // the splat contributes at all (~50 (pixel, splat) pairs of a frame at workload D; the
                            // reference's walk then carries one factor 1 / (1 - 1/255) more or less than its forward
                            // did, and so must the segments), and in 1 - alpha for alpha near 1 (6e-8 / (1 - alpha)
                            // relative on the walk's factor).  There the backward's alpha is evaluated as well
                            // (a few percent of the visits); elsewhere the forward's is within 6e-7 of it.
                            // Values for the backward only: contraction allowed.
#pragma clang fp contract(fast)
                            T ab = alpha;
                            const bool near = __builtin_fabsf(alpha - Thr<T>::alpha_min()) < T(4e-8) || alpha > T(0.9);
                            if (near) {   // (skipped by the whole wave when no lane is near: s_cbranch_execz)
                                const T mh_b = mh_num * r.g2.x;
                                const T e_b = exp_neg_half(mh_b);
                                ab = r.g0.w * ((mh_b > T(0)) ? e_b : T(0));
                            }
                            // branch-free: a splat the backward skips enters with alpha 0 (changes nothing)
                            const bool cb = ab >= Thr<T>::alpha_min();
                            const T cap = Thr<T>::alpha_cap();
                            if (b_cur > 0) {   // (wave-uniform; nobody reads segment 0's record: no segment lies in front of it)
                                const T ac = cb ? ((ab > cap) ? cap : ab) : T(0);   // the backward caps alpha
                                const T aL = ac * segL;
                                sg0 += r.g2.y * aL; sg1 += r.g2.z * aL; sg2 += r.g2.w * aL;
                                segL -= aL;   // segL (1 - ac)
                            }
                            kend = cb ? base + i + 1 : kend;
                        }
                        if (mh_pos & !(alpha < Thr<T>::alpha_min())) {      // render.cu:145
                            GS_STAT_SET(st_hit);
                            // render.cu:149-150: final_weight = 1.0 - acc and weight = alpha * (1.0 - acc), the literal
                            // making both double expressions that are narrowed to float -- six fp64-rate instructions
                            // per contributing visit when written that way.  acc is 0 or >= 1/255 here (it only grows,
                            // and the first weight is an alpha >= 1/255), so 1.0 - (double)acc is EXACT; hence
                            // (a) its narrowing is the correctly rounded float difference, i.e. the float subtraction;
                            // (b) alpha (1 - acc) = alpha - alpha acc as real numbers, and one double fma rounds that
                            // once, exactly as the reference's double multiplication rounds its exact operands.
                            // Same bits, four fp64-rate instructions and one fp32.
                            fw = 1.0f - acc;
                            const T weight = (T)__builtin_fma((double)alpha, -(double)acc, (double)alpha);
                            img[0] += r.g2.y * weight;
                            img[1] += r.g2.z * weight;
                            img[2] += r.g2.w * weight;
                            acc += weight;
                            if (acc > Thr<T>::sat_gt()) {   // saturated: the next splat's check fails
                                nsp = base + i + 1;
                            }
                        }
                    }
                }
                GS_STAT(3, ballot(st_in) != 0);
                GS_STAT(4, ballot(st_hit) != 0);
                GS_STAT(5, __popcll(ballot(st_hit)));
                GS_HALF_VISIT(ballot(st_in));
            };
            for (int word = 0; word < NW && word * 64 < cnt; word++) {
                if constexpr (CK) {
                    // crossing into the next depth segment: leave the state at the boundary behind
                    const int sidx = min((base + word * 64) / SEG_LEN, SEG_MAX - 1);
                    if (sidx != b_cur && !wave_fin) {
                        write_record();
                        b_cur = sidx;
                    }
                }
                if (ballot(!is_done()) == 0) {   // wave-uniform: every pixel of the patch saturated
                    if constexpr (CK) {
                        if (!wave_fin) write_record();
                        wave_fin = true;
                    }
                    break;
                }
                unsigned long long m = wave_uniform(s_mask[wave][word]);
                if (m == 0) continue;
                LdsRecord ra, rb;
                int cur = word * 64 + __builtin_ctzll(m), nxt;
                m &= m - 1;
                lds_record_fetch(ra, s_geom + cur * GS_PACKED_WIDTH);
                while (true) {
                    nxt = -1;
                    if (m) {
                        nxt = word * 64 + __builtin_ctzll(m);
                        m &= m - 1;
                        lds_record_fetch(rb, s_geom + nxt * GS_PACKED_WIDTH);
                    }
                    visit(ra, cur);
                    if (nxt < 0) break;
                    cur = nxt;
                    nxt = -1;
                    if (m) {
                        nxt = word * 64 + __builtin_ctzll(m);
                        m &= m - 1;
                        lds_record_fetch(ra, s_geom + nxt * GS_PACKED_WIDTH);
                    }
                    visit(rb, cur);
                    if (nxt < 0) break;
                    cur = nxt;
                }
            }
        } else {
            for (int word = 0; word < NW && word * 64 < cnt; word++) {
                if (ballot(!is_done()) == 0) break;   // wave-uniform: every pixel of the patch saturated
                unsigned long long m = wave_uniform(s_mask[wave][word]);
                while (m) {
                    const int i = word * 64 + __builtin_ctzll(m);
                    m &= m - 1;
                    GS_STAT(2, 1);                        // visits (touch-mask bits walked)
                    GS_STAT(6, __popcll(ballot(!is_done())));   // live lanes at the visit
                    GS_STAT_FLAG(st_in);
                    GS_STAT_FLAG(st_hit);
                    if (!is_done()) {
                        const T* rec = s_geom + i * GS_PACKED_WIDTH;
                        const Vec4<T> g0 = *reinterpret_cast<const Vec4<T>*>(rec);   // u v r2 opacity
                        const T du = pu - g0.x, dv = pv - g0.y;
                        // beyond the cutoff radius alpha < 1/255 is certain: same outcome as :145-148
                        if (!(fast && du * du + dv * dv > g0.z)) {
                            GS_STAT_SET(st_in);
                            const Vec4<T> g1 = *reinterpret_cast<const Vec4<T>*>(rec + 4);   // a b c det
                            const Vec4<T> g2 = *reinterpret_cast<const Vec4<T>*>(rec + 8);   // 1/det, colour
                            const T a = g1.x, b = g1.y, c = g1.z, det = g1.w;
                            const T mh = div_by_reciprocal(c * du * du - (b + b) * du * dv + a * dv * dv, det, g2.x);
                            T alpha = g0.w * exp_neg_half(mh);
                            alpha = (mh > T(0)) ? alpha : T(0);                 // render.cu:133
                            if (!(fast && alpha < Thr<T>::alpha_min())) {       // render.cu:145
                                GS_STAT_SET(st_hit);
                                fw = 1.0 - acc;
                                const T weight = alpha * (1.0 - acc);           // double, narrowed
                                T col[3];
                                splat_colour<T, N_SH>(s_geom, s_col, i, Y, col);
#pragma unroll
                                for (int ch = 0; ch < 3; ch++) img[ch] += col[ch] * weight;
                                acc += weight;
                                if (acc > Thr<T>::sat_gt()) {   // saturated: the next splat's check fails
                                    nsp = base + i + 1;
                                }
                            }
                        }
                    }
                    GS_STAT(3, ballot(st_in) != 0);           // visits with a lane inside the cutoff circle
                    GS_STAT(4, ballot(st_hit) != 0);          // visits with a contributing lane
                    GS_STAT(5, __popcll(ballot(st_hit)));     // contributing (pixel, splat) pairs
                }
            }
        }
        GS_PHASE(2);
        GS_HALF_CHUNK(9);
        all_done = __syncthreads_and(is_done());
        GS_PHASE(3);
        if (all_done) break;
    }
    GS_STAT_FLUSH(0);
    if (tile_cost != nullptr && tid == 0)
        tile_cost[tile] = (int)min((__builtin_readcyclecounter() - cost_c0) >> 4, 0x3fffffffull);
    // an unsaturated pixel at the end of the prefix / of the cut list: the tile is redone from its full list
    if (tile_flags != nullptr && !flagged_only && tid == 0) {
        const bool flag = (prefix_only || truncated) && !all_done;
        tile_flags[tile] = flag;
        if (flag && flag_counter != nullptr) atomicAdd(flag_counter, 1);
    }

    if constexpr (CK) {
        if (!wave_fin) write_record();   // the segment the list (or its ordered prefix) ended in
        if (valid) {
            // the pixel's last contributor once more, in the BACKWARD's arithmetic (k_render_bwd: alpha from
            // mh * (1 / det), capped at 0.9999): what its first step does to weight and colour_accum
            T oma_last = 1, bgw = 0;
            if (kend > 0) {
                const int g = sorted[s0 + kend - 1];
                const Vec4<T>* rec = reinterpret_cast<const Vec4<T>*>(packed + (size_t)g * GS_PACKED_WIDTH);
                const Vec4<T> g0 = rec[0], g1 = rec[1], g2 = rec[2];
                const T du = pu - g0.x, dv = pv - g0.y;
                const T mh = (g1.z * du * du - (g1.y + g1.y) * du * dv + g1.x * dv * dv) * g2.x;
                const T e = exp_neg_half(mh);
                T alpha = g0.w * ((mh > T(0)) ? e : T(0));
                if (alpha > Thr<T>::sat_gt()) alpha = Thr<T>::alpha_cap();
                const T bw = background_weight<T>(alpha, fw);   // render_backward.cu:172-181
                if (bw > Thr<T>::bgw_gt()) bgw = bw;
                oma_last = T(1) - alpha;
            }
            const size_t p = (size_t)px.v * W + px.u - seg.pix0;
            seg.kend[p] = kend;
            seg.oma_last[p] = oma_last;
            seg.bgw[p] = bgw;
        }
    }
    if (valid) {
        if (acc < Thr<T>::bg_lt()) {   // render.cu:169
#pragma unroll
            for (int ch = 0; ch < 3; ch++) img[ch] += bg[ch] * (1.0 - acc);
        }
        const size_t p = (size_t)px.v * W + px.u;
        nsp_out[p] = nsp;
        fw_out[p] = fw;
        image[p * 3 + 0] = img[0];
        image[p * 3 + 1] = img[1];
        image[p * 3 + 2] = img[2];
    }
}

template <typename T, int N_SH>
__global__ __launch_bounds__(RB) void k_render_fwd(
    const T* __restrict__ packed, const T* __restrict__ rgb, const T* __restrict__ view_dir,
    const int* __restrict__ ranges, const int* __restrict__ sorted, const T* __restrict__ bg,
    int W, int H, int ntx, int tile0, int nt, int* __restrict__ nsp_out, T* __restrict__ fw_out,
    T* __restrict__ image, int sort_prefix, int* __restrict__ tile_flags, int64_t cap,
    int* __restrict__ tile_cost, const T* __restrict__ src_opacity, const T* __restrict__ src_conic,
    const int* __restrict__ full_ranges, int* __restrict__ flag_counter, unsigned long long* __restrict__ touch_masks) {
    const int t_local = tile_of_block(blockIdx.x, nt);
    if (t_local >= nt) return;
    render_tile_fwd<T, N_SH>(tile0 + t_local, packed, rgb, view_dir, ranges, sorted, bg, W, H, ntx,
                             nsp_out, fw_out, image, sort_prefix, tile_flags, false, cap, tile_cost, src_opacity,
                             src_conic, SEG_NONE, full_ranges, flag_counter, touch_masks);
}

// the fused renderer's forward that also leaves the state for the depth-segmented backward
#ifndef GS_FWD_CK_WAVES
#define GS_FWD_CK_WAVES 5   // no spills at 5 (90 VGPRs); the kernel runs where a band leaves 2-3 waves per SIMD anyway
#endif
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(GS_FWD_CK_WAVES, 8))) void k_render_fwd_ck(
    const float* __restrict__ packed, const float* __restrict__ rgb, const int* __restrict__ ranges,
    const int* __restrict__ sorted, const float* __restrict__ bg, int W, int H, int ntx, int tile0, int nt,
    int* __restrict__ nsp_out, float* __restrict__ fw_out, float* __restrict__ image, int sort_prefix,
    int* __restrict__ tile_flags, int64_t cap, int* __restrict__ tile_cost, const SegState seg,
    unsigned long long* __restrict__ touch_masks) {
    const int t_local = tile_of_block(blockIdx.x, nt);
    if (t_local >= nt) return;
    render_tile_fwd<float, 1, true>(tile0 + t_local, packed, rgb, nullptr, ranges, sorted, bg, W, H, ntx, nsp_out,
                                    fw_out, image, sort_prefix, tile_flags, false, cap, tile_cost, nullptr, nullptr,
                                    seg, nullptr, nullptr, touch_masks);
}

// repair pass of the prefix mode / of the depth cut, ONE launch: a small grid walks the flags; a flagged tile's list is
// first sorted in full by the workgroup (sort_keys != nullptr: lists up to SORT_MAX_LDS_KEYS entries in LDS, longer
// ones -- sort_beyond -- in global memory; in prefix mode, sort_prefix > 0, only the lists that were prefix-sorted need
// it) and then rendered again from it.  (Two launches until round 5 -- a sort kernel and this one: ~5 us each on the
// frames where no tile is flagged, which is nearly all of them.)
template <bool CK>
__global__ __launch_bounds__(RB) __attribute__((amdgpu_waves_per_eu(GS_FWD_CK_WAVES, 8))) void k_render_fwd_flagged(
    const float* __restrict__ packed, const float* __restrict__ rgb, const int* __restrict__ ranges,
    int* __restrict__ sorted, const float* __restrict__ bg, int W, int H, int ntx, int tile0,
    int nt, int* __restrict__ nsp_out, float* __restrict__ fw_out, float* __restrict__ image,
    int* __restrict__ tile_flags, int64_t cap, int* __restrict__ tile_cost, const SegState seg,
    const int* __restrict__ flag_counter, int* __restrict__ host_flagged, uint64_t* __restrict__ sort_keys,
    int sort_prefix, int sort_beyond, unsigned long long* __restrict__ touch_masks) {
    extern __shared__ uint64_t s_sort_keys[];
    static_assert(RB == SORT_BLOCK, "the repair workgroup sorts with the binning's network");
    // (depth cut) how many tiles of the frame had to be repaired, into the caller's pinned host word: what the
    // orchestration's policy looks at before later frames (never waited for)
    if (host_flagged != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *host_flagged = *flag_counter;
    if (flag_counter != nullptr && *flag_counter == 0) return;   // (depth cut: no tile of the frame is flagged)
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int tile = tile0 + t;
        if (tile_flags[tile] == 0) continue;
        if (sort_keys != nullptr) {
            const int s0 = ranges[tile];
            const int n = ranges[tile + 1] - s0;
            const bool wanted = sort_prefix > 0 ? prefix_sorted_tile(n, sort_prefix) : n > 1;
            if (wanted && (int64_t)s0 + n <= cap) {   // (block-uniform)
                if (n <= SORT_MAX_LDS_KEYS) {
                    for (int i = tid; i < n; i += SORT_BLOCK) s_sort_keys[slot(i)] = sort_keys[s0 + i];
                    __syncthreads();
                    lds_bitonic_sort(s_sort_keys, n, tid);
                    for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = (int)(uint32_t)s_sort_keys[slot(i)];
                } else if (sort_beyond) {
                    global_sort_tile(sort_keys + s0, sorted + s0, n, tid);
                }
            }
            __syncthreads();   // the list is in place for every thread of the workgroup (workgroup-scope fence + barrier)
        }
        render_tile_fwd<float, 1, CK>(tile, packed, rgb, nullptr, ranges, sorted, bg, W, H, ntx,
                                      nsp_out, fw_out, image, 0, tile_flags, true, cap, tile_cost, nullptr, nullptr,
                                      seg, nullptr, nullptr, touch_masks);
        __syncthreads();
    }
}
// the repair kernel's dynamic LDS (the sort's keys): the limit is raised once per device
template <bool CK> static void repair_attr_once() {
    static std::atomic<uint64_t> seen{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    const uint64_t bit = 1ull << dev;
    if ((seen.fetch_or(bit) & bit) == 0)
        (void)hipFuncSetAttribute((const void*)k_render_fwd_flagged<CK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sort_lds_bytes(SORT_MAX_LDS_KEYS));
}

// ---------------------------------------------------------------------------------------------------
// wave reductions
// ---------------------------------------------------------------------------------------------------
// Sum over the 64 lanes, result valid in lane 63.  Four in-row DPP shifts (Hillis-Steele inside
// each 16-lane row), then row_bcast:15 / row_bcast:31 carry the row sums across.
#define GS_DPP(x, ctrl, row_mask, bound)                                                           \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl),  \
                                                          (row_mask), 0xf, (bound)))
__device__ inline float wave_sum(float v) {
    v += GS_DPP(v, 0x111, 0xf, true);    // row_shr:1
    v += GS_DPP(v, 0x112, 0xf, true);    // row_shr:2
    v += GS_DPP(v, 0x114, 0xf, true);    // row_shr:4
    v += GS_DPP(v, 0x118, 0xf, true);    // row_shr:8  -> lane 15 of each row = row sum
    v += GS_DPP(v, 0x142, 0xa, false);   // row_bcast:15 into rows 1 and 3
    v += GS_DPP(v, 0x143, 0xc, false);   // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return v;
}
__device__ inline double wave_sum(double v);
// Sum within each 16-lane row only: lanes 15, 31, 47, 63 hold the four row sums.  The backward
// lets those four lanes issue the LDS atomic (the LDS pipe is otherwise idle there), which saves the
// two cross-row DPP steps per value.
__device__ inline float row_sum(float v) {
    v += GS_DPP(v, 0x111, 0xf, true);
    v += GS_DPP(v, 0x112, 0xf, true);
    v += GS_DPP(v, 0x114, 0xf, true);
#if GS_BWD_GROUP >= 16
    v += GS_DPP(v, 0x118, 0xf, true);
#endif
#if GS_BWD_GROUP >= 32
    v += GS_DPP(v, 0x142, 0xa, false);   // row_bcast:15 into rows 1 and 3: lanes 31, 63 hold half sums
#endif
    return v;
}
__device__ inline double row_sum(double v) { return wave_sum(v); }
template <typename T> __device__ inline bool row_leader(int lane) {
    return sizeof(T) == 4 ? (lane & (GS_BWD_GROUP - 1)) == (GS_BWD_GROUP - 1) : lane == 63;
}

__device__ inline double wave_sum(double v) {   // gradcheck-only path: plain shuffles
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}

template <typename T> __device__ inline void lds_add(T* p, T v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename T> __device__ inline void global_add(T* p, T v) { unsafeAtomicAdd(p, v); }


// v_permlane16_swap_b32 a, b: rows 1 and 3 (16-lane rows) of a are exchanged with rows 0 and 2 of b;
// v_permlane32_swap_b32 a, b: lanes 32..63 of a with lanes 0..31 of b (gfx950).  Inline assembly: this
// toolchain's builtin hands back its first result twice (scripts/ubench/permlane_test.hip checks the
// semantics on the device).  Must run with all 64 lanes enabled.
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// Sums nine per-lane values over the 64 lanes of the wave and stores the nine totals to slot[0..8]
// (LDS, owned by this wave for this splat: plain stores, no atomics -- an LDS float atomic costs
// ~3.5 cycles PER ACTIVE LANE on gfx950, which made the accumulators the bound of the kernel).
// A TRANSPOSING reduction: instead of summing all 9 values in every lane (54 cross-lane adds), lanes
// trade halves of the value set -- after the stride-4 step a lane keeps 4 of the first 8 values, after the
// stride-8 step 2 (see the function body) -- one v_permlane16_swap puts the even values' row pairs in
// rows 0/2 and the odd values' in rows 1/3, one v_permlane32_swap joins the halves: totals of the even
// values end in row 0, of the odd values in row 1, the ninth in lanes 48..63 (rows 1 + 3, joined by a row_bcast:15 add); nine lanes store with
// ONE ds_write_b32 (slot_lane_offset gives each its element, -1 elsewhere).
// Lanes that do not contribute must hold zeros.
__device__ __forceinline__ int slot_lane_offset(int lane) {
    // after the reduction: row 0 holds the totals of the even values, row 1 of the odd ones, bank k of a
    // row (lanes 4k..4k+3) the pair e(k) = (0, 4, 2, 6)[k]; lanes 48..63 hold the ninth value (lane 63 stores it)
    if (lane == 63) return 8;
    if (lane >= 32 || (lane & 3) != 0) return -1;
    const int bank = (lane >> 2) & 3;
    const int e = ((bank & 1) ? 4 : 0) + ((bank & 2) ? 2 : 0);
    return lane < 16 ? e : e + 1;
}
__device__ __forceinline__ void reduce9_to_slot(const float* val, bool stores, float* slots, int lane_slot) {
    // Rows first (16 lanes), hand-scheduled.  A DPP add with a BANK mask writes only the enabled quads
    // (bank = the four lanes i/4 of a row), so the trade "keep one half of the values, send the other"
    // needs no select when it is done ACROSS quads:
    //   stride 4   r_j = banks 0,2: v_j + v_j@(i+4)          banks 1,3: v_{j+4} + v_{j+4}@(i-4)     (j = 0..3)
    //   stride 8   s_j = banks 0,1: r_j + r_j@(i+8)          banks 2,3: r_{j+2} + r_{j+2}@(i-8)     (j = 0, 1)
    //   inside the quads the two survivors are summed plainly (xor 1, xor 2): every lane of bank k then
    //   holds the ROW totals of values e(k), e(k)+1 with e = (0, 4, 2, 6); the ninth value takes xor 1,
    //   xor 2, row_ror 12, row_ror 8 (row total in every lane).
    // 20 DPP adds and no v_cndmask (the compiler's form of the same trade: 12 selects + 17 DPP ops).
    // Every DPP source was written at least two instructions earlier (the DPP read-after-VALU-write hazard
    // needs two wait states; the s_nop covers the compiler's code in front of the block).
    float r0, r1, r2, r3, s2[2], s8;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %9, %9 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s2[0]), "=&v"(s2[1]), "=&v"(s8)
        : "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]),
          "v"(val[8]));
    // rows -> wave
    float x = s2[0], y = s2[1];
    permlane16_swap(x, y);           // x: rows (x0, y0, x2, y2); y: rows (x1, y1, x3, y3)
    float z = x + y;                 // rows 0, 2: even values of row pairs (0,1), (2,3); rows 1, 3: odd values
    // ninth value (its row total in every lane): rows 1 and 3 add lane 15 of the row in front of them (DPP row_bcast:15)
    // -- rows 0+1 and 2+3 as the two-register lane swap + add formed them, for one DPP add instead of a move, an
    // 8-cycle swap with its hazard padding and an add.  Lane 63 stores it: with the third step of the row reduction
    // rotating by 12 (quad q + quad q+1) lane 15 of every row holds ((Q3+Q0)+(Q1+Q2)), bit for bit the association
    // lane 0 held -- and lane 32 stored -- when that step rotated by 4
    float e = s8;
    asm volatile("v_add_f32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0" : "+v"(e) : "v"(s8));
    permlane32_swap(z, e);           // z: (z.lo, e.lo); e: (z.hi, e.hi)
    const float total = z + e;       // lanes 0..31: even / odd values' totals; lanes 48..63: the ninth (rows 0+1 + rows 2+3)
    // (lane_slot: the wave's slot of this splat + slot_lane_offset(lane), formed by the caller from a per-lane base that
    // holds everything but the splat -- one vector add per visit where (wave, splat, lane) -> address took three)
    if (stores) slots[lane_slot] = total;
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// At most 7 waves per SIMD (no minimum: the per-pixel-SH and fp64 instantiations sit far below).  Squeezed into the 64
// registers of 8 waves, k_render_bwd<float, 1> reloads a coefficient of the exponential in every visit; with the 72
// of 7 it stays resident (104 -> 103 vector instructions per visit, 0.4506 -> 0.4466 ms at D alternating on one box),
// and the kernel never held 8 waves anyway (5-7).
#ifdef GS_BWD_WAVES
#define GS_BWD_OCC __attribute__((amdgpu_waves_per_eu(GS_BWD_WAVES, GS_BWD_WAVES)))
#else
#define GS_BWD_OCC __attribute__((amdgpu_waves_per_eu(1, 7)))
#endif
template <typename T, int N_SH>
__global__ __launch_bounds__(RB) GS_BWD_OCC void k_render_bwd(
    const T* __restrict__ packed, const T* __restrict__ rgb, const T* __restrict__ view_dir,
    const int* __restrict__ ranges, const int* __restrict__ sorted, const T* __restrict__ bg,
    const int* __restrict__ nsp_in, const T* __restrict__ fw_in, const T* __restrict__ grad_image,
    int W, int H, int ntx, int tile0, int nt, T* __restrict__ g_rgb, T* __restrict__ g_opa,
    T* __restrict__ g_uv, T* __restrict__ g_conic, int slab, int exact, const int* __restrict__ tile_order,
    const T* __restrict__ src_opacity, const T* __restrict__ src_conic, const SegState seg,
    const int* __restrict__ cut_flags, const int* __restrict__ full_ranges, const int* __restrict__ overflow_sorted,
    const unsigned long long* __restrict__ touch_masks) {
    constexpr bool fast = sizeof(T) == 4;
    constexpr int CW = ColW<N_SH>::value;
    constexpr int C = 3 * N_SH;
    constexpr int NV = C + 6;   // rgb coeffs, opacity, u, v, conic x3
    constexpr int REF_CH = ref_chunk<T>(N_SH);
    // SLOTS: the fp32 kernels.  Every wave owns a slot of nine sums per staged splat and writes it once
    // (plain store after a full-wave reduction), the flush adds the slots of the waves that wrote -- no LDS
    // atomics, no zero fill.  The fp64 instantiations (gradcheck) keep one shared accumulator row per splat.
    // SHMM: fp32 with per-pixel SH (render_backward.cu:422-488).  The gradient of coefficient (ch, s) of a
    // splat is sum over pixels of Y_s(p) gi_ch(p) * aw(p): a contraction over the wave's 64 pixels whose left
    // factor does not depend on the splat -- a GEMM [16 x 64] x [64 x 16 splats] per channel, done with
    // v_mfma_f32_16x16x4_f32 (exact fp32) on batches of 16 contributing splats instead of 3 * N_SH cross-lane
    // reductions per visit.  The slots then carry only the six geometric sums.
    constexpr bool SLOTS = fast;
    constexpr bool SHMM = fast && N_SH > 1;
    constexpr int SV = 9;   // width of a slot: colour 3 (unused with SHMM) | w, w du, w dv | conic terms 3
    constexpr int RCHUNK = SLOTS ? GS_BWD_CHUNK : Chunk<T, N_SH>::value;
    constexpr int NWORD = RCHUNK / 64 > 0 ? RCHUNK / 64 : 1;
    constexpr int MB = 16;        // splats per MFMA batch (the N of 16x16x4)
    constexpr int BROW = 64 + 4;  // floats per row of the batch's aw matrix [slot][pixel] (16-byte aligned rows)
    __shared__ alignas(16) T s_geom[RCHUNK * GS_PACKED_WIDTH];
    __shared__ alignas(16) T s_col[N_SH > 1 ? RCHUNK * CW : 4];
    __shared__ int s_idx[RCHUNK];
    __shared__ T s_acc[SLOTS ? 4 * RCHUNK * SV : RCHUNK * NV];   // SLOTS: [wave][splat][9]
    __shared__ alignas(16) float s_B[SHMM ? 4 * MB * BROW : 4];  // SHMM: [wave][slot][pixel] aw of the open batch
    __shared__ int s_bidx[SHMM ? 4 * MB : 1];                     // SHMM: [wave][slot] Gaussian index of the splat
    __shared__ alignas(16) float s_gi[SHMM ? 4 * 3 * 64 : 4];    // SHMM: [wave][ch][pixel] grad_image
    __shared__ int s_max[4];
    __shared__ unsigned long long s_mask[4][NWORD];
    __shared__ unsigned long long s_hit[SLOTS ? 4 : 1][NWORD];   // SLOTS: slots written by each wave

    // depth segments (fused renderer, seg.rec != nullptr): work item = (tile, segment), block index =
    // segment * grid + block of the tile; segment 0 -- every pixel active, the most work -- starts first
    int seg_id = 0, blk = blockIdx.x;
    const bool seg_on = SLOTS && seg.rec != nullptr;
    if (seg_on) {
        const int g0 = render_grid(nt);
        seg_id = blockIdx.x / g0;
        blk = blockIdx.x - seg_id * g0;
    }
    const int t_local = tile_order ? tile_order[blk] : tile_of_block(blk, nt);
    if (t_local < 0 || t_local >= nt) return;
    const int tile = tile0 + t_local;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PixelMap px = pixel_of_thread(tile % ntx, tile / ntx, tid);
    const bool valid = px.u < W && px.v < H;
    int s0 = ranges[tile];
    int n_tile = ranges[tile + 1] - s0;
    // depth cut: a tile the forward had to redo from its complete list reads that list from the overflow buffer
    if (cut_flags != nullptr && cut_flags[tile] != 0) {
        s0 = full_ranges[tile];
        n_tile = full_ranges[tile + 1] - s0;
        sorted = overflow_sorted;
    }
    if (n_tile <= 0) return;

    int nsp = 0;
    T weight = 0;
    T gi[3] = {0, 0, 0};
    T Y[N_SH];
    int kend = 0;                 // segments: index after the pixel's last contributor, and what the backward's
    float oma_last = 1, bgw = 0;  // first step at that splat does (written by the forward's epilogue)
    {
        T d[3] = {0, 0, 0};
        if (valid) {
            const size_t p = (size_t)px.v * W + px.u;
            nsp = nsp_in[p];
            weight = fw_in[p];
            if constexpr (SLOTS) {
                if (seg_on) {
                    kend = seg.kend[p - seg.pix0];
                    oma_last = seg.oma_last[p - seg.pix0];
                    bgw = seg.bgw[p - seg.pix0];
                }
            }
            gi[0] = grad_image[p * 3 + 0];
            gi[1] = grad_image[p * 3 + 1];
            gi[2] = grad_image[p * 3 + 2];
            if constexpr (N_SH > 1) {
                d[0] = view_dir[p * 3 + 0]; d[1] = view_dir[p * 3 + 1]; d[2] = view_dir[p * 3 + 2];
            }
        }
        if constexpr (N_SH > 1) sh_basis<T, N_SH>(d, Y);
        else Y[0] = T(GS_SH_0);
    }
    // the tile's deepest used splat (render_backward.cu:131 makes everything beyond it a no-op)
    int m = nsp;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d));
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    const int n_used = min(n_tile, max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3])));
    if (n_used <= 0) return;

    const T pu = T(px.u), pv = T(px.v);
    const int slot_off = slot_lane_offset(lane);
    const bool slot_stores = slot_off >= 0;
    const int slot_lane_base = wave * RCHUNK * SV + (slot_stores ? slot_off : 0);
    T color_accum[3] = {0, 0, 0};
    bool bg_init = false;
    const T bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

    // SHMM: the batch GEMM, per channel  D[s][j] = sum_k A[s][k] B[k][j]  with
    //   A[s][k] = Y_s(pixel k)                  (constant over the walk: registers y_op)
    //   B[k][j] = aw of batch slot j at pixel k * grad_image_ch(pixel k)
    // The contraction index of MFMA step t, sub-index q (= lane >> 4) is pixel 16 q + t of the wave -- any bijection
    // serves a sum -- so that lane 16 q + j finds its sixteen B values (slot j, pixels 16 q .. 16 q + 15) and its
    // sixteen grad_image values contiguous in LDS:   y_op[t] = Y_{lane & 15}(pixel 16 q + t)   (rows s >= N_SH: 0)
    float y_op[SHMM ? 16 : 1];
    int nb = 0;   // SHMM: filled slots of the wave's open batch (wave-uniform)
    if constexpr (SHMM) {
        float* tmp = s_B + wave * MB * BROW;   // [pixel][17]: 64 * 17 == MB * BROW floats of this wave
        static_assert(64 * 17 <= MB * BROW && MB * 48 <= MB * BROW, "scratch uses of the wave's B rows");
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) tmp[lane * 17 + s2] = s2 < N_SH ? float(Y[s2 < N_SH ? s2 : 0]) : 0.0f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) s_gi[(wave * 3 + ch) * 64 + lane] = float(gi[ch]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 16; t++) y_op[t] = tmp[(16 * (lane >> 4) + t) * 17 + (lane & 15)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // SHMM: the GEMM of the open batch; its results are the wave's complete sums for (tile patch, splat): they go
    // straight to the global gradient rows (an LDS accumulator shared by the four waves needs 768 LDS float
    // atomics per batch, which cost more than the MFMAs: 0.15 of 0.69 ms at workload B)
    auto mma_flush = [&](int n_filled) {
        if constexpr (SHMM) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int j = lane & 15, q = lane >> 4;
            float* mine = s_B + wave * MB * BROW;
            const Vec4<float>* brow = reinterpret_cast<const Vec4<float>*>(mine + j * BROW + 16 * q);
            float b[16];
#pragma unroll
            for (int m4 = 0; m4 < 4; m4++) {
                const Vec4<float> v4 = brow[m4];
                b[4 * m4 + 0] = v4.x; b[4 * m4 + 1] = v4.y; b[4 * m4 + 2] = v4.z; b[4 * m4 + 3] = v4.w;
            }
            f32x4 acc[3];
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const Vec4<float>* grow = reinterpret_cast<const Vec4<float>*>(s_gi + (wave * 3 + ch) * 64 + 16 * q);
                float g[16];
#pragma unroll
                for (int m4 = 0; m4 < 4; m4++) {
                    const Vec4<float> v4 = grow[m4];
                    g[4 * m4 + 0] = v4.x; g[4 * m4 + 1] = v4.y; g[4 * m4 + 2] = v4.z; g[4 * m4 + 3] = v4.w;
                }
                f32x4 even = {0, 0, 0, 0}, odd = {0, 0, 0, 0};   // two chains: the dependent-issue latency is 40 cycles
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    even = __builtin_amdgcn_mfma_f32_16x16x4f32(y_op[t], b[t] * g[t], even, 0, 0, 0);
                    odd = __builtin_amdgcn_mfma_f32_16x16x4f32(y_op[t + 1], b[t + 1] * g[t + 1], odd, 0, 0, 0);
                }
                acc[ch] = even + odd;
            }
            // D: column j = lane & 15 (the batch slot), rows s = 4 q + r  ->  the wave's scratch as [slot][ch][16]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // every lane has read its B values
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                Vec4<float> v4;
                v4.x = acc[ch][0]; v4.y = acc[ch][1]; v4.z = acc[ch][2]; v4.w = acc[ch][3];
                *reinterpret_cast<Vec4<float>*>(mine + j * 48 + ch * 16 + 4 * q) = v4;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // consecutive lanes, consecutive words of a gradient row
            for (int k = lane; k < n_filled * 48; k += 64) {
                const int row = k / 48, c = k - row * 48;
                const int ch = c >> 4, s2 = c & 15;
                if (s2 < N_SH) global_add(g_rgb + (size_t)s_bidx[wave * MB + row] * C + ch * N_SH + s2, mine[k]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };

    GS_STAT_DECL;
    GS_HALF_DECL;
    GS_STAT(0, 1);          // waves
    GS_STAT(7, n_tile);
    GS_STAT(8, n_used);
    // the workgroup's part of the list: [seg_lo, seg_hi) (the whole used part without segments)
    int seg_lo = 0, seg_hi = n_used;
    int reach_end = nsp;   // the lane looks at list entries k < reach_end (render_backward.cu:131)
    if constexpr (SLOTS) {
        if (seg_on) {
            seg_lo = seg_id * SEG_LEN;
            if (seg_lo >= n_used) return;
            if (seg_id < SEG_MAX - 1) seg_hi = min(n_used, seg_lo + SEG_LEN);
            if (kend <= seg_lo) {
                reach_end = 0;   // the pixel's walk ended in front of this segment
            } else if (kend > seg_hi) {
                // the pixel's walk comes from behind: resume it with the state it has at the boundary
                const Vec4<float>* rec = seg.rec + (size_t)(tile - seg.tile0) * SEG_MAX * RB + tid;
                const int e_p = min((kend - 1) / SEG_LEN, SEG_MAX - 1);        // segment of its last contributor
                const int e_tile = min((n_used - 1) / SEG_LEN, SEG_MAX - 1);   // wave-uniform bound
                const int k_m = kend - 1;
                const bool first_divides = (exact ? k_m : k_m % REF_CH) < nsp - 1;   // Q1 at the first contributor
                // weight after the walk's first step, with the last contributor's own factor taken out again
                // (it is inside P of its segment)
                T wb = (first_divides ? weight * fast_rcp(oma_last) : weight) * oma_last;
                T d0 = 0, d1 = 0, d2 = 0;
                for (int s2 = e_tile; s2 > seg_id; s2--) {   // deepest first, as the walk accumulates
                    if (s2 <= e_p) {
                        const Vec4<float> r4 = rec[s2 * RB];
                        wb = wb * fast_rcp(r4.x);   // the walk's weight at the near boundary of segment s2
                        d0 += wb * r4.y; d1 += wb * r4.z; d2 += wb * r4.w;
                    }
                }
                weight = wb;
                color_accum[0] = bg0 * bgw + d0;
                color_accum[1] = bg1 * bgw + d1;
                color_accum[2] = bg2 * bgw + d2;
                bg_init = true;
            }
        }
    }
    const int first_chunk = seg_lo / RCHUNK;
    const int last_chunk = (seg_hi - 1) / RCHUNK;
    // Q1 (render_backward.cu:185 compares a chunk-local index): k % REF_CH for k = base + i, i < RCHUNK <= REF_CH, is
    // base % REF_CH + i with at most one wrap -- one scalar division per CHUNK instead of a multiply-high sequence
    // (9 scalar instructions) per visit; exact mode never wraps
    static_assert(RCHUNK <= REF_CH, "one wrap per chunk");
    const int q1_wrap = exact ? 0x7fffffff : REF_CH;
    for (int chunk = last_chunk; chunk >= first_chunk; chunk--) {
        const int base = chunk * RCHUNK;
        const int base_q = exact ? base : base % REF_CH;
        const int cnt = min(RCHUNK, seg_hi - base);
        GS_STAT(1, 1);
        __syncthreads();   // previous chunk fully flushed
        stage_chunk<T, N_SH, SLOTS ? 1 : 0>(packed, rgb, sorted, s0 + base, cnt, tid, s_geom, s_col, s_idx, src_opacity, src_conic);
        if constexpr (!SLOTS)
            for (int k = tid; k < cnt * NV; k += RB) s_acc[k] = 0;
        GS_PHASE(0);
        if constexpr (SLOTS && RCHUNK == 64) {
            if (touch_masks != nullptr && chunk < GS_MASK_WORDS) {
                // the forward's own masks of this word (GS_MASK_WORDS); entries beyond the used part are not staged here
                if (tid < 4) {
                    const unsigned long long keep = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
                    s_mask[tid][0] = touch_masks[((size_t)tile * GS_MASK_WORDS + chunk) * 4 + tid] & keep;
                }
            } else {
                // one patch per wave (round 6): the records wave 0 staged have to be visible to the other three first
                __syncthreads();
                build_touch_masks_by_patch<T>(s_geom, cnt, tid, tile % ntx, tile / ntx, s_mask);
            }
        } else {
            // (no barrier in between: thread t tests the record thread t staged)
            build_touch_masks<T, RCHUNK>(s_geom, cnt, tid, tile % ntx, tile / ntx, s_mask);
        }
        __syncthreads();
        GS_PHASE(1);

        for (int word = (cnt - 1) >> 6; word >= 0; word--) {
          unsigned long long m = s_mask[wave][word];
          m = wave_uniform(m);
          unsigned long long hit = 0;   // SLOTS: the splats of this word whose slot the wave wrote
          while (m) {
            const int bit = 63 - __builtin_clzll(m);
            m &= ~(1ull << bit);
            const int i = (word << 6) + bit;
            const int k = base + i;
            int kq = base_q + i;   // = exact ? k : k % REF_CH
            kq = kq >= q1_wrap ? kq - q1_wrap : kq;
            const bool reach = k < reach_end;   // render_backward.cu:131 (nsp == 0 outside the image)
            GS_STAT(2, 1);                          // visits
            if (ballot(reach) == 0) continue;      // wave-uniform: no lane reaches this splat
            GS_STAT(9, 1);                          // visits with a reaching lane
            GS_STAT(6, __popcll(ballot(reach)));
            GS_STAT_FLAG(st_in);
            if constexpr (SLOTS) {
                // ---- fp32, one colour coefficient per channel: the fused renderer's kernel -------------
                // Per lane: aw = alpha * weight (colour gradient = aw * Y0 grad_image[ch]),
                // w = norm_prob * grad_alpha (the opacity gradient term), and with t = k w,
                // k = -0.5 opacity / det a per-splat constant applied in the flush:
                //   grad_u = -2 (c du - b dv) t, grad_v = -2 (a dv - b du) t     (render_backward.cu:216-219)
                //   grad_conic = (dv^2 - c mh, b mh - du dv, du^2 - a mh) t      (:221-229, cf = mh / det)
                // The uv pair only needs the sums of w du and w dv (the flush forms the combinations);
                // the conic terms are formed per pixel as the reference does -- their cancellation is
                // benign there and would not be after the sum.  All nine are 0 for a lane that does not
                // contribute, so the reduction needs no zero-filled value array.
                const T* rec = s_geom + i * GS_PACKED_WIDTH;
                const Vec4<T> g0 = *reinterpret_cast<const Vec4<T>*>(rec);       // u v r2 opacity
                // the whole 48-byte record at once (one LDS round trip per visit instead of three dependent
                // ones; LDS bandwidth is no longer what bounds this kernel): 0.70 -> 0.69 ms
                const Vec4<T> g1 = *reinterpret_cast<const Vec4<T>*>(rec + 4);   // a b c det
                const Vec4<T> g2 = *reinterpret_cast<const Vec4<T>*>(rec + 8);   // 1/det, colour
                asm volatile("" ::"v"(g1.x), "v"(g1.y), "v"(g1.z), "v"(g1.w), "v"(g2.x), "v"(g2.y), "v"(g2.z), "v"(g2.w));
                const T du = pu - g0.x, dv = pv - g0.y;
                const T du2 = du * du, dv2 = dv * dv;
                // aw, w and mh are formed by the lanes that pass the alpha test only; the others are cut out of the
                // nine sums by `contrib` below -- no zero-filled registers in front of (and, for the ones the
                // exponential reuses, again inside) the branches: they were 11 of the visit's ~120 vector instructions
                T aw = unset<T>(), w = unset<T>(), mh = unset<T>(), duv = unset<T>();
                bool contrib = false;
                if (reach && !(du2 + dv2 > g0.z)) {   // inside the cutoff radius
                    GS_STAT_SET(st_in);
                    // render_backward.cu:153-165 (multiplies by 1/det; the forward divides)
                    duv = du * dv;
                    mh = (g1.z * du * du - g1.w * du * dv + g1.x * dv * dv) * g2.x;   // g1.w = b + b (stage_chunk)
                    // norm_prob = 0 unless mh > 0 (render_backward.cu:158-165) -- and then alpha = 0 fails the 1/255 test:
                    // `mh > 0` joins the test's lane mask (a scalar and), behind it norm_prob IS the exponential
                    const T norm_prob = exp_neg_half(mh);
                    T alpha = g0.w * norm_prob;
                    if (alpha > Thr<T>::sat_gt()) alpha = Thr<T>::alpha_cap();   // min(0.9999, .)
                    if ((mh > T(0)) & (alpha >= Thr<T>::alpha_min())) {
                        contrib = true;
                        if (!bg_init) {   // render_backward.cu:172-181
                            const T bw = background_weight<T>(alpha, weight);
                            if (bw > Thr<T>::bgw_gt()) {
                                color_accum[0] += bg0 * bw;
                                color_accum[1] += bg1 * bw;
                                color_accum[2] += bg2 * bw;
                            }
                            bg_init = true;
                        }
                        // values only from here on (no threshold depends on them): contraction allowed
                        {
#pragma clang fp contract(fast)
                            const T r1ma = fast_rcp(T(1) - alpha);
                            if (kq < nsp - 1) weight = weight * r1ma;   // Q1 (chunk-local index) unless exact
                            aw = alpha * weight;
                            // grad_alpha (render_backward.cu:196-203); the record's colour is Y0 * coefficient
                            T c0 = g2.y, c1 = g2.z, c2 = g2.w;
                            if constexpr (SHMM) {   // colour at this pixel's view direction
                                T col[3];
                                sh_to_rgb_contracted<T, N_SH>(s_col + i * CW, Y, col);
                                c0 = col[0]; c1 = col[1]; c2 = col[2];
                            }
                            const T ga = (c0 * weight - color_accum[0] * r1ma) * gi[0] +
                                         (c1 * weight - color_accum[1] * r1ma) * gi[1] +
                                         (c2 * weight - color_accum[2] * r1ma) * gi[2];
                            color_accum[0] += c0 * aw;
                            color_accum[1] += c1 * aw;
                            color_accum[2] += c2 * aw;
                            w = norm_prob * ga;
                        }
                    }
                }
                // a lane contributes iff it passed the alpha test; aw > 0 there (alpha >= 1/255, weight > 0)
                const T awz = contrib ? aw : T(0);
                const unsigned long long cmask = ballot(awz != T(0));
                GS_STAT(3, ballot(st_in) != 0);
                GS_STAT(4, cmask != 0);
                GS_STAT(5, __popcll(cmask));
                GS_HALF_VISIT(ballot(st_in));
                if (cmask == 0) continue;   // every reaching lane skipped the splat
                T val[9];
                const T wz = contrib ? w : T(0);
                const T awy = awz * Y[0];   // (the compiler re-formed Y0 * gi[ch] at every visit to save registers)
                val[0] = awy * gi[0]; val[1] = awy * gi[1]; val[2] = awy * gi[2];
                val[3] = wz; val[4] = wz * du; val[5] = wz * dv;
                // (mh and duv of a lane that stayed outside are whatever an earlier visit left: 0 * x = 0 for every x)
                {
#pragma clang fp contract(fast)
                    val[6] = mul_zero_wins((dv2 - g1.z * mh), wz);
                    val[7] = mul_zero_wins((g1.y * mh - duv), wz);
                    val[8] = mul_zero_wins((du2 - g1.x * mh), wz);
                }
                int slot_i = i * SV;   // (kept scalar: the compiler otherwise folds it into a 64-bit vector multiply-add)
                asm volatile("" : "+s"(slot_i));
                reduce9_to_slot(val, slot_stores, reinterpret_cast<float*>(s_acc), slot_lane_base + slot_i);
                hit |= 1ull << bit;
                if constexpr (SHMM) {
                    // column nb of the batch's B: this splat's aw at the wave's 64 pixels (0 where it does not contribute)
                    s_B[(wave * MB + nb) * BROW + lane] = awz;
                    if (lane == 0) s_bidx[wave * MB + nb] = s_idx[i];
                    if (++nb == MB) {
                        mma_flush(MB);
                        nb = 0;
                    }
                }
                continue;
            }
            T val[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) val[j] = 0;
            bool contrib = false;
            if (reach) {
                const T* rec = s_geom + i * GS_PACKED_WIDTH;
                const Vec4<T> g0 = *reinterpret_cast<const Vec4<T>*>(rec);       // u v r2 opacity
                const Vec4<T> g1 = *reinterpret_cast<const Vec4<T>*>(rec + 4);   // a b c det
                const T du = pu - g0.x, dv = pv - g0.y;
                const T a = g1.x, b = g1.y, c = g1.z, rdet = rec[8], opa = g0.w;
                T norm_prob = 0, alpha = 0, mh = 0;
                if (!(fast && du * du + dv * dv > g0.z)) {   // inside the cutoff radius
                    GS_STAT_SET(st_in);
                    // render_backward.cu:153-165 (multiplies by 1/det; forward divides)
                    mh = (c * du * du - (b + b) * du * dv + a * dv * dv) * rdet;
                    if (mh > T(0)) norm_prob = gexp<T>(T(-0.5) * mh);
                    alpha = opa * norm_prob;
                    if (alpha > Thr<T>::sat_gt()) alpha = Thr<T>::alpha_cap();   // min(0.9999, .)
                }
                if (!fast || alpha >= Thr<T>::alpha_min()) {
                    contrib = true;
                    if (!bg_init) {   // render_backward.cu:172-181
                        const T bw = background_weight<T>(alpha, weight);
                        if (bw > Thr<T>::bgw_gt()) {
                            color_accum[0] += bg0 * bw;
                            color_accum[1] += bg1 * bw;
                            color_accum[2] += bg2 * bw;
                        }
                        bg_init = true;
                    }
                    // From here on only gradient VALUES are formed (no threshold depends on
                    // them): evaluated in T with the reference's formulas factored
                    // (render_backward.cu:183-234) and contraction allowed; checked at 1e-4.
                    {
#pragma clang fp contract(fast)
                        const T r1ma = fast_rcp(T(1) - alpha);
                        if (kq < nsp - 1) weight = weight * r1ma;   // Q1 (chunk-local index) unless exact
                        T col[3];
                        splat_colour<T, N_SH>(s_geom, s_col, i, Y, col);
                        const T aw = alpha * weight;
                        T grad_alpha = 0;
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            const T grl = aw * gi[ch];
#pragma unroll
                            for (int s = 0; s < N_SH; s++) val[N_SH * ch + s] = Y[s] * grl;
                            grad_alpha += (col[ch] * weight - color_accum[ch] * r1ma) * gi[ch];
                            color_accum[ch] += col[ch] * aw;
                        }
                        val[C + 0] = norm_prob * grad_alpha;
                        // d alpha / d mh^2 = -alpha_unclamped / 2; t = rdet * grad_mh
                        const T t = T(-0.5) * norm_prob * opa * grad_alpha * rdet;
                        const T A = c * du - b * dv;
                        const T B = a * dv - b * du;
                        val[C + 1] = T(-2) * A * t;           // :216-217
                        val[C + 2] = T(-2) * B * t;           // :218-219
                        val[C + 3] = (dv * dv - c * mh) * t;  // :221-229 with cf = mh^2 * rdet
                        val[C + 4] = (b * mh - du * dv) * t;
                        val[C + 5] = (du * du - a * mh) * t;
                    }
                }
            }
            const unsigned long long cmask = __ballot(contrib);
            GS_STAT(3, __ballot(st_in) != 0);
            GS_STAT(4, cmask != 0);
            GS_STAT(5, __popcll(cmask));
            if (cmask == 0) continue;   // every reaching lane skipped the splat
            GS_STAT(10, __popcll(cmask) <= 4);
            if (fast && __popcll(cmask) <= 4) {
                // few contributors: they add their own values (<= 4 lanes per LDS atomic, the same
                // conflict degree as the row-leader form) and the DPP reduction is skipped
                if (contrib) {
#pragma unroll
                    for (int j = 0; j < NV; j++) lds_add(&s_acc[i * NV + j], val[j]);
                }
                continue;
            }
            {
#pragma unroll
                for (int j = 0; j < NV; j++) val[j] = row_sum(val[j]);
                if (row_leader<T>(lane)) {
#pragma unroll
                    for (int j = 0; j < NV; j++) lds_add(&s_acc[i * NV + j], val[j]);
                }
            }
          }
          if constexpr (SLOTS)
              if (lane == 0) s_hit[wave][word] = hit;
        }
        if constexpr (SHMM) {
            if (nb > 0) {
                mma_flush(nb);
                nb = 0;
            }
        }
        GS_PHASE(2);
        GS_HALF_CHUNK(12);
        __syncthreads();
        GS_PHASE(3);
        // one global atomic per value per (splat, tile)
        if constexpr (SLOTS) {
            // Phase 1, thread = splat: add the slots of the waves that wrote this splat, in wave order
            // (deterministic per tile), apply the per-splat factors, and park the row in wave 0's slot of the
            // same splat (each thread only ever touches its own splat's slots: no barrier needed before).
            if (tid < cnt) {
                T a[SV];
#pragma unroll
                for (int j = 0; j < SV; j++) a[j] = 0;
#pragma unroll
                for (int w4 = 0; w4 < 4; w4++) {
                    if ((s_hit[w4][tid >> 6] >> (tid & 63)) & 1ull) {
                        const T* sl = s_acc + (w4 * RCHUNK + tid) * SV;
#pragma unroll
                        for (int j = 0; j < SV; j++) a[j] += sl[j];
                    }
                }
                // sums of (w, w du, w dv) and of the per-pixel conic terms -> gradients (see the loop)
                const T* rec = s_geom + tid * GS_PACKED_WIDTH;
                const T ca = rec[4], cb = rec[5], cc = rec[6];
                const T k = T(-0.5) * rec[3] * rec[8];
                const T Mu = a[4], Mv = a[5];
                a[4] = T(-2) * k * (cc * Mu - cb * Mv);
                a[5] = T(-2) * k * (ca * Mv - cb * Mu);
                a[6] *= k;
                a[7] *= k;
                a[8] *= k;
                T* row = s_acc + tid * SV;
#pragma unroll
                for (int j = 0; j < SV; j++) row[j] = a[j];
            }
            __syncthreads();
            // Phase 2, nine lanes = one row: a wave's atomic instruction then covers seven whole 36-byte
            // rows with consecutive addresses, which the L2 handles per cache line -- 6x the rate of one
            // thread per row with nine instructions (scripts/ubench/atomic_rows.hip: 0.094 vs 0.557 ms for
            // the 1.25 M rows of a frame; the per-row form had become 0.17 ms of this kernel).
            const int sub = lane / SV, col = lane - sub * SV;   // lane 63: idle
            for (int r0 = wave * 7; r0 < cnt; r0 += 28) {
                const int r = r0 + sub;
                const bool in = lane < 63 && r < cnt && !(SHMM && col < 3);   // SHMM: the colour sums went out with the batches
                const T v = in ? s_acc[r * SV + col] : T(0);
                const unsigned long long nz = ballot(v != T(0));
                const bool any = in && ((nz >> (sub * SV)) & 0x1ffull) != 0;   // rows of zeros stay untouched
                GS_STAT(11, __popcll(ballot(any && col == 0)));   // flushed rows
                if (any) {
                    const int g = s_idx[r];
                    T* dst;
                    if (slab) dst = g_rgb + (size_t)g * SV + col;   // [V, 9]: rgb 3 | opacity 1 | uv 2 | conic 3
                    else if (col < 3) dst = g_rgb + (size_t)g * 3 + col;
                    else if (col == 3) dst = g_opa + g;
                    else if (col < 6) dst = g_uv + (size_t)g * 2 + (col - 4);
                    else dst = g_conic + (size_t)g * 3 + (col - 6);
                    global_add(dst, v);
                }
            }
        } else if (tid < cnt) {
            const int g = s_idx[tid];
            const T* a = s_acc + tid * NV;
            bool any = false;
#pragma unroll
            for (int j = 0; j < NV; j++) any |= (a[j] != T(0));
            GS_STAT(11, __popcll(__ballot(any)));   // flushed rows
            if (any) {
#pragma unroll
                for (int j = 0; j < C; j++) global_add(g_rgb + (size_t)g * C + j, a[j]);
                global_add(g_opa + g, a[C + 0]);
                global_add(g_uv + (size_t)g * 2 + 0, a[C + 1]);
                global_add(g_uv + (size_t)g * 2 + 1, a[C + 2]);
                global_add(g_conic + (size_t)g * 3 + 0, a[C + 3]);
                global_add(g_conic + (size_t)g * 3 + 1, a[C + 4]);
                global_add(g_conic + (size_t)g * 3 + 2, a[C + 5]);
            }
        }
        GS_PHASE(4);
    }
    GS_STAT_FLUSH(16);
}

// ---------------------------------------------------------------------------------------------------
// depth (depth.cu:7-115), fp32 only
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RB) void k_render_depth(const float* __restrict__ packed,
                                                     const float* __restrict__ xyz_cam,
                                                     const int* __restrict__ ranges,
                                                     const int* __restrict__ sorted, int W, int H,
                                                     int ntx, int nt, float alpha_threshold,
                                                     float* __restrict__ depth) {
#ifndef GS_DEPTH_CHUNK
#define GS_DEPTH_CHUNK 256
#endif
    constexpr int RCHUNK = GS_DEPTH_CHUNK;
    __shared__ alignas(16) float s_geom[RCHUNK * GS_PACKED_WIDTH];
    __shared__ int s_idx[RCHUNK];
    const int tile = tile_of_block(blockIdx.x, nt);
    if (tile >= nt) return;
    const int tid = threadIdx.x;
    const PixelMap px = pixel_of_thread(tile % ntx, tile / ntx, tid);
    const bool valid = px.u < W && px.v < H;
    const int s0 = ranges[tile];
    const int n_tile = ranges[tile + 1] - s0;
    float acc = 0.0f;
    bool done = !valid;
    const float pu = (float)px.u, pv = (float)px.v;
    for (int base = 0; base < n_tile; base += RCHUNK) {
        const int cnt = min(RCHUNK, n_tile - base);
        if (tid < cnt) {
            const int g = sorted[s0 + base + tid];
            const Vec4<float>* src =
                reinterpret_cast<const Vec4<float>*>(packed + (size_t)g * GS_PACKED_WIDTH);
            Vec4<float>* dst = reinterpret_cast<Vec4<float>*>(s_geom + tid * GS_PACKED_WIDTH);
            dst[0] = src[0];
            dst[1] = src[1];
            s_idx[tid] = g;
        }
        __syncthreads();
        for (int i = 0; i < cnt; i++) {
            if (__ballot(!done) == 0) break;
            if (!done) {
                const Vec4<float> g0 =
                    *reinterpret_cast<const Vec4<float>*>(s_geom + i * GS_PACKED_WIDTH);       // u v r2 opa
                const Vec4<float> g1 =
                    *reinterpret_cast<const Vec4<float>*>(s_geom + i * GS_PACKED_WIDTH + 4);   // a b c det
                const float du = pu - g0.x, dv = pv - g0.y;
                const float a = g1.x, b = g1.y, c = g1.z, det = g1.w, opa = g0.w;
                const float mh = (c * du * du - (b + b) * du * dv + a * dv * dv) / det;
                float alpha = 0.0f;
                if (mh > 0.0f) alpha = opa * det_expf(-0.5f * mh);
                const float weight = alpha * (1.0 - acc);
                acc += weight;
                if (acc > alpha_threshold) {
                    const int g = s_idx[i];
                    const float x = xyz_cam[g * 3 + 0], y = xyz_cam[g * 3 + 1],
                                z = xyz_cam[g * 3 + 2];
                    depth[(size_t)px.v * W + px.u] = __builtin_sqrtf(x * x + y * y + z * z);
                    done = true;
                }
            }
        }
        if (__syncthreads_and(done)) break;
    }
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

static int check_rows(int H, int row0, int row1) {
    const int nty = (H + 15) / 16;
    if (row0 < 0 || row1 > nty || row0 > row1) {
        gs::set_error("bad tile row range [%d, %d) for %d tile rows", row0, row1, nty);
        return GS_EINVAL;
    }
    return GS_OK;
}

// process-wide DEFAULT of the render backward's gradient mode; every entry point takes the mode per call
// (ABI 5) and reads this only when asked for GS_BACKWARD_DEFAULT, once, at the call
static std::atomic<int> g_backward_mode{GS_BACKWARD_COMPAT};
static int resolve_backward_mode(int mode, int* exact) {
    if (mode == GS_BACKWARD_DEFAULT) mode = g_backward_mode.load(std::memory_order_relaxed);
    if (mode != GS_BACKWARD_COMPAT && mode != GS_BACKWARD_EXACT) {
        gs::set_error("backward mode must be GS_BACKWARD_DEFAULT, GS_BACKWARD_COMPAT or GS_BACKWARD_EXACT");
        return GS_EINVAL;
    }
    *exact = mode == GS_BACKWARD_EXACT;
    return GS_OK;
}

extern "C" {

int gs_set_backward_mode(int mode) {
    GS_REQUIRE(mode == GS_BACKWARD_COMPAT || mode == GS_BACKWARD_EXACT, "backward mode must be GS_BACKWARD_COMPAT or GS_BACKWARD_EXACT");
    g_backward_mode.store(mode, std::memory_order_relaxed);
    return GS_OK;
}
int gs_get_backward_mode(void) { return g_backward_mode.load(std::memory_order_relaxed); }

#ifdef GS_STATS
// instrumented build only: copies the 32 counters to the host (and clears them when reset != 0)
int gs_debug_render_stats(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return GS_EHIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_render_stats), 32 * sizeof(unsigned long long)) != hipSuccess)
        return GS_EHIP;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gs::g_render_stats), z, sizeof(z)) != hipSuccess) return GS_EHIP;
    }
    return GS_OK;
}
// instrumented build only: copies the wave timeline (2 kernels x GS_TIMELINE_CAP slots x GS_TIMELINE_W u64; unused
// slots are zero) to the host and clears it
int gs_debug_render_timeline(unsigned long long* out, int cap_waves) {
    if (hipDeviceSynchronize() != hipSuccess) return GS_EHIP;
    if (cap_waves != gs::GS_TIMELINE_CAP) return GS_EINVAL;
    const size_t bytes = (size_t)2 * gs::GS_TIMELINE_CAP * gs::GS_TIMELINE_W * sizeof(unsigned long long);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gs::g_timeline), bytes) != hipSuccess) return GS_EHIP;
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(gs::g_timeline)) != hipSuccess) return GS_EHIP;
    if (hipMemset(p, 0, bytes) != hipSuccess) return GS_EHIP;
    return GS_OK;
}
#endif

static int launch_render_fwd(const void* packed_or_uvs, const void* opacity, const void* conic, const void* rgb,
                             const void* view_dir_by_pixel, const int32_t* tile_ranges,
                             const int32_t* sorted_gaussians, const void* background_rgb, int W, int H, int n_sh,
                             int tile_row0, int tile_row1, int32_t* num_splats_per_pixel,
                             void* final_weight_per_pixel, void* image, int dtype, void* stream,
                             void* segment_state = nullptr) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16;
    const int nt = (tile_row1 - tile_row0) * ntx;
    if (nt == 0) return GS_OK;
    const int grid = render_grid(nt);
    if (segment_state != nullptr) {   // the fused renderer's kernel with the state for the segmented backward
        GS_REQUIRE(dtype == GS_F32 && n_sh == 1 && opacity == nullptr,
                   "segment_state needs float32 packed records and one colour coefficient per channel");
        GS_REQUIRE(((uintptr_t)segment_state & 15) == 0, "segment_state must be 16-byte aligned");
        k_render_fwd_ck<<<grid, RB, 0, s>>>(
            (const float*)packed_or_uvs, (const float*)rgb, tile_ranges, sorted_gaussians,
            (const float*)background_rgb, W, H, ntx, tile_row0 * ntx, nt, num_splats_per_pixel,
            (float*)final_weight_per_pixel, (float*)image, 0, nullptr, INT64_MAX, nullptr,
            seg_state_of(segment_state, W, H, tile_row0, tile_row1), nullptr);
        return check_launch("render_tiles");
    }
    DISPATCH_T(dtype, DISPATCH_SH(n_sh, (k_render_fwd<T, N_SH><<<grid, RB, 0, s>>>(
                                            (const T*)packed_or_uvs, (const T*)rgb,
                                            (const T*)view_dir_by_pixel, tile_ranges,
                                            sorted_gaussians, (const T*)background_rgb, W, H, ntx,
                                            tile_row0 * ntx, nt, num_splats_per_pixel,
                                            (T*)final_weight_per_pixel, (T*)image, 0,
                                            nullptr, INT64_MAX, nullptr, (const T*)opacity, (const T*)conic, nullptr,
                                            nullptr, nullptr))));
    return check_launch("render_tiles");
}

int gs_render_tiles(const void* uvs, const void* opacity, const void* rgb, const void* conic,
                    const void* view_dir_by_pixel, const int32_t* tile_ranges,
                    const int32_t* sorted_gaussians, const void* background_rgb,
                    int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image, int W, int H,
                    int n_sh, int tile_row0, int tile_row1, int dtype, void* stream) {
    GS_REQUIRE(opacity != nullptr && conic != nullptr, "opacity and conic must not be null");
    return launch_render_fwd(uvs, opacity, conic, rgb, view_dir_by_pixel, tile_ranges, sorted_gaussians,
                             background_rgb, W, H, n_sh, tile_row0, tile_row1, num_splats_per_pixel,
                             final_weight_per_pixel, image, dtype, stream);
}

int gs_render_tiles_packed(const void* packed, const void* rgb, const void* view_dir_by_pixel,
                           const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                           const void* background_rgb, int W, int H, int n_sh, int tile_row0,
                           int tile_row1, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                           void* image, int dtype, void* segment_state, void* stream) {
    return launch_render_fwd(packed, nullptr, nullptr, rgb, view_dir_by_pixel, tile_ranges, sorted_gaussians,
                             background_rgb, W, H, n_sh, tile_row0, tile_row1, num_splats_per_pixel,
                             final_weight_per_pixel, image, dtype, stream, segment_state);
}

size_t gs_render_segment_workspace_bytes(int W, int H, int tile_row0, int tile_row1) {
    if (W <= 0 || H <= 0 || tile_row0 < 0 || tile_row1 <= tile_row0 || tile_row1 > (H + 15) / 16) return 0;
    const size_t Tb = (size_t)((W + 15) / 16) * (size_t)(tile_row1 - tile_row0);
    return Tb * SEG_MAX * 256 * sizeof(Vec4<float>) + seg_band_pixels(W, H, tile_row0, tile_row1) * 12;
}

int gs_render_tiles_prefix(const void* packed, const void* rgb, const int32_t* tile_ranges,
                           int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                           const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                           int32_t* tile_flags, int32_t* num_splats_per_pixel,
                           void* final_weight_per_pixel, void* image, int32_t* tile_cost, void* segment_state,
                           void* stream) {
    return gs_render_tiles_prefix_phased(packed, rgb, tile_ranges, sorted_gaussians, keys, S, background_rgb, W, H, tile_row0,
                                         tile_row1, tile_flags, num_splats_per_pixel, final_weight_per_pixel, image,
                                         tile_cost, segment_state, GS_PREFIX_RENDER | GS_PREFIX_REPAIR, stream);
}

int gs_render_tiles_prefix_phased_m(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                                  const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                                  int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                                  void* image, int32_t* tile_cost, void* segment_state, int phases, uint64_t* touch_masks, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    GS_REQUIRE(phases >= 1 && phases <= 3, "phases: GS_PREFIX_RENDER, GS_PREFIX_REPAIR or both");
    GS_REQUIRE(tile_flags != nullptr, "tile_flags must not be null");
    GS_REQUIRE(segment_state == nullptr || ((uintptr_t)segment_state & 15) == 0, "segment_state must be 16-byte aligned");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16;
    const int nt = (tile_row1 - tile_row0) * ntx;
    if (nt == 0) return GS_OK;
    const int grid = render_grid(nt);
    const int t0 = tile_row0 * ntx;
    const SegState seg = seg_state_of(segment_state, W, H, tile_row0, tile_row1);
    // 1. provisional pass over the ordered prefixes; raises the flags
    if (phases & GS_PREFIX_RENDER) {
        if (segment_state)
            k_render_fwd_ck<<<grid, RB, 0, s>>>(
                (const float*)packed, (const float*)rgb, tile_ranges, sorted_gaussians, (const float*)background_rgb, W, H,
                ntx, t0, nt, num_splats_per_pixel, (float*)final_weight_per_pixel, (float*)image, GS_SORT_PREFIX,
                tile_flags, S, tile_cost, seg, (unsigned long long*)touch_masks);
        else
            k_render_fwd<float, 1><<<grid, RB, GS_FWD_LDS_PAD, s>>>(
                (const float*)packed, (const float*)rgb, nullptr, tile_ranges, sorted_gaussians,
                (const float*)background_rgb, W, H, ntx, t0, nt, num_splats_per_pixel,
                (float*)final_weight_per_pixel, (float*)image, GS_SORT_PREFIX, tile_flags, S, tile_cost, nullptr, nullptr,
                nullptr, nullptr, (unsigned long long*)touch_masks);
    }
    if ((phases & GS_PREFIX_REPAIR) && S > GS_SORT_PREFIX) {
        // 2. flagged tiles: full sort + render again, one launch (exits at once on a dense scene)
        const size_t lds = sort_lds_bytes(S <= 4096 ? 4096 : SORT_MAX_LDS_KEYS);
        if (segment_state) {
            repair_attr_once<true>();
            k_render_fwd_flagged<true><<<nt < 512 ? nt : 512, RB, lds, s>>>(
                (const float*)packed, (const float*)rgb, tile_ranges, sorted_gaussians, (const float*)background_rgb, W, H, ntx,
                t0, nt, num_splats_per_pixel, (float*)final_weight_per_pixel, (float*)image, tile_flags, S, tile_cost, seg,
                nullptr, nullptr, const_cast<uint64_t*>(keys), GS_SORT_PREFIX, 0, (unsigned long long*)touch_masks);
        } else {
            repair_attr_once<false>();
            k_render_fwd_flagged<false><<<nt < 512 ? nt : 512, RB, lds, s>>>(
                (const float*)packed, (const float*)rgb, tile_ranges, sorted_gaussians, (const float*)background_rgb, W, H, ntx,
                t0, nt, num_splats_per_pixel, (float*)final_weight_per_pixel, (float*)image, tile_flags, S, tile_cost, seg,
                nullptr, nullptr, const_cast<uint64_t*>(keys), GS_SORT_PREFIX, 0, (unsigned long long*)touch_masks);
        }
    }
    return check_launch("render_tiles_prefix");
}

int gs_render_tiles_prefix_phased(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  int32_t* sorted_gaussians, const uint64_t* keys, int64_t S,
                                  const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                                  int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel,
                                  void* image, int32_t* tile_cost, void* segment_state, int phases, void* stream) {
    return gs_render_tiles_prefix_phased_m(packed, rgb, tile_ranges, sorted_gaussians, keys, S, background_rgb, W, H, tile_row0,
                                           tile_row1, tile_flags, num_splats_per_pixel, final_weight_per_pixel, image, tile_cost,
                                           segment_state, phases, nullptr, stream);
}

// the fused renderer's forward on depth-cut lists (binning.hip "depth cut"; gs_tile_count_cut / gs_tile_emit_sort_cut)
int gs_render_tiles_cut_m(const void* packed, const void* rgb, const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                        int64_t S, const int32_t* full_ranges, const void* bin_records, int N, float mh_dist,
                        int32_t* workspace, int32_t* cut_workspace, uint64_t* overflow_keys, int32_t* overflow_sorted,
                        int64_t overflow_capacity, const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                        int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image,
                        int32_t* tile_cost, int32_t* host_flagged, uint64_t* touch_masks, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    GS_REQUIRE(tile_flags != nullptr && full_ranges != nullptr, "tile_flags and full_ranges must not be null");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
    const int nt = (tile_row1 - tile_row0) * ntx;
    if (nt == 0) return GS_OK;
    const int grid = render_grid(nt);
    const int t0 = tile_row0 * ntx;
    int* flag_counter = depth_cut_flag_counter(cut_workspace, N, ntx * nty);
    // 1. every tile from its kept (completely sorted) depth prefix; raises the flags
    k_render_fwd<float, 1><<<grid, RB, GS_FWD_LDS_PAD, s>>>(
        (const float*)packed, (const float*)rgb, nullptr, tile_ranges, sorted_gaussians, (const float*)background_rgb, W, H,
        ntx, t0, nt, num_splats_per_pixel, (float*)final_weight_per_pixel, (float*)image, 0, tile_flags, S, tile_cost,
        nullptr, nullptr, full_ranges, flag_counter, (unsigned long long*)touch_masks);
    // 2. + 3. flagged tiles: complete lists into the overflow buffers; sorted and rendered again by one workgroup each
    // (both launches exit at once while the frame has no flagged tile)
    if (int e = depth_cut_repair((const float*)bin_records, N, ntx, nty, mh_dist, tile_row0, tile_row1, full_ranges, workspace,
                                 cut_workspace, overflow_keys, overflow_capacity, overflow_sorted, tile_flags, s, false))
        return e;
    repair_attr_once<false>();
    k_render_fwd_flagged<false><<<nt < 512 ? nt : 512, RB, sort_lds_bytes(SORT_MAX_LDS_KEYS), s>>>(
        (const float*)packed, (const float*)rgb, full_ranges, overflow_sorted, (const float*)background_rgb, W, H, ntx, t0,
        nt, num_splats_per_pixel, (float*)final_weight_per_pixel, (float*)image, tile_flags, overflow_capacity, tile_cost,
        SEG_NONE, flag_counter, host_flagged, overflow_keys, 0, 1, (unsigned long long*)touch_masks);
    return check_launch("render_tiles_cut");
}

int gs_render_tiles_cut(const void* packed, const void* rgb, const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                        int64_t S, const int32_t* full_ranges, const void* bin_records, int N, float mh_dist,
                        int32_t* workspace, int32_t* cut_workspace, uint64_t* overflow_keys, int32_t* overflow_sorted,
                        int64_t overflow_capacity, const void* background_rgb, int W, int H, int tile_row0, int tile_row1,
                        int32_t* tile_flags, int32_t* num_splats_per_pixel, void* final_weight_per_pixel, void* image,
                        int32_t* tile_cost, int32_t* host_flagged, void* stream) {
    return gs_render_tiles_cut_m(packed, rgb, tile_ranges, sorted_gaussians, S, full_ranges, bin_records, N, mh_dist, workspace,
                                 cut_workspace, overflow_keys, overflow_sorted, overflow_capacity, background_rgb, W, H, tile_row0,
                                 tile_row1, tile_flags, num_splats_per_pixel, final_weight_per_pixel, image, tile_cost,
                                 host_flagged, nullptr, stream);
}

static int launch_render_bwd(const void* packed_or_uvs, const void* opacity, const void* conic, const void* rgb,
                             const void* view_dir_by_pixel, const int32_t* tile_ranges,
                             const int32_t* sorted_gaussians, const void* background_rgb,
                             const int32_t* num_splats_per_pixel, const void* final_weight_per_pixel,
                             const void* grad_image, int W, int H, int n_sh, int tile_row0, int tile_row1,
                             void* grad_rgb, void* grad_opacity, void* grad_uv, void* grad_conic, int dtype,
                             int backward_mode, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    int exact = 0;
    if (int e = resolve_backward_mode(backward_mode, &exact)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16;
    const int nt = (tile_row1 - tile_row0) * ntx;
    if (nt == 0) return GS_OK;
    const int grid = render_grid(nt);
    DISPATCH_T(dtype,
               DISPATCH_SH(n_sh, (k_render_bwd<T, N_SH><<<grid, RB, 0, s>>>(
                                     (const T*)packed_or_uvs, (const T*)rgb, (const T*)view_dir_by_pixel,
                                     tile_ranges, sorted_gaussians, (const T*)background_rgb,
                                     num_splats_per_pixel, (const T*)final_weight_per_pixel,
                                     (const T*)grad_image, W, H, ntx, tile_row0 * ntx, nt,
                                     (T*)grad_rgb, (T*)grad_opacity, (T*)grad_uv,
                                     (T*)grad_conic, 0, exact, nullptr, (const T*)opacity,
                                     (const T*)conic, SEG_NONE, nullptr, nullptr, nullptr, nullptr))));
    return check_launch("render_tiles_backward");
}

int gs_render_tiles_backward(const void* uvs, const void* opacity, const void* rgb, const void* conic,
                             const void* view_dir_by_pixel, const int32_t* tile_ranges,
                             const int32_t* sorted_gaussians, const void* background_rgb,
                             const int32_t* num_splats_per_pixel, const void* final_weight_per_pixel,
                             const void* grad_image, void* grad_rgb, void* grad_opacity, void* grad_uv,
                             void* grad_conic, int W, int H, int n_sh, int tile_row0, int tile_row1, int dtype,
                             int backward_mode, void* stream) {
    GS_REQUIRE(opacity != nullptr && conic != nullptr, "opacity and conic must not be null");
    return launch_render_bwd(uvs, opacity, conic, rgb, view_dir_by_pixel, tile_ranges, sorted_gaussians,
                             background_rgb, num_splats_per_pixel, final_weight_per_pixel, grad_image, W, H, n_sh,
                             tile_row0, tile_row1, grad_rgb, grad_opacity, grad_uv, grad_conic, dtype, backward_mode,
                             stream);
}

int gs_render_tiles_backward_packed(const void* packed, const void* rgb, const void* view_dir_by_pixel,
                                    const int32_t* tile_ranges, const int32_t* sorted_gaussians,
                                    const void* background_rgb, const int32_t* num_splats_per_pixel,
                                    const void* final_weight_per_pixel, const void* grad_image, int W,
                                    int H, int n_sh, int tile_row0, int tile_row1, void* grad_rgb,
                                    void* grad_opacity, void* grad_uv, void* grad_conic, int dtype,
                                    int backward_mode, void* stream) {
    return launch_render_bwd(packed, nullptr, nullptr, rgb, view_dir_by_pixel, tile_ranges, sorted_gaussians,
                             background_rgb, num_splats_per_pixel, final_weight_per_pixel, grad_image, W, H, n_sh,
                             tile_row0, tile_row1, grad_rgb, grad_opacity, grad_uv, grad_conic, dtype, backward_mode,
                             stream);
}

static int launch_bwd_prologue(void* grad_slab, int64_t zero_slab_rows, const int32_t* tile_cost, int32_t* tile_order,
                               int ntx, int tile_row0, int nt, hipStream_t s) {
    const int grid = render_grid(nt);
    const bool ordered = tile_cost != nullptr && nt >= GS_LPT_MIN_TILES;
    const size_t n_floats = (size_t)zero_slab_rows * 9;
    if (tile_order == nullptr && n_floats == 0) return GS_OK;
    const size_t want = (n_floats / 4 + 1023) / 1024;
    const int fill_blocks =
        n_floats == 0 ? 0 : (int)(want < (size_t)GS_PROLOGUE_FILL_BLOCKS ? (want ? want : 1) : GS_PROLOGUE_FILL_BLOCKS);
    k_bwd_prologue<<<(tile_order != nullptr ? 1 : 0) + fill_blocks, 1024, 0, s>>>(
        tile_cost, tile_row0 * ntx, nt, grid, tile_order, ordered ? 1 : 0, (float*)grad_slab, n_floats);
    return GS_OK;
}

int gs_render_backward_prologue(void* grad_slab, int64_t zero_slab_rows, const int32_t* tile_cost, int32_t* tile_order,
                                int W, int H, int tile_row0, int tile_row1, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    GS_REQUIRE(tile_cost == nullptr || tile_order != nullptr, "tile_cost needs tile_order");
    GS_REQUIRE(zero_slab_rows >= 0 && (zero_slab_rows == 0 || (grad_slab != nullptr && ((uintptr_t)grad_slab & 15) == 0)),
               "zero_slab_rows must be >= 0 and the slab 16-byte aligned");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    const int ntx = (W + 15) / 16;
    if (int e = launch_bwd_prologue(grad_slab, zero_slab_rows, tile_cost, tile_order, ntx, tile_row0,
                                    (tile_row1 - tile_row0) * ntx, (hipStream_t)stream))
        return e;
    return check_launch("render_backward_prologue");
}

int gs_render_tiles_backward_slab_m(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  const int32_t* sorted_gaussians, const void* background_rgb,
                                  const int32_t* num_splats_per_pixel,
                                  const void* final_weight_per_pixel, const void* grad_image, int W,
                                  int H, int tile_row0, int tile_row1, void* grad_slab, int64_t zero_slab_rows,
                                  const int32_t* tile_cost, int32_t* tile_order, const void* segment_state,
                                  const int32_t* cut_flags, const int32_t* full_ranges, const int32_t* overflow_sorted,
                                  int backward_mode, const uint64_t* touch_masks, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    GS_REQUIRE(tile_cost == nullptr || tile_order != nullptr, "tile_cost needs tile_order (tile_cost and tile_order go together)");
    GS_REQUIRE((cut_flags == nullptr) == (full_ranges == nullptr) && (cut_flags == nullptr) == (overflow_sorted == nullptr),
               "cut_flags, full_ranges and overflow_sorted go together");
    GS_REQUIRE(segment_state == nullptr || ((uintptr_t)segment_state & 15) == 0, "segment_state must be 16-byte aligned");
    GS_REQUIRE(zero_slab_rows >= 0 && (zero_slab_rows == 0 || ((uintptr_t)grad_slab & 15) == 0),
               "zero_slab_rows must be >= 0 and the slab 16-byte aligned");
    if (int e = check_rows(H, tile_row0, tile_row1)) return e;
    int exact = 0;
    if (int e = resolve_backward_mode(backward_mode, &exact)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16;
    const int nt = (tile_row1 - tile_row0) * ntx;
    const int grid = render_grid(nt);
    // tile_order without tile_cost: the order gs_render_backward_prologue left there.  With tile_cost (or rows to
    // clear) the prologue runs here.
    const bool order_given = tile_cost == nullptr && tile_order != nullptr;
    const bool use_order = segment_state == nullptr && (order_given || (tile_cost != nullptr && nt >= GS_LPT_MIN_TILES));
    if (!order_given)
        if (int e = launch_bwd_prologue(grad_slab, zero_slab_rows,
                                        segment_state == nullptr && nt >= GS_LPT_MIN_TILES ? tile_cost : nullptr,
                                        use_order ? tile_order : nullptr, ntx, tile_row0, nt, s))
            return e;
    if (order_given && zero_slab_rows > 0)
        if (int e = launch_bwd_prologue(grad_slab, zero_slab_rows, nullptr, nullptr, ntx, tile_row0, nt, s)) return e;
    if (nt == 0) return check_launch("render_tiles_backward_slab");
    if (segment_state != nullptr) {
        // (tile, depth segment) work items from the state the forward left (gs_render_tiles_prefix / _packed
        // with the same segment_state); segment-major launch order = most work first, no order kernel
        k_render_bwd<float, 1><<<grid * SEG_MAX, RB, 0, s>>>(
            (const float*)packed, (const float*)rgb, nullptr, tile_ranges, sorted_gaussians,
            (const float*)background_rgb, num_splats_per_pixel, (const float*)final_weight_per_pixel,
            (const float*)grad_image, W, H, ntx, tile_row0 * ntx, nt, (float*)grad_slab, nullptr,
            nullptr, nullptr, 1, exact, nullptr, nullptr, nullptr, seg_state_of((void*)segment_state, W, H, tile_row0, tile_row1),
            cut_flags, full_ranges, overflow_sorted, (const unsigned long long*)touch_masks);
        return check_launch("render_tiles_backward_slab");
    }
    k_render_bwd<float, 1><<<grid, RB, GS_BWD_LDS_PAD, s>>>(
        (const float*)packed, (const float*)rgb, nullptr, tile_ranges, sorted_gaussians,
        (const float*)background_rgb, num_splats_per_pixel, (const float*)final_weight_per_pixel,
        (const float*)grad_image, W, H, ntx, tile_row0 * ntx, nt, (float*)grad_slab, nullptr,
        nullptr, nullptr, 1, exact, use_order ? tile_order : nullptr, nullptr, nullptr,
        SEG_NONE, cut_flags, full_ranges, overflow_sorted, (const unsigned long long*)touch_masks);
    return check_launch("render_tiles_backward_slab");
}

int gs_render_tiles_backward_slab(const void* packed, const void* rgb, const int32_t* tile_ranges,
                                  const int32_t* sorted_gaussians, const void* background_rgb,
                                  const int32_t* num_splats_per_pixel,
                                  const void* final_weight_per_pixel, const void* grad_image, int W,
                                  int H, int tile_row0, int tile_row1, void* grad_slab, int64_t zero_slab_rows,
                                  const int32_t* tile_cost, int32_t* tile_order, const void* segment_state,
                                  const int32_t* cut_flags, const int32_t* full_ranges, const int32_t* overflow_sorted,
                                  int backward_mode, void* stream) {
    return gs_render_tiles_backward_slab_m(packed, rgb, tile_ranges, sorted_gaussians, background_rgb, num_splats_per_pixel,
                                           final_weight_per_pixel, grad_image, W, H, tile_row0, tile_row1, grad_slab, zero_slab_rows,
                                           tile_cost, tile_order, segment_state, cut_flags, full_ranges, overflow_sorted,
                                           backward_mode, nullptr, stream);
}

int gs_render_depth(const void* packed, const void* xyz_camera_frame, const int32_t* tile_ranges,
                    const int32_t* sorted_gaussians, int W, int H, float alpha_threshold,
                    void* depth_image, void* stream) {
    GS_REQUIRE(W > 0 && H > 0, "image must be non-empty");
    hipStream_t s = (hipStream_t)stream;
    const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
    const int nt = ntx * nty;
    const int grid = render_grid(nt);
    k_render_depth<<<grid, RB, 0, s>>>((const float*)packed, (const float*)xyz_camera_frame,
                                       tile_ranges, sorted_gaussians, W, H, ntx, nt,
                                       alpha_threshold, (float*)depth_image);
    return check_launch("render_depth");
}

}  // extern "C"
