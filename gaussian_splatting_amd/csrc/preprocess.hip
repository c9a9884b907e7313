// preprocess.hip -- fused per-Gaussian stage of a frame, forward and backward.
//
// Replaces the PyTorch glue and the six per-Gaussian kernels of the reference's rasterize()
// (splat_py/rasterize.py:29-99 + projection.cu, precompute_sh.cu; SURVEY.md rows a1-a6, a9 and, for
// the backward, a12-a17): world->camera transform, pinhole projection, frustum cull, stream
// compaction of the survivors (order preserving, so visible index == the reference's boolean-mask
// index), sigmoid(opacity), Sigma_world, projection Jacobian, 2D covariance ("conic"), view-dependent
// colour from SH, and the packed record the render kernels read.  One pass over the parameters:
// 12 B (xyz) per Gaussian for the cull, then 32 + 4*C B in and ~120 B out per visible Gaussian.
// The reference runs a batched 4x4 matmul, ~10 boolean-mask gathers, a cat and six kernels with
// host synchronisations in between.
//
//   k_cull_count      cull predicate per Gaussian, per-workgroup survivor counts; its first thread also
//                     leaves the camera centre = -A^-1 t of camera_T_world (fp64; the reference takes
//                     torch.inverse() on the host side, rasterize.py:92) for k_preprocess
//   k_scan_counts     exclusive prefix over workgroups (+ total V)
//   k_preprocess      recompute the predicate, rank survivors (wave ballot + prefix), compute and
//                     write everything at the compacted index
//   k_preprocess_bwd  one thread per Gaussian of the FULL set: survivors chain the render gradients
//                     (uv, conic, opacity, rgb) back to the parameters, culled rows get zeros --
//                     the dense [N, ...] gradient tensors are written exactly once, no memset,
//                     no index_put
//
// Every forward quantity is computed with the same device functions and operation order as the
// stand-alone kernels (pg_math.h), so given the same camera-frame coordinates the results are
// bit-identical to them and to the CPU restatement.
#include "pg_math.h"
#include "tile_math.h"

namespace gs {

constexpr int PP_BLOCK = 256;

struct Frustum {
    float near, far, u_lo, u_hi, v_lo, v_hi;   // rasterize.py:33-49, compared in fp32
};

__device__ inline void camera_center(const float* __restrict__ M, float* __restrict__ center) {
    const double a = M[0], b = M[1], c = M[2], d = M[4], e = M[5], f = M[6], g = M[8], h = M[9],
                 i = M[10];
    const double tx = M[3], ty = M[7], tz = M[11];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double inv[9] = {A / det, -(b * i - c * h) / det, (b * f - c * e) / det,
                           B / det, (a * i - c * g) / det, -(a * f - c * d) / det,
                           C / det, -(a * h - b * g) / det, (a * e - b * d) / det};
    center[0] = (float)(-(inv[0] * tx + inv[1] * ty + inv[2] * tz));
    center[1] = (float)(-(inv[3] * tx + inv[4] * ty + inv[5] * tz));
    center[2] = (float)(-(inv[6] * tx + inv[7] * ty + inv[8] * tz));
}

// utils.py:60-72 as explicit fp32 arithmetic: ((m0*x + m1*y) + m2*z) + m3
__device__ inline void to_camera(const float* __restrict__ M, float x, float y, float z, float* c) {
    c[0] = M[0] * x + M[1] * y + M[2] * z + M[3];
    c[1] = M[4] * x + M[5] * y + M[6] * z + M[7];
    c[2] = M[8] * x + M[9] * y + M[10] * z + M[11];
}

__device__ inline bool is_culled(const float* c, const float* __restrict__ K, const Frustum& fr,
                                 float* uv) {
    uv[0] = K[0] * c[0] / c[2] + K[2];   // projection.cu:16-18
    uv[1] = K[4] * c[1] / c[2] + K[5];
    return (c[2] < fr.near) | (c[2] > fr.far) | (uv[0] < fr.u_lo) | (uv[0] > fr.u_hi) |
           (uv[1] < fr.v_lo) | (uv[1] > fr.v_hi);
}

// sortable bits of a depth (monotone float -> uint) and its bin in the quantile histogram: GS_CUT_HIST_BINS bins over
// the sortable-bit range of [near, far] (a visible Gaussian's depth lies inside)
__device__ inline uint32_t depth_bits(float z) {
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline int depth_bin(float z, const Frustum& fr) {
    const uint32_t lo = depth_bits(fr.near), hi = depth_bits(fr.far);
    const uint64_t range = (uint64_t)(hi - lo) + 1;
    const uint32_t x = min(max(depth_bits(z), lo), hi);
    return (int)(((uint64_t)(x - lo) * GS_CUT_HIST_BINS) / range);
}

__global__ __launch_bounds__(PP_BLOCK) void k_cull_count(const float* __restrict__ xyz,
                                                         const float* __restrict__ M,
                                                         const float* __restrict__ K, int N,
                                                         Frustum fr, int* __restrict__ block_counts,
                                                         float* __restrict__ center,
                                                         int* __restrict__ depth_hist, int sample_stride) {
    __shared__ int s_cnt[PP_BLOCK / GS_WAVE];
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (g == 0) camera_center(M, center);
    bool vis = false;
    if (g < N) {
        float c[3], uv[2];
        to_camera(M, xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2], c);
        vis = !is_culled(c, K, fr, uv);
        // depth-bucketed binning (binning.hip "depth cut"): a histogram of the depths of every sample_stride-th
        // Gaussian that is visible, over [near, far] in sortable-bit (log-like) space; the bucket boundaries are
        // its quantiles (k_scan_counts).  ~65 k device-scope atomics per frame, behind the kernel's xyz stream.
        if (depth_hist != nullptr && vis && g % sample_stride == 0)
            atomicAdd(&depth_hist[depth_bin(c[2], fr)], 1);
    }
    const int n = __popcll(__ballot(vis));
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive prefix of counts[n] in place of offsets[n]; total -> total_out[0].  One workgroup (CUT: a second one).
// depth_hist != nullptr (depth-bucketed binning): also turns k_cull_count's depth histogram into GS_CUT_BUCKETS - 1
// bucket boundaries of about equal population -- boundary k is where the cumulative sample count reaches
// (k + 1) / GS_CUT_BUCKETS, interpolated linearly inside its bin -- and returns the histogram to zero for the next
// frame.  bounds[k] = inclusive upper bound of bucket k in sortable depth bits, bounds[last] = all ones.
template <bool CUT>
__global__ __launch_bounds__(1024) void k_scan_counts(const int* __restrict__ counts, int n,
                                                      int* __restrict__ offsets,
                                                      int* __restrict__ total_out, int* __restrict__ depth_hist,
                                                      uint32_t* __restrict__ bounds, Frustum fr) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    __shared__ int s_cum[CUT ? GS_CUT_HIST_BINS : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // CUT: two workgroups -- 0 scans the counts, 1 makes the boundaries (independent; one after the other in a single
    // workgroup was 12 us, each alone is 5-7)
    if constexpr (CUT) if (blockIdx.x == 1) {
        constexpr int PER = GS_CUT_HIST_BINS / 1024;
        static_assert(GS_CUT_HIST_BINS % 1024 == 0 && GS_CUT_BUCKETS == 1024, "one thread per bucket, PER bins per thread");
        int v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            v[k] = depth_hist[tid * PER + k];
            depth_hist[tid * PER + k] = 0;
            sum += v[k];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int run = incl - sum, valid = 0;
        for (int w = 0; w < 16; w++) {
            if (w < wave) run += s_wave[w];
            valid += s_wave[w];
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
            run += v[k];
            s_cum[tid * PER + k] = run;
        }
        __syncthreads();
        uint32_t bound = 0xffffffffu;
        if (tid < GS_CUT_BUCKETS - 1 && valid > 0) {
            // target count as a fraction: (tid + 1) * valid / BUCKETS, kept exact as target_num / BUCKETS
            const int64_t target_num = (int64_t)(tid + 1) * valid;
            int lo_i = 0, len = GS_CUT_HIST_BINS;
            while (len > 0) {   // first bin j with cum[j] * BUCKETS >= target_num
                const int half = len >> 1;
                if ((int64_t)s_cum[lo_i + half] * GS_CUT_BUCKETS < target_num) {
                    lo_i += half + 1;
                    len -= half + 1;
                } else {
                    len = half;
                }
            }
            const int j = min(lo_i, GS_CUT_HIST_BINS - 1);
            const int64_t prev = j > 0 ? s_cum[j - 1] : 0, h = max(s_cum[j] - (int)prev, 1);
            // position inside bin j: (target - prev) / h of its width
            const uint32_t lo = depth_bits(fr.near), hi = depth_bits(fr.far);
            const uint64_t range = (uint64_t)(hi - lo) + 1;
            // (double arithmetic: the boundaries only have to be ascending and the same for everybody who reads
            // them -- this workgroup is their one source; every step below is monotone in the position)
            const double frac = (double)(target_num - prev * GS_CUT_BUCKETS) / ((double)h * GS_CUT_BUCKETS);   // (0, 1]
            const double pos = ((double)j + frac) / (double)GS_CUT_HIST_BINS;                                // (0, 1]
            const uint64_t b = (uint64_t)lo + (uint64_t)(pos * (double)range);
            bound = (uint32_t)min(b, (uint64_t)hi);
        }
        if (tid < GS_CUT_BUCKETS) bounds[tid] = bound;
        return;
    }
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            v[k] = i < n ? counts[i] : 0;
            sum += v[k];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = s_carry;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        int run = off + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            if (i < n) offsets[i] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) total_out[0] = s_carry;
}

// activation of the opacity logit (rasterize.py:60-62: torch.sigmoid), IEEE-only form
__device__ inline float sigmoid_det(float x) { return 1.0f / (1.0f + det_expf(-x)); }

struct PreOut {
    float* uv;        // [V,2]
    float* conic;     // [V,3]
    float* opacity;   // [V,1]  sigmoid(logit)
    float* rgb;       // [V,3]  colour the renderer consumes
    float* xyz_cam;   // [V,3]
    float* packed;    // [V,12]
    int* vis_idx;     // [V]    visible -> Gaussian
    int* rank;        // [N]    Gaussian -> visible index or -1
    uint8_t* culled;  // [N]    culling mask (1 = culled), rasterize.py:33-49
    // depth-bucketed binning (binning.hip "depth cut"), all three or none:
    float* bin_rec;             // [V,8]  u v conic0 conic1 | conic2 z 0 0 -- what the binning reads, one 32-byte sector
    const uint32_t* bounds;     // [GS_CUT_BUCKETS] bucket boundaries (k_scan_counts)
    uint16_t* bucket_of;        // [V]    the visible Gaussian's depth bucket
};

// tile-row band of a multi-GPU rank and what is needed to evaluate the candidate window
struct Band {
    int row0, row1, ntx, nty;
    float mh;
};

// BAND == true (multi-GPU): the SH colour -- 180 B of the 236 B a Gaussian reads at degree 3 -- is
// evaluated only for Gaussians whose candidate tile window (tile_culling.cu:138-156) reaches the
// rank's rows; the others cannot appear in any of the rank's tile lists.  SH rows are then read
// directly (no LDS staging: most rows are skipped).
template <int N_SH, bool BAND>
__global__ __launch_bounds__(PP_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_preprocess(
    const float* __restrict__ xyz, const float* __restrict__ quat, const float* __restrict__ scale,
    const float* __restrict__ opacity, const float* __restrict__ rgb, const float* __restrict__ sh,
    const float* __restrict__ M, const float* __restrict__ K, const float* __restrict__ center,
    int N, Frustum fr, const int* __restrict__ block_offsets, PreOut o, Band band) {
    __shared__ int s_cnt[PP_BLOCK / GS_WAVE];
    // The workgroup's SH coefficients are one contiguous block of 256 * 3 * (N_SH-1) floats: fetched
    // with coalesced 16-byte loads into LDS (a per-thread walk over its own 180-byte row makes every
    // load instruction touch 64 cache lines); rows are then read at an odd word stride
    // (conflict-free).  Half of the block at a time: 23 KiB instead of 46 KiB of LDS per workgroup
    // lets five waves per SIMD stay resident instead of three on this HBM-bound kernel.
    constexpr int SHW = 3 * (N_SH - 1);
    constexpr int HALF = PP_BLOCK / 2;
    __shared__ alignas(16) float s_sh[(N_SH > 1 && !BAND) ? HALF * SHW : 4];
    __shared__ uint32_t s_bounds[BAND ? 1 : GS_CUT_BUCKETS];
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (!BAND) {
        if (o.bin_rec != nullptr)
            for (int k = threadIdx.x; k < GS_CUT_BUCKETS; k += PP_BLOCK) s_bounds[k] = o.bounds[k];
    }
    bool vis = false;
    float c[3] = {0, 0, 1}, uv[2] = {0, 0}, p[3] = {0, 0, 0};
    // Every load of the Gaussian's own parameters and the workgroup's whole SH block are requested before
    // anything is computed (loads return in order: what is asked for first can be used first, the SH block lands while
    // the covariance is projected).  A workgroup that asks for its 23 KiB of SH only after its geometry is done has
    // 3-8 KiB in flight for half of its life: the kernel ran at 4.8 TB/s where its backward twin, whose traffic is
    // stores, reaches 5.9 (round 6).
    typedef float vfloat4 __attribute__((ext_vector_type(4)));
    constexpr int SH_PRE = (N_SH > 1 && !BAND) ? (HALF * SHW / 4 + PP_BLOCK - 1) / PP_BLOCK : 1;
    vfloat4 pre[2][SH_PRE];   // both halves: 48 registers at degree 3 -- the LDS staging, not the registers, sets the occupancy
    float q4[4] = {0, 0, 0, 1}, s3[3] = {0, 0, 0}, opa_raw = 0;
    if (g < N) {
        p[0] = xyz[g * 3 + 0]; p[1] = xyz[g * 3 + 1]; p[2] = xyz[g * 3 + 2];
        if constexpr (!BAND) {
            q4[0] = quat[g * 4 + 0]; q4[1] = quat[g * 4 + 1]; q4[2] = quat[g * 4 + 2]; q4[3] = quat[g * 4 + 3];
            s3[0] = scale[g * 3 + 0]; s3[1] = scale[g * 3 + 1]; s3[2] = scale[g * 3 + 2];
            opa_raw = opacity[g];
        }
    }
    if constexpr (N_SH > 1 && !BAND) {   // the SH block -> pre[][] (coalesced 16-byte non-temporal loads)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int g0 = blockIdx.x * PP_BLOCK + half * HALF;
            const int count = max(0, min(HALF, N - g0)) * SHW;
            const vfloat4* src4 = reinterpret_cast<const vfloat4*>(sh + (size_t)g0 * SHW);   // 16-byte aligned: g0 % 128 == 0
#pragma unroll
            for (int k = 0; k < SH_PRE; k++) {
                const int i = k * PP_BLOCK + (int)threadIdx.x;
                if (i < (count >> 2)) pre[half][k] = __builtin_nontemporal_load(src4 + i);
            }
        }
    }
    if (g < N) {
        to_camera(M, p[0], p[1], p[2], c);
        vis = !is_culled(c, K, fr, uv);
    }
    const unsigned long long bal = __ballot(vis);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int v = block_offsets[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wave; w++) v += s_cnt[w];
    if (g < N) {
        o.culled[g] = vis ? 0 : 1;
        o.rank[g] = vis ? v : -1;
    }
    const bool act = g < N && vis;   // no early return: the SH staging below has workgroup barriers

    float c3[3] = {0, 0, 0}, opa = 0;
    bool in_band = true;
    if (act) {
        if (o.vis_idx != nullptr) o.vis_idx[v] = g;
        if (o.uv != nullptr) {
            o.uv[v * 2 + 0] = uv[0];
            o.uv[v * 2 + 1] = uv[1];
        }
        if (o.xyz_cam != nullptr) {
            o.xyz_cam[v * 3 + 0] = c[0];
            o.xyz_cam[v * 3 + 1] = c[1];
            o.xyz_cam[v * 3 + 2] = c[2];
        }

        if constexpr (BAND) {
            q4[0] = quat[g * 4 + 0]; q4[1] = quat[g * 4 + 1]; q4[2] = quat[g * 4 + 2]; q4[3] = quat[g * 4 + 3];
            s3[0] = scale[g * 3 + 0]; s3[1] = scale[g * 3 + 1]; s3[2] = scale[g * 3 + 2];
            opa_raw = opacity[g];
        }
        float S9[9], W[9], J6[6];
        sigma_world_of(q4, s3, S9);
        load_rotation(M, W);
        J6[0] = K[0] / c[2];                       // projection.cu:169-174
        J6[1] = 0;
        J6[2] = -K[0] * c[0] / (c[2] * c[2]);
        J6[3] = 0;
        J6[4] = K[4] / c[2];
        J6[5] = -K[4] * c[1] / (c[2] * c[2]);
        conic_of(J6, W, S9, c3);
        if (o.conic != nullptr) {
            o.conic[v * 3 + 0] = c3[0];
            o.conic[v * 3 + 1] = c3[1];
            o.conic[v * 3 + 2] = c3[2];
        }

        opa = sigmoid_det(opa_raw);
        o.opacity[v] = opa;
        if constexpr (!BAND) {
            if (o.bin_rec != nullptr) {
                float4* br = reinterpret_cast<float4*>(o.bin_rec) + 2 * (size_t)v;
                br[0] = make_float4(uv[0], uv[1], c3[0], c3[1]);
                br[1] = make_float4(c3[2], c[2], 0.0f, 0.0f);
                // the depth bucket: first boundary >= the depth's sortable bits (equal depths share a bucket)
                const uint32_t x = depth_bits(c[2]);
                int lo_i = 0, len = GS_CUT_BUCKETS - 1;
                while (len > 0) {
                    const int half = len >> 1;
                    if (s_bounds[lo_i + half] < x) {
                        lo_i += half + 1;
                        len -= half + 1;
                    } else {
                        len = half;
                    }
                }
                o.bucket_of[v] = (uint16_t)lo_i;
            }
        }

        if constexpr (BAND) {
            const float a = c3[0] + 0.25f, b = c3[1] / 2.0f, cc = c3[2] + 0.25f;   // tile_culling.cu:142-144
            const Obb obb = compute_obb(uv[0], uv[1], a, b, cc, band.mh);
            const Window w = candidate_window(uv[0], uv[1], obb.radius_tiles, band.ntx, band.nty, band.row0,
                                              band.row1);
            in_band = w.sx < w.ex && w.sy < w.ey;
        }
    }

    float col[3] = {0, 0, 0};
    if constexpr (N_SH == 1) {
        if (act) { col[0] = rgb[g * 3 + 0]; col[1] = rgb[g * 3 + 1]; col[2] = rgb[g * 3 + 2]; }
    } else {
        // precompute_sh.cu:28-55 with coefficient 0 = rgb and 1.. = sh (rasterize.py:89)
        float Y[N_SH];
        if (act && in_band) {
            float d[3] = {p[0] - center[0], p[1] - center[1], p[2] - center[2]};
            const float r = 1.0f / __builtin_sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= r; d[1] *= r; d[2] *= r;
            sh_basis<float, N_SH>(d, Y);
        }
        auto colour_from = [&](const float* shg) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float t = 0;
                t += Y[0] * rgb[g * 3 + ch];
#pragma unroll
                for (int s = 1; s < N_SH; s++) t += Y[s] * shg[(N_SH - 1) * ch + (s - 1)];
                t *= GS_R_SH_0;
                col[ch] = t;
            }
        };
        if constexpr (BAND) {
            if (act && in_band) colour_from(sh + (size_t)g * SHW);
        } else {
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int g0 = blockIdx.x * PP_BLOCK + half * HALF;
                const int rows = max(0, min(HALF, N - g0));
                const int count = rows * SHW;
                const float* src = sh + (size_t)g0 * SHW;   // 16-byte aligned: g0 is a multiple of 128
                // (read once per frame: non-temporal, so that the 0.5 GB stream does not push the records this kernel
                // writes -- and the render gathers -- out of the caches).  pre[half][] holds it, requested at the kernel's top
                vfloat4* dst4 = reinterpret_cast<vfloat4*>(s_sh);
                if (half) __syncthreads();   // the first half has been consumed
#pragma unroll
                for (int k = 0; k < SH_PRE; k++) {
                    const int i = k * PP_BLOCK + (int)threadIdx.x;
                    if (i < (count >> 2)) dst4[i] = pre[half][k];
                }
                for (int i = (count & ~3) + threadIdx.x; i < count; i += PP_BLOCK) s_sh[i] = src[i];
                __syncthreads();
                if (act && (threadIdx.x / HALF) == half) colour_from(s_sh + (threadIdx.x % HALF) * SHW);
            }
        }
    }
    if (!act) return;
    if (o.rgb != nullptr) {   // (the renderer reads the colour from the packed record; the array serves callers)
        o.rgb[v * 3 + 0] = col[0];
        o.rgb[v * 3 + 1] = col[1];
        o.rgb[v * 3 + 2] = col[2];
    }

    // packed render record, identical to k_pack
    float pk[GS_PACKED_WIDTH];
    pack_record<float>(uv[0], uv[1], c3, opa, col, pk);
    float4* dst = reinterpret_cast<float4*>(o.packed + (size_t)v * GS_PACKED_WIDTH);
    dst[0] = make_float4(pk[0], pk[1], pk[2], pk[3]);
    dst[1] = make_float4(pk[4], pk[5], pk[6], pk[7]);
    dst[2] = make_float4(pk[8], pk[9], pk[10], pk[11]);
}


// ---- multi-GPU: band-compact per-Gaussian stage ------------------------------------------------------------
// A rank renders 1/G of the tile rows but gs_preprocess_forward evaluates and writes ~100 B for every visible
// Gaussian of the frame (only the SH colour is band-limited): 0.126 of a rank's 0.53 ms at 8 ranks (workload D).
// Here the full evaluation runs only for the Gaussians that can reach the rank's band, into arrays compacted to
// exactly those rows:
//   k_band_project     every Gaussian: transform, projection, frustum cull, ordered compaction (visible index v as in
//                      k_preprocess) -> culling mask, rank, vis_idx, uv[v], sigmoid(opacity)[v], and mask[v]: bit s =
//                      the Gaussian CAN reach the band of rank s.  The test needs no covariance: the candidate window of
//                      tile_culling.cu:138-156 has radius_tiles = ceil(mh sqrt(l1) / 16) + 1 with l1 = lambda_max(Sigma_2D)
//                      + 0.25 and Sigma_2D = (J W) Sigma (J W)^T <= s_max^2 |W|^2 J J^T, so l1 <= s_max^2 |W|^2
//                      lambda_max(J J^T) + 0.25 (s_max = the largest scale, |W| = the rotation block's norm, 1 for a
//                      rigid pose): a superset of the exact window, the same on every rank.  A row that is exchanged
//                      for nothing is a row of zeros.
//   (gs_halo_plan_masked: the exchange plan from these masks; its send list -- the visible Gaussians with this
//   rank's bit, ascending -- is the list of rows of the compact arrays)
//   k_preprocess_list  row l of the list: Sigma, J, conic, SH colour, packed record of Gaussian vis_idx[send[l]],
//                      written at row l.  Same device functions and operation order as k_preprocess: the same bits.
// Binning, sort and render then work on the compact arrays (row index l, monotone in v: depth ties keep their
// order, the tile lists are the single-GPU lists with v relabelled), and the band's render-gradient slab [L, 9]
// IS the send buffer of the gradient exchange.
struct BandRows {
    int v[GS_MAX_RANKS + 1];
};

__device__ inline float rotation_norm2_bound(const float* __restrict__ M) {
    // lambda_max(W^T W) <= its largest absolute row sum (1 + rounding for a rotation)
    float best = 0;
    for (int i = 0; i < 3; i++) {
        float row = 0;
        for (int j = 0; j < 3; j++) {
            const float d = M[0 + i] * M[0 + j] + M[4 + i] * M[4 + j] + M[8 + i] * M[8 + j];
            row += __builtin_fabsf(d);
        }
        best = fmaxf(best, row);
    }
    return best;
}

// bit s = the Gaussian CAN reach the tile rows of band s: the candidate window of tile_culling.cu:138-156 with its
// radius bounded from the largest scale (see the block comment above) -- a superset of the exact window, the same
// on every rank
// rot2 = rotation_norm2_bound(M), the same value for every Gaussian of the frame (callers form it once per workgroup)
__device__ inline uint32_t band_mask_bound(const float* c, const float* uv, const float* __restrict__ scale3, float rot2,
                                           const float* __restrict__ K, float mh, int nty, const BandRows& rows, int G) {
    // A BOUND, not a reference quantity: it carries 0.1 % of slack twice over, so the hardware's approximate
    // exponential, reciprocal and square root (1-2 ulp) serve -- every rank issues these same instructions on the same
    // inputs, so the masks agree across ranks, and the old and the fused frontends share this function.  One
    // exponential of the largest log-scale instead of three and their maximum (round 6: ~40 instructions less per
    // visible Gaussian in the kernels that stream all N)
    const float s_max = __builtin_amdgcn_exp2f(fmaxf(fmaxf(scale3[0], scale3[1]), scale3[2]) * 1.44269504088896341f);
    const float iz = __builtin_amdgcn_rcpf(c[2]);
    const float jx = K[0] * iz, jy = K[4] * iz, tx = K[0] * c[0] * iz * iz, ty = K[4] * c[1] * iz * iz;
    const float A = jx * jx + tx * tx, C = jy * jy + ty * ty, B = tx * ty;
    const float lam = 0.5f * (A + C) + __builtin_amdgcn_sqrtf(0.25f * (A - C) * (A - C) + B * B);
    const float l1 = s_max * s_max * rot2 * lam * 1.001f + 0.25f;
    const float rt = __builtin_ceilf(mh * __builtin_amdgcn_sqrtf(l1) * 1.001f * 0.0625f) + 1.0f;
    uint32_t m = 0;
    if (rt < 1.0e6f) {
        const int r = (int)rt;
        const int py = f2i(__builtin_floorf(uv[1] / 16.0f));
        const int sy = max(0, py - r), ey = min(nty, py + r);
        for (int s2 = 0; s2 < G; s2++)
            if (sy < rows.v[s2 + 1] && ey > rows.v[s2] && sy < ey) m |= 1u << s2;
    } else {
        m = (1u << G) - 1;   // unbounded or not a number: every band (a superset is always safe)
    }
    return m;
}

template <int PASS>   // 0: count per block; 1: write
__global__ __launch_bounds__(PP_BLOCK) void k_band_project(
    const float* __restrict__ xyz, const float* __restrict__ scale, const float* __restrict__ opacity,
    const float* __restrict__ M, const float* __restrict__ K, int N, Frustum fr, float mh, int nty, BandRows rows, int G,
    const int* __restrict__ block_offsets, uint8_t* __restrict__ culled, int* __restrict__ rank,
    int* __restrict__ vis_idx, float* __restrict__ uv_out, float* __restrict__ opa_out, uint32_t* __restrict__ mask,
    int* __restrict__ bit_counts /*[G][nblk] by visible-index block*/, int nblk) {
    __shared__ int s_cnt[PP_BLOCK / GS_WAVE];
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool vis = false;
    float c[3] = {0, 0, 1}, uv[2] = {0, 0};
    if (g < N) {
        to_camera(M, xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2], c);
        vis = !is_culled(c, K, fr, uv);
    }
    const unsigned long long bal = __ballot(vis);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int v = block_offsets[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wave; w++) v += s_cnt[w];
    if (g < N) {
        culled[g] = vis ? 0 : 1;
        rank[g] = vis ? v : -1;
    }
    if (!(g < N && vis)) return;
    vis_idx[v] = g;
    uv_out[v * 2 + 0] = uv[0];
    uv_out[v * 2 + 1] = uv[1];
    opa_out[v] = sigmoid_det(opacity[g]);
    mask[v] = band_mask_bound(c, uv, scale + (size_t)g * 3, rotation_norm2_bound(M), K, mh, nty, rows, G);
}

// per-256-block counts of every mask bit, by VISIBLE index (the layout gs_halo_plan's scan expects)
__global__ __launch_bounds__(PP_BLOCK) void k_band_bit_counts(const uint32_t* __restrict__ mask,
                                                              const int* __restrict__ visible_count, int G,
                                                              int* __restrict__ blk_counts, int nblk) {
    __shared__ int s_cnt[GS_MAX_RANKS][PP_BLOCK / GS_WAVE];
    const int v = blockIdx.x * PP_BLOCK + threadIdx.x;
    const uint32_t m = v < *visible_count ? mask[v] : 0;
    for (int s2 = 0; s2 < G; s2++) {
        const int n = __popcll(__ballot((m >> s2) & 1u));
        if ((threadIdx.x & 63) == 0) s_cnt[s2][threadIdx.x >> 6] = n;
    }
    __syncthreads();
    if (threadIdx.x < G)
        blk_counts[threadIdx.x * nblk + blockIdx.x] =
            s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
}

template <int N_SH>
__global__ __launch_bounds__(PP_BLOCK) void k_preprocess_list(
    const float* __restrict__ xyz, const float* __restrict__ quat, const float* __restrict__ scale,
    const float* __restrict__ rgb, const float* __restrict__ sh, const float* __restrict__ M,
    const float* __restrict__ K, const float* __restrict__ center, const int* __restrict__ list,
    const int* __restrict__ list_count, const int* __restrict__ vis_idx, const float* __restrict__ uv_v,
    const float* __restrict__ opa_v, float* __restrict__ uv_l, float* __restrict__ xyz_cam_l,
    float* __restrict__ conic_l, float* __restrict__ packed_l) {
    constexpr int SHW = 3 * (N_SH - 1);
    // (a workgroup-wide fetch of the rows' SH coefficients through LDS, as k_preprocess does for contiguous rows, was
    // measured and is slower here: 0.062 -> 0.089 ms at 8 ranks -- scattered 180-byte rows, one word per lane)
    const int l = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (l >= *list_count) return;
    const int v = list[l];
    const int g = vis_idx[v];
    const float p[3] = {xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2]};
    float c[3];
    to_camera(M, p[0], p[1], p[2], c);
    const float uv[2] = {uv_v[v * 2 + 0], uv_v[v * 2 + 1]};   // == K c / z + c0 of k_band_project (same expression)
    uv_l[l * 2 + 0] = uv[0];
    uv_l[l * 2 + 1] = uv[1];
    xyz_cam_l[l * 3 + 0] = c[0];
    xyz_cam_l[l * 3 + 1] = c[1];
    xyz_cam_l[l * 3 + 2] = c[2];
    const float q4[4] = {quat[g * 4 + 0], quat[g * 4 + 1], quat[g * 4 + 2], quat[g * 4 + 3]};
    const float s3[3] = {scale[g * 3 + 0], scale[g * 3 + 1], scale[g * 3 + 2]};
    float S9[9], W[9], J6[6], c3[3];
    sigma_world_of(q4, s3, S9);
    load_rotation(M, W);
    J6[0] = K[0] / c[2];                       // projection.cu:169-174
    J6[1] = 0;
    J6[2] = -K[0] * c[0] / (c[2] * c[2]);
    J6[3] = 0;
    J6[4] = K[4] / c[2];
    J6[5] = -K[4] * c[1] / (c[2] * c[2]);
    conic_of(J6, W, S9, c3);
    conic_l[l * 3 + 0] = c3[0];
    conic_l[l * 3 + 1] = c3[1];
    conic_l[l * 3 + 2] = c3[2];
    float col[3];
    if constexpr (N_SH == 1) {
        col[0] = rgb[g * 3 + 0]; col[1] = rgb[g * 3 + 1]; col[2] = rgb[g * 3 + 2];
    } else {
        float Y[N_SH];
        float d[3] = {p[0] - center[0], p[1] - center[1], p[2] - center[2]};
        const float r = 1.0f / __builtin_sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] *= r; d[1] *= r; d[2] *= r;
        sh_basis<float, N_SH>(d, Y);
        const float* shg = sh + (size_t)g * SHW;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float t = 0;
            t += Y[0] * rgb[g * 3 + ch];
#pragma unroll
            for (int s2 = 1; s2 < N_SH; s2++) t += Y[s2] * shg[(N_SH - 1) * ch + (s2 - 1)];
            t *= GS_R_SH_0;
            col[ch] = t;
        }
    }
    float pk[GS_PACKED_WIDTH];
    pack_record<float>(uv[0], uv[1], c3, opa_v[v], col, pk);
    float4* dst = reinterpret_cast<float4*>(packed_l + (size_t)l * GS_PACKED_WIDTH);
    dst[0] = make_float4(pk[0], pk[1], pk[2], pk[3]);
    dst[1] = make_float4(pk[4], pk[5], pk[6], pk[7]);
    dst[2] = make_float4(pk[8], pk[9], pk[10], pk[11]);
}

// ---- multi-GPU: the fused band frontend (ABI 8) ---------------------------------------------------------------
// gs_band_project + gs_halo_plan_masked + gs_preprocess_forward_list were nine launches (cull count, scan, project,
// bit counts, scan of G rows, send list + bounds, list evaluation: 0.14 of a rank's 0.49 ms at 8 ranks) that read xyz
// three times and chained list -> visible index -> Gaussian index gathers.  Four launches here, every count by
// GAUSSIAN-index block -- owner slices are whole blocks of the Gaussian index, so the exchange plan is 2 G
// differences of scanned offsets and the receive layout needs no visible-index bookkeeping:
//   k_band_count   every Gaussian: transform, cull, band mask (band_mask_bound) -> gmask[g] (bit 15 = visible, bit s =
//                  can reach band s) and per-block counts of the visible Gaussians (row 0) and of every band's
//                  Gaussians (row 1 + s); the camera centre
//   k_band_scan    one workgroup per row: exclusive prefix per block, total at [nb]
//   k_band_write   every Gaussian again: culling mask, rank, uv[v], sigmoid(opacity)[v] (API + owned backward), and
//                  for the Gaussians of MY band -- rows l = row (1 + me)'s prefix, ascending in g, hence in v: the
//                  send list -- send_index[l] = v, list_g[l] = g, uv_l[l], xyz_cam_l[l], the opacity slot of
//                  packed_l[l] and the world position (staged in conic_l[l]); its extra workgroup writes the plan
//   k_band_rows    row l: Sigma, J, conic, SH colour, packed record (k_preprocess_list's evaluation: the same device
//                  functions in the same order, the same bits) from coalesced reads of what k_band_write left plus
//                  gathers of quaternion / scale / rgb / sh
struct BandOwners {
    int v[GS_MAX_RANKS + 1];   // owner slices in 256-blocks of the Gaussian index
};

__global__ __launch_bounds__(PP_BLOCK) void k_band_count(
    const float* __restrict__ xyz, const float* __restrict__ scale, const float* __restrict__ M,
    const float* __restrict__ K, int N, Frustum fr, float mh, int nty, BandRows rows, int G,
    int* __restrict__ counts, int nbp, uint16_t* __restrict__ gmask, float* __restrict__ center) {
    __shared__ int s_cnt[GS_MAX_RANKS + 1][PP_BLOCK / GS_WAVE];
    __shared__ float s_rot2;
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    if (g == 0) camera_center(M, center);
    if (threadIdx.x == 0) s_rot2 = rotation_norm2_bound(M);   // (once per workgroup instead of once per lane)
    __syncthreads();
    uint32_t m = 0;
    if (g < N) {
        float c[3], uv[2];
        to_camera(M, xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2], c);
        if (!is_culled(c, K, fr, uv)) m = 0x8000u | band_mask_bound(c, uv, scale + (size_t)g * 3, s_rot2, K, mh, nty, rows, G);
        gmask[g] = (uint16_t)m;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const int n = __popcll(__ballot((m >> 15) & 1u));
        if (lane == 0) s_cnt[0][wave] = n;
    }
    for (int s2 = 0; s2 < G; s2++) {
        const int n = __popcll(__ballot((m >> s2) & 1u));
        if (lane == 0) s_cnt[1 + s2][wave] = n;
    }
    __syncthreads();
    if (threadIdx.x <= G)
        counts[threadIdx.x * nbp + blockIdx.x] =
            s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
}

// row r of counts[.][nbp] (nb entries used) -> exclusive prefix in row r of offsets, the total at [nb]
__global__ __launch_bounds__(1024) void k_band_scan(const int* __restrict__ counts, int nb, int nbp,
                                                    int* __restrict__ offsets) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    counts += (size_t)blockIdx.x * nbp;
    offsets += (size_t)blockIdx.x * nbp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 4096) {
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            v[k] = i < nb ? counts[i] : 0;
            sum += v[k];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int off = s_carry;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        int run = off + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            if (i < nb) offsets[i] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) offsets[nb] = s_carry;
}

__global__ __launch_bounds__(PP_BLOCK) void k_band_write(
    const float* __restrict__ xyz, const float* __restrict__ opacity, const float* __restrict__ M,
    const float* __restrict__ K, int N, Frustum fr, const uint16_t* __restrict__ gmask,
    const int* __restrict__ offsets, int nb, int nbp, int G, int me, BandOwners owners,
    uint8_t* __restrict__ culled, int* __restrict__ rank, float* __restrict__ uv_out, float* __restrict__ opa_out,
    int* __restrict__ send_index, int* __restrict__ list_g, float* __restrict__ uv_l, float* __restrict__ xyz_cam_l,
    float* __restrict__ p_stage /* conic_l: the world position until k_band_rows replaces it */,
    float* __restrict__ packed_l, int* __restrict__ plan, int* __restrict__ plan_host) {
    if ((int)blockIdx.x == nb) {
        // the extra workgroup: the frame's plan record (rows of my list, V, v_lo, v_hi, send[G], recv[G]) from the
        // scanned offsets at the owner slices' block boundaries
        const int t = threadIdx.x;
        auto at = [&](int row, int blk) { return offsets[row * nbp + min(blk, nb)]; };
        int val = 0;
        if (t == 0) val = at(1 + me, nb);
        else if (t == 1) val = at(0, nb);
        else if (t == 2) val = at(0, owners.v[me]);
        else if (t == 3) val = at(0, owners.v[me + 1]);
        else if (t < 4 + G) val = at(1 + me, owners.v[t - 4 + 1]) - at(1 + me, owners.v[t - 4]);          // rows I send to owner t - 4
        else if (t < 4 + 2 * G) val = at(1 + t - 4 - G, owners.v[me + 1]) - at(1 + t - 4 - G, owners.v[me]);   // rows from sender t - 4 - G
        if (t < 4 + 2 * G) {
            plan[t] = val;
            if (plan_host != nullptr) plan_host[t] = val;
        }
        if (plan_host != nullptr) __threadfence_system();
        return;
    }
    __shared__ int s_vis[PP_BLOCK / GS_WAVE], s_mine[PP_BLOCK / GS_WAVE];
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t m = g < N ? gmask[g] : 0u;
    const bool vis = (m >> 15) & 1u, mine = (m >> me) & 1u;
    const unsigned long long bv = __ballot(vis), bm = __ballot(mine);
    if (lane == 0) {
        s_vis[wave] = __popcll(bv);
        s_mine[wave] = __popcll(bm);
    }
    __syncthreads();
    if (g >= N) return;
    culled[g] = vis ? 0 : 1;
    int v = offsets[blockIdx.x] + __popcll(bv & ((1ull << lane) - 1));
    int l = offsets[(1 + me) * nbp + blockIdx.x] + __popcll(bm & ((1ull << lane) - 1));
    for (int w = 0; w < wave; w++) {
        v += s_vis[w];
        l += s_mine[w];
    }
    rank[g] = vis ? v : -1;
    if (!vis) return;
    const float p[3] = {xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2]};
    float c[3], uv[2];
    to_camera(M, p[0], p[1], p[2], c);
    (void)is_culled(c, K, fr, uv);   // (uv: the expression every other kernel uses)
    uv_out[v * 2 + 0] = uv[0];
    uv_out[v * 2 + 1] = uv[1];
    const float opa = sigmoid_det(opacity[g]);
    opa_out[v] = opa;
    if (!mine) return;
    send_index[l] = v;
    list_g[l] = g;
    uv_l[l * 2 + 0] = uv[0];
    uv_l[l * 2 + 1] = uv[1];
    xyz_cam_l[l * 3 + 0] = c[0];
    xyz_cam_l[l * 3 + 1] = c[1];
    xyz_cam_l[l * 3 + 2] = c[2];
    p_stage[l * 3 + 0] = p[0];
    p_stage[l * 3 + 1] = p[1];
    p_stage[l * 3 + 2] = p[2];
    packed_l[(size_t)l * GS_PACKED_WIDTH + 3] = opa;
}

template <int N_SH>
__global__ __launch_bounds__(PP_BLOCK) void k_band_rows(
    const float* __restrict__ quat, const float* __restrict__ scale, const float* __restrict__ rgb,
    const float* __restrict__ sh, const float* __restrict__ M, const float* __restrict__ K,
    const float* __restrict__ center, const int* __restrict__ list_g, const int* __restrict__ list_count,
    const float* __restrict__ uv_l, const float* __restrict__ xyz_cam_l, float* __restrict__ conic_l,
    float* __restrict__ packed_l) {
    constexpr int SHW = 3 * (N_SH - 1);
    // (the SH row is walked by its own lane: a row-cooperative fetch -- twelve lanes per row, 16-byte chunks through
    // wave-private LDS -- was measured and is 24 % slower, scripts/experiments/list_coop_sh_fetch.patch)
    const int l = blockIdx.x * PP_BLOCK + threadIdx.x;
    const bool act = l < *list_count;
    if (!act) return;
    const int g = list_g[l];
    float col[3] = {0, 0, 0}, p[3] = {0, 0, 0}, c[3] = {0, 0, 1}, uv[2] = {0, 0}, opa = 0;
    float q4[4] = {1, 0, 0, 0}, s3[3] = {0, 0, 0}, c0[3] = {0, 0, 0};
    if (act) {
        p[0] = conic_l[l * 3 + 0]; p[1] = conic_l[l * 3 + 1]; p[2] = conic_l[l * 3 + 2];
        c[0] = xyz_cam_l[l * 3 + 0]; c[1] = xyz_cam_l[l * 3 + 1]; c[2] = xyz_cam_l[l * 3 + 2];
        uv[0] = uv_l[l * 2 + 0]; uv[1] = uv_l[l * 2 + 1];
        opa = packed_l[(size_t)l * GS_PACKED_WIDTH + 3];
        const float4 q = *reinterpret_cast<const float4*>(quat + (size_t)g * 4);
        q4[0] = q.x; q4[1] = q.y; q4[2] = q.z; q4[3] = q.w;
        s3[0] = scale[g * 3 + 0]; s3[1] = scale[g * 3 + 1]; s3[2] = scale[g * 3 + 2];
        c0[0] = rgb[g * 3 + 0]; c0[1] = rgb[g * 3 + 1]; c0[2] = rgb[g * 3 + 2];
    }
    if constexpr (N_SH == 1) {
        col[0] = c0[0]; col[1] = c0[1]; col[2] = c0[2];
    } else {
        float Y[N_SH];
        if (act) {
            float d[3] = {p[0] - center[0], p[1] - center[1], p[2] - center[2]};
            const float r = 1.0f / __builtin_sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= r; d[1] *= r; d[2] *= r;
            sh_basis<float, N_SH>(d, Y);
        }
        auto colour_from = [&](const float* shg) {   // precompute_sh.cu:28-55, coefficient 0 = rgb (rasterize.py:89)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                float t = 0;
                t += Y[0] * c0[ch];
#pragma unroll
                for (int s2 = 1; s2 < N_SH; s2++) t += Y[s2] * shg[(N_SH - 1) * ch + (s2 - 1)];
                t *= GS_R_SH_0;
                col[ch] = t;
            }
        };
        if (act) colour_from(sh + (size_t)g * SHW);
    }
    if (!act) return;
    float S9[9], W[9], J6[6], c3[3];
    sigma_world_of(q4, s3, S9);
    load_rotation(M, W);
    J6[0] = K[0] / c[2];                       // projection.cu:169-174
    J6[1] = 0;
    J6[2] = -K[0] * c[0] / (c[2] * c[2]);
    J6[3] = 0;
    J6[4] = K[4] / c[2];
    J6[5] = -K[4] * c[1] / (c[2] * c[2]);
    conic_of(J6, W, S9, c3);
    conic_l[l * 3 + 0] = c3[0];
    conic_l[l * 3 + 1] = c3[1];
    conic_l[l * 3 + 2] = c3[2];
    float pk[GS_PACKED_WIDTH];
    pack_record<float>(uv[0], uv[1], c3, opa, col, pk);
    float4* dst = reinterpret_cast<float4*>(packed_l + (size_t)l * GS_PACKED_WIDTH);
    dst[0] = make_float4(pk[0], pk[1], pk[2], pk[3]);
    dst[1] = make_float4(pk[4], pk[5], pk[6], pk[7]);
    dst[2] = make_float4(pk[8], pk[9], pk[10], pk[11]);
}

// the receive side of the gradient exchange by Gaussian-index blocks: sender s's rows for my slice are its band's
// Gaussians inside [256 owners[me], 256 owners[me + 1]), ascending, so the row of Gaussian g from sender s sits at
// (offsets[1 + s][block of g] - offsets[1 + s][owners[me]]) + (the bit-s Gaussians of g's block in front of g)
struct GatherPlan {
    const uint16_t* gmask;   // [N]
    const int* offsets;      // [G + 1][nbp]
    const float* recv;       // [sum of recv counts][9]
    int nbp, G, blk0;        // blk0 = owners[me]
    int recv_off[GS_MAX_RANKS];
};
// -> true and row[9] = the sum over the senders (ascending) of the received rows of Gaussian g; every thread of the
// workgroup must call it (ballots + a barrier).  blk = the 256-block of g, tid = its index inside the block
__device__ inline uint32_t gather_row(const GatherPlan& gp, int blk, int g, bool in_range, float* row,
                                      int (*s_cnt)[PP_BLOCK / GS_WAVE]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t m = in_range ? gp.gmask[g] : 0u;
    unsigned long long bal[GS_MAX_RANKS];
    for (int s2 = 0; s2 < gp.G; s2++) {
        bal[s2] = __ballot((m >> s2) & 1u);
        if (lane == 0) s_cnt[s2][wave] = __popcll(bal[s2]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 9; j++) row[j] = 0;
    for (int s2 = 0; s2 < gp.G; s2++) {
        if (((m >> s2) & 1u) == 0) continue;
        int pos = gp.offsets[(1 + s2) * gp.nbp + blk] - gp.offsets[(1 + s2) * gp.nbp + gp.blk0];
        for (int w = 0; w < wave; w++) pos += s_cnt[s2][w];
        pos += __popcll(bal[s2] & ((1ull << lane) - 1));
        const float* src = gp.recv + (size_t)(gp.recv_off[s2] + pos) * 9;
#pragma unroll
        for (int j = 0; j < 9; j++) row[j] += src[j];
    }
    return m;
}

// the owned rows materialised ([v_hi - v_lo, 9] by visible index: what uv.grad of the owned Gaussians is a view of)
__global__ __launch_bounds__(PP_BLOCK) void k_band_gather_sum(GatherPlan gp, int N, const int* __restrict__ rank, int v_lo,
                                                              float* __restrict__ out) {
    __shared__ int s_cnt[GS_MAX_RANKS][PP_BLOCK / GS_WAVE];
    const int blk = gp.blk0 + blockIdx.x;
    const int g = blk * PP_BLOCK + threadIdx.x;
    float row[9];
    const uint32_t m = gather_row(gp, blk, g, g < N, row, s_cnt);
    if (!((m >> 15) & 1u)) return;
    float* o = out + (size_t)(rank[g] - v_lo) * 9;
#pragma unroll
    for (int j = 0; j < 9; j++) o[j] = row[j];
}

struct PreGrad {
    float* xyz;         // [N,3]
    float* quaternion;  // [N,4]
    float* scale;       // [N,3]
    float* opacity;     // [N,1]
    float* rgb;         // [N,3]
    float* sh;          // [N,3,N_SH-1] or null
};

// GATHER (multi-GPU, owner-sliced backward): the render-gradient row of a Gaussian is not read from a slab but summed
// on the spot from the rows the all_to_all delivered (gather_row: the senders in ascending order, as
// k_band_gather_sum adds them -- the same bits) -- no [owned, 9] buffer written and read back, one launch less
template <int N_SH, bool GATHER = false>
__global__ __launch_bounds__(PP_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_preprocess_bwd(
    const float* __restrict__ xyz, const float* __restrict__ quat, const float* __restrict__ scale,
    const float* __restrict__ M, const float* __restrict__ K, const float* __restrict__ center,
    const int* __restrict__ rank, const float* __restrict__ opacity_act,
    const float* __restrict__ g_slab, int v_base, int N, PreGrad o, GatherPlan gp = GatherPlan{}) {
    constexpr int SHW = 3 * (N_SH - 1);
    __shared__ int s_gather[GATHER ? GS_MAX_RANKS : 1][PP_BLOCK / GS_WAVE];
    float gathered[9];
    if constexpr (GATHER) {
        const int lg = blockIdx.x * PP_BLOCK + threadIdx.x;   // the slice starts at Gaussian 256 gp.blk0
        gather_row(gp, gp.blk0 + blockIdx.x, gp.blk0 * PP_BLOCK + lg, lg < N, gathered, s_gather);
    }
    // SH gradients (180 B per Gaussian at degree 3) are staged in LDS and written back as one
    // contiguous block with coalesced 16-byte stores -- half of the workgroup's rows at a time
    // (23 KiB of LDS instead of 46: more resident waves on this HBM-bound kernel)
    constexpr int HALF = PP_BLOCK / 2;
    __shared__ alignas(16) float s_sh[N_SH > 1 ? HALF * SHW : 4];
    const int g = blockIdx.x * PP_BLOCK + threadIdx.x;
    const bool in_range = g < N;
    const int v = in_range ? rank[g] : -1;
    float gx[3] = {0, 0, 0}, gq[4] = {0, 0, 0, 0}, gs[3] = {0, 0, 0}, go = 0, gc[3] = {0, 0, 0};
    float Y[N_SH];
    float gl3[3] = {0, 0, 0};   // d colour / d(sh row) factors of this Gaussian; 0 for culled rows
#pragma unroll
    for (int s_ = 0; s_ < N_SH; s_++) Y[s_] = 0;
    if (v >= 0) {
        const float p[3] = {xyz[g * 3 + 0], xyz[g * 3 + 1], xyz[g * 3 + 2]};
        float c[3];
        to_camera(M, p[0], p[1], p[2], c);
        // colour: precompute_sh.cu:61-111, then the split of cat(rgb, sh) (rasterize.py:89)
        // render gradients of visible Gaussian v: row v - v_base of the [*, 9] slab
        // (rgb 3 | opacity 1 | uv 2 | conic 3)
        const float* gsl = GATHER ? gathered : g_slab + (size_t)(v - v_base) * 9;
        const float gr[3] = {gsl[0], gsl[1], gsl[2]};
        if constexpr (N_SH == 1) {
            gc[0] = gr[0]; gc[1] = gr[1]; gc[2] = gr[2];
        } else {
            float d[3] = {p[0] - center[0], p[1] - center[1], p[2] - center[2]};
            const float r = 1.0f / __builtin_sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] *= r; d[1] *= r; d[2] *= r;
            sh_basis<float, N_SH>(d, Y);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                gl3[ch] = gr[ch] * GS_R_SH_0;
                gc[ch] = gl3[ch] * Y[0];
            }
        }
        // opacity: d sigmoid = y (1 - y)
        const float y = opacity_act[v];
        go = gsl[3] * (1.0f - y) * y;
        // conic -> Sigma_world, J (projection_backward.cu:385-471)
        const float q4[4] = {quat[g * 4 + 0], quat[g * 4 + 1], quat[g * 4 + 2], quat[g * 4 + 3]};
        const float s3[3] = {scale[g * 3 + 0], scale[g * 3 + 1], scale[g * 3 + 2]};
        float S9[9], W[9], J6[6], gS9[9], gJ6[6];
        sigma_world_of(q4, s3, S9);
        load_rotation(M, W);
        const float z = c[2], z2 = c[2] * c[2], z3 = c[2] * c[2] * c[2];
        const float fx = K[0], fy = K[4];
        J6[0] = fx / z; J6[1] = 0; J6[2] = -fx * c[0] / z2;
        J6[3] = 0; J6[4] = fy / z; J6[5] = -fy * c[1] / z2;
        const float gc3[3] = {gsl[6], gsl[7], gsl[8]};
        conic_bwd_of(J6, W, S9, gc3, gS9, gJ6);
        // Sigma_world -> quaternion, scale (projection_backward.cu:174-315)
        sigma_world_bwd_of(q4, s3, gS9, gq, gs);
        // J -> camera-frame xyz (projection_backward.cu:93-120)
        float gcam[3];
        gcam[0] = gJ6[2] * -fx / z2;
        gcam[1] = gJ6[5] * -fy / z2;
        gcam[2] = gJ6[0] * -fx / z2 + gJ6[4] * -fy / z2 + gJ6[2] * 2 * c[0] * fx / z3 +
                  gJ6[5] * 2 * c[1] * fy / z3;
        // uv -> camera-frame xyz (projection_backward.cu:9-36; nothing when z <= 0, Q10)
        if (z > 0.0f) {
            const float gu = gsl[4], gv = gsl[5];
            gcam[0] += gu * (fx / z);
            gcam[1] += gv * (fy / z);
            gcam[2] += gu * (-fx * c[0] / z2) + gv * (-fy * c[1] / z2);
        }
        // camera frame -> world: transpose of the rotation block (backward of utils.py:64)
        gx[0] = M[0] * gcam[0] + M[4] * gcam[1] + M[8] * gcam[2];
        gx[1] = M[1] * gcam[0] + M[5] * gcam[1] + M[9] * gcam[2];
        gx[2] = M[2] * gcam[0] + M[6] * gcam[1] + M[10] * gcam[2];
    }
    if constexpr (N_SH > 1) {
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if (half) __syncthreads();   // the first half has been written out
            if ((int)(threadIdx.x / HALF) == half) {
                float* gsh = s_sh + (threadIdx.x % HALF) * SHW;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
#pragma unroll
                    for (int s = 1; s < N_SH; s++) gsh[(N_SH - 1) * ch + (s - 1)] = gl3[ch] * Y[s];
            }
            __syncthreads();
            const int g0 = blockIdx.x * PP_BLOCK + half * HALF;
            const int count = max(0, min(HALF, N - g0)) * SHW;
            float* dst = o.sh + (size_t)g0 * SHW;   // 16-byte aligned: g0 is a multiple of 128
            // streamed out once (the optimizer reads it much later): non-temporal 16-byte stores
            typedef float vfloat4 __attribute__((ext_vector_type(4)));
            vfloat4* dst4 = reinterpret_cast<vfloat4*>(dst);
            const vfloat4* src4 = reinterpret_cast<const vfloat4*>(s_sh);
            for (int i = threadIdx.x; i < (count >> 2); i += PP_BLOCK) __builtin_nontemporal_store(src4[i], dst4 + i);
            for (int i = (count & ~3) + threadIdx.x; i < count; i += PP_BLOCK) dst[i] = s_sh[i];
        }
    }
    if (!in_range) return;
    o.xyz[g * 3 + 0] = gx[0]; o.xyz[g * 3 + 1] = gx[1]; o.xyz[g * 3 + 2] = gx[2];
    o.quaternion[g * 4 + 0] = gq[0]; o.quaternion[g * 4 + 1] = gq[1];
    o.quaternion[g * 4 + 2] = gq[2]; o.quaternion[g * 4 + 3] = gq[3];
    o.scale[g * 3 + 0] = gs[0]; o.scale[g * 3 + 1] = gs[1]; o.scale[g * 3 + 2] = gs[2];
    o.opacity[g] = go;
    o.rgb[g * 3 + 0] = gc[0]; o.rgb[g * 3 + 1] = gc[1]; o.rgb[g * 3 + 2] = gc[2];
}

}  // namespace gs

using namespace gs;

static Frustum make_frustum(int W, int H, float near, float far, float padding) {
    Frustum fr;
    fr.near = near;
    fr.far = far;
    fr.u_lo = -1.0f * padding;
    fr.u_hi = (float)W + padding;
    fr.v_lo = -1.0f * padding;
    fr.v_hi = (float)H + padding;
    return fr;
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

size_t gs_preprocess_workspace_ints(int N) { return (size_t)div_up(N > 0 ? N : 1, PP_BLOCK) * 2 + 8; }

int gs_preprocess_forward_cut(const void* xyz, const void* quaternion, const void* scale,
                              const void* opacity, const void* rgb, const void* sh, int n_sh,
                              const void* camera_T_world, const void* K, int N, int W, int H,
                              float near_thresh, float far_thresh, float cull_mask_padding,
                              float mh_dist, int band_row0, int band_row1,
                              int32_t* workspace, void* camera_center, int32_t* visible_count,
                              uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv,
                              void* xyz_camera_frame, void* conic, void* opacity_act, void* rgb_render,
                              void* packed, void* bin_records, int32_t* cut_workspace, int32_t* depth_hist,
                              int sample_stride, void* stream) {
    GS_REQUIRE((bin_records == nullptr) == (cut_workspace == nullptr) && (bin_records == nullptr) == (depth_hist == nullptr),
               "bin_records, cut_workspace and depth_hist go together");
    GS_REQUIRE(depth_hist == nullptr || sample_stride >= 1, "sample_stride must be gs_cut_sample_stride(N)");
    GS_REQUIRE(bin_records != nullptr || (uv != nullptr && xyz_camera_frame != nullptr && conic != nullptr),
               "uv, xyz_camera_frame and conic may only be omitted together with bin_records");
    GS_REQUIRE(n_sh == 1 || sh != nullptr, "sh must be given when n_sh > 1");
    hipStream_t s = (hipStream_t)stream;
    const Frustum fr = make_frustum(W, H, near_thresh, far_thresh, cull_mask_padding);
    const int nb = div_up(N > 0 ? N : 1, PP_BLOCK);
    int* block_counts = workspace;
    int* block_offsets = workspace + nb;
    k_cull_count<<<nb, PP_BLOCK, 0, s>>>((const float*)xyz, (const float*)camera_T_world,
                                         (const float*)K, N, fr, block_counts, (float*)camera_center, depth_hist,
                                         sample_stride > 0 ? sample_stride : 1);
    const int n_tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    CutState cs{};
    if (cut_workspace != nullptr) {
        cs = cut_state_of(cut_workspace, N, n_tiles);
        k_scan_counts<true><<<2, 1024, 0, s>>>(block_counts, nb, block_offsets, visible_count, depth_hist, cs.bounds, fr);
    } else {
        k_scan_counts<false><<<1, 1024, 0, s>>>(block_counts, nb, block_offsets, visible_count, nullptr, nullptr, fr);
    }
    PreOut o;
    o.uv = (float*)uv;
    o.conic = (float*)conic;
    o.opacity = (float*)opacity_act;
    o.rgb = (float*)rgb_render;
    o.xyz_cam = (float*)xyz_camera_frame;
    o.packed = (float*)packed;
    o.vis_idx = vis_idx;
    o.rank = rank;
    o.culled = culling_mask;
    o.bin_rec = (float*)bin_records;
    o.bounds = cs.bounds;
    o.bucket_of = cs.bucket_of;
    Band band;
    band.ntx = (W + GS_TILE - 1) / GS_TILE;
    band.nty = (H + GS_TILE - 1) / GS_TILE;
    band.row0 = band_row0;
    band.row1 = band_row1;
    band.mh = mh_dist;
    GS_REQUIRE(band_row0 >= 0 && band_row1 <= band.nty && band_row0 <= band_row1, "bad tile row band");
    const bool banded = !(band_row0 == 0 && band_row1 == band.nty);
    GS_REQUIRE(!(banded && bin_records != nullptr), "the depth-bucketed binning serves whole frames");
    if (banded) {
        DISPATCH_SH(n_sh, (k_preprocess<N_SH, true><<<nb, PP_BLOCK, 0, s>>>(
                              (const float*)xyz, (const float*)quaternion, (const float*)scale,
                              (const float*)opacity, (const float*)rgb, (const float*)sh,
                              (const float*)camera_T_world, (const float*)K,
                              (const float*)camera_center, N, fr, block_offsets, o, band)));
    } else {
        DISPATCH_SH(n_sh, (k_preprocess<N_SH, false><<<nb, PP_BLOCK, 0, s>>>(
                              (const float*)xyz, (const float*)quaternion, (const float*)scale,
                              (const float*)opacity, (const float*)rgb, (const float*)sh,
                              (const float*)camera_T_world, (const float*)K,
                              (const float*)camera_center, N, fr, block_offsets, o, band)));
    }
    return check_launch("preprocess_forward");
}


int gs_band_project(const void* xyz, const void* scale, const void* opacity, const void* camera_T_world, const void* K,
                    int N, int W, int H, float near_thresh, float far_thresh, float cull_mask_padding, float mh_dist,
                    const int32_t* band_rows, int G, int32_t* workspace, void* camera_center, int32_t* visible_count,
                    uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv, void* opacity_act, uint32_t* mask,
                    int32_t* halo_workspace, void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "band_project: 1 <= G <= %d", GS_MAX_RANKS);
    hipStream_t s = (hipStream_t)stream;
    const Frustum fr = make_frustum(W, H, near_thresh, far_thresh, cull_mask_padding);
    const int nb = div_up(N > 0 ? N : 1, PP_BLOCK);
    int* block_counts = workspace;
    int* block_offsets = workspace + nb;
    k_cull_count<<<nb, PP_BLOCK, 0, s>>>((const float*)xyz, (const float*)camera_T_world, (const float*)K, N, fr,
                                         block_counts, (float*)camera_center, nullptr, 1);
    k_scan_counts<false><<<1, 1024, 0, s>>>(block_counts, nb, block_offsets, visible_count, nullptr, nullptr, fr);
    BandRows rows;
    for (int i = 0; i <= GS_MAX_RANKS; i++) rows.v[i] = i <= G ? band_rows[i] : 0;
    const int nty = (H + GS_TILE - 1) / GS_TILE;
    k_band_project<1><<<nb, PP_BLOCK, 0, s>>>((const float*)xyz, (const float*)scale, (const float*)opacity,
                                              (const float*)camera_T_world, (const float*)K, N, fr, mh_dist, nty, rows, G,
                                              block_offsets, culling_mask, rank, vis_idx, (float*)uv, (float*)opacity_act,
                                              mask, nullptr, nb);
    // per-block bit counts by visible index: the first block of gs_halo_workspace_ints' layout
    k_band_bit_counts<<<nb, PP_BLOCK, 0, s>>>(mask, visible_count, G, halo_workspace, nb);
    return check_launch("band_project");
}

int gs_preprocess_forward(const void* xyz, const void* quaternion, const void* scale,
                          const void* opacity, const void* rgb, const void* sh, int n_sh,
                          const void* camera_T_world, const void* K, int N, int W, int H,
                          float near_thresh, float far_thresh, float cull_mask_padding,
                          float mh_dist, int band_row0, int band_row1,
                          int32_t* workspace, void* camera_center, int32_t* visible_count,
                          uint8_t* culling_mask, int32_t* rank, int32_t* vis_idx, void* uv,
                          void* xyz_camera_frame, void* conic, void* opacity_act, void* rgb_render,
                          void* packed, void* stream) {
    return gs_preprocess_forward_cut(xyz, quaternion, scale, opacity, rgb, sh, n_sh, camera_T_world, K, N, W, H, near_thresh,
                                     far_thresh, cull_mask_padding, mh_dist, band_row0, band_row1, workspace, camera_center,
                                     visible_count, culling_mask, rank, vis_idx, uv, xyz_camera_frame, conic, opacity_act,
                                     rgb_render, packed, nullptr, nullptr, nullptr, 0, stream);
}

int gs_preprocess_forward_list(const void* xyz, const void* quaternion, const void* scale, const void* rgb, const void* sh,
                               int n_sh, const void* camera_T_world, const void* K, const void* camera_center,
                               const int32_t* list, const int32_t* list_count, int capacity, const int32_t* vis_idx,
                               const void* uv, const void* opacity_act, void* uv_l, void* xyz_camera_frame_l, void* conic_l,
                               void* packed_l, void* stream) {
    GS_REQUIRE(n_sh == 1 || sh != nullptr, "sh must be given when n_sh > 1");
    if (capacity <= 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_SH(n_sh, (k_preprocess_list<N_SH><<<div_up(capacity, PP_BLOCK), PP_BLOCK, 0, s>>>(
                          (const float*)xyz, (const float*)quaternion, (const float*)scale, (const float*)rgb,
                          (const float*)sh, (const float*)camera_T_world, (const float*)K, (const float*)camera_center, list,
                          list_count, vis_idx, (const float*)uv, (const float*)opacity_act, (float*)uv_l,
                          (float*)xyz_camera_frame_l, (float*)conic_l, (float*)packed_l)));
    return check_launch("preprocess_forward_list");
}

// ---- the fused band frontend (ABI 8) ----
namespace {
struct FrontendWs {
    int nb, nbp;
    int32_t *counts, *offsets;
    uint16_t* gmask;
};
FrontendWs frontend_ws(int32_t* workspace, int N, int G) {
    FrontendWs w;
    w.nb = div_up(N > 0 ? N : 1, PP_BLOCK);
    w.nbp = w.nb + 1;
    w.counts = workspace;
    w.offsets = workspace + (size_t)(G + 1) * w.nbp;
    w.gmask = reinterpret_cast<uint16_t*>(workspace + 2 * (size_t)(G + 1) * w.nbp);
    return w;
}
GatherPlan gather_plan(const FrontendWs& w, int G, int rank, const int32_t* owner_blocks, const void* recv,
                       const int32_t* recv_offsets) {
    GatherPlan gp;
    gp.gmask = w.gmask;
    gp.offsets = w.offsets;
    gp.recv = (const float*)recv;
    gp.nbp = w.nbp;
    gp.G = G;
    gp.blk0 = owner_blocks[rank] < w.nb ? owner_blocks[rank] : w.nb;
    for (int i = 0; i < GS_MAX_RANKS; i++) gp.recv_off[i] = i < G ? recv_offsets[i] : 0;
    return gp;
}
}  // namespace

size_t gs_band_frontend_workspace_ints(int N, int G) {
    const size_t nbp = (size_t)div_up(N > 0 ? N : 1, PP_BLOCK) + 1;
    return 2 * (size_t)(G + 1) * nbp + ((size_t)(N > 0 ? N : 1) + 1) / 2;
}

int gs_band_frontend(const void* xyz, const void* quaternion, const void* scale, const void* opacity, const void* rgb,
                     const void* sh, int n_sh, const void* camera_T_world, const void* K, int N, int W, int H,
                     float near_thresh, float far_thresh, float cull_mask_padding, float mh_dist, const int32_t* band_rows,
                     const int32_t* owner_blocks, int G, int rank, int32_t* workspace, void* camera_center,
                     uint8_t* culling_mask, int32_t* rank_out, void* uv, void* opacity_act, int32_t* send_index,
                     int32_t* list_g, void* uv_l, void* xyz_camera_frame_l, void* conic_l, void* packed_l, int32_t* plan,
                     int32_t* plan_host, void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "band_frontend: 1 <= G <= %d", GS_MAX_RANKS);
    GS_REQUIRE(rank >= 0 && rank < G, "band_frontend: bad rank");
    GS_REQUIRE(N > 0, "band_frontend: N must be positive");
    GS_REQUIRE(n_sh == 1 || sh != nullptr, "sh must be given when n_sh > 1");
    hipStream_t s = (hipStream_t)stream;
    const Frustum fr = make_frustum(W, H, near_thresh, far_thresh, cull_mask_padding);
    const FrontendWs w = frontend_ws(workspace, N, G);
    BandRows rows;
    BandOwners owners;
    for (int i = 0; i <= GS_MAX_RANKS; i++) {
        rows.v[i] = i <= G ? band_rows[i] : 0;
        owners.v[i] = i <= G ? owner_blocks[i] : 0;
        GS_REQUIRE(i == 0 || i > G || owner_blocks[i] >= owner_blocks[i - 1], "band_frontend: owner_blocks must ascend");
    }
    const int nty = (H + GS_TILE - 1) / GS_TILE;
    k_band_count<<<w.nb, PP_BLOCK, 0, s>>>((const float*)xyz, (const float*)scale, (const float*)camera_T_world,
                                           (const float*)K, N, fr, mh_dist, nty, rows, G, w.counts, w.nbp, w.gmask,
                                           (float*)camera_center);
    k_band_scan<<<G + 1, 1024, 0, s>>>(w.counts, w.nb, w.nbp, w.offsets);
    k_band_write<<<w.nb + 1, PP_BLOCK, 0, s>>>((const float*)xyz, (const float*)opacity, (const float*)camera_T_world,
                                               (const float*)K, N, fr, w.gmask, w.offsets, w.nb, w.nbp, G, rank, owners,
                                               culling_mask, rank_out, (float*)uv, (float*)opacity_act, send_index, list_g,
                                               (float*)uv_l, (float*)xyz_camera_frame_l, (float*)conic_l, (float*)packed_l,
                                               plan, plan_host);
    // rows of my list: offsets[1 + rank][nb] (device side); the grid covers the capacity, the surplus exits at once
    const int32_t* list_count = w.offsets + (size_t)(1 + rank) * w.nbp + w.nb;
    DISPATCH_SH(n_sh, (k_band_rows<N_SH><<<w.nb, PP_BLOCK, 0, s>>>(
                          (const float*)quaternion, (const float*)scale, (const float*)rgb, (const float*)sh,
                          (const float*)camera_T_world, (const float*)K, (const float*)camera_center, list_g, list_count,
                          (const float*)uv_l, (const float*)xyz_camera_frame_l, (float*)conic_l, (float*)packed_l)));
    return check_launch("band_frontend");
}

int gs_band_gather_sum(const int32_t* workspace, int N, int G, int rank, const int32_t* owner_blocks,
                       const int32_t* rank_of_gaussian, int v_lo, const void* recv, const int32_t* recv_offsets, void* out,
                       void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS && rank >= 0 && rank < G, "band_gather_sum: bad G / rank");
    const FrontendWs w = frontend_ws(const_cast<int32_t*>(workspace), N, G);
    const GatherPlan gp = gather_plan(w, G, rank, owner_blocks, recv, recv_offsets);
    const int blk1 = owner_blocks[rank + 1] < w.nb ? owner_blocks[rank + 1] : w.nb;
    if (blk1 <= gp.blk0) return GS_OK;
    k_band_gather_sum<<<blk1 - gp.blk0, PP_BLOCK, 0, (hipStream_t)stream>>>(gp, N, rank_of_gaussian, v_lo, (float*)out);
    return check_launch("band_gather_sum");
}

int gs_preprocess_backward_gathered(const void* xyz, const void* quaternion, const void* scale, int n_sh,
                                    const void* camera_T_world, const void* K, const void* camera_center,
                                    const int32_t* rank_of_gaussian, const void* opacity_act, const int32_t* workspace,
                                    int N_total, int G, int rank, const int32_t* owner_blocks, const void* recv,
                                    const int32_t* recv_offsets, int n, void* grad_xyz, void* grad_quaternion,
                                    void* grad_scale, void* grad_opacity_logit, void* grad_rgb_param, void* grad_sh,
                                    void* stream) {
    GS_REQUIRE(n_sh == 1 || grad_sh != nullptr, "grad_sh must be given when n_sh > 1");
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS && rank >= 0 && rank < G, "preprocess_backward_gathered: bad G / rank");
    if (n <= 0) return GS_OK;
    const FrontendWs w = frontend_ws(const_cast<int32_t*>(workspace), N_total, G);
    const GatherPlan gp = gather_plan(w, G, rank, owner_blocks, recv, recv_offsets);
    GS_REQUIRE((size_t)gp.blk0 * PP_BLOCK + (size_t)n <= (size_t)N_total, "preprocess_backward_gathered: slice beyond N");
    PreGrad o;
    o.xyz = (float*)grad_xyz;
    o.quaternion = (float*)grad_quaternion;
    o.scale = (float*)grad_scale;
    o.opacity = (float*)grad_opacity_logit;
    o.rgb = (float*)grad_rgb_param;
    o.sh = (float*)grad_sh;
    DISPATCH_SH(n_sh, (k_preprocess_bwd<N_SH, true><<<div_up(n, PP_BLOCK), PP_BLOCK, 0, (hipStream_t)stream>>>(
                          (const float*)xyz, (const float*)quaternion, (const float*)scale, (const float*)camera_T_world,
                          (const float*)K, (const float*)camera_center, rank_of_gaussian, (const float*)opacity_act, nullptr,
                          0, n, o, gp)));
    return check_launch("preprocess_backward_gathered");
}

int gs_preprocess_backward(const void* xyz, const void* quaternion, const void* scale, int n_sh,
                           const void* camera_T_world, const void* K, const void* camera_center,
                           const int32_t* rank, const void* opacity_act, const void* grad_slab,
                           int v_base, int N, void* grad_xyz, void* grad_quaternion, void* grad_scale,
                           void* grad_opacity_logit, void* grad_rgb_param, void* grad_sh,
                           void* stream) {
    GS_REQUIRE(n_sh == 1 || grad_sh != nullptr, "grad_sh must be given when n_sh > 1");
    hipStream_t s = (hipStream_t)stream;
    if (N <= 0) return GS_OK;
    PreGrad o;
    o.xyz = (float*)grad_xyz;
    o.quaternion = (float*)grad_quaternion;
    o.scale = (float*)grad_scale;
    o.opacity = (float*)grad_opacity_logit;
    o.rgb = (float*)grad_rgb_param;
    o.sh = (float*)grad_sh;
    DISPATCH_SH(n_sh, (k_preprocess_bwd<N_SH><<<div_up(N, PP_BLOCK), PP_BLOCK, 0, s>>>(
                          (const float*)xyz, (const float*)quaternion, (const float*)scale,
                          (const float*)camera_T_world, (const float*)K,
                          (const float*)camera_center, rank, (const float*)opacity_act,
                          (const float*)grad_slab, v_base, N, o)));
    return check_launch("preprocess_backward");
}

}  // extern "C"
