// binning.hip -- Gaussian -> tile assignment and per-tile depth ordering for gfx950.
//
// Replaces get_sorted_gaussian_list (tile_culling.cu:124-340).  The reference emits fp64 keys
// z + (max_z+1)*tile in Gaussian order and runs one global torch::sort over all instances.  Here
// the count pass already yields every tile's segment [tile_ranges[t], tile_ranges[t+1]), so
//   1. k_bin_count     SAT test per (Gaussian, candidate tile).  Each of NB workgroups owns a
//                      contiguous slice of the Gaussians and counts its hits per tile in an LDS
//                      histogram (ds_add, no global atomics: device-scope atomics resolve at the
//                      memory side at ~14 G/s chip-wide, which bounded the first version), then
//                      stores the histogram as row b of hist[NB][T]
//   2. k_bin_colscan   per tile: exclusive prefix over the NB workgroups (hist becomes each
//                      workgroup's start offset inside the tile's segment) and the tile total
//   3. k_scan_tiles    exclusive prefix of the tile totals (one workgroup) -> tile_ranges
//   4. k_bin_emit      same slices, same test; LDS cursors start at ranges[t] + hist[b][t];
//                      scatters key = (sortable z bits << 32 | gaussian) into the tile's segment
//                      (a counting sort on the tile digit)
//   5. k_tile_sort*    one workgroup per tile orders its segment on unique 64-bit keys
//                      (deterministic) and writes the Gaussian indices: wave-level sorts up to
//                      1024 entries, and above that either the register-blocked bitonic network
//                      in LDS (full mode, what get_sorted_gaussian_list returns) or a radix select
//                      + sort of the 1024 nearest entries (prefix mode of the fused renderer, with
//                      a flag-and-repair pass that keeps it exact); see "per-tile sort",
//                      "wave-level sorts" and "prefix sort" below.
// The Gaussians walked can be restricted to an index list (multi-GPU band mode, struct Items).
// Tile grids too large for an LDS histogram (T > 16384) fall back to global-atomic counters
// (k_tile_count / k_tile_emit).
// HBM traffic: 20 B in per Gaussian per pass, 8 B out + 8 B in + 4 B out per instance,
// + 2 * NB * T * 4 B for the histogram matrix.
//
// The intersection test is bit-identical to the CPU restatement: fp32, no contraction, the same
// operation order as tile_culling.cu:8-122, cos/sin of the OBB angle formed algebraically from
// the eigenvector (b, lambda1 - a) instead of atan2f/cosf/sinf (whose last bits differ between
// libdevice, OCML and glibc).
#include "tile_math.h"
#include "tile_sort.h"

#include <atomic>

namespace gs {

constexpr int BIN_BLOCK = 256;

// per-Gaussian part of the separating-axis test (tile_culling.cu:14-15,20-21,27-28,38-41,47-48,
// 58-62): everything that does not depend on the tile
struct Sat {
    float mnx, mxx, mny, mxy;
    float ax[2], ay[2], mn_o[2], mx_o[2];
    bool separable;   // the per-candidate test may take the separable form (sat_separable below)
};

__device__ inline Sat sat_setup(const Obb& o) {
    Sat s;
    const float* p = o.p;
    s.mnx = fminf(fminf(p[0], p[2]), fminf(p[4], p[6]));
    s.mxx = fmaxf(fmaxf(p[0], p[2]), fmaxf(p[4], p[6]));
    s.mny = fminf(fminf(p[1], p[3]), fminf(p[5], p[7]));
    s.mxy = fmaxf(fmaxf(p[1], p[3]), fmaxf(p[5], p[7]));
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int q = k == 0 ? 0 : 6;
        s.ax[k] = p[2] - p[q];
        s.ay[k] = p[3] - p[q + 1];
        const float p0 = s.ax[k] * p[2] + s.ay[k] * p[3];
        const float p1 = s.ax[k] * p[q] + s.ay[k] * p[q + 1];
        s.mn_o[k] = fminf(p0, p1);
        s.mx_o[k] = fmaxf(p0, p1);
    }
    return s;
}

// tile_culling.cu:8-66 for tile bounds [l, r] x [t, b]
__device__ inline bool sat_overlaps(const Sat& s, float l, float r, float t, float b) {
    if (s.mnx > r || s.mxx < l) return false;
    if (s.mny > b || s.mxy < t) return false;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float tl = s.ax[k] * l + s.ay[k] * t;
        const float tr = s.ax[k] * r + s.ay[k] * t;
        const float bl = s.ax[k] * l + s.ay[k] * b;
        const float br = s.ax[k] * r + s.ay[k] * b;
        const float mn_t = fminf(fminf(tl, tr), fminf(bl, br));
        const float mx_t = fmaxf(fmaxf(tl, tr), fmaxf(bl, br));
        if (mn_t > s.mx_o[k] || mx_t < s.mn_o[k]) return false;
    }
    return true;
}

// The separable form of the two OBB-axis tests.  With tile corners (l|r, t|b) the reference projects all four onto
// the axis, ax*x + ay*y, and takes their least and greatest.  Rounding is monotone: fl(ax*x) is least at the same x
// for every y, and fl(p + q) does not decrease when p or q grows -- so, as long as no product is NaN or infinite,
//     min over corners fl(fl(ax*x) + fl(ay*y))  ==  fl( min(fl(ax*l), fl(ax*r)) + min(fl(ay*t), fl(ay*b)) )
// bit for bit, and likewise the greatest: 4 products and 2 sums per axis instead of 8 and 4, the x half the same for
// a whole tile column, and a tile's bottom products its lower neighbour's top products.  sat_separable() says when
// that holds: axes and extents finite and far from overflow (tile coordinates are < 2^24, so |axis| < 1e30 keeps
// every product and sum finite).  Such a Gaussian's candidate window has also been shrunk to the tiles that pass the
// x and y tests (tile_walk_setup_vals), so those two are not evaluated again; the extents must not be denormal for
// that (the shrink divides them by 16).  Any other Gaussian takes sat_overlaps, the reference's form.
__device__ inline bool sat_separable(const Sat& s) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 2; k++) ok = ok && __builtin_fabsf(s.ax[k]) < 1.0e30f && __builtin_fabsf(s.ay[k]) < 1.0e30f;
    const float e[4] = {s.mnx, s.mxx, s.mny, s.mxy};
#pragma unroll
    for (int k = 0; k < 4; k++) ok = ok && __builtin_fabsf(e[k]) < 1.0e30f && (e[k] == 0.0f || __builtin_fabsf(e[k]) > 1.0e-30f);
    return ok;   // (a NaN fails every comparison above)
}
// the x half of both axes for the tile column [l, r]
struct SatColumn {
    float mn[2], mx[2];
};
__device__ inline SatColumn sat_column(const Sat& s, float l, float r) {
    SatColumn c;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float pl = s.ax[k] * l, pr = s.ax[k] * r;
        c.mn[k] = fminf(pl, pr);
        c.mx[k] = fmaxf(pl, pr);
    }
    return c;
}
// yt[k] = ay[k] * t, yb[k] = ay[k] * b
__device__ inline bool sat_overlaps_separable(const Sat& s, const SatColumn& c, const float* yt, const float* yb) {
    // (all four comparisons, then one decision: ~20 instructions straight through cost less than four exec-mask
    // branches that most candidates leave at different points)
    int rejected = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float mn_t = c.mn[k] + fminf(yt[k], yb[k]);
        const float mx_t = c.mx[k] + fmaxf(yt[k], yb[k]);
        rejected |= (int)(mn_t > s.mx_o[k]) | (int)(mx_t < s.mn_o[k]);
    }
    return rejected == 0;
}

// one candidate tile (the walks that hand a lane one tile at a time)
__device__ inline bool sat_overlaps_tile(const Sat& s, int tx, int ty) {
    const float l = (float)tx * 16.0f, r = (float)(tx + 1) * 16.0f, t = (float)ty * 16.0f, b = (float)(ty + 1) * 16.0f;
    if (!s.separable) return sat_overlaps(s, l, r, t, b);
    const float yt[2] = {s.ay[0] * t, s.ay[1] * t}, yb[2] = {s.ay[0] * b, s.ay[1] * b};
    return sat_overlaps_separable(s, sat_column(s, l, r), yt, yb);
}

// Per-Gaussian part of the tile walk: OBB, separating-axis constants and the candidate window
// (empty window: w.sx >= w.ex).
struct TileWalk {
    Sat s;
    Window w;
};

__device__ inline TileWalk tile_walk_setup_vals(float u, float v, float conic0, float conic1, float conic2,
                                                int ntx, int nty, float mh, int row0, int row1) {
    TileWalk tw;
    const float a = conic0 + 0.25f;
    const float b = conic1 / 2.0f;
    const float c = conic2 + 0.25f;
    const Obb o = compute_obb(u, v, a, b, c, mh);
    Window w = candidate_window(u, v, o.radius_tiles, ntx, nty, row0, row1);
    tw.s = sat_setup(o);
    const Sat& s = tw.s;
    // The first two axes of the test (tile_culling.cu:14-25) are solved for the tile index instead
    // of being evaluated per candidate: tile tx passes  mnx <= 16 (tx+1)  and  mxx >= 16 tx  iff
    // ceil(mnx/16) - 1 <= tx <= floor(mxx/16)  (16 tx is exact in fp32, /16 is exact), likewise in y.
    // The window shrinks from the reference's (2r)^2 square to the OBB's bounding box; the set of
    // accepted tiles is unchanged.  Non-finite extents keep the full window (NaN compares pass).
    if (w.sx < w.ex && w.sy < w.ey && s.mnx == s.mnx && s.mxx == s.mxx && s.mny == s.mny && s.mxy == s.mxy) {
        // (clamped in float before the conversion: +-inf extents must not overflow the int math;
        // tile indices are < 2^20, where float arithmetic on integers is exact)
        const float big = 1.0e9f;
        w.sx = max(w.sx, f2i(fminf(fmaxf(__builtin_ceilf(s.mnx / 16.0f) - 1.0f, -big), big)));
        w.ex = min(w.ex, f2i(fminf(fmaxf(__builtin_floorf(s.mxx / 16.0f) + 1.0f, -big), big)));
        w.sy = max(w.sy, f2i(fminf(fmaxf(__builtin_ceilf(s.mny / 16.0f) - 1.0f, -big), big)));
        w.ey = min(w.ey, f2i(fminf(fmaxf(__builtin_floorf(s.mxy / 16.0f) + 1.0f, -big), big)));
    }
    if (w.sx >= w.ex || w.sy >= w.ey) w.sx = w.ex = w.sy = w.ey = 0;
    tw.s.separable = sat_separable(s);
    tw.w = w;
    return tw;
}

__device__ inline TileWalk tile_walk_setup(const float* __restrict__ uvs, const float* __restrict__ conic, int g,
                                           int ntx, int nty, float mh, int row0, int row1) {
    return tile_walk_setup_vals(uvs[g * 2], uvs[g * 2 + 1], conic[g * 3], conic[g * 3 + 1], conic[g * 3 + 2], ntx, nty, mh,
                                row0, row1);
}

// the 32-byte binning record of the depth-bucketed path (k_preprocess: u v conic0 conic1 | conic2 z - -): one
// sector per Gaussian when the records are gathered through a bucket's index list
struct BinRec {
    float4 a, b;
};
__device__ inline BinRec load_bin_record(const float* __restrict__ rec, int g) {
    const float4* p = reinterpret_cast<const float4*>(rec) + 2 * (size_t)g;
    BinRec r;
    r.a = p[0];
    r.b = p[1];
    return r;
}
__device__ inline TileWalk tile_walk_setup(const BinRec& r, int ntx, int nty, float mh, int row0, int row1) {
    return tile_walk_setup_vals(r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, ntx, nty, mh, row0, row1);
}

__device__ inline float lane_bcast(float x, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
}
__device__ inline int lane_bcast(int x, int src) { return __builtin_amdgcn_readlane(x, src); }

// The tile walk of one Gaussian per lane, for a whole wave (every lane of the wave must call it;
// `active` = the lane holds a Gaussian): emit(tile, payload) for every tile the Gaussian's OBB overlaps.
// A window of up to COOP_MIN candidate tiles is walked by the Gaussian's own lane; a larger one is walked
// by all 64 lanes together, 64 candidate tiles per step, after broadcasting the Gaussian's constants -- a
// lane looping alone over the thousands of tiles of a screen-filling Gaussian stalled its whole workgroup
// (training scenes have such Gaussians: binning went from 0.2 to 3 ms on them).  The accepted set and
// the payloads are unchanged; only the order of the emit calls differs (the per-tile sort fixes order).
// LPG > 1 (a power of two): LPG consecutive lanes hold the SAME Gaussian (sub = its lane's index among them) and
// share a small window's candidates round-robin -- for frames with few Gaussians, where one lane per Gaussian leaves
// the chip a wave or two per SIMD, each a serial chain of tests and atomics (workload B: 90 k Gaussians).
// want(tile): evaluated BEFORE the separating-axis test of a candidate (depth-bucketed path: a tile whose list is
// already cut in front of this workgroup's depth bucket costs one LDS read instead of the test).
constexpr int COOP_MIN = 48;
struct AnyTile {
    __device__ bool operator()(int) const { return true; }
};
template <int LPG = 1, typename F, typename P = AnyTile>
__device__ inline void wave_for_each_tile(bool active, const TileWalk& tw, int ntx, uint64_t payload, F emit, int sub = 0,
                                          P want = P()) {
    const Window& w = tw.w;
    const int area = active ? (w.ex - w.sx) * (w.ey - w.sy) : 0;
    if (area > 0 && area <= COOP_MIN) {
        if constexpr (LPG == 1) {
            if (tw.s.separable) {
                for (int tx = w.sx; tx < w.ex; tx++) {
                    const SatColumn col = sat_column(tw.s, (float)tx * 16.0f, (float)(tx + 1) * 16.0f);
                    // (the top products are formed again rather than carried over from the tile above: min / max of a
                    // loop-carried value first pay a canonicalising v_max each, twice the cost of the multiplication)
                    float t = (float)w.sy * 16.0f;
                    for (int ty = w.sy; ty < w.ey; ty++) {
                        const float b2 = t + 16.0f;   // exact: tile coordinates are integers < 2^24
                        const float yt[2] = {tw.s.ay[0] * t, tw.s.ay[1] * t};
                        const float yb[2] = {tw.s.ay[0] * b2, tw.s.ay[1] * b2};
                        if (want(ty * ntx + tx) && sat_overlaps_separable(tw.s, col, yt, yb)) emit(ty * ntx + tx, payload);
                        t = b2;
                    }
                }
            } else {
                for (int tx = w.sx; tx < w.ex; tx++) {
                    const float l = (float)tx * 16.0f, r = (float)(tx + 1) * 16.0f;
                    for (int ty = w.sy; ty < w.ey; ty++) {
                        const float t = (float)ty * 16.0f, b2 = (float)(ty + 1) * 16.0f;
                        if (want(ty * ntx + tx) && sat_overlaps(tw.s, l, r, t, b2)) emit(ty * ntx + tx, payload);
                    }
                }
            }
        } else {
            const int h = w.ey - w.sy;
            for (int c = sub; c < area; c += LPG) {
                const int cx = c / h;
                const int tx = w.sx + cx, ty = w.sy + (c - cx * h);
                if (want(ty * ntx + tx) && sat_overlaps_tile(tw.s, tx, ty)) emit(ty * ntx + tx, payload);
            }
        }
    }
    unsigned long long big = __builtin_amdgcn_ballot_w64(area > COOP_MIN && sub == 0);
    const int lane = threadIdx.x & 63;
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1;
        Sat s;
        s.mnx = lane_bcast(tw.s.mnx, src); s.mxx = lane_bcast(tw.s.mxx, src);
        s.mny = lane_bcast(tw.s.mny, src); s.mxy = lane_bcast(tw.s.mxy, src);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            s.ax[k] = lane_bcast(tw.s.ax[k], src); s.ay[k] = lane_bcast(tw.s.ay[k], src);
            s.mn_o[k] = lane_bcast(tw.s.mn_o[k], src); s.mx_o[k] = lane_bcast(tw.s.mx_o[k], src);
        }
        s.separable = lane_bcast((int)tw.s.separable, src) != 0;
        const int sx = lane_bcast(w.sx, src), ex = lane_bcast(w.ex, src), sy = lane_bcast(w.sy, src),
                  ey = lane_bcast(w.ey, src);
        const uint64_t pl = ((uint64_t)(uint32_t)lane_bcast((int)(payload >> 32), src) << 32) |
                            (uint32_t)lane_bcast((int)(uint32_t)payload, src);
        const int h = ey - sy, n = (ex - sx) * h;
        for (int t = lane; t < n; t += 64) {
            const int cx = t / h;
            const int tx = sx + cx, ty = sy + (t - cx * h);
            if (want(ty * ntx + tx) && sat_overlaps_tile(s, tx, ty)) emit(ty * ntx + tx, pl);
        }
    }
}

// Which Gaussians a binning kernel walks: all V rows, the first *v_dev of them (fill level known
// only on the device), or the *subset_n entries of an index list (multi-GPU: the Gaussians whose
// candidate window reaches the rank's band; any list that contains every Gaussian with a tile in
// the rows [row0, row1) gives the same tile lists).
struct Items {
    const int* v_dev;
    const int* subset;
    const int* subset_n;
};
__device__ inline int item_count(const Items& it, int V) {
    return it.subset ? *it.subset_n : (it.v_dev ? *it.v_dev : V);
}
__device__ inline int item_at(const Items& it, int i) { return it.subset ? it.subset[i] : i; }

template <int LPG>
__global__ __launch_bounds__(BIN_BLOCK) void k_tile_count(const float* __restrict__ uvs,
                                                          const float* __restrict__ conic, int V,
                                                          int ntx, int nty, float mh, int row0,
                                                          int row1, int* __restrict__ counts,
                                                          Items items) {
    const int t = blockIdx.x * BIN_BLOCK + threadIdx.x;
    const int i = t / LPG, sub = t % LPG;
    const bool active = i < item_count(items, V);
    TileWalk tw;
    if (active) tw = tile_walk_setup(uvs, conic, item_at(items, i), ntx, nty, mh, row0, row1);
    wave_for_each_tile<LPG>(active, tw, ntx, 0, [&](int tile, uint64_t) { atomicAdd(counts + tile, 1); }, sub);
}

// exclusive prefix of counts[T] -> ranges[T+1]; single workgroup of 1024 threads.  Only counts[t0, t0 + Tb) are
// read (a band's tiles: the others are empty and need not have been written)
// clear != 0: every count read is set to 0 (the atomic-counter path: the array is k_tile_emit's cursor next)
__global__ __launch_bounds__(1024) void k_scan_tiles(int* __restrict__ counts, int T,
                                                     int* __restrict__ ranges,
                                                     const int* __restrict__ v_dev, int t0, int Tb, int clear,
                                                     int* __restrict__ host_mirror) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    __shared__ int s_longest;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_carry = 0;
        s_longest = 0;
    }
    __syncthreads();
    int longest = 0;
    for (int base = 0; base < T; base += 1024 * 4) {
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            v[k] = (i >= t0 && i < t0 + Tb) ? counts[i] : 0;
            if (clear && i >= t0 && i < t0 + Tb) counts[i] = 0;
            sum += v[k];
            longest = max(longest, v[k]);
        }
        int incl = sum;   // inclusive wave scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int n = __shfl_up(incl, d);
            if (lane >= d) incl += n;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int w = 0; w < wave; w++) wave_off += s_wave[w];
        int run = s_carry + wave_off + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            if (i < T) ranges[i] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (host_mirror != nullptr) {   // (workgroup-uniform)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) longest = max(longest, __shfl_xor(longest, d));
        if (lane == 0) atomicMax(&s_longest, longest);
        __syncthreads();
    }
    if (tid == 0) {
        ranges[T] = s_carry;
        const int V = v_dev ? *v_dev : 0;
        if (v_dev) ranges[T + 1] = V;
        // the frame's host read without a copy in the stream: (S, V, longest list) straight into the caller's pinned
        // buffer (a device -> host hipMemcpyAsync of 8 bytes is a ~10 us blit kernel the next launch queues behind)
        if (host_mirror != nullptr) {
            host_mirror[0] = s_carry;
            host_mirror[1] = V;
            host_mirror[2] = s_longest;
            __threadfence_system();
        }
    }
}

// ---- privatized (LDS histogram) count / emit -----------------------------------------------------
constexpr int PRIV_BLOCK = 512;
constexpr int PRIV_MAX_TILES = 16384;   // 64 KiB of LDS
constexpr int PRIV_NB = 1024;           // workgroups = slices of the Gaussian list

// slice of workgroup b.  A tile's segment is filled slice by slice (hist row = slice) and a slice's run in
// it is a few keys long.  Giving the workgroups of one XCD (b % 8) consecutive slices, so that runs sharing a cache
// line go through the same L2, measured 0.293 -> 0.287 ms for the emit at workload D (and runs of 8 depth buckets per
// XCD in the depth-bucketed emit 87.8 -> 87.2 us, round 6); fewer, larger slices (512 x 1024 threads, 256 x 1024) are
// no faster either: the 2.6x write amplification of the scattered 8-byte keys is not what bounds the emit.  The A/B
// branches are kept as scripts/experiments/binning_macro_experiments.patch.
__device__ inline int slice_index(int b) { return b; }
__device__ inline void slice_of(int sl, int V, int& g0, int& g1) {
    const int chunk = (V + PRIV_NB - 1) / PRIV_NB;
    g0 = min(V, sl * chunk);
    g1 = min(V, g0 + chunk);
}

// (t0, Tb): the tiles of the rows [row0, row1) -- the histogram covers only those (a multi-GPU rank's band is an
// eighth of the grid: 2.4 MB of histogram matrix instead of 17.8 at workload D)
__global__ __launch_bounds__(PRIV_BLOCK) void k_bin_count(const float* __restrict__ uvs,
                                                          const float* __restrict__ conic, int V,
                                                          int ntx, int nty, float mh, int row0,
                                                          int row1, int* __restrict__ hist,
                                                          Items items) {
    extern __shared__ int s_hist[];
    const int t0 = row0 * ntx, Tb = (row1 - row0) * ntx;
    for (int t = threadIdx.x; t < Tb; t += PRIV_BLOCK) s_hist[t] = 0;
    __syncthreads();
    int g0, g1;
    const int sl = slice_index(blockIdx.x);
    slice_of(sl, item_count(items, V), g0, g1);
    for (int base = g0; base < g1; base += PRIV_BLOCK) {   // wave-uniform trip count
        const int i = base + threadIdx.x;
        const bool active = i < g1;
        TileWalk tw;
        if (active) tw = tile_walk_setup(uvs, conic, item_at(items, i), ntx, nty, mh, row0, row1);
        wave_for_each_tile(active, tw, ntx, 0, [&](int tile, uint64_t) { atomicAdd(&s_hist[tile - t0], 1); });
    }
    __syncthreads();
    int* row = hist + (size_t)sl * Tb;
    for (int t = threadIdx.x; t < Tb; t += PRIV_BLOCK) row[t] = s_hist[t];
}

// hist[b][t] <- sum_{b' < b} hist[b'][t];  counts[t] <- column total.
// 1024-thread workgroup = 64 tiles x 16 row segments of 64 rows: every thread loads and sums its segment
// (coalesced: a wave reads 64 consecutive tiles of one row), the 16 segment sums of a tile are
// prefixed through LDS, then the thread writes the exclusive offsets of its segment.
constexpr int GS_CS_TILES = 64;   // 68 workgroups at workload D; 32 or 16 tiles per workgroup (more, narrower ones) are no faster
// CS_TILES tiles x (1024 / CS_TILES) row segments per workgroup.  The narrow form (16 tiles) is for a multi-GPU
// rank's band: ~570 tiles are 9 workgroups of 64 tiles -- 12 us of a 0.5 ms frame -- but 36 of 16
template <int CS_TILES>
__global__ __launch_bounds__(1024) void k_bin_colscan(int* __restrict__ hist, int T, int* __restrict__ counts) {
    constexpr int CS_SEGS = 1024 / CS_TILES;
    constexpr int CS_ROWS = PRIV_NB / CS_SEGS;
    __shared__ int s_seg[CS_SEGS][CS_TILES];
    const int lt = threadIdx.x & (CS_TILES - 1);
    const int seg = threadIdx.x / CS_TILES;
    const int t = blockIdx.x * CS_TILES + lt;
    int* col = hist + (size_t)seg * CS_ROWS * T + t;
    // the segment's values stay in registers between the two walks (CS_ROWS <= 64 of them): the matrix is read once
    int v[CS_ROWS], sum = 0;
    if (t < T) {
#pragma unroll
        for (int r = 0; r < CS_ROWS; r++) {
            v[r] = col[(size_t)r * T];
            sum += v[r];
        }
    }
    s_seg[seg][lt] = sum;
    __syncthreads();
    int run = 0;
    for (int s2 = 0; s2 < seg; s2++) run += s_seg[s2][lt];
    if (t < T) {
        if (seg == CS_SEGS - 1) counts[t] = run + sum;
#pragma unroll
        for (int r = 0; r < CS_ROWS; r++) {
            col[(size_t)r * T] = run;
            run += v[r];
        }
    }
}

__device__ inline uint32_t sortable_bits(float z) {   // monotone float -> uint map
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// cursor[T] must be zero on entry and is zero again on exit: the hit that takes a tile's last slot resets its cursor
// (gs_tile_count leaves the zeros: k_scan_tiles clears each count it has read), so a repeated emit of the same
// frame -- a capacity miss -- starts clean as well and no fill kernel runs per frame.
template <int LPG>
__global__ __launch_bounds__(BIN_BLOCK) void k_tile_emit(
    const float* __restrict__ uvs, const float* __restrict__ xyz_cam,
    const float* __restrict__ conic, int V, int ntx, int nty, float mh, int row0, int row1,
    const int* __restrict__ ranges, int* __restrict__ cursor, uint64_t* __restrict__ keys,
    Items items, int64_t cap) {
    const int t = blockIdx.x * BIN_BLOCK + threadIdx.x;
    const int i = t / LPG, sub = t % LPG;
    const bool active = i < item_count(items, V);
    TileWalk tw;
    uint64_t key = 0;
    if (active) {
        const int g = item_at(items, i);
        key = ((uint64_t)sortable_bits(xyz_cam[g * 3 + 2]) << 32) | (uint32_t)g;
        tw = tile_walk_setup(uvs, conic, g, ntx, nty, mh, row0, row1);
    }
    wave_for_each_tile<LPG>(active, tw, ntx, key, [&](int tile, uint64_t k) {
        const int first = ranges[tile];
        const int old = atomicAdd(cursor + tile, 1);
        if (old + 1 == ranges[tile + 1] - first) cursor[tile] = 0;   // the tile's last hit: nobody else comes
        if (first + old < cap) keys[first + old] = k;
    }, sub);
}

__global__ __launch_bounds__(PRIV_BLOCK) void k_bin_emit(
    const float* __restrict__ uvs, const float* __restrict__ xyz_cam,
    const float* __restrict__ conic, int V, int ntx, int nty, float mh, int row0, int row1,
    const int* __restrict__ ranges, const int* __restrict__ hist, uint64_t* __restrict__ keys,
    Items items, int64_t cap) {
    extern __shared__ int s_cursor[];
    const int t0 = row0 * ntx, Tb = (row1 - row0) * ntx;   // the histogram matrix covers the rows [row0, row1)
    const int sl = slice_index(blockIdx.x);
    const int* row = hist + (size_t)sl * Tb;
    for (int t = threadIdx.x; t < Tb; t += PRIV_BLOCK) s_cursor[t] = ranges[t0 + t] + row[t];
    __syncthreads();
    int g0, g1;
    slice_of(sl, item_count(items, V), g0, g1);
    for (int base = g0; base < g1; base += PRIV_BLOCK) {   // wave-uniform trip count
        const int i = base + threadIdx.x;
        const bool active = i < g1;
        TileWalk tw;
        uint64_t key = 0;
        if (active) {
            const int g = item_at(items, i);
            key = ((uint64_t)sortable_bits(xyz_cam[g * 3 + 2]) << 32) | (uint32_t)g;
            tw = tile_walk_setup(uvs, conic, g, ntx, nty, mh, row0, row1);
        }
        // (storing a hit's key only after the NEXT hit's cursor atomic has been issued, so that the LDS round
        // trip overlaps the following separating-axis test, changes nothing: 0.293 vs 0.291 ms)
        wave_for_each_tile(active, tw, ntx, key, [&](int tile, uint64_t k) {
            const int pos = atomicAdd(&s_cursor[tile - t0], 1);
            if (pos < cap) keys[pos] = k;
        });
    }
}

// ---- depth-bucketed binning ("depth cut") ---------------------------------------------------------------
// The lists of a dense frame are several times longer than what the render consumes: at workload D a tile has
// ~2800 entries, no pixel of any tile composites deeper than the 743rd, and the prefix sort already orders only the
// 1024 nearest -- yet count, emit and sort handle all 12.2 M instances.  Not emitting what lies behind a tile's
// 1024 nearest entries needs, per tile, the DEPTH below which there are at most 1024 of them -- before the emit.
// The privatised count pass can provide exactly that if its workgroups are DEPTH BUCKETS instead of arbitrary
// slices of the Gaussians: row b of the histogram matrix hist[b][t] is then "entries of tile t in depth bucket b",
// the column scan that turns the rows into offsets is a cumulative depth histogram, and the cut of tile t is the
// last bucket b*(t) at which it still holds <= GS_SORT_PREFIX entries -- exact counts, no estimate:
//   k_depth_hist      the visible Gaussians are split into NBK = 1024 depth buckets of about equal population
//                     (boundaries = quantiles of ~100 k sampled depths that k_cull_count left behind; every
//                     workgroup derives the same boundaries from a fine histogram) and counted per (partition
//                     workgroup, bucket)
//   k_depth_scatter   counting sort of the Gaussian indices by bucket: list[] + bucket offsets boff[NBK + 1]
//   k_bin_count       workgroup b walks bucket b (32-byte binning records gathered through the list: one sector per
//                     Gaussian); rows of hist as before
//   k_bin_colscan     + per tile: b*(t), n'(t) = entries in buckets <= b*(t) (<= 1024; everything if the tile has
//                     no more than that), n(t) = all entries
//   k_scan_tiles_cut  tile_ranges from n' (the lists that are emitted), full_ranges from n (where the complete list
//                     of a tile goes if it has to be repaired), the deepest bucket any tile still wants
//   k_bin_emit<CUT>   workgroups behind the deepest wanted bucket exit; a candidate tile that is cut in front of the
//                     bucket is skipped before its separating-axis test
// A truncated list is a true depth prefix of the complete one (buckets are depth intervals; equal depths share a
// bucket), completely sorted by the ordinary <= 1024 sort.  It stays exact the way the prefix sort did: the
// forward raises tile_flags[t] when a truncated tile reaches the end of its list with an unsaturated pixel, and
// the same call enqueues k_bin_emit<REST> (complete lists of the flagged tiles into an overflow buffer at
// full_ranges), their sort and a second render; the backward reads a flagged tile's list from the overflow.
// All of it is enqueued without a host read; the repair kernels exit at once while nothing is flagged.
constexpr int NBK = PRIV_NB;            // depth buckets == workgroups of the count / emit passes
constexpr int DC_BLOCK = 1024;
static_assert(NBK == GS_CUT_BUCKETS && NBK == DC_BLOCK, "one thread per bucket in the partition kernels");
static_assert(GS_SORT_PREFIX == 1024, "a truncated list must fit the <= 1024 sort of k_tile_sort");

__device__ inline uint32_t sortable_bits_u(float z) {   // (defined again below as sortable_bits: same map)
    const uint32_t u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// first index k in [0, n) with a[k] >= x (n if none); a ascending, in LDS
__device__ inline int lower_bound_u32(const uint32_t* a, int n, uint32_t x) {
    int lo = 0, len = n;
    while (len > 0) {
        const int half = len >> 1;
        if (a[lo + half] < x) {
            lo += half + 1;
            len -= half + 1;
        } else {
            len = half;
        }
    }
    return lo;
}

// block-wide exclusive prefix of one int per thread (1024 threads); returns the prefix, *total = the sum
__device__ inline int block_scan_1024(int v, int* s_wave /*[17]*/, int* total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();   // s_wave may still be read from a previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = 0, sum = 0;
    for (int w = 0; w < 16; w++) {
        if (w < wave) off += s_wave[w];
        sum += s_wave[w];
    }
    if (total) *total = sum;
    return off + incl - v;
}

// per (partition workgroup, bucket) counts of the visible Gaussians' bucket ids (written by the per-Gaussian stage:
// preprocess.hip, k_preprocess).  The ids of a thread's elements are loaded before the LDS atomics are issued: one
// workgroup per CU leaves 4 waves per SIMD, not enough to hide a load -> atomic chain per element.
__global__ __launch_bounds__(DC_BLOCK) void k_depth_hist(const int* __restrict__ v_dev, CutState cs) {
    __shared__ int s_cnt[NBK];
    const int tid = threadIdx.x;
    s_cnt[tid] = 0;
    __syncthreads();
    const int V = *v_dev;
    const int chunk = (V + DC_PART - 1) / DC_PART;
    const int v0 = min(V, (int)blockIdx.x * chunk), v1 = min(V, v0 + chunk);
    for (int base = v0; base < v1; base += 4 * DC_BLOCK) {
        int b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = base + k * DC_BLOCK + tid;
            b[k] = v < v1 ? (int)cs.bucket_of[v] : -1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (b[k] >= 0) atomicAdd(&s_cnt[b[k]], 1);
    }
    __syncthreads();
    cs.phist[(size_t)blockIdx.x * NBK + tid] = s_cnt[tid];
}

// phist[w][b] <- entries of bucket b in the partition workgroups before w; boff[b + 1] <- bucket b's total
// (k_depth_scatter turns the totals into offsets itself: 4 KB per workgroup).  64 buckets x 16 row segments per
// workgroup, as k_bin_colscan: sixteen workgroups share the 1 MB matrix -- one workgroup alone reads ~100 GB/s,
// and every scatter workgroup summing the 256 rows for itself was 256 MB of L2 reads (25 of that kernel's 39 us).
__global__ __launch_bounds__(1024) void k_depth_colscan(CutState cs) {
    constexpr int CT = 64, SEGS = 1024 / CT, ROWS = DC_PART / SEGS;
    __shared__ int s_seg[SEGS][CT];
    const int lt = threadIdx.x & (CT - 1), seg = threadIdx.x / CT;
    const int b = blockIdx.x * CT + lt;
    int* col = cs.phist + (size_t)seg * ROWS * NBK + b;
    int v[ROWS], sum = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        v[r] = col[(size_t)r * NBK];
        sum += v[r];
    }
    s_seg[seg][lt] = sum;
    __syncthreads();
    int run = 0, total = 0;
    for (int s2 = 0; s2 < SEGS; s2++) {
        if (s2 < seg) run += s_seg[s2][lt];
        total += s_seg[s2][lt];
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        col[(size_t)r * NBK] = run;
        run += v[r];
    }
    if (seg == 0) cs.boff[b + 1] = total;
}

// counting sort of the visible indices by bucket: workgroup w places its chunk.  The chunk is first ordered by
// bucket in LDS (local cursor per bucket), then written out: consecutive LDS slots of one bucket go to consecutive
// list positions, so a wave's store covers runs of ~10 entries instead of 64 unrelated 4-byte sectors.  Inside a
// (workgroup, bucket) run the order is whatever the LDS atomics hand out -- no result depends on it (tile lists are
// ordered by unique keys afterwards; counts per (bucket, tile) do not depend on the order).
constexpr int DC_STAGE = 12 * 1024;   // LDS slots: the chunk of a partition workgroup (V / 256) up to 3.1 M visible
__global__ __launch_bounds__(DC_BLOCK) void k_depth_scatter(const int* __restrict__ v_dev, CutState cs) {
    __shared__ int s_lbase[NBK + 1];   // first LDS slot of each bucket's run
    __shared__ int s_cur[NBK];
    __shared__ int s_gbase[NBK];
    __shared__ int s_wave[17];
    __shared__ int s_list[DC_STAGE];
    const int tid = threadIdx.x;
    const int V = *v_dev;
    const int chunk = (V + DC_PART - 1) / DC_PART;
    const int v0 = min(V, (int)blockIdx.x * chunk), v1 = min(V, v0 + chunk);
    const int n = v1 - v0;
    // bucket offsets from the totals k_depth_colscan left in boff[1 ..] (every workgroup scans the 4 KB itself;
    // workgroup 0 publishes the result as boff2 for the count and emit passes)
    const int total = cs.boff[tid + 1];
    int vis;
    const int base = block_scan_1024(total, s_wave, &vis);
    if (blockIdx.x == 0) {
        cs.boff2[tid] = base;
        if (tid == 0) cs.boff2[NBK] = vis;
    }
    const int before = cs.phist[(size_t)blockIdx.x * NBK + tid];
    const int gb = base + before;
    // this workgroup's count of bucket tid = the distance to the next workgroup's offset (or the bucket's total)
    const int nxt = (int)blockIdx.x + 1 < DC_PART ? cs.phist[(size_t)(blockIdx.x + 1) * NBK + tid] : total;
    const int cnt = nxt - before;
    int sum;
    const int lb = block_scan_1024(cnt, s_wave, &sum);
    s_lbase[tid] = lb;
    if (tid == 0) s_lbase[NBK] = sum;
    s_cur[tid] = lb;
    s_gbase[tid] = gb;
    __syncthreads();
    const bool staged = n <= DC_STAGE;   // (workgroup-uniform; larger chunks fall back to direct scattered stores)
    for (int b0 = v0; b0 < v1; b0 += 4 * DC_BLOCK) {
        int b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = b0 + k * DC_BLOCK + tid;
            b[k] = v < v1 ? (int)cs.bucket_of[v] : -1;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (b[k] < 0) continue;
            const int slot = atomicAdd(&s_cur[b[k]], 1);
            const int v = b0 + k * DC_BLOCK + tid;
            if (staged) s_list[slot] = v;
            else cs.list[s_gbase[b[k]] + slot - s_lbase[b[k]]] = v;
        }
    }
    if (!staged) return;
    __syncthreads();
    for (int i = tid; i < n; i += DC_BLOCK) {
        // the bucket of LDS slot i: last b with lbase[b] <= i
        int lo = 0, len = NBK;
        while (len > 1) {
            const int half = len >> 1;
            if (s_lbase[lo + half] <= i) {
                lo += half;
                len -= half;
            } else {
                len = half;
            }
        }
        cs.list[s_gbase[lo] + i - s_lbase[lo]] = s_list[i];
    }
}

// k_bin_count of the depth-bucketed path: workgroup b walks the Gaussians of bucket b
__global__ __launch_bounds__(PRIV_BLOCK) void k_bin_count_buckets(const float* __restrict__ rec, int ntx, int nty, float mh,
                                                                  int row0, int row1, int* __restrict__ hist,
                                                                  CutState cs) {
    extern __shared__ int s_hist[];
    const int t0 = row0 * ntx, Tb = (row1 - row0) * ntx;
    for (int t = threadIdx.x; t < Tb; t += PRIV_BLOCK) s_hist[t] = 0;
    __syncthreads();
    const int sl = blockIdx.x;
    const int g0 = cs.boff2[sl], g1 = cs.boff2[sl + 1];
    // (requesting the NEXT trip's record -- list entry, then a 32-byte gather -- before this trip's tiles are walked
    // makes the kernel slower, 77 -> 87 us at workload D; reading the records in list order instead of gathering
    // them: 73 us; the rest of the distance to k_bin_count's 61 us is the buckets' unequal populations (max / mean
    // 1.4 with 64 samples per bucket).  profiles/r04/kbench_cut_count_variants.txt)
    for (int base = g0; base < g1; base += PRIV_BLOCK) {   // wave-uniform trip count
        const int i = base + threadIdx.x;
        const bool active = i < g1;
        TileWalk tw;
        if (active) tw = tile_walk_setup(load_bin_record(rec, cs.list[i]), ntx, nty, mh, row0, row1);
        wave_for_each_tile(active, tw, ntx, 0, [&](int tile, uint64_t) { atomicAdd(&s_hist[tile - t0], 1); });
    }
    __syncthreads();
    int* row = hist + (size_t)sl * Tb;
    for (int t = threadIdx.x; t < Tb; t += PRIV_BLOCK) row[t] = s_hist[t];
}

// k_bin_colscan + the cut: b*(t) = the last NON-EMPTY bucket at which the tile's cumulative count is <= kcut (the
// qualifying buckets are a prefix: counts only grow), n'(t) = the count there, n(t) = all.  -1 / 0 when already the
// first non-empty bucket exceeds kcut (such a tile is rendered from its complete list by the repair pass).
template <int CS_TILES>
__global__ __launch_bounds__(1024) void k_bin_colscan_cut(int* __restrict__ hist, int T, int* __restrict__ counts,
                                                          int* __restrict__ totals, int* __restrict__ bstar, int kcut) {
    constexpr int CS_SEGS = 1024 / CS_TILES;
    constexpr int CS_ROWS = NBK / CS_SEGS;
    __shared__ int s_seg[CS_SEGS][CS_TILES];
    __shared__ int s_best[CS_SEGS][CS_TILES];
    __shared__ int s_bn[CS_SEGS][CS_TILES];
    const int lt = threadIdx.x & (CS_TILES - 1);
    const int seg = threadIdx.x / CS_TILES;
    const int t = blockIdx.x * CS_TILES + lt;
    int* col = hist + (size_t)seg * CS_ROWS * T + t;
    int v[CS_ROWS], sum = 0;   // (registers between the two walks, as in k_bin_colscan)
    if (t < T) {
#pragma unroll
        for (int r = 0; r < CS_ROWS; r++) {
            v[r] = col[(size_t)r * T];
            sum += v[r];
        }
    }
    s_seg[seg][lt] = sum;
    __syncthreads();
    int run = 0;
    for (int s2 = 0; s2 < seg; s2++) run += s_seg[s2][lt];
    int best = -1, bn = 0;
    if (t < T) {
#pragma unroll
        for (int r = 0; r < CS_ROWS; r++) {
            const int h = v[r];
            col[(size_t)r * T] = run;
            run += h;
            if (h > 0 && run <= kcut) {
                best = seg * CS_ROWS + r;
                bn = run;
            }
        }
    }
    s_best[seg][lt] = best;
    s_bn[seg][lt] = bn;
    __syncthreads();
    if (t < T && seg == CS_SEGS - 1) {
        for (int s2 = 0; s2 < CS_SEGS; s2++)
            if (s_best[s2][lt] > best) {
                best = s_best[s2][lt];
                bn = s_bn[s2][lt];
            }
        counts[t] = bn;
        totals[t] = run;   // the last segment's running count is the column total
        bstar[t] = best;
    }
}

// exclusive prefixes of the kept counts (-> ranges) and of the totals (-> full_ranges), the deepest wanted bucket;
// clears the frame's flag counter.  One workgroup.  Only [t0, t0 + Tb) was written by the column scan.
// ranges[T] = S' (entries emitted), ranges[T + 1] = V, ranges[T + 2] = S (all entries): the frame's host read
__global__ __launch_bounds__(1024) void k_scan_tiles_cut(const int* __restrict__ counts, const int* __restrict__ totals,
                                                         const int* __restrict__ bstar, int T, int* __restrict__ ranges,
                                                         int* __restrict__ full_ranges, const int* __restrict__ v_dev,
                                                         int t0, int Tb, int* __restrict__ ctrl,
                                                         int* __restrict__ host_mirror) {
    // both prefixes in one scan: (complete count << 32 | kept count) -- neither half exceeds 2^31
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_carry;
    __shared__ int s_bmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_bmax = -1;
        s_carry = 0;
    }
    __syncthreads();
    int bmax = -1;
    for (int base = 0; base < T; base += 4 * 1024) {
        unsigned long long v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            const bool in = i >= t0 && i < t0 + Tb;
            v[k] = in ? ((unsigned long long)(unsigned)totals[i] << 32) | (unsigned)counts[i] : 0ull;
            if (in) bmax = max(bmax, bstar[i]);
            sum += v[k];
        }
        unsigned long long incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long n = __shfl_up(incl, d);
            if (lane >= d) incl += n;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long run = s_carry + incl - sum;
        for (int w = 0; w < wave; w++) run += s_wave[w];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = base + tid * 4 + k;
            if (i < T) {
                ranges[i] = (int)(unsigned)run;
                full_ranges[i] = (int)(run >> 32);
            }
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) bmax = max(bmax, __shfl_xor(bmax, d));
    if (lane == 0) atomicMax(&s_bmax, bmax);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long c = s_carry;
        ranges[T] = (int)(unsigned)c;
        ranges[T + 1] = v_dev ? *v_dev : 0;
        ranges[T + 2] = (int)(c >> 32);
        full_ranges[T] = (int)(c >> 32);
        ctrl[0] = s_bmax;
        ctrl[1] = 0;
        if (host_mirror != nullptr) {   // (see k_scan_tiles)
            host_mirror[0] = (int)(unsigned)c;
            host_mirror[1] = v_dev ? *v_dev : 0;
            host_mirror[2] = (int)(c >> 32);
            __threadfence_system();
        }
    }
}

// k_bin_emit of the depth-bucketed path.  MODE 1 (cut): keys of the buckets <= b*(t) into the tile's (truncated)
// segment of `keys`; MODE 2 (rest): the COMPLETE lists of the flagged tiles into the overflow buffer at
// full_ranges -- the repair pass, exits at once while the frame has no flagged tile.  hist holds, after the column
// scan, the tile's entries in front of bucket b: the offset in either layout.
// Only the buckets in front of the deepest cut do anything (42 % of them at workload D): with the count pass's 512
// threads per bucket a CU is left with one or two 8-wave workgroups; 1024 threads per bucket halve the trips.
constexpr int GS_CUT_EMIT_BLOCK = 1024;
template <int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_bin_emit_buckets(const float* __restrict__ rec, int ntx, int nty, float mh,
                                                                 int row0, int row1, const int* __restrict__ ranges,
                                                                 const int* __restrict__ hist, uint64_t* __restrict__ keys,
                                                                 int64_t cap, const int* __restrict__ flags, CutState cs) {
    extern __shared__ int s_cursor[];
    const int sl = blockIdx.x;
    // (the flag counter belongs to the render that follows this emit: cleared here so that a repeated emit + render
    // of the same frame -- a capacity miss -- starts from zero as well)
    if (MODE == 1 && sl == 0 && threadIdx.x == 0) cs.ctrl[1] = 0;
    if (MODE == 1 && sl > cs.ctrl[0]) return;
    if (MODE == 2 && cs.ctrl[1] == 0) return;
    const int t0 = row0 * ntx, Tb = (row1 - row0) * ntx;
    const int* row = hist + (size_t)sl * Tb;
    for (int t = threadIdx.x; t < Tb; t += BLOCK) {
        const bool want = MODE == 1 ? sl <= cs.bstar[t0 + t] : flags[t0 + t] != 0;
        s_cursor[t] = want ? ranges[t0 + t] + row[t] : -1;
    }
    const int g0 = cs.boff2[sl], g1 = cs.boff2[sl + 1];
    __syncthreads();
    // (the next trip's record in flight during this trip's walk: no faster, as in the count pass)
    for (int base = g0; base < g1; base += BLOCK) {   // wave-uniform trip count
        const bool active = base + (int)threadIdx.x < g1;
        const int g = active ? cs.list[base + threadIdx.x] : 0;
        BinRec r{};
        if (active) r = load_bin_record(rec, g);
        TileWalk tw;
        uint64_t key = 0;
        if (active) {
            key = ((uint64_t)sortable_bits_u(r.b.y) << 32) | (uint32_t)g;
            tw = tile_walk_setup(r, ntx, nty, mh, row0, row1);
        }
        wave_for_each_tile(
            active, tw, ntx, key,
            [&](int tile, uint64_t k) {
                const int pos = atomicAdd(&s_cursor[tile - t0], 1);
                if (pos < cap) keys[pos] = k;
            },
            0, [&](int tile) { return s_cursor[tile - t0] >= 0; });
    }
}

// ---- per-tile sort -------------------------------------------------------------------------------
// Bitonic network in its all-ascending form on the tile's unique 64-bit keys: phase k starts with
// a "flip" step (element i of a k-block against element k-1-i) followed by half-cleaners of stride
// j = k/4 .. 1.  Every comparator puts the smaller key at the lower index, so the +inf padding at
// indices >= n never moves and stays virtual.
//
// Register blocking: a thread owns SORT_R = 16 consecutive elements.  All phases with k <= 16 and
// the half-cleaners with j < 16 of every later phase run in registers (one LDS round trip for 16
// elements, 32-80 compare-exchanges in between); only the flip and the j >= 16 half-cleaners of
// the phases k >= 32 go through LDS pair-wise.  For a 4096-key tile that is 36 pair steps + 9
// register passes instead of 78 pair steps, and the small strides that bank-conflict in a
// pair-wise step never touch LDS.  Element i lives at LDS slot i + i/16 (one pad slot per block),
// which makes the 16-element block reads conflict-free (lane stride 34 banks).
//
// Size classes (separate launches, a workgroup whose tile is in another class exits at once) keep
// LDS per workgroup -- and with it the number of resident workgroups per CU -- matched to the tile:
//   n <= 1024: 8.5 KiB    n <= 4096: 34 KiB    n <= 8192: 68 KiB    larger: pair-wise on global memory.
template <int CAP_LO, int CAP_HI>
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort_lds(const int* __restrict__ ranges,
                                                              const uint64_t* __restrict__ keys,
                                                              int* __restrict__ sorted, int tile0,
                                                              int64_t cap) {
    extern __shared__ uint64_t s_keys[];
    const int tile = tile0 + blockIdx.x;
    const int s0 = ranges[tile];
    const int n = ranges[tile + 1] - s0;
    if (n <= CAP_LO || n > CAP_HI || (int64_t)s0 + n > cap) return;
    const int tid = threadIdx.x;
    if (n == 1) {
        if (tid == 0) sorted[s0] = (int)(uint32_t)keys[s0];
        return;
    }
    for (int i = tid; i < n; i += SORT_BLOCK) s_keys[slot(i)] = keys[s0 + i];
    __syncthreads();
    lds_bitonic_sort(s_keys, n, tid);
    for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = (int)(uint32_t)s_keys[slot(i)];
}

// ---- wave-level sorts (lists of up to 1024 entries) ------------------------------------------------
// The register-blocked network above keeps only n/16 threads busy in its register passes: one
// wave out of four for 1024 keys, four lanes for the ~50-entry lists of a sparse scene.  Short
// lists therefore use waves as the unit:
//   wave_sort<R>      one wave sorts 64*R keys held R per lane (element e = R*lane + r): classic
//                     bitonic network with alternating directions, strides < R in registers, the
//                     others as lane-xor exchanges.  No LDS storage, no workgroup barrier.
//   sort1024_4waves   each of the 4 waves sorts a run of 256 (wave_sort<4>); the sorted runs go to
//                     LDS and every key finds its rank in the three other runs by a branch-free
//                     binary search (12 independent searches per thread, 9 dependent LDS reads
//                     each) and lands at own index + ranks: a 4-way merge without a merge loop.
// Padding is explicit here (~0 keys, larger than any real key); real keys are unique (they carry
// the Gaussian index), so ranks are unambiguous and padding lands at positions >= n.
__device__ __forceinline__ void ce_dir(uint64_t& a, uint64_t& b, bool asc) {
    const bool sw = (a > b) == asc;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}

template <int R>
__device__ __forceinline__ void wave_sort(uint64_t (&v)[R], int lane) {
    constexpr int NK = 64 * R;
#pragma unroll
    for (int k = 2; k <= NK; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
            if (j >= R) {
                const int lx = j / R;
                const bool lower = (lane & lx) == 0;
                const bool asc = k == NK || (lane & (k / R)) == 0;   // k > j >= R
                const bool keep_min = lower == asc;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint64_t o = __shfl_xor(v[r], lx);
                    const bool take = keep_min ? (o < v[r]) : (o > v[r]);
                    v[r] = take ? o : v[r];
                }
            } else {
                // partner r ^ j inside the lane; direction from bit k of e = R*lane + r
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if ((r & j) != 0) continue;
                    const bool asc = k == NK || (k >= R ? (lane & (k / R)) == 0 : (r & k) == 0);
                    ce_dir(v[r], v[r | j], asc);
                }
            }
        }
    }
}

// one wave, n <= 64 * R keys straight from / to global memory
template <int R>
__device__ __forceinline__ void wave_sort_tile(const uint64_t* __restrict__ keys, int* __restrict__ sorted,
                                      int n, int lane) {
    uint64_t v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = (R * lane + r < n) ? keys[R * lane + r] : ~0ull;
    wave_sort<R>(v, lane);
#pragma unroll
    for (int r = 0; r < R; r++)
        if (R * lane + r < n) sorted[R * lane + r] = (int)(uint32_t)v[r];
}

// s_run: 1024 keys (linear, padded with ~0 beyond the list); s_out: 1024 ints
__device__ __forceinline__ void sort1024_4waves(uint64_t* s_run, int* s_out, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    uint64_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = s_run[4 * tid + r];
    wave_sort<4>(v, lane);
    __syncthreads();   // every wave has read its unsorted run
#pragma unroll
    for (int r = 0; r < 4; r++) s_run[4 * tid + r] = v[r];
    __syncthreads();
    int pos[4][3];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int o = 0; o < 3; o++) pos[r][o] = 0;
#pragma unroll
    for (int step = 128; step >= 1; step >>= 1) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int o = 0; o < 3; o++) {
                const int run = (wave + 1 + o) & 3;
                if (s_run[run * 256 + pos[r][o] + step - 1] < v[r]) pos[r][o] += step;
            }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int at = 4 * lane + r;
#pragma unroll
        for (int o = 0; o < 3; o++) {
            const int run = (wave + 1 + o) & 3;
            at += pos[r][o] + (s_run[run * 256 + pos[r][o]] < v[r] ? 1 : 0);   // reaches 256
        }
        s_out[at] = (int)(uint32_t)v[r];   // padding lands in [n, 1024), possibly on top of other padding
    }
    __syncthreads();
}

// ---- prefix sort -----------------------------------------------------------------------------------
// The render consumes a tile's list front to back and stops when every pixel of the tile is
// saturated; on dense scenes that is a small part of the list (2.86 M Gaussians at 1297x840: lists
// of ~2800, deepest pixel of any tile at 743).  In prefix mode a tile with more than
// GS_SORT_PREFIX = 1024 entries gets only its 1024 smallest keys ordered:
//   1. the workgroup holds the tile's keys in registers (KPT per thread)
//   2. MSB-first radix select of the 1024th smallest key: 8-bit digits starting at the highest bit
//      in which the tile's keys differ (so the first histogram already spreads), LDS histogram +
//      one-wave scan per digit, until the pivot's bucket is needed in full
//   3. ballot-compaction of the keys <= pivot into LDS (exactly 1024: the keys are unique)
//   4. sort1024_4waves on those 1024
// The result is exact, not approximate: k_render_fwd raises a per-tile flag if it reaches the end
// of the prefix with an unsaturated pixel, and the repair pass (k_tile_sort_flagged + a
// flagged-only render) redoes such a tile from its full list, all enqueued without a host read.
struct SortLds {
    uint64_t sel[GS_SORT_PREFIX];
    int out[GS_SORT_PREFIX];
    int hist[256];
    unsigned long long diff;
    int state[3];
    int cnt;
};

template <int KPT>
__device__ __forceinline__ void prefix_sort_tile(const uint64_t* __restrict__ keys, int* __restrict__ sorted,
                                        int n, int tid, SortLds& L) {
    constexpr int K = GS_SORT_PREFIX;
    const int lane = tid & 63;
    uint64_t k[KPT];
    const uint64_t key0 = keys[0];
    uint64_t diff = 0;
#pragma unroll
    for (int e = 0; e < KPT; e++) {
        const int i = e * SORT_BLOCK + tid;
        k[e] = i < n ? keys[i] : key0;   // the filler never counts: every use is guarded by i < n
        diff |= k[e] ^ key0;
    }
    if (tid == 0) {
        L.diff = 0;
        L.cnt = 0;
    }
    __syncthreads();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) diff |= __shfl_xor(diff, d);
    if (lane == 0) atomicOr(&L.diff, (unsigned long long)diff);
    __syncthreads();
    int hi = 64 - __builtin_clzll(L.diff);   // bits [hi, 64) are common to all keys; unique keys: hi >= 1
    uint64_t prefix = hi < 64 ? ((key0 >> hi) << hi) : 0ull;
    int r = K;
    while (true) {
        const int w = hi < 8 ? hi : 8;
        const int shift = hi - w;
        L.hist[tid] = 0;   // SORT_BLOCK == 256 bins
        __syncthreads();
#pragma unroll
        for (int e = 0; e < KPT; e++) {
            const int i = e * SORT_BLOCK + tid;
            if (i < n && (((k[e] ^ prefix) >> shift) >> w) == 0)
                atomicAdd(&L.hist[(int)((k[e] >> shift) & ((1u << w) - 1))], 1);
        }
        __syncthreads();
        if (tid < 64) {
            const int h0 = L.hist[4 * lane], h1 = L.hist[4 * lane + 1], h2 = L.hist[4 * lane + 2],
                      h3 = L.hist[4 * lane + 3];
            const int t = h0 + h1 + h2 + h3;
            int incl = t;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            int below = incl - t;
            if (below < r && r <= incl) {   // exactly one lane: the candidates number at least r
                int d = 4 * lane, cd = h0;
                if (r > below + h0) {
                    below += h0; d++; cd = h1;
                    if (r > below + h1) {
                        below += h1; d++; cd = h2;
                        if (r > below + h2) { below += h2; d++; cd = h3; }
                    }
                }
                L.state[0] = d;
                L.state[1] = r - below;
                L.state[2] = cd;
            }
        }
        __syncthreads();
        const int d = L.state[0], cd = L.state[2];
        r = L.state[1];
        prefix |= (uint64_t)d << shift;
        hi = shift;
        if (cd == r || hi == 0) break;   // the whole bucket belongs to the prefix (at hi == 0: one key)
    }
    const uint64_t pivot_hi = prefix >> hi;
#pragma unroll
    for (int e = 0; e < KPT; e++) {
        const int i = e * SORT_BLOCK + tid;
        const bool sel = i < n && (k[e] >> hi) <= pivot_hi;
        const unsigned long long m = __ballot(sel);
        if (m == 0) continue;
        int base = 0;
        if (lane == __builtin_ctzll(m)) base = atomicAdd(&L.cnt, __builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (sel) L.sel[base + __builtin_popcountll(m & ((1ull << lane) - 1))] = k[e];
    }
    __syncthreads();
    sort1024_4waves(L.sel, L.out, tid);
    for (int i = tid; i < K; i += SORT_BLOCK) sorted[i] = L.out[i];
}

// One workgroup per tile of the band.  By list length n:
//   n <= 64, <= 256   wave 0 alone (wave_sort<1>, <4>)
//   n <= 1024         sort1024_4waves on the padded list
//   n <= 4096         prefix mode: radix select + sort of the nearest 1024; full mode: left to
//                     k_tile_sort_lds
//   larger            k_tile_sort_big
template <bool PREFIX>
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort(
    const int* __restrict__ ranges,
                                                          const uint64_t* __restrict__ keys,
                                                          int* __restrict__ sorted, int tile0,
                                                          int64_t cap, int64_t lim) {
    __shared__ SortLds L;
    const int tile = tile0 + blockIdx.x;
    const int s0 = ranges[tile];
    const int n = ranges[tile + 1] - s0;
    if (n <= 0 || (int64_t)s0 + n > cap) return;
    const int tid = threadIdx.x;
    if (n > (PREFIX ? 4096 : 1024)) {
        // a longer list belongs to one of the kernels launch_tile_sort enqueues behind this one -- IF the caller's
        // bound on the list lengths (lim) let it enqueue that kernel.  A bound that was only a guess, and too small,
        // leaves the list to nobody: the caller will find out and repeat the call, but a render it enqueued in
        // between must at least read valid indices -- the list's entries in emit order.
        const int lower = n <= 4096 ? 1024 : (n <= SORT_MAX_LDS_KEYS ? 4096 : SORT_MAX_LDS_KEYS);
        const bool covered = PREFIX ? lim > 4096 : lim > lower;
        if (!covered)
            for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = (int)(uint32_t)keys[s0 + i];
        return;
    }
    if (n <= 256) {
        if (tid >= 64) return;
        if (n <= 64) wave_sort_tile<1>(keys + s0, sorted + s0, n, tid);
        else wave_sort_tile<4>(keys + s0, sorted + s0, n, tid);
    } else if (n <= 1024) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = r * SORT_BLOCK + tid;
            L.sel[i] = i < n ? keys[s0 + i] : ~0ull;
        }
        __syncthreads();
        sort1024_4waves(L.sel, L.out, tid);
        for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = L.out[i];
    } else if (PREFIX) {
        prefix_sort_tile<4096 / SORT_BLOCK>(keys + s0, sorted + s0, n, tid, L);
    }
}

// Sort of a depth-cut tile list (<= 1024 entries).  The emit lays a tile's kept entries out bucket by bucket, and
// buckets are depth intervals in ascending order: the list is a concatenation of short unordered runs (2.7 entries on
// average at workload D) whose runs are already in order -- every key is at most (its run's length - 1) places from
// its final position.  An odd-even transposition network sorts exactly that in as many passes as the longest run is
// long: four consecutive keys per thread in registers, the pair across two threads through LDS, until a pass swaps
// nothing (then every adjacent pair is in order).  Correct for any input (n passes sort anything); a list that has
// not settled after RUN_PASSES passes goes through the general 1024-key sort instead.
constexpr int GS_CUT_RUN_PASSES = 24;
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort_runs(const int* __restrict__ ranges,
                                                               const uint64_t* __restrict__ keys,
                                                               int* __restrict__ sorted, int tile0, int64_t cap) {
    __shared__ SortLds L;   // first / last key of every thread in L.sel[0 .. 512); the fallback's buffers
    const int tile = tile0 + blockIdx.x;
    const int s0 = ranges[tile];
    const int n = ranges[tile + 1] - s0;
    if (n <= 0 || n > GS_SORT_PREFIX || (int64_t)s0 + n > cap) return;
    const int tid = threadIdx.x;
    uint64_t k[4];
#pragma unroll
    for (int r = 0; r < 4; r++) k[r] = 4 * tid + r < n ? keys[s0 + 4 * tid + r] : ~0ull;
    uint64_t* s_first = L.sel;
    uint64_t* s_last = L.sel + SORT_BLOCK;
    bool settled = false;
    for (int pass = 0; pass < GS_CUT_RUN_PASSES; pass++) {
        bool sw = false;
#define GS_CE_SW(x, y)                                                                             \
    do {                                                                                           \
        const uint64_t _a = (x), _b = (y);                                                         \
        const bool _s = _a > _b;                                                                   \
        (x) = _s ? _b : _a;                                                                        \
        (y) = _s ? _a : _b;                                                                        \
        sw |= _s;                                                                                  \
    } while (0)
        GS_CE_SW(k[0], k[1]);
        GS_CE_SW(k[2], k[3]);
        GS_CE_SW(k[1], k[2]);
        s_first[tid] = k[0];
        s_last[tid] = k[3];
        __syncthreads();
        const uint64_t right = tid + 1 < SORT_BLOCK ? s_first[tid + 1] : ~0ull;
        const uint64_t left = tid > 0 ? s_last[tid - 1] : 0ull;
        if (right < k[3]) {   // (k[3] of this thread, k[0] of the next): the smaller stays here
            k[3] = right;
            sw = true;
        }
        if (left > k[0]) {    // ... the larger goes there
            k[0] = left;
            sw = true;
        }
        if (!__syncthreads_or(sw)) {
            settled = true;
            break;
        }
    }
#undef GS_CE_SW
    if (settled) {
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * tid + r < n) sorted[s0 + 4 * tid + r] = (int)(uint32_t)k[r];
        return;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) L.sel[4 * tid + r] = k[r];   // (any permutation of the list, padded with ~0)
    __syncthreads();
    sort1024_4waves(L.sel, L.out, tid);
    for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = L.out[i];
}

// The rare long lists: a small grid walks the tiles.  4096 < n <= 8192: prefix mode only (full mode:
// k_tile_sort_lds); n > 8192: full sort in global memory in both modes.
template <bool PREFIX>
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort_big(const int* __restrict__ ranges,
                                                              uint64_t* __restrict__ keys,
                                                              int* __restrict__ sorted, int tile0,
                                                              int nt, int64_t cap) {
    __shared__ SortLds L;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int s0 = ranges[tile0 + t];
        const int n = ranges[tile0 + t + 1] - s0;
        if (n <= 4096 || (int64_t)s0 + n > cap) continue;
        if (n <= SORT_MAX_LDS_KEYS) {
            if (!PREFIX) continue;
            prefix_sort_tile<SORT_MAX_LDS_KEYS / SORT_BLOCK>(keys + s0, sorted + s0, n, tid, L);
        } else {
            global_sort_tile(keys + s0, sorted + s0, n, tid);
        }
        __syncthreads();
    }
}

// full sort of the flagged tiles (repair pass of the prefix mode): a small grid walks the flags
template <int CAP_HI>
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort_flagged(const int* __restrict__ ranges,
                                                                  const uint64_t* __restrict__ keys,
                                                                  int* __restrict__ sorted, int tile0,
                                                                  int nt, int64_t cap,
                                                                  const int* __restrict__ flags) {
    extern __shared__ uint64_t s_keys[];
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int tile = tile0 + t;
        if (flags[tile] == 0) continue;
        const int s0 = ranges[tile];
        const int n = ranges[tile + 1] - s0;
        if (!prefix_sorted_tile(n, GS_SORT_PREFIX) || n > CAP_HI || (int64_t)s0 + n > cap) continue;
        for (int i = tid; i < n; i += SORT_BLOCK) s_keys[slot(i)] = keys[s0 + i];
        __syncthreads();
        lds_bitonic_sort(s_keys, n, tid);
        for (int i = tid; i < n; i += SORT_BLOCK) sorted[s0 + i] = (int)(uint32_t)s_keys[slot(i)];
        __syncthreads();
    }
}


// hipFuncSetAttribute applies to the CURRENT device: once per device, not once per process (a second device in the
// same process otherwise never gets the larger dynamic-LDS limit; round-4 advisor finding)
static bool first_call_on_this_device(std::atomic<uint64_t>& seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const uint64_t bit = 1ull << dev;
    return (seen.fetch_or(bit) & bit) == 0;
}
static void sort_attr_once() {
    static std::atomic<uint64_t> seen{0};
    if (first_call_on_this_device(seen)) {
        (void)hipFuncSetAttribute((const void*)k_tile_sort_lds<4096, 8192>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sort_lds_bytes(8192));
        (void)hipFuncSetAttribute((const void*)k_tile_sort_flagged<SORT_MAX_LDS_KEYS>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sort_lds_bytes(SORT_MAX_LDS_KEYS));
    }
}

constexpr int WALK_GRID = 512;   // kernels whose work is rare walk the tiles with this many workgroups

static int launch_tile_sort(const int* ranges, uint64_t* keys, int* sorted, int tile0, int nt,
                            int64_t S, int sort_prefix, hipStream_t s, int64_t longest = -1) {
    if (nt <= 0) return GS_OK;
    const int walk = nt < WALK_GRID ? nt : WALK_GRID;
    // larger classes can only be populated if the instance count -- or the caller's bound on the longest list (< 0:
    // none) -- allows it (S itself stays the capacity the kernels check their writes against)
    const int64_t lim = longest >= 0 && longest < S ? longest : S;
    if (sort_prefix > 0) {
        k_tile_sort<true><<<nt, SORT_BLOCK, 0, s>>>(ranges, keys, sorted, tile0, S, lim);
        if (lim > 4096) k_tile_sort_big<true><<<walk, SORT_BLOCK, 0, s>>>(ranges, keys, sorted, tile0, nt, S);
        return GS_OK;
    }
    sort_attr_once();
    k_tile_sort<false><<<nt, SORT_BLOCK, 0, s>>>(ranges, keys, sorted, tile0, S, lim);
    if (lim > 1024)
        k_tile_sort_lds<1024, 4096><<<nt, SORT_BLOCK, sort_lds_bytes(4096), s>>>(ranges, keys, sorted,
                                                                             tile0, S);
    if (lim > 4096)
        k_tile_sort_lds<4096, 8192><<<nt, SORT_BLOCK, sort_lds_bytes(8192), s>>>(ranges, keys, sorted,
                                                                             tile0, S);
    if (lim > SORT_MAX_LDS_KEYS)
        k_tile_sort_big<false><<<walk, SORT_BLOCK, 0, s>>>(ranges, keys, sorted, tile0, nt, S);
    return GS_OK;
}

// repair pass: tiles of the (1024, 4096] class get the smaller LDS allocation
static int launch_sort_flagged(const int* ranges, const uint64_t* keys, int* sorted, int tile0,
                               int nt, int64_t S, const int* flags, hipStream_t s) {
    if (nt <= 0 || S <= GS_SORT_PREFIX) return GS_OK;   // no tile can have been prefix-sorted
    sort_attr_once();
    const int grid = nt < WALK_GRID ? nt : WALK_GRID;
    if (S <= 4096)
        k_tile_sort_flagged<4096><<<grid, SORT_BLOCK, sort_lds_bytes(4096), s>>>(
            ranges, keys, sorted, tile0, nt, S, flags);
    else
        k_tile_sort_flagged<SORT_MAX_LDS_KEYS><<<grid, SORT_BLOCK, sort_lds_bytes(SORT_MAX_LDS_KEYS), s>>>(
            ranges, keys, sorted, tile0, nt, S, flags);
    return GS_OK;
}

int sort_flagged_tiles(const int* ranges, const uint64_t* keys, int* sorted, int tile0, int nt,
                       int64_t S, const int* flags, hipStream_t s) {
    return launch_sort_flagged(ranges, keys, sorted, tile0, nt, S, flags, s);
}

// repair pass of the depth cut: complete sort of the flagged tiles' lists in the overflow buffer (any length);
// a small grid walks the flags and exits at once while the frame has none
__global__ __launch_bounds__(SORT_BLOCK) void k_tile_sort_overflow(const int* __restrict__ full_ranges,
                                                                   uint64_t* __restrict__ okeys, int* __restrict__ osorted,
                                                                   int tile0, int nt, int64_t cap,
                                                                   const int* __restrict__ flags,
                                                                   const int* __restrict__ ctrl) {
    extern __shared__ uint64_t s_keys[];
    if (ctrl[1] == 0) return;
    const int tid = threadIdx.x;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
        const int tile = tile0 + t;
        if (flags[tile] == 0) continue;
        const int s0 = full_ranges[tile];
        const int n = full_ranges[tile + 1] - s0;
        if (n <= 0 || (int64_t)s0 + n > cap) continue;
        if (n <= SORT_MAX_LDS_KEYS) {
            for (int i = tid; i < n; i += SORT_BLOCK) s_keys[slot(i)] = okeys[s0 + i];
            __syncthreads();
            if (n > 1) lds_bitonic_sort(s_keys, n, tid);
            for (int i = tid; i < n; i += SORT_BLOCK) osorted[s0 + i] = (int)(uint32_t)s_keys[slot(i)];
        } else {
            global_sort_tile(okeys + s0, osorted + s0, n, tid);
        }
        __syncthreads();
    }
}

// sort_too = false: only the emit; the caller's repair render sorts each flagged tile itself (render.hip
// k_render_fwd_flagged: one launch less on the frames without a flagged tile, i.e. nearly all)
int depth_cut_repair(const float* bin_records, int N, int ntx, int nty, float mh, int row0, int row1,
                     const int* full_ranges, int32_t* workspace, int32_t* cut_ws, uint64_t* okeys, int64_t ocap,
                     int* osorted, const int* flags, hipStream_t s, bool sort_too) {
    const int T = ntx * nty, t0 = row0 * ntx, Tb = (row1 - row0) * ntx;
    if (Tb <= 0) return GS_OK;
    const CutState cs = cut_state_of(cut_ws, N, T);
    k_bin_emit_buckets<2, PRIV_BLOCK><<<NBK, PRIV_BLOCK, sizeof(int) * (size_t)Tb, s>>>(
        bin_records, ntx, nty, mh, row0, row1, full_ranges, workspace + T, okeys, ocap, flags, cs);
    if (!sort_too) return GS_OK;
    static std::atomic<uint64_t> seen{0};
    if (first_call_on_this_device(seen))
        (void)hipFuncSetAttribute((const void*)k_tile_sort_overflow, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)sort_lds_bytes(SORT_MAX_LDS_KEYS));
    k_tile_sort_overflow<<<Tb < WALK_GRID ? Tb : WALK_GRID, SORT_BLOCK, sort_lds_bytes(SORT_MAX_LDS_KEYS), s>>>(
        full_ranges, okeys, osorted, t0, Tb, ocap, flags, cs.ctrl);
    return GS_OK;
}

int* depth_cut_flag_counter(int32_t* cut_ws, int N, int T) { return cut_state_of(cut_ws, N, T).ctrl + 1; }

}  // namespace gs

using namespace gs;

extern "C" {

// LDS-histogram mode pays when there are many instances per tile; with few Gaussians the
// NB x T histogram matrix costs more than the global atomics it saves.  The decision depends only
// on (T, V) so that gs_tile_count and gs_tile_emit_sort agree.
// The atomic-counter kernels give a Gaussian a group of lanes while the frame has few of them (see wave_for_each_tile)
// (workload B, 90 k visible Gaussians, gs_tile_count / gs_tile_emit_sort entries with their scans and sort: one lane
// per Gaussian 41 / 58 us, 4 lanes 32.3 / 46.2, 8 lanes 30.3 / 40.1, 16 lanes 28.1 / 34.6, 32 lanes 36.4 / 43.9;
// every lane of a group repeats the Gaussian's setup, so the group shrinks as the frame grows)
constexpr int SUBGROUP_SMALL = 16, SUBGROUP_SMALL_MAX_V = 1 << 17;
constexpr int SUBGROUP_MID = 8, SUBGROUP_MID_MAX_V = 1 << 18;
// Tb: tiles of the rows binned (a band, or the grid).  The histogram matrix (PRIV_NB x Tb ints) lives behind the
// grid's T counters; gs_tile_workspace_ints reserves room for the largest matrix any row range of the grid can ask
// for, PRIV_NB x min(T, PRIV_MAX_TILES) -- a band of a grid that is itself too large for the LDS histogram (4K and
// up: T > 16384) still takes the LDS path.
static bool use_private(int Tb, int V) {
    return Tb > 0 && Tb <= PRIV_MAX_TILES && (int64_t)V * 8 > (int64_t)PRIV_NB * Tb;
}

size_t gs_tile_workspace_ints(int n_tiles) {
    const size_t T = n_tiles > 0 ? (size_t)n_tiles : 1;
    return T + (size_t)PRIV_NB * (T <= (size_t)PRIV_MAX_TILES ? T : (size_t)PRIV_MAX_TILES);
}

int gs_tile_count(const void* uvs, const void* conic, int V, const int32_t* visible_count,
                  const int32_t* subset, const int32_t* subset_count, int n_tiles_x, int n_tiles_y,
                  float mh_dist, int tile_row0, int tile_row1, int32_t* workspace,
                  int32_t* tile_ranges, int32_t* host_mirror, void* stream) {
    GS_REQUIRE((subset == nullptr) == (subset_count == nullptr),
               "subset and subset_count go together");
    const Items items{visible_count, subset, subset_count};
    GS_REQUIRE(n_tiles_x > 0 && n_tiles_y > 0, "tile grid must be positive");
    GS_REQUIRE(tile_row0 >= 0 && tile_row1 <= n_tiles_y && tile_row0 <= tile_row1,
               "bad tile row range");
    hipStream_t s = (hipStream_t)stream;
    const int T = n_tiles_x * n_tiles_y;
    const int t0 = tile_row0 * n_tiles_x, Tb = (tile_row1 - tile_row0) * n_tiles_x;   // the rows' tiles
    int32_t* counts = workspace;
    const bool private_hist = use_private(Tb, V);
    if (private_hist) {
        int32_t* hist = workspace + T;
        k_bin_count<<<PRIV_NB, PRIV_BLOCK, sizeof(int) * (size_t)Tb, s>>>(
            (const float*)uvs, (const float*)conic, V, n_tiles_x, n_tiles_y, mh_dist, tile_row0,
            tile_row1, hist, items);
        if (Tb >= 2048) k_bin_colscan<GS_CS_TILES><<<div_up(Tb, GS_CS_TILES), 1024, 0, s>>>(hist, Tb, counts + t0);
        else k_bin_colscan<16><<<div_up(Tb, 16), 1024, 0, s>>>(hist, Tb, counts + t0);
    } else {
        if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)T, s) != hipSuccess) {
            gs::set_error("tile_count: memset failed");
            return GS_EHIP;
        }
        if (V > 0) {
            if (V <= SUBGROUP_SMALL_MAX_V)
                k_tile_count<SUBGROUP_SMALL><<<div_up(V * SUBGROUP_SMALL, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                    (const float*)uvs, (const float*)conic, V, n_tiles_x, n_tiles_y, mh_dist,
                    tile_row0, tile_row1, counts, items);
            else if (V <= SUBGROUP_MID_MAX_V)
                k_tile_count<SUBGROUP_MID><<<div_up(V * SUBGROUP_MID, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                    (const float*)uvs, (const float*)conic, V, n_tiles_x, n_tiles_y, mh_dist,
                    tile_row0, tile_row1, counts, items);
            else
                k_tile_count<1><<<div_up(V, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                    (const float*)uvs, (const float*)conic, V, n_tiles_x, n_tiles_y, mh_dist,
                    tile_row0, tile_row1, counts, items);
        }
    }
    // (the histogram path writes the counts of the rows' tiles only; the atomic path zero-fills all of them)
    k_scan_tiles<<<1, 1024, 0, s>>>(counts, T, tile_ranges, visible_count, private_hist ? t0 : 0, private_hist ? Tb : T,
                                    private_hist ? 0 : 1, host_mirror);
    return check_launch("tile_count");
}

int gs_tile_emit_sort(const void* uvs, const void* xyz_camera_frame, const void* conic, int V,
                      const int32_t* visible_count, const int32_t* subset,
                      const int32_t* subset_count, int n_tiles_x, int n_tiles_y, float mh_dist,
                      int tile_row0, int tile_row1, const int32_t* tile_ranges, int32_t* workspace,
                      uint64_t* keys, int64_t S, int32_t* sorted_gaussians, int sort_prefix,
                      void* stream) {
    return gs_tile_emit_sort_bounded(uvs, xyz_camera_frame, conic, V, visible_count, subset, subset_count, n_tiles_x,
                                     n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, workspace, keys, S,
                                     sorted_gaussians, sort_prefix, -1, stream);
}

int gs_tile_emit_sort_bounded(const void* uvs, const void* xyz_camera_frame, const void* conic, int V,
                              const int32_t* visible_count, const int32_t* subset, const int32_t* subset_count,
                              int n_tiles_x, int n_tiles_y, float mh_dist, int tile_row0, int tile_row1,
                              const int32_t* tile_ranges, int32_t* workspace, uint64_t* keys, int64_t S,
                              int32_t* sorted_gaussians, int sort_prefix, int64_t longest_list, void* stream) {
    GS_REQUIRE(tile_row0 >= 0 && tile_row1 <= n_tiles_y && tile_row0 <= tile_row1,
               "bad tile row range");
    GS_REQUIRE((subset == nullptr) == (subset_count == nullptr),
               "subset and subset_count go together");
    const Items items{visible_count, subset, subset_count};
    GS_REQUIRE(sort_prefix == 0 || sort_prefix == GS_SORT_PREFIX,
               "sort_prefix must be 0 or GS_SORT_PREFIX (%d)", GS_SORT_PREFIX);
    hipStream_t s = (hipStream_t)stream;
    const int T = n_tiles_x * n_tiles_y;
    const int Tb = (tile_row1 - tile_row0) * n_tiles_x;
    if (S <= 0 || V <= 0) return GS_OK;
    if (use_private(Tb, V)) {
        const int32_t* hist = workspace + T;
        k_bin_emit<<<PRIV_NB, PRIV_BLOCK, sizeof(int) * (size_t)Tb, s>>>(
            (const float*)uvs, (const float*)xyz_camera_frame, (const float*)conic, V, n_tiles_x,
            n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, hist, keys, items, S);
    } else {
        int32_t* cursor = workspace;   // zero since gs_tile_count's scan, and again after every emit (k_tile_emit)
        if (V <= SUBGROUP_SMALL_MAX_V)
            k_tile_emit<SUBGROUP_SMALL><<<div_up(V * SUBGROUP_SMALL, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                (const float*)uvs, (const float*)xyz_camera_frame, (const float*)conic, V, n_tiles_x,
                n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, cursor, keys, items, S);
        else if (V <= SUBGROUP_MID_MAX_V)
            k_tile_emit<SUBGROUP_MID><<<div_up(V * SUBGROUP_MID, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                (const float*)uvs, (const float*)xyz_camera_frame, (const float*)conic, V, n_tiles_x,
                n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, cursor, keys, items, S);
        else
            k_tile_emit<1><<<div_up(V, BIN_BLOCK), BIN_BLOCK, 0, s>>>(
                (const float*)uvs, (const float*)xyz_camera_frame, (const float*)conic, V, n_tiles_x,
                n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, cursor, keys, items, S);
    }
    const int t0 = tile_row0 * n_tiles_x;
    const int nt = (tile_row1 - tile_row0) * n_tiles_x;
    launch_tile_sort(tile_ranges, keys, sorted_gaussians, t0, nt, S, sort_prefix, s, longest_list);
    return check_launch("tile_emit_sort");
}

int gs_tile_sort_flagged(const int32_t* tile_ranges, const uint64_t* keys, int64_t S,
                         const int32_t* tile_flags, int n_tiles_x, int tile_row0, int tile_row1,
                         int32_t* sorted_gaussians, void* stream) {
    GS_REQUIRE(tile_flags != nullptr, "tile_flags must not be null");
    GS_REQUIRE(tile_row0 >= 0 && tile_row0 <= tile_row1, "bad tile row range");
    launch_sort_flagged(tile_ranges, keys, sorted_gaussians, tile_row0 * n_tiles_x,
                        (tile_row1 - tile_row0) * n_tiles_x, S, tile_flags, (hipStream_t)stream);
    return check_launch("tile_sort_flagged");
}

// ---- depth-bucketed binning (see "depth cut" above) -------------------------------------------------------
size_t gs_cut_workspace_ints(int n_gaussians, int n_tiles) { return cut_ws_ints(n_gaussians, n_tiles); }

int gs_cut_sample_stride(int n_gaussians) {
    return n_gaussians <= GS_CUT_MAX_SAMPLES ? 1 : (n_gaussians + GS_CUT_MAX_SAMPLES - 1) / GS_CUT_MAX_SAMPLES;
}

int gs_cut_supported(int n_tiles_x, int tile_row0, int tile_row1, int n_gaussians) {
    return use_private((tile_row1 - tile_row0) * n_tiles_x, n_gaussians) ? 1 : 0;
}

int gs_tile_count_cut(const void* bin_records, int N, const int32_t* visible_count, int n_tiles_x, int n_tiles_y,
                      float mh_dist, int tile_row0, int tile_row1, int32_t* workspace, int32_t* cut_workspace,
                      int32_t* tile_ranges, int32_t* full_ranges, int32_t* host_mirror, void* stream) {
    GS_REQUIRE(n_tiles_x > 0 && n_tiles_y > 0, "tile grid must be positive");
    GS_REQUIRE(tile_row0 >= 0 && tile_row1 <= n_tiles_y && tile_row0 <= tile_row1, "bad tile row range");
    GS_REQUIRE(visible_count != nullptr, "visible_count must not be null");
    hipStream_t s = (hipStream_t)stream;
    const int T = n_tiles_x * n_tiles_y;
    const int t0 = tile_row0 * n_tiles_x, Tb = (tile_row1 - tile_row0) * n_tiles_x;
    GS_REQUIRE(use_private(Tb, N), "the depth-bucketed binning needs the LDS-histogram regime (gs_cut_supported)");
    const CutState cs = cut_state_of(cut_workspace, N, T);
    int32_t* counts = workspace;
    int32_t* hist = workspace + T;
    k_depth_hist<<<DC_PART, DC_BLOCK, 0, s>>>(visible_count, cs);
    k_depth_colscan<<<NBK / 64, 1024, 0, s>>>(cs);
    k_depth_scatter<<<DC_PART, DC_BLOCK, 0, s>>>(visible_count, cs);
    k_bin_count_buckets<<<NBK, PRIV_BLOCK, sizeof(int) * (size_t)Tb, s>>>((const float*)bin_records, n_tiles_x, n_tiles_y,
                                                                          mh_dist, tile_row0, tile_row1, hist, cs);
    if (Tb >= 2048)
        k_bin_colscan_cut<GS_CS_TILES><<<div_up(Tb, GS_CS_TILES), 1024, 0, s>>>(hist, Tb, counts + t0, cs.totals + t0,
                                                                                 cs.bstar + t0, GS_SORT_PREFIX);
    else
        k_bin_colscan_cut<16><<<div_up(Tb, 16), 1024, 0, s>>>(hist, Tb, counts + t0, cs.totals + t0, cs.bstar + t0,
                                                              GS_SORT_PREFIX);
    k_scan_tiles_cut<<<1, 1024, 0, s>>>(counts, cs.totals, cs.bstar, T, tile_ranges, full_ranges, visible_count, t0, Tb,
                                        cs.ctrl, host_mirror);
    return check_launch("tile_count_cut");
}

int gs_tile_emit_sort_cut(const void* bin_records, int N, int n_tiles_x, int n_tiles_y, float mh_dist, int tile_row0,
                          int tile_row1, const int32_t* tile_ranges, int32_t* workspace, int32_t* cut_workspace,
                          uint64_t* keys, int64_t S, int32_t* sorted_gaussians, void* stream) {
    GS_REQUIRE(tile_row0 >= 0 && tile_row1 <= n_tiles_y && tile_row0 <= tile_row1, "bad tile row range");
    hipStream_t s = (hipStream_t)stream;
    const int T = n_tiles_x * n_tiles_y;
    const int t0 = tile_row0 * n_tiles_x, Tb = (tile_row1 - tile_row0) * n_tiles_x;
    if (S <= 0 || Tb <= 0) return GS_OK;
    const CutState cs = cut_state_of(cut_workspace, N, T);
    k_bin_emit_buckets<1, GS_CUT_EMIT_BLOCK><<<NBK, GS_CUT_EMIT_BLOCK, sizeof(int) * (size_t)Tb, s>>>(
        (const float*)bin_records, n_tiles_x, n_tiles_y, mh_dist, tile_row0, tile_row1, tile_ranges, workspace + T, keys,
        S, nullptr, cs);
    // every kept list has at most GS_SORT_PREFIX entries, laid out as short runs in depth order
    k_tile_sort_runs<<<Tb, SORT_BLOCK, 0, s>>>(tile_ranges, keys, sorted_gaussians, t0, S);
    return check_launch("tile_emit_sort_cut");
}

// test / tool access to the cut's per-tile results: copies b*(t), n(t) (device -> device) and the bucket bounds
int gs_cut_debug_views(int32_t* cut_workspace, int N, int n_tiles, int32_t** bstar, int32_t** totals, uint32_t** bounds,
                       int32_t** bucket_offsets, int32_t** ctrl) {
    const CutState cs = cut_state_of(cut_workspace, N, n_tiles);
    if (bstar) *bstar = cs.bstar;
    if (totals) *totals = cs.totals;
    if (bounds) *bounds = cs.bounds;
    if (bucket_offsets) *bucket_offsets = cs.boff2;
    if (ctrl) *ctrl = cs.ctrl;
    return GS_OK;
}

}  // extern "C"
