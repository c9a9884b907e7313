// halo.hip -- multi-GPU bookkeeping for tile-row sharded frames with owner-sliced gradients
// (gaussian_splatting_amd/sharded.py; BASELINE.json north_star).
//
// Rank b renders the band of tile rows [row[b], row[b+1]) and its render backward yields PARTIAL
// sums of the 9 render gradients (rgb 3 | opacity 1 | uv 2 | conic 3) for every Gaussian that
// reaches its band.  Rank r owns the Gaussians of an index slice (multiples of 256) and runs the
// per-Gaussian backward for them only, so it needs, for each of its Gaussians, the partial rows
// of exactly those ranks whose band the Gaussian reaches.  With replicated parameters every rank
// can work that out for itself: bit s of mask[v] says whether visible Gaussian v reaches band s
// (candidate tile window of tile_culling.cu:138-156 against the band, a superset of the tiles the
// binning accepts, so a row that is not exchanged is provably zero).  From the masks:
//   send list of rank b   {v : bit b}, ascending v, i.e. grouped by owner because the compaction of
//                         the frustum cull preserves the Gaussian order; the slice of the list that
//                         goes to owner r has send_cnt[r] rows
//   receive layout        from sender s: {v in my visible range : bit s}, ascending, recv_cnt[s]
//                         rows; row of v sits at P_s(v) - P_s(v_lo) with P_s the exclusive prefix
//                         count of bit s
// so one all_to_all of ~36 B x (Gaussians reaching the band) replaces the all-reduce of 36 B x V.
//   k_halo_mask        mask[v] + per-256-block counts of every bit
//   k_halo_scan        exclusive prefix of the block counts, one workgroup per bit
//   k_halo_bounds      visible-index bounds of the owner slices, P_s at those bounds, the split
//                      sizes, and the frame's host-read record (rows to send, V, v_lo, v_hi, send,
//                      recv)
//   k_halo_send_index  the send list
//   k_halo_gather_sum  out[v - v_lo] = sum over senders of the received rows of v (no atomics)
#include "tile_math.h"

namespace gs {

constexpr int HB = 256;   // == PP_BLOCK of preprocess.hip: owner slices are whole blocks of it

struct RankInts {
    int v[GS_MAX_RANKS + 1];
};

__device__ inline uint32_t band_mask_of(const float* __restrict__ uvs,
                                        const float* __restrict__ conic, int g, int ntx, int nty,
                                        float mh, const RankInts& rows, int G) {
    const float u = uvs[g * 2], v = uvs[g * 2 + 1];
    const float a = conic[g * 3] + 0.25f;
    const float b = conic[g * 3 + 1] / 2.0f;
    const float c = conic[g * 3 + 2] + 0.25f;
    const Obb o = compute_obb(u, v, a, b, c, mh);
    const Window w = candidate_window(u, v, o.radius_tiles, ntx, nty, 0, nty);
    uint32_t m = 0;
    if (w.sx < w.ex) {
        for (int s = 0; s < G; s++)
            if (w.sy < rows.v[s + 1] && w.ey > rows.v[s] && w.sy < w.ey) m |= 1u << s;
    }
    return m;
}

// per-bit counts of one 256-block; result valid in threads [0, G)
__device__ inline int block_bit_count(uint32_t m, int G, int tid, int (*s_cnt)[HB / GS_WAVE]) {
    for (int s = 0; s < G; s++) {
        const int n = __popcll(__ballot((m >> s) & 1u));
        if ((tid & 63) == 0) s_cnt[s][tid >> 6] = n;
    }
    __syncthreads();
    return tid < G ? s_cnt[tid][0] + s_cnt[tid][1] + s_cnt[tid][2] + s_cnt[tid][3] : 0;
}

__global__ __launch_bounds__(HB) void k_halo_mask(const float* __restrict__ uvs,
                                                  const float* __restrict__ conic, int N,
                                                  const int* __restrict__ visible_count, int ntx,
                                                  int nty, float mh, RankInts rows, int G,
                                                  uint32_t* __restrict__ mask,
                                                  int* __restrict__ blk_counts, int nblk) {
    __shared__ int s_cnt[GS_MAX_RANKS][HB / GS_WAVE];
    const int v = blockIdx.x * HB + threadIdx.x;
    uint32_t m = 0;
    if (v < *visible_count) m = band_mask_of(uvs, conic, v, ntx, nty, mh, rows, G);
    if (v < N) mask[v] = m;
    const int n = block_bit_count(m, G, threadIdx.x, s_cnt);
    if (threadIdx.x < G) blk_counts[threadIdx.x * nblk + blockIdx.x] = n;
}

// row s of counts -> exclusive prefix in row s of offsets; one workgroup of 1024 per row
__global__ __launch_bounds__(1024) void k_halo_scan(const int* __restrict__ counts, int n,
                                                    int* __restrict__ offsets) {
    __shared__ int s_wave[16];
    __shared__ int s_carry;
    counts += (size_t)blockIdx.x * n;
    offsets += (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
}

// exclusive prefix count of bit s at visible index v: whole blocks from offsets, the rest counted
// by the calling wave
__device__ inline int prefix_at(const uint32_t* __restrict__ mask, const int* __restrict__ offsets,
                                int nblk, int s, int v, int lane) {
    const int blk = min(v / HB, nblk - 1);   // v == nblk * HB: the last block counted in full
    int n = 0;
    for (int i = blk * HB + lane; i < v; i += GS_WAVE) n += (mask[i] >> s) & 1u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) n += __shfl_xor(n, d);
    return offsets[s * nblk + blk] + n;
}

// wave r handles boundary r.  pre_offsets: exclusive visible-count prefix per 256-block of the
// Gaussian index (gs_preprocess_forward's workspace), so the visible index of the first Gaussian
// of block k is pre_offsets[k].
// (device function: runs as the one workgroup of k_halo_bounds, or as the extra last workgroup of
// k_halo_send_index -- both only need the scan -- with n_waves waves taking the boundaries round-robin)
__device__ inline void halo_bounds_block(
    const uint32_t* __restrict__ mask, const int* __restrict__ offsets, int nblk,
    const int* __restrict__ visible_count, const int* __restrict__ pre_offsets, const RankInts& owner_blk,
    int G, int me, int* __restrict__ vb /*[G+1]*/,
    int* __restrict__ Pb /*[G][G+1]*/, int* __restrict__ plan /*[4+2G]*/, int n_waves,
    int* s_vb /*[GS_MAX_RANKS + 1]*/, int (*s_P)[GS_MAX_RANKS + 1], int* __restrict__ plan_host = nullptr) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int V = *visible_count;
    // Two rounds of loads for the whole workgroup: (1) every boundary's visible index, (2) per boundary the block of
    // mask words it cuts (the same words for all s) and lane s's scanned offset -- a wave issues the loads of ALL
    // its boundaries before it reduces any (dependent rounds of ~2 us each were what this workgroup's time was).
    constexpr int MAXB = (GS_MAX_RANKS + 1 + 3) / 4;   // boundaries per wave with four waves
    int bnd[MAXB], off[MAXB];
    uint32_t words[MAXB][HB / GS_WAVE];
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int r = wave + j * n_waves;
        const int ob = owner_blk.v[min(r, G)];   // (r > G: no boundary, the value is not used)
        bnd[j] = (r > G) ? 0 : ((r == G || ob >= nblk) ? V : min(V, pre_offsets[ob]));
    }
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int r = wave + j * n_waves;
        const int b = bnd[j];
        const int blk = min(b / HB, nblk - 1);   // b == nblk * HB: the last block counted in full
#pragma unroll
        for (int k = 0; k < HB / GS_WAVE; k++) {
            const int i = blk * HB + k * GS_WAVE + lane;
            words[j][k] = (r <= G && i < b) ? mask[i] : 0u;
        }
        off[j] = (r <= G && lane < G) ? offsets[lane * nblk + blk] : 0;
    }
#pragma unroll
    for (int j = 0; j < MAXB; j++) {
        const int r = wave + j * n_waves;
        if (r > G) continue;   // wave-uniform
        if (lane == 0) {
            s_vb[r] = bnd[j];
            vb[r] = bnd[j];
        }
        for (int s = 0; s < G; s++) {   // = prefix_at(mask, offsets, nblk, s, b, lane)
            int n = 0;
#pragma unroll
            for (int k = 0; k < HB / GS_WAVE; k++) n += (words[j][k] >> s) & 1u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) n += __shfl_xor(n, d);
            const int p = __shfl(off[j], s) + n;
            if (lane == 0) {
                s_P[s][r] = p;
                Pb[s * (G + 1) + r] = p;
            }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t == 0) {
        plan[0] = s_P[me][G];   // rows of the send list: every visible Gaussian that reaches my band
        plan[1] = V;
        plan[2] = s_vb[me];
        plan[3] = s_vb[me + 1];
    }
    if (t < G) {
        plan[4 + t] = s_P[me][t + 1] - s_P[me][t];       // rows I send to owner t
        plan[4 + G + t] = s_P[t][me + 1] - s_P[t][me];   // rows I receive from sender t
    }
    // the same record straight into the caller's pinned host buffer (the frame's host read without a copy kernel
    // in the stream; the caller waits for an event behind a LATER kernel of the stream)
    if (plan_host != nullptr) {
        if (t == 0) {
            plan_host[0] = s_P[me][G];
            plan_host[1] = V;
            plan_host[2] = s_vb[me];
            plan_host[3] = s_vb[me + 1];
        }
        if (t < G) {
            plan_host[4 + t] = s_P[me][t + 1] - s_P[me][t];
            plan_host[4 + G + t] = s_P[t][me + 1] - s_P[t][me];
        }
        __threadfence_system();
    }
}

__global__ __launch_bounds__(GS_WAVE*(GS_MAX_RANKS + 1)) void k_halo_bounds(
    const uint32_t* __restrict__ mask, const int* __restrict__ offsets, int nblk,
    const int* __restrict__ visible_count, const int* __restrict__ pre_offsets, RankInts owner_blk,
    int G, int me, int* __restrict__ vb /*[G+1]*/,
    int* __restrict__ Pb /*[G][G+1]*/, int* __restrict__ plan /*[4+2G]*/) {
    __shared__ int s_vb[GS_MAX_RANKS + 1];
    __shared__ int s_P[GS_MAX_RANKS][GS_MAX_RANKS + 1];
    halo_bounds_block(mask, offsets, nblk, visible_count, pre_offsets, owner_blk, G, me, vb, Pb, plan,
                      GS_MAX_RANKS + 1, s_vb, s_P);
}

// gs_halo_plan_masked's form: workgroups [0, nblk) build the send list, workgroup nblk the bounds and the plan
__global__ __launch_bounds__(HB) void k_halo_send_index_bounds(
    const uint32_t* __restrict__ mask, const int* __restrict__ offsets, int nblk,
    const int* __restrict__ visible_count, const int* __restrict__ pre_offsets, RankInts owner_blk, int G, int me,
    int* __restrict__ send_index, int* __restrict__ vb, int* __restrict__ Pb, int* __restrict__ plan,
    int* __restrict__ plan_host) {
    __shared__ int s_cnt[HB / GS_WAVE];
    __shared__ int s_vb[GS_MAX_RANKS + 1];
    __shared__ int s_P[GS_MAX_RANKS][GS_MAX_RANKS + 1];
    if ((int)blockIdx.x == nblk) {
        halo_bounds_block(mask, offsets, nblk, visible_count, pre_offsets, owner_blk, G, me, vb, Pb, plan,
                          HB / GS_WAVE, s_vb, s_P, plan_host);
        return;
    }
    const int v = blockIdx.x * HB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool bit = v < *visible_count && ((mask[v] >> me) & 1u);
    const unsigned long long bal = __ballot(bit);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (!bit) return;
    int pos = offsets[me * nblk + blockIdx.x];
    for (int w = 0; w < wave; w++) pos += s_cnt[w];
    pos += __popcll(bal & ((1ull << lane) - 1));
    send_index[pos] = v;
}

__global__ __launch_bounds__(HB) void k_halo_send_index(const uint32_t* __restrict__ mask,
                                                        const int* __restrict__ offsets, int nblk,
                                                        const int* __restrict__ visible_count,
                                                        int me, int* __restrict__ send_index) {
    __shared__ int s_cnt[HB / GS_WAVE];
    const int v = blockIdx.x * HB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool bit = v < *visible_count && ((mask[v] >> me) & 1u);
    const unsigned long long bal = __ballot(bit);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (!bit) return;
    int pos = offsets[me * nblk + blockIdx.x];
    for (int w = 0; w < wave; w++) pos += s_cnt[w];
    pos += __popcll(bal & ((1ull << lane) - 1));
    send_index[pos] = v;
}

// grid: the 256-blocks of the visible index that intersect [v_lo, v_hi)
__global__ __launch_bounds__(HB) void k_halo_gather_sum(
    const uint32_t* __restrict__ mask, const int* __restrict__ offsets, int nblk,
    const int* __restrict__ Pb, int G, int me, int v_lo, int v_hi, const float* __restrict__ recv,
    RankInts recv_off, float* __restrict__ out) {
    __shared__ int s_cnt[GS_MAX_RANKS][HB / GS_WAVE];
    const int blk = v_lo / HB + blockIdx.x;
    const int v = blk * HB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool mine = v >= v_lo && v < v_hi;
    const uint32_t m = (v < v_hi) ? mask[v] : 0;   // the prefix counts every v of the block below v_hi
    unsigned long long bal[GS_MAX_RANKS];
    for (int s = 0; s < G; s++) {
        bal[s] = __ballot((m >> s) & 1u);
        if (lane == 0) s_cnt[s][wave] = __popcll(bal[s]);
    }
    __syncthreads();
    if (!mine) return;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < G; s++) {
        if (((m >> s) & 1u) == 0) continue;
        int p = offsets[s * nblk + blk];
        for (int w = 0; w < wave; w++) p += s_cnt[s][w];
        p += __popcll(bal[s] & ((1ull << lane) - 1));
        const float* row = recv + (size_t)(recv_off.v[s] + p - Pb[s * (G + 1) + me]) * 9;
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] += row[j];
    }
    float* o = out + (size_t)(v - v_lo) * 9;
#pragma unroll
    for (int j = 0; j < 9; j++) o[j] = acc[j];
}

// ---- cost-balanced bands (sharded.py band_policy "cost") -----------------------------------
// Bands of unequal height cannot be all-gathered in place: rank r contributes a chunk of chunk_rows pixel rows
// (16 x the tallest band + 1) that starts at its band; the chunk's LAST row carries, as floats, the cost of every
// tile row of the rank's band (0 elsewhere) -- what the next frame's bands are balanced on.
//   k_band_row_costs   cost[r] = sum over the row's tiles of min(list length, cap) + tile_cost per tile
//   k_band_assemble    gathered chunks -> the frame (pixel row y comes from the band that owns tile row y / 16);
//                      its last workgroup sums the cost rows straight into the caller's pinned host buffer
__global__ __launch_bounds__(GS_WAVE) void k_band_row_costs(const int* __restrict__ ranges, int ntx, int row0, int row1,
                                                            int tile_cost, int cap, float* __restrict__ cost_row) {
    const int r = blockIdx.x, lane = threadIdx.x;
    int sum = 0;
    if (r >= row0 && r < row1) {
        for (int t = lane; t < ntx; t += GS_WAVE) sum += min(ranges[r * ntx + t + 1] - ranges[r * ntx + t], cap);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d);
        sum += tile_cost * ntx;
    }
    if (lane == 0) cost_row[r] = (float)sum;
}

__global__ __launch_bounds__(HB) void k_band_assemble(const float* __restrict__ gathered, RankInts bounds, int G,
                                                      int chunk_rows, int row_floats, int H, int R,
                                                      float* __restrict__ image, float* __restrict__ host_costs) {
    const int y = blockIdx.x;
    if (y == H) {   // the extra workgroup: next frame's row costs
        for (int q = threadIdx.x; q < R; q += HB) {
            float c = 0;
            for (int r = 0; r < G; r++) c += gathered[((size_t)r * chunk_rows + chunk_rows - 1) * row_floats + q];
            host_costs[q] = c;
        }
        __threadfence_system();
        return;
    }
    const int ty = y >> 4;
    int r = 0;
    while (r + 1 < G && ty >= bounds.v[r + 1]) r++;
    const float* src = gathered + ((size_t)r * chunk_rows + (y - 16 * bounds.v[r])) * row_floats;
    float* dst = image + (size_t)y * row_floats;
    for (int i = threadIdx.x; i < row_floats; i += HB) dst[i] = src[i];
}

static RankInts rank_ints(const int32_t* a, int n) {
    RankInts r;
    for (int i = 0; i <= GS_MAX_RANKS; i++) r.v[i] = i < n ? a[i] : 0;
    return r;
}

}  // namespace gs

using namespace gs;

extern "C" {

// layout: blk_counts[G*nblk] | offsets[G*nblk] | vb[G+1] | Pb[G*(G+1)]
size_t gs_halo_workspace_ints(int N, int G) {
    const size_t nblk = (size_t)div_up(N > 0 ? N : 1, HB);
    return 2 * (size_t)G * nblk + (size_t)(G + 1) * (G + 1);
}

int gs_halo_plan(const void* uvs, const void* conic, int N, const int32_t* visible_count,
                 const int32_t* preprocess_workspace, int n_tiles_x, int n_tiles_y, float mh_dist,
                 const int32_t* band_rows, const int32_t* owner_blocks, int G, int rank,
                 uint32_t* mask, int32_t* workspace, int32_t* send_index, int32_t* plan,
                 void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "halo_plan: 1 <= G <= %d", GS_MAX_RANKS);
    GS_REQUIRE(rank >= 0 && rank < G, "halo_plan: bad rank");
    GS_REQUIRE(N > 0, "halo_plan: N must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = div_up(N, HB);
    int32_t* blk_counts = workspace;
    int32_t* offsets = workspace + (size_t)G * nblk;
    int32_t* vb = offsets + (size_t)G * nblk;
    int32_t* Pb = vb + (G + 1);
    // gs_preprocess_forward's workspace: counts[nblk] | offsets[nblk]
    const int32_t* pre_offsets = preprocess_workspace + nblk;
    k_halo_mask<<<nblk, HB, 0, s>>>((const float*)uvs, (const float*)conic, N, visible_count,
                                    n_tiles_x, n_tiles_y, mh_dist, rank_ints(band_rows, G + 1), G,
                                    mask, blk_counts, nblk);
    k_halo_scan<<<G, 1024, 0, s>>>(blk_counts, nblk, offsets);
    k_halo_bounds<<<1, GS_WAVE*(GS_MAX_RANKS + 1), 0, s>>>(
        mask, offsets, nblk, visible_count, pre_offsets, rank_ints(owner_blocks, G + 1), G, rank,
        vb, Pb, plan);
    k_halo_send_index<<<nblk, HB, 0, s>>>(mask, offsets, nblk, visible_count, rank, send_index);
    return check_launch("halo_plan");
}

int gs_halo_plan_masked(const uint32_t* mask, int N, const int32_t* visible_count,
                        const int32_t* preprocess_workspace, const int32_t* owner_blocks, int G, int rank,
                        int32_t* workspace, int32_t* send_index, int32_t* plan, int32_t* plan_host, void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "halo_plan: 1 <= G <= %d", GS_MAX_RANKS);
    GS_REQUIRE(rank >= 0 && rank < G, "halo_plan: bad rank");
    GS_REQUIRE(N > 0, "halo_plan: N must be positive");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = div_up(N, HB);
    int32_t* blk_counts = workspace;   // filled by gs_band_project
    int32_t* offsets = workspace + (size_t)G * nblk;
    int32_t* vb = offsets + (size_t)G * nblk;
    int32_t* Pb = vb + (G + 1);
    const int32_t* pre_offsets = preprocess_workspace + nblk;
    k_halo_scan<<<G, 1024, 0, s>>>(blk_counts, nblk, offsets);
    k_halo_send_index_bounds<<<nblk + 1, HB, 0, s>>>(mask, offsets, nblk, visible_count, pre_offsets,
                                                     rank_ints(owner_blocks, G + 1), G, rank, send_index, vb, Pb, plan,
                                                     plan_host);
    return check_launch("halo_plan_masked");
}

int gs_band_row_costs(const int32_t* tile_ranges, int n_tiles_x, int n_tile_rows, int tile_row0, int tile_row1,
                      int tile_cost, int cap, void* cost_row, void* stream) {
    GS_REQUIRE(n_tiles_x > 0 && n_tile_rows > 0 && tile_row0 >= 0 && tile_row0 <= tile_row1 && tile_row1 <= n_tile_rows,
               "band_row_costs: bad tile rows");
    k_band_row_costs<<<n_tile_rows, GS_WAVE, 0, (hipStream_t)stream>>>(tile_ranges, n_tiles_x, tile_row0, tile_row1, tile_cost,
                                                                       cap, (float*)cost_row);
    return check_launch("band_row_costs");
}

int gs_band_assemble(const void* gathered, const int32_t* band_rows, int G, int chunk_rows, int W, int H, int n_tile_rows,
                     void* image, void* host_costs, void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "band_assemble: 1 <= G <= %d", GS_MAX_RANKS);
    GS_REQUIRE(W > 0 && H > 0 && chunk_rows > 0 && n_tile_rows <= 3 * W, "band_assemble: bad sizes");
    GS_REQUIRE(host_costs != nullptr, "band_assemble: host_costs must not be null");
    k_band_assemble<<<H + 1, HB, 0, (hipStream_t)stream>>>((const float*)gathered, rank_ints(band_rows, G + 1), G, chunk_rows,
                                                           3 * W, H, n_tile_rows, (float*)image, (float*)host_costs);
    return check_launch("band_assemble");
}

int gs_halo_gather_sum(const uint32_t* mask, const int32_t* workspace, int N, int G, int rank,
                       int v_lo, int v_hi, const void* recv, const int32_t* recv_offsets,
                       void* out, void* stream) {
    GS_REQUIRE(G >= 1 && G <= GS_MAX_RANKS, "halo_gather_sum: 1 <= G <= %d", GS_MAX_RANKS);
    GS_REQUIRE(v_lo >= 0 && v_lo <= v_hi, "halo_gather_sum: bad visible range");
    if (v_hi == v_lo) return GS_OK;
    const int nblk = div_up(N, HB);
    const int32_t* offsets = workspace + (size_t)G * nblk;
    const int32_t* Pb = offsets + (size_t)G * nblk + (G + 1);
    const int grid = (v_hi - 1) / HB - v_lo / HB + 1;
    k_halo_gather_sum<<<grid, HB, 0, (hipStream_t)stream>>>(
        mask, offsets, nblk, Pb, G, rank, v_lo, v_hi, (const float*)recv,
        rank_ints(recv_offsets, G), (float*)out);
    return check_launch("halo_gather_sum");
}

}  // extern "C"
