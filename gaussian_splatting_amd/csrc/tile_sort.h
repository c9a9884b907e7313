// tile_sort.h -- per-tile sort of 64-bit tile-list keys in LDS (register-blocked bitonic network) and in global
// memory: shared by the binning kernels (binning.hip "per-tile sort") and the repair kernel of the fused renderer
// (render.hip: a flagged tile is sorted in full and rendered again by one workgroup).
#pragma once
#include "gs_common.h"

namespace gs {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_R = 16;
static_assert(SORT_BLOCK == 256, "the radix-select histogram has one bin per thread");

__device__ inline int slot(int i) { return i + (i >> 4); }

#define GS_CE(x, y)                                                                                \
    do {                                                                                           \
        const uint64_t _a = (x), _b = (y);                                                         \
        const bool _sw = _a > _b;                                                                  \
        (x) = _sw ? _b : _a;                                                                       \
        (y) = _sw ? _a : _b;                                                                       \
    } while (0)

// half-cleaners of stride J, J/2, ..., 1 on 16 registers
template <int J>
__device__ inline void reg_half_cleaners(uint64_t (&v)[SORT_R]) {
#pragma unroll
    for (int j = J; j > 0; j >>= 1) {
#pragma unroll
        for (int p = 0; p < SORT_R / 2; p++) {
            const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1));
            GS_CE(v[lo], v[lo + j]);
        }
    }
}

// all phases k = 2 .. 16 on 16 registers
__device__ inline void reg_sort16(uint64_t (&v)[SORT_R]) {
#pragma unroll
    for (int k = 2; k <= SORT_R; k <<= 1) {
        const int h = k >> 1;
#pragma unroll
        for (int p = 0; p < SORT_R / 2; p++) {   // flip
            const int blk = p / h, off = p & (h - 1);
            GS_CE(v[blk * k + off], v[blk * k + (k - 1 - off)]);
        }
#pragma unroll
        for (int j = k >> 2; j > 0; j >>= 1) {
#pragma unroll
            for (int p = 0; p < SORT_R / 2; p++) {
                const int lo = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                GS_CE(v[lo], v[lo + j]);
            }
        }
    }
}

template <bool FULL_SORT>
__device__ __forceinline__ void register_pass(uint64_t* s, int n, int n_pad, int tid) {
    for (int blk = tid; blk < (n_pad >> 4); blk += SORT_BLOCK) {
        const int base = blk << 4;
        if (base >= n) continue;   // an all-padding block is already in order
        uint64_t v[SORT_R];
        const int sb = slot(base);
#pragma unroll
        for (int e = 0; e < SORT_R; e++) v[e] = (base + e < n) ? s[sb + e] : ~0ull;
        if (FULL_SORT) reg_sort16(v);
        else reg_half_cleaners<SORT_R / 2>(v);
#pragma unroll
        for (int e = 0; e < SORT_R; e++)
            if (base + e < n) s[sb + e] = v[e];
    }
}

// one pair-wise step on LDS (flip when j == 0), four pairs in flight per thread
__device__ __forceinline__ void pair_step_lds(uint64_t* s, int n, int n_pad, int k, int j, int tid) {
    const int pairs = n_pad >> 1;
    for (int p0 = tid; p0 < pairs; p0 += 4 * SORT_BLOCK) {
        int lo[4], hi[4];
        uint64_t a[4], b[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * SORT_BLOCK;
            if (j == 0) {
                const int h = k >> 1;
                const int blk = p / h, off = p & (h - 1);
                lo[u] = blk * k + off;
                hi[u] = blk * k + (k - 1 - off);
            } else {
                lo[u] = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                hi[u] = lo[u] + j;
            }
            ok[u] = p < pairs && hi[u] < n;
            if (ok[u]) {
                a[u] = s[slot(lo[u])];
                b[u] = s[slot(hi[u])];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (ok[u] && a[u] > b[u]) {
                s[slot(lo[u])] = b[u];
                s[slot(hi[u])] = a[u];
            }
        }
    }
}

// sorts s[slot(0..n)) ascending; every thread of the workgroup calls it, keys already in LDS
__device__ __forceinline__ void lds_bitonic_sort(uint64_t* s_keys, int n, int tid) {
    int n_pad = SORT_R;
    while (n_pad < n) n_pad <<= 1;
    register_pass<true>(s_keys, n, n_pad, tid);
    __syncthreads();
    for (int k = 2 * SORT_R; k <= n_pad; k <<= 1) {
        pair_step_lds(s_keys, n, n_pad, k, 0, tid);
        __syncthreads();
        for (int j = k >> 2; j >= SORT_R; j >>= 1) {
            pair_step_lds(s_keys, n, n_pad, k, j, tid);
            __syncthreads();
        }
        register_pass<false>(s_keys, n, n_pad, tid);
        __syncthreads();
    }
}

// pair-wise bitonic network on global memory, in place (lists beyond the LDS classes)
__device__ __forceinline__ void global_sort_tile(uint64_t* gk, int* __restrict__ sorted, int n, int tid) {
    int n_pad = 2;
    while (n_pad < n) n_pad <<= 1;
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = 0, first = 1; first || j > 0; first = 0) {
            for (int p = tid; p < (n_pad >> 1); p += SORT_BLOCK) {
                int lo, hi;
                if (j == 0) {
                    const int h = k >> 1;
                    const int blk = p / h, off = p & (h - 1);
                    lo = blk * k + off;
                    hi = blk * k + (k - 1 - off);
                } else {
                    lo = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    hi = lo + j;
                }
                if (hi >= n) continue;
                const uint64_t a = gk[lo], b = gk[hi];
                if (a > b) {
                    gk[lo] = b;
                    gk[hi] = a;
                }
            }
            __threadfence_block();
            __syncthreads();
            j = (j == 0) ? (k >> 2) : (j >> 1);
        }
    }
    for (int i = tid; i < n; i += SORT_BLOCK) sorted[i] = (int)(uint32_t)gk[i];
}

inline size_t sort_lds_bytes(int cap) { return (size_t)(cap + cap / SORT_R) * sizeof(uint64_t); }

}  // namespace gs
