// lds_int_atomics.hip -- what an integer LDS atomic costs on gfx950, per CU, by kind and by active lanes.
// The tile count pass (csrc/binning.hip: k_bin_count*, one ds_add_u32 per (Gaussian, tile) hit into a per-workgroup
// histogram) was priced in round 6 with the figure measured for ds_add_f32 (scripts/ubench/valu_rate.hip: ~3.5 cycles
// per ACTIVE LANE per CU).  This measures the integer forms it actually issues and the ones it could issue instead:
//   ds_add_u32 (no return), ds_add_rtn_u32 (the emit pass's cursors), ds_add_u64 (four 16-bit counters per atomic),
//   with 64 / 16 / 4 / 1 active lanes and scattered addresses in a 32 KiB histogram, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-atomic-optimizer-strategy=None -o lds_int_atomics lds_int_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int HIST = 8192;   // 32 KiB of 32-bit counters

// KIND 0: ds_add_u32   1: ds_add_rtn_u32   2: ds_add_u64 (16 KiB of 64-bit words, the same 32 KiB)   3: ds_write_b32
template <int KIND, int ACTIVE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    __shared__ unsigned long long hist64[HIST / 2];
    unsigned* hist = reinterpret_cast<unsigned*>(hist64);
    for (int i = threadIdx.x; i < HIST; i += 256) hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = ACTIVE == 64 ? true : ACTIVE == 16 ? (lane & 3) == 3 : ACTIVE == 4 ? (lane & 15) == 15 : lane == 63;
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 9176u;
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;   // (2 vector instructions per atomic: the address)
            if (on) {
                if constexpr (KIND == 0) {
                    __hip_atomic_fetch_add(&hist[(x >> 8) & (HIST - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if constexpr (KIND == 1) {
                    acc += __hip_atomic_fetch_add(&hist[(x >> 8) & (HIST - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if constexpr (KIND == 2) {
                    __hip_atomic_fetch_add(&hist64[(x >> 8) & (HIST / 2 - 1)], 0x0001000000010001ull + ((unsigned long long)(x & 1) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    hist[(x >> 8) & (HIST - 1)] = x;
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc + hist[threadIdx.x] + x;
}

template <int KIND, int ACTIVE> int run(const char* name) {
    const int iters = 2000, wg_per_cu = 8;
    const int grid = 256 * wg_per_cu;
    unsigned* out;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<KIND, ACTIVE><<<grid, 256>>>(out, 10, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<KIND, ACTIVE><<<grid, 256>>>(out, iters, 1u);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // wave-level atomic instructions per CU: 8 per trip, 4 waves per workgroup, wg_per_cu workgroups per CU
    const double inst_per_cu = (double)iters * 8 * 4 * wg_per_cu;
    const double ns = ms * 1e6 / inst_per_cu;
    printf("%-18s %2d active lanes  %8.3f ms  %7.2f ns per wave-instruction per CU = %6.1f cycles at 2.4 GHz = %5.2f per active lane\n",
           name, ACTIVE, ms, ns, ns * 2.4, ns * 2.4 / ACTIVE);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<0, 64>("ds_add_u32"); run<0, 16>("ds_add_u32"); run<0, 4>("ds_add_u32"); run<0, 1>("ds_add_u32");
    run<1, 64>("ds_add_rtn_u32"); run<1, 16>("ds_add_rtn_u32"); run<1, 4>("ds_add_rtn_u32"); run<1, 1>("ds_add_rtn_u32");
    run<2, 64>("ds_add_u64"); run<2, 16>("ds_add_u64"); run<2, 4>("ds_add_u64"); run<2, 1>("ds_add_u64");
    run<3, 64>("ds_write_b32"); run<3, 16>("ds_write_b32"); run<3, 4>("ds_write_b32"); run<3, 1>("ds_write_b32");
    return 0;
}
