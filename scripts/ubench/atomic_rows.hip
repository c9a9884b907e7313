// atomic_rows.hip -- how should a 36-byte row of nine floats be added into a [V, 9] slab with global atomics?
// R rows at random positions of a 2.75 M-row slab (the render backward's flush: 1.25 M rows per frame at D).
//   A  one thread per row, nine atomic instructions (lanes of one instruction hit 64 different rows)
//   B  nine lanes per row: one atomic instruction covers seven whole rows (lanes hit consecutive addresses)
//   C  as A with plain stores (the bandwidth reference)
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_rows atomic_rows.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_a(float* slab, const int* rows, const float* vals, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float* dst = slab + (size_t)rows[r] * 9;
#pragma unroll
    for (int j = 0; j < 9; j++) unsafeAtomicAdd(dst + j, vals[(size_t)r * 9 + j]);
}
__global__ void k_b(float* slab, const int* rows, const float* vals, int R) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane >= 63) return;
    const int r = wave * 7 + lane / 9, col = lane % 9;
    if (r >= R) return;
    unsafeAtomicAdd(slab + (size_t)rows[r] * 9 + col, vals[(size_t)r * 9 + col]);
}
__global__ void k_c(float* slab, const int* rows, const float* vals, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float* dst = slab + (size_t)rows[r] * 9;
#pragma unroll
    for (int j = 0; j < 9; j++) dst[j] = vals[(size_t)r * 9 + j];
}

int main() {
    const int V = 2750000, R = 1250000;
    std::vector<int> rows(R);
    srand(1);
    for (int i = 0; i < R; i++) rows[i] = (int)(((long long)rand() * 32768 + rand()) % V);
    float *slab, *vals;
    int* d_rows;
    CHECK(hipMalloc(&slab, (size_t)V * 36));
    CHECK(hipMalloc(&vals, (size_t)R * 36));
    CHECK(hipMalloc(&d_rows, (size_t)R * 4));
    CHECK(hipMemset(slab, 0, (size_t)V * 36));
    CHECK(hipMemset(vals, 0, (size_t)R * 36));
    CHECK(hipMemcpy(d_rows, rows.data(), (size_t)R * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int which = 0; which < 3; which++) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; rep++) {
            CHECK(hipEventRecord(a));
            if (which == 0) k_a<<<(R + 255) / 256, 256>>>(slab, d_rows, vals, R);
            if (which == 1) k_b<<<((R + 6) / 7 * 64 + 255) / 256, 256>>>(slab, d_rows, vals, R);
            if (which == 2) k_c<<<(R + 255) / 256, 256>>>(slab, d_rows, vals, R);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%s: %.4f ms for %d rows of 9 floats (%.1f G lane-atomics/s)\n",
               which == 0 ? "A thread per row, 9 atomic instructions  " : which == 1 ? "B nine lanes per row, 1 atomic instruction" :
               "C thread per row, plain stores            ", best, R, 9.0 * R / best / 1e6);
    }
    return 0;
}
