// valu_rate.hip -- issue-rate microbenchmark for the instruction kinds the render kernels are made of.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_rate valu_rate.hip ; run on the GPU box.
// Every kernel runs `iters` trips of an unrolled body of independent chains on 256 CUs x `waves` waves
// per SIMD; prints cycles per wave-instruction per SIMD at the measured clock-free rate
// (instructions / s / SIMD -> ns per instruction) so that the numbers do not depend on DVFS guesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float2v pm = {m, m}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (KIND == 0) {          // v_fma_f32 x8
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
                a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
            } else if constexpr (KIND == 1) {   // v_pk_fma_f32 x4 (8 fmas)
                p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
                p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
            } else if constexpr (KIND == 2) {   // v_add_f32_dpp row_shr:1 x8
#define DPPADD(x) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true))
                DPPADD(a0); DPPADD(a1); DPPADD(a2); DPPADD(a3); DPPADD(a4); DPPADD(a5); DPPADD(a6); DPPADD(a7);
            } else if constexpr (KIND == 3) {   // v_mul_f32 x8
                a0 *= m; a1 *= m; a2 *= m; a3 *= m; a4 *= m; a5 *= m; a6 *= m; a7 *= m;
            } else if constexpr (KIND == 4) {   // v_rcp_f32 x8
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
            } else if constexpr (KIND == 5) {   // v_cndmask x8 (select on a lane-varying condition)
                const bool q = ((threadIdx.x + i) >> u) & 1;
                a0 = q ? a0 : a1; a1 = q ? a1 : a2; a2 = q ? a2 : a3; a3 = q ? a3 : a4;
                a4 = q ? a4 : a5; a5 = q ? a5 : a6; a6 = q ? a6 : a7; a7 = q ? a7 : a0;
            } else if constexpr (KIND == 6) {   // v_pk_mul_f32 x4
                p0 *= pm; p1 *= pm; p2 *= pm; p3 *= pm;
            } else if constexpr (KIND == 7) {   // v_exp_f32 x8
                a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
            } else if constexpr (KIND == 8) {   // v_ldexp_f32 x8
                const int e = (u & 1) ? -(i & 7) : (i & 7);
                a0 = __builtin_ldexpf(a0, e); a1 = __builtin_ldexpf(a1, e); a2 = __builtin_ldexpf(a2, e); a3 = __builtin_ldexpf(a3, e);
                a4 = __builtin_ldexpf(a4, e); a5 = __builtin_ldexpf(a5, e); a6 = __builtin_ldexpf(a6, e); a7 = __builtin_ldexpf(a7, e);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND> int run(const char* name, int insts_per_trip, int wg_per_cu) {
    const int iters = 4000;
    const int grid = 256 * wg_per_cu;
    float* out;
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    k<KIND><<<grid, 256>>>(out, 10, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<KIND><<<grid, 256>>>(out, iters, 1.0f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per SIMD: each workgroup = 4 waves, one per SIMD; wg_per_cu waves per SIMD
    const double inst_per_simd = (double)iters * insts_per_trip * wg_per_cu;
    const double ns_per_inst = ms * 1e6 / inst_per_simd;
    printf("%-28s waves/SIMD %d  %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n",
           name, wg_per_cu, ms, ns_per_inst, ns_per_inst * 2.4);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_pk_fma_f32", 32, w);
        run<3>("v_mul_f32", 64, w);
        run<6>("v_pk_mul_f32", 32, w);
        run<2>("v_add_f32_dpp row_shr:1", 64, w);
        run<5>("v_cndmask_b32", 64, w);
        run<4>("v_rcp_f32", 64, w);
        run<7>("v_exp_f32", 64, w);
        run<8>("v_ldexp_f32", 64, w);
    }
    return 0;
}
