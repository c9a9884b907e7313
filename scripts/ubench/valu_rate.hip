// valu_rate.hip -- issue-rate microbenchmark for the instruction kinds the render kernels are made of.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -o valu_rate valu_rate.hip ; run on the GPU box.
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
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    const double dm = 1e-9 + seed;
    const unsigned long long selmask = __ballot((threadIdx.x * 2654435761u >> 13) & 1);
    __shared__ float4 lds[512];
    lds[threadIdx.x] = make_float4(a0, a1, a2, a3);
    __syncthreads();
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
            } else if constexpr (KIND == 9) {   // v_cndmask_b32_e64 with a loop-invariant SGPR-pair mask x8
#define CND(x, y) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(y), "s"(selmask))
                CND(a0, a1); CND(a1, a2); CND(a2, a3); CND(a3, a4); CND(a4, a5); CND(a5, a6); CND(a6, a7); CND(a7, a0);
            } else if constexpr (KIND == 10) {  // v_max_f32 x8
                a0 = __builtin_fmaxf(a0, a1); a1 = __builtin_fmaxf(a1, a2); a2 = __builtin_fmaxf(a2, a3); a3 = __builtin_fmaxf(a3, a4);
                a4 = __builtin_fmaxf(a4, a5); a5 = __builtin_fmaxf(a5, a6); a6 = __builtin_fmaxf(a6, a7); a7 = __builtin_fmaxf(a7, a0);
            } else if constexpr (KIND == 11) {  // v_rndne_f32 x8
                a0 = __builtin_rintf(a0 * m); a1 = __builtin_rintf(a1 * m); a2 = __builtin_rintf(a2 * m); a3 = __builtin_rintf(a3 * m);
                a4 = __builtin_rintf(a4 * m); a5 = __builtin_rintf(a5 * m); a6 = __builtin_rintf(a6 * m); a7 = __builtin_rintf(a7 * m);
            } else if constexpr (KIND == 12) {  // f64 add: v_add_f64 x8
                d0 += dm; d1 += dm; d2 += dm; d3 += dm; d4 += dm; d5 += dm; d6 += dm; d7 += dm;
            } else if constexpr (KIND == 13) {  // f32->f64->f32 round trip: v_cvt_f64_f32 + v_cvt_f32_f64 x4 (8 cvts)
                a0 = (float)((double)a0 + dm); a1 = (float)((double)a1 + dm); a2 = (float)((double)a2 + dm); a3 = (float)((double)a3 + dm);
            } else if constexpr (KIND == 14) {  // v_mov_b32_dpp quad_perm x8 (xor 1)
#define DPPMOV(x) x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true))
                DPPMOV(a0); DPPMOV(a1); DPPMOV(a2); DPPMOV(a3); DPPMOV(a4); DPPMOV(a5); DPPMOV(a6); DPPMOV(a7);
            } else if constexpr (KIND == 15) {  // ds_read_b128, wave-uniform address (broadcast) x8 -> 4 regs each
                const float4 q0 = lds[(i + u) & 255], q1 = lds[(i + u + 64) & 255];
                a0 += q0.x; a1 += q0.y; a2 += q0.z; a3 += q0.w; a4 += q1.x; a5 += q1.y; a6 += q1.z; a7 += q1.w;
            } else if constexpr (KIND == 16) {  // ds_add_f32 from ONE lane per 16 (4 lanes/wave) to a wave-uniform address x8
                if ((threadIdx.x & 15) == 15) {
                    float* dst = reinterpret_cast<float*>(lds) + ((i + u) & 127) * 8;
                    __hip_atomic_fetch_add(dst + 0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 1, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 2, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 3, a3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 4, a4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 5, a5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 6, a6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 7, a7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else if constexpr (KIND == 20 || KIND == 21 || KIND == 22 || KIND == 23) {
                // ds_add_f32 x8: 20 = one lane (63); 21 = 4 lanes (15, 31, 47, 63), four different addresses;
                // 22 = all 64 lanes, 64 different addresses (conflict-free); 23 = lanes 60..63, different addresses
                const bool on = KIND == 20 ? (threadIdx.x & 63) == 63 : KIND == 21 ? (threadIdx.x & 15) == 15 :
                                KIND == 22 ? true : (threadIdx.x & 63) >= 60;
                if (on) {
                    float* dst = reinterpret_cast<float*>(lds) + ((i + u) & 3) * 8 * 32 + (KIND == 20 ? 0 : (threadIdx.x & 63));
                    __hip_atomic_fetch_add(dst + 0 * 32 * 0, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 64, a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 128, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 192, a3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 1, a4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 65, a5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 129, a6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(dst + 193, a7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else if constexpr (KIND == 24 || KIND == 25) {
                // plain stores from lanes 60..63: 24 = ds_write_b32 x8, 25 = ds_write_b64 x4 (counted as 8)
                if ((threadIdx.x & 63) >= 60) {
                    float* dst = reinterpret_cast<float*>(lds) + ((i + u) & 3) * 8 * 32 + (threadIdx.x & 63) * 2;
                    if constexpr (KIND == 24) {
                        dst[0] = a0; dst[128] = a1; dst[256] = a2; dst[384] = a3; dst[1] = a4; dst[129] = a5; dst[257] = a6; dst[385] = a7;
                    } else {
                        *reinterpret_cast<float2*>(dst) = make_float2(a0, a4); *reinterpret_cast<float2*>(dst + 128) = make_float2(a1, a5);
                        *reinterpret_cast<float2*>(dst + 256) = make_float2(a2, a6); *reinterpret_cast<float2*>(dst + 384) = make_float2(a3, a7);
                    }
                    asm volatile("" ::: "memory");
                }
            } else if constexpr (KIND == 30) {  // ONE dependent chain per lane: 8 v_fma, each reading the previous result
                a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
                a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaf(a0, m, c);
            } else if constexpr (KIND == 31) {  // two interleaved dependent chains
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c);
                a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c);
            } else if constexpr (KIND == 32) {  // dependent chain of mixed ops the render loops use: mul, sub, fma, max, cndmask(e64), dpp add
                a0 = a0 * m; a0 = a0 - a1; a0 = __builtin_fmaf(a0, m, c); a0 = __builtin_fmaxf(a0, a2);
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a0) : "v"(a0), "v"(a3), "s"(selmask));
                a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x111, 0xf, 0xf, true));
                a0 = a0 * m; a0 = __builtin_fmaf(a0, m, c);
            } else if constexpr (KIND == 33) {  // v_permlane16_swap_b32 x8 (pairs of registers)
#define PSWAP16(x, y) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y))
                PSWAP16(a0, a1); PSWAP16(a2, a3); PSWAP16(a4, a5); PSWAP16(a6, a7); PSWAP16(a0, a2); PSWAP16(a1, a3); PSWAP16(a4, a6); PSWAP16(a5, a7);
            } else if constexpr (KIND == 34) {  // v_permlane32_swap_b32 x8
#define PSWAP32(x, y) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y))
                PSWAP32(a0, a1); PSWAP32(a2, a3); PSWAP32(a4, a5); PSWAP32(a6, a7); PSWAP32(a0, a2); PSWAP32(a1, a3); PSWAP32(a4, a6); PSWAP32(a5, a7);
            } else if constexpr (KIND == 18) {  // v_readfirstlane_b32 x8 (value goes back to a VGPR through an s_add)
                a0 += __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a1)));
                a1 += __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a2)));
                a2 += __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a3)));
                a3 += __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a0)));
            } else if constexpr (KIND == 19) {  // v_div (IEEE fp32 division) x8
                a0 = a0 / a1; a1 = a1 / a2; a2 = a2 / a3; a3 = a3 / a4; a4 = a4 / a5; a5 = a5 / a6; a6 = a6 / a7; a7 = a7 / a0;
            } else if constexpr (KIND == 8) {   // v_ldexp_f32 x8
                const int e = (u & 1) ? -(i & 7) : (i & 7);
                a0 = __builtin_ldexpf(a0, e); a1 = __builtin_ldexpf(a1, e); a2 = __builtin_ldexpf(a2, e); a3 = __builtin_ldexpf(a3, e);
                a4 = __builtin_ldexpf(a4, e); a5 = __builtin_ldexpf(a5, e); a6 = __builtin_ldexpf(a6, e); a7 = __builtin_ldexpf(a7, e);
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                          (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + lds[threadIdx.x ^ 1].x;
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
    for (int w : {1, 2, 3, 4, 6, 8}) {
        run<30>("dependent v_fma chain", 64, w);
        run<31>("two dependent v_fma chains", 64, w);
        run<32>("dependent mixed chain (mul sub fma max cnd dpp mul fma)", 64, w);
    }
    for (int w : {4, 8}) {
        run<33>("v_permlane16_swap_b32", 64, w);
        run<34>("v_permlane32_swap_b32", 64, w);
    }
    for (int w : {8}) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_pk_fma_f32", 32, w);
        run<3>("v_mul_f32", 64, w);
        run<6>("v_pk_mul_f32", 32, w);
        run<2>("v_add_f32_dpp row_shr:1", 64, w);
        run<5>("v_cndmask_b32", 64, w);
        run<4>("v_rcp_f32", 64, w);
        run<7>("v_exp_f32", 64, w);
        run<8>("v_ldexp_f32", 64, w);
        run<9>("v_cndmask_b32_e64 sgpr mask", 64, w);
        run<10>("v_max_f32", 64, w);
        run<11>("v_mul+v_rndne_f32 pairs", 64, w);
        run<12>("v_add_f64", 64, w);
        run<13>("cvt f32->f64, add_f64, cvt back (x4)", 32, w);
        run<14>("v_mov_b32_dpp quad_perm", 64, w);
        run<15>("ds_read_b128 uniform (x2 per 8 adds)", 16, w);
        run<16>("ds_add_f32 4 lanes, one address", 64, w);
        run<20>("ds_add_f32 1 lane", 64, w);
        run<21>("ds_add_f32 4 lanes, 4 addresses", 64, w);
        run<22>("ds_add_f32 64 lanes, 64 addresses", 64, w);
        run<23>("ds_add_f32 lanes 60-63, 4 addresses", 64, w);
        run<24>("ds_write_b32 lanes 60-63", 64, w);
        run<25>("ds_write_b64 lanes 60-63 (4 per 8)", 32, w);
        run<18>("v_readfirstlane + v_add (x4)", 32, w);
        run<19>("fp32 IEEE division", 64, w);
    }
    return 0;
}
