// ring_step.hip -- what would a SPLAT-PARALLEL render backward cost per (64 pixels x 1 splat) against the visit of
// today's pixel-parallel kernel?  (Review r05 "Next" 3; record: profiles/r06/splat_parallel_backward_ab.txt.)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o ring_step ring_step.hip ; run on the GPU box.
//
// Two kernels execute, per trip, the work of ONE (splat, 8x8 patch) pair with the arithmetic of csrc/render.hip's fp32
// backward (alpha from the conic, det_expf-style exponential, capped alpha, weight / colour-accumulator recurrence,
// grad_alpha, the nine per-pixel terms) on synthetic records held in LDS:
//   visit   today's form: lanes = the patch's 64 pixels, the splat's record read at a wave-uniform LDS address, the
//           nine values summed over the wave by the transposing DPP / permlane-swap reduction and stored to an LDS
//           slot by nine lanes (k_render_bwd's contributing visit; its mask walk and chunk flush are NOT in here)
//   ring    the proposed form: lanes = 64 splats (records in registers), the patch's pixels walked in a skewed
//           pipeline -- pixel state (weight, three colour sums) handed from lane j+1 to lane j with four DPP
//           wave-rotate moves per step, pixel constants (grad_image, num_splats) read from an LDS table by slot,
//           nine register accumulators per lane; every step ONE lane finishes its splat: it stores its nine sums to an
//           LDS row, clears them and takes the next record from an LDS queue (single-lane exec-masked instructions,
//           which still cost the wave their issue slots)
// Both with W waves per SIMD on every CU; prints ns per trip per wave and trips/s of the chip.  The ring's trip does
// the work of one visit (64 pixel-splat pairs), so the ratio of the two rates bounds what the restructuring can win
// before list generation, fill/drain and load balance are paid for.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

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
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ int slot_lane_offset(int lane) {
    if (lane == 32) return 8;
    if (lane >= 32 || (lane & 3) != 0) return -1;
    const int bank = (lane >> 2) & 3;
    const int e = ((bank & 1) ? 4 : 0) + ((bank & 2) ? 2 : 0);
    return lane < 16 ? e : e + 1;
}
__device__ __forceinline__ void reduce9_to_slot(const float* val, int lane_offset, float* slot) {
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
        "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s2[0]), "=&v"(s2[1]), "=&v"(s8)
        : "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]),
          "v"(val[8]));
    float x = s2[0], y = s2[1];
    permlane16_swap(x, y);
    float z = x + y;
    float t = s8, u = s8;
    permlane16_swap(t, u);
    float e = t + u;
    permlane32_swap(z, e);
    const float total = z + e;
    if (lane_offset >= 0) slot[lane_offset] = total;
}

struct Rec {   // the packed record: u v r2 opacity | a b c det | 1/det colour
    float4 g0, g1, g2;
};
constexpr int NREC = 64;

// the per-pair arithmetic of k_render_bwd's fp32 visit (render.hip), shared by both forms.  Returns the nine values;
// updates weight / colour_accum.  kq_lt: the Q1 condition of the pair; reach: k < num_splats of the pixel
__device__ __forceinline__ void pair(const float4 g0, const float4 g1, const float4 g2, float pu, float pv, bool reach,
                                     bool kq_lt, const float* gi, float& weight, float* ca, float* val) {
    const float du = pu - g0.x, dv = pv - g0.y;
    const float du2 = du * du, dv2 = dv * dv;
    float aw = 0, w = 0, q0 = 0, q1 = 0, q2 = 0;
    if (reach && !(du2 + dv2 > g0.z)) {
        const float duv = du * dv;
        const float mh = (g1.z * du * du - (g1.y + g1.y) * du * dv + g1.x * dv * dv) * g2.x;
        const float e = exp_neg_half(mh);
        const float norm_prob = (mh > 0.0f) ? e : 0.0f;
        float alpha = g0.w * norm_prob;
        if (alpha > 0.9999f) alpha = 0.9999f;
        if (alpha >= 0.00392156862f) {
#pragma clang fp contract(fast)
            const float r1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
            if (kq_lt) weight = weight * r1ma;
            aw = alpha * weight;
            const float ga = (g2.y * weight - ca[0] * r1ma) * gi[0] + (g2.z * weight - ca[1] * r1ma) * gi[1] +
                             (g2.w * weight - ca[2] * r1ma) * gi[2];
            ca[0] += g2.y * aw;
            ca[1] += g2.z * aw;
            ca[2] += g2.w * aw;
            w = norm_prob * ga;
            q0 = (dv2 - g1.z * mh) * w;
            q1 = (g1.y * mh - duv) * w;
            q2 = (du2 - g1.x * mh) * w;
        }
    }
    const float awy = aw * 0.28209479177387814f;
    val[0] = awy * gi[0]; val[1] = awy * gi[1]; val[2] = awy * gi[2];
    val[3] = w; val[4] = w * du; val[5] = w * dv;
    val[6] = q0; val[7] = q1; val[8] = q2;
}

__device__ __forceinline__ Rec make_rec(int i, float seed) {
    // centres spread over and around an 8x8 patch, 1.5-4 px sigma: about 40 % of the (pixel, splat) pairs inside the cutoff
    const float h = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f, h2 = (float)((i * 40503u + 17) & 0xffff) / 65536.0f;
    const float sg = 1.5f + 2.5f * h2, a = sg * sg, c = sg * sg * (0.6f + 0.8f * h), b = 0.3f * sg * sg * (h - 0.5f);
    const float det = a * c - b * b;
    Rec r;
    r.g0 = make_float4(-3.0f + 14.0f * h + seed, -3.0f + 14.0f * h2, 9.0f * 1.05f * (a > c ? a : c), 0.15f + 0.8f * h);
    r.g1 = make_float4(a, b, c, det);
    r.g2 = make_float4(1.0f / det, 0.2f + h, 0.9f - h2, 0.5f * h + 0.3f * h2);
    return r;
}

// ---- today's visit ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_visit(float* out, int iters, float seed) {
    __shared__ Rec s_rec[NREC];
    __shared__ float s_acc[4 * NREC * 9];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < NREC) s_rec[tid] = make_rec(tid, seed);
    __syncthreads();
    const float pu = (float)(lane & 7), pv = (float)(lane >> 3);
    const float gi[3] = {0.01f + 1e-4f * lane, -0.02f, 0.015f};
    const int nsp = 40 + (lane * 7) % 24;
    const int slot_off = slot_lane_offset(lane);
    float acc_out = 0;
    for (int it = 0; it < iters; it++) {
        float weight = 0.02f + 1e-4f * lane, ca[3] = {0, 0, 0};
        for (int i = NREC - 1; i >= 0; i--) {   // one "visit" per trip, records at wave-uniform addresses
            const Rec r = s_rec[i];
            float val[9];
            pair(r.g0, r.g1, r.g2, pu, pv, i < nsp, i < nsp - 1, gi, weight, ca, val);
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(val[3] != 0.0f || val[0] != 0.0f);
            if (cm == 0) continue;
            reduce9_to_slot(val, slot_off, &s_acc[(wave * NREC + i) * 9]);
        }
        acc_out += weight + ca[0];
    }
    out[blockIdx.x * 256 + tid] = acc_out + s_acc[tid];
}

// ---- the ring -----------------------------------------------------------------------------------------------------
#define DPP_ROL1(x) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x134, 0xf, 0xf, false))
__global__ __launch_bounds__(256) void k_ring(float* out, int iters, float seed) {
    __shared__ Rec s_queue[4][NREC];            // per wave: the records waiting to enter the ring
    __shared__ float4 s_pix[4][64];             // per wave: grad_image (3) | num_splats, by pixel slot
    __shared__ float s_rows[4][16 * 9];         // per wave: finished rows on their way out
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_queue[wave][lane] = make_rec(lane, seed);
    s_pix[wave][lane] = make_float4(0.01f + 1e-4f * lane, -0.02f, 0.015f, (float)(40 + (lane * 7) % 24));
    __syncthreads();
    Rec r = s_queue[wave][lane];
    int k = 63 - lane;                           // list index of the lane's splat (deepest in lane 63 ... )
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    float weight = 0.02f, ca0 = 0, ca1 = 0, ca2 = 0;
    float acc_out = 0;
    const int steps = iters * NREC;              // one step = one (splat, 64 pixels) of work for the wave
    for (int t = 0; t < steps; t++) {
        // 1. the pixel state arrives from lane j + 1 (lane 63 from lane 0: the slot starts its next revolution)
        weight = DPP_ROL1(weight); ca0 = DPP_ROL1(ca0); ca1 = DPP_ROL1(ca1); ca2 = DPP_ROL1(ca2);
        // 2. the slot this lane holds at step t, its constants
        const int q = (t + lane + 1) & 63;
        const float4 px = s_pix[wave][q];
        const float gi[3] = {px.x, px.y, px.z};
        const int nsp = (int)px.w;
        const float pu = (float)(q & 7), pv = (float)(q >> 3);
        float ca[3] = {ca0, ca1, ca2};
        float val[9];
        pair(r.g0, r.g1, r.g2, pu, pv, k < nsp, k < nsp - 1, gi, weight, ca, val);
        ca0 = ca[0]; ca1 = ca[1]; ca2 = ca[2];
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] += val[j];
        // 3. one lane has seen all 64 pixels: row out, sums cleared, next record in (single-lane instructions)
        const int sw = 63 - (t & 63);
        if (lane == sw) {
            float* row = &s_rows[wave][(t & 15) * 9];
#pragma unroll
            for (int j = 0; j < 9; j++) { row[j] = acc[j]; acc[j] = 0; }
            r = s_queue[wave][(t + 7) & (NREC - 1)];
            k = (k + 37) & 63;
        }
    }
    acc_out = weight + ca0 + acc[0] + acc[8] + r.g0.x;
    out[blockIdx.x * 256 + tid] = acc_out + s_rows[wave][lane];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%d CUs, %d trips of %d (splat, patch) pairs per wave\n", cus, iters, NREC);
    for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD = workgroups (4 waves) per CU
        for (int kind = 0; kind < 2; kind++) {
            const int grid = cus * wps;
            float ms_best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CHECK(hipEventRecord(e0));
                if (kind == 0) k_visit<<<grid, 256>>>(out, iters, 0.001f * rep);
                else k_ring<<<grid, 256>>>(out, iters, 0.001f * rep);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < ms_best) ms_best = ms;
            }
            const double trips_per_wave = (double)iters * NREC;
            const double waves = (double)grid * 4;
            printf("%-6s %d waves/SIMD: %8.3f ms  %7.1f ns per trip per wave  %7.2f G trips/s chip  (%.2f ns per trip per SIMD)\n",
                   kind == 0 ? "visit" : "ring", wps, ms_best, ms_best * 1e6 / trips_per_wave,
                   trips_per_wave * waves / (ms_best * 1e-3) / 1e9, ms_best * 1e6 / (trips_per_wave * wps));
        }
    }
    return 0;
}
