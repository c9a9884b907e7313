// train_ops.hip -- the training-loop operations right behind the rasterizer:
//
//   gs_adam_step              torch.optim.Adam.step() over the reference's parameter groups
//                             (splat_py/optimizer_manager.py:15-42 builds the optimizer, trainer.py:376
//                             steps it): one launch for all groups instead of ~10 elementwise kernels
//                             per group
//   gs_accumulate_grad_stats  the densification statistics of trainer.py:378-385: |uv.grad| scaled by
//                             the focal lengths and scattered to the visible Gaussians, |xyz.grad|,
//                             and the view counter -- without the boolean-mask index_put (which costs
//                             a nonzero() and a host sync per frame in PyTorch)
//
// Both are HBM streaming kernels: 28 B moved per parameter element for Adam (p, g, m, v read; p, m, v
// written), so a step over 2.86 M Gaussians x 59 parameters moves 4.7 GB.
#include "gs_common.h"

namespace gs {

constexpr int GS_ADAM_MAX_GROUPS = 8;

struct AdamGroups {
    float* p[GS_ADAM_MAX_GROUPS];
    const float* g[GS_ADAM_MAX_GROUPS];
    float* m[GS_ADAM_MAX_GROUPS];
    float* v[GS_ADAM_MAX_GROUPS];
    long long numel[GS_ADAM_MAX_GROUPS];
    long long chunk_end[GS_ADAM_MAX_GROUPS];   // inclusive prefix of ceil(numel / 4)
    float neg_step_size[GS_ADAM_MAX_GROUPS];   // -lr / (1 - beta1^step)
    float bc2_sqrt[GS_ADAM_MAX_GROUPS];        // sqrt(1 - beta2^step)
    int vec_ok[GS_ADAM_MAX_GROUPS];            // all four base pointers 16-byte aligned
    int n;
};

// torch/optim/adam.py _single_tensor_adam (amsgrad=False, weight_decay=0, maximize=False), with the
// operation order of the ATen CPU kernels:
//   exp_avg.lerp_(grad, 1 - beta1)                         a + w (b - a)         (|w| < 0.5)
//   exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2) self + (value t1) t2
//   denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   param.addcdiv_(exp_avg, denom, value=-step_size)       self + (value t1) / t2
__device__ inline void adam_update(float& p, float g, float& m, float& v, float w1, float beta2,
                                   float w2, float bc2_sqrt, float eps, float neg_step) {
    m = m + w1 * (g - m);
    v = v * beta2 + (w2 * g) * g;
    const float denom = __builtin_sqrtf(v) / bc2_sqrt + eps;
    p = p + (neg_step * m) / denom;
}

typedef float vfloat4 __attribute__((ext_vector_type(4)));
__device__ inline float4 nt_load4(const float* p) {
    const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline void nt_store4(float* p, const float4& a) {
    vfloat4 v = {a.x, a.y, a.z, a.w};
    __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(p));
}

// one float4 chunk (or the scalar tail of a tensor)
struct AdamChunk {
    int k;
    long long i;
    bool vec;
    float4 p, g, m, v;
};

__device__ inline void adam_locate(const AdamGroups& G, long long c, AdamChunk& ch) {
    int k = 0;
    while (c >= G.chunk_end[k]) k++;
    ch.k = k;
    ch.i = (c - (k ? G.chunk_end[k - 1] : 0)) * 4;
    ch.vec = G.vec_ok[k] && ch.i + 4 <= G.numel[k];
}

__device__ inline void adam_load(const AdamGroups& G, AdamChunk& ch) {
    if (!ch.vec) return;
    ch.p = nt_load4(G.p[ch.k] + ch.i);
    ch.g = nt_load4(G.g[ch.k] + ch.i);
    ch.m = nt_load4(G.m[ch.k] + ch.i);
    ch.v = nt_load4(G.v[ch.k] + ch.i);
}

__device__ inline void adam_finish(const AdamGroups& G, AdamChunk& ch, float w1, float beta2,
                                   float w2, float eps) {
    const int k = ch.k;
    const float neg_step = G.neg_step_size[k], bc2 = G.bc2_sqrt[k];
    if (ch.vec) {
        adam_update(ch.p.x, ch.g.x, ch.m.x, ch.v.x, w1, beta2, w2, bc2, eps, neg_step);
        adam_update(ch.p.y, ch.g.y, ch.m.y, ch.v.y, w1, beta2, w2, bc2, eps, neg_step);
        adam_update(ch.p.z, ch.g.z, ch.m.z, ch.v.z, w1, beta2, w2, bc2, eps, neg_step);
        adam_update(ch.p.w, ch.g.w, ch.m.w, ch.v.w, w1, beta2, w2, bc2, eps, neg_step);
        nt_store4(G.p[k] + ch.i, ch.p);
        nt_store4(G.m[k] + ch.i, ch.m);
        nt_store4(G.v[k] + ch.i, ch.v);
    } else {
        for (long long j = ch.i; j < ch.i + 4 && j < G.numel[k]; j++) {
            float p = G.p[k][j], m = G.m[k][j], v = G.v[k][j];
            adam_update(p, G.g[k][j], m, v, w1, beta2, w2, bc2, eps, neg_step);
            G.p[k][j] = p;
            G.m[k][j] = m;
            G.v[k][j] = v;
        }
    }
}

// grid-stride over the float4 chunks of all tensors, two chunks (8 x 16 B of loads) in flight per
// thread
__global__ __launch_bounds__(256) void k_adam(AdamGroups G, float w1, float beta2, float w2,
                                              float eps) {
    const long long total = G.chunk_end[G.n - 1];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < total;
         c += 2 * stride) {
        AdamChunk a, b;
        const bool two = c + stride < total;
        adam_locate(G, c, a);
        if (two) adam_locate(G, c + stride, b);
        adam_load(G, a);
        if (two) adam_load(G, b);
        adam_finish(G, a, w1, beta2, w2, eps);
        if (two) adam_finish(G, b, w1, beta2, w2, eps);
    }
}

// trainer.py:378-385.  One thread per Gaussian index i:
//   uv_grad_accum[i]  += |uv_grad[rank[i]] * (fx, fy)|    if i is visible (rank[i] >= 0)
//   xyz_grad_accum[i] += |xyz_grad[i]|
//   grad_accum_count[i] += visible
__global__ __launch_bounds__(256) void k_grad_stats(const float* __restrict__ uv_grad,
                                                    int uv_row_stride, const int* __restrict__ rank,
                                                    const float* __restrict__ xyz_grad,
                                                    const float* __restrict__ K, int N,
                                                    float* __restrict__ uv_accum,
                                                    float* __restrict__ xyz_accum,
                                                    int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int v = rank[i];
    if (v >= 0) {
        const float fx = K[0], fy = K[4];   // camera.K[0, 0], camera.K[1, 1], read on the device
        const float* gr = uv_grad + (size_t)v * uv_row_stride;
        uv_accum[i * 2 + 0] += __builtin_fabsf(gr[0] * fx);
        uv_accum[i * 2 + 1] += __builtin_fabsf(gr[1] * fy);
        count[i] += 1;
    }
    if (xyz_grad != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; k++) xyz_accum[i * 3 + k] += __builtin_fabsf(xyz_grad[i * 3 + k]);
    }
}

// A plain float4 stream copy (bench.py: the practical HBM roof of THIS box, next to the 8 TB/s nominal -- a library
// copy_() measures the library, MI355X_MICROARCH.md quotes 6.29 TB/s for this kernel shape): grid-stride, 16 bytes per
// lane per trip, non-temporal so that the 256 MiB Infinity Cache does not serve part of the stream.
typedef float copy_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const copy_f4* __restrict__ src, copy_f4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const copy_f4 v = __builtin_nontemporal_load(&src[i]);
        __builtin_nontemporal_store(v, &dst[i]);
    }
}

}  // namespace gs

using namespace gs;

extern "C" {

int gs_adam_step(int n_groups, void* const* params, const void* const* grads, void* const* exp_avg,
                 void* const* exp_avg_sq, const int64_t* numel, const double* lr,
                 const int64_t* step, double beta1, double beta2, double eps, void* stream) {
    GS_REQUIRE(n_groups >= 1 && n_groups <= GS_ADAM_MAX_GROUPS, "adam_step: 1 <= n_groups <= %d",
               GS_ADAM_MAX_GROUPS);
    AdamGroups G;
    long long chunks = 0;
    for (int k = 0; k < n_groups; k++) {
        GS_REQUIRE(numel[k] >= 0 && step[k] >= 1, "adam_step: bad numel / step of group %d", k);
        G.p[k] = (float*)params[k];
        G.g[k] = (const float*)grads[k];
        G.m[k] = (float*)exp_avg[k];
        G.v[k] = (float*)exp_avg_sq[k];
        G.numel[k] = numel[k];
        chunks += (numel[k] + 3) / 4;
        G.chunk_end[k] = chunks;
        // torch/optim/adam.py: bias corrections and the step size are Python floats (fp64), the
        // kernels then take them as fp32 scalars
        const double bc1 = 1.0 - pow(beta1, (double)step[k]);
        const double bc2 = 1.0 - pow(beta2, (double)step[k]);
        G.neg_step_size[k] = (float)(-(lr[k] / bc1));
        G.bc2_sqrt[k] = (float)sqrt(bc2);
        const uintptr_t bits = (uintptr_t)params[k] | (uintptr_t)grads[k] | (uintptr_t)exp_avg[k] |
                               (uintptr_t)exp_avg_sq[k];
        G.vec_ok[k] = (bits & 15) == 0;
    }
    for (int k = n_groups; k < GS_ADAM_MAX_GROUPS; k++) {
        G.p[k] = nullptr; G.g[k] = nullptr; G.m[k] = nullptr; G.v[k] = nullptr;
        G.numel[k] = 0; G.chunk_end[k] = chunks; G.neg_step_size[k] = 0; G.bc2_sqrt[k] = 1;
        G.vec_ok[k] = 0;
    }
    G.n = n_groups;
    if (chunks == 0) return GS_OK;
    const long long want = (chunks + 255) / 256;
    const int grid = (int)(want < 8192 ? want : 8192);
    // torch passes `1 - beta1`, `beta2`, `1 - beta2`, `eps` as Python floats (fp64) that the fp32
    // kernels round once
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    k_adam<<<grid, 256, 0, (hipStream_t)stream>>>(G, w1, (float)beta2, w2, (float)eps);
    return check_launch("adam_step");
}

int gs_stream_copy(void* dst, const void* src, size_t bytes, int blocks, void* stream) {
    GS_REQUIRE((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0, "stream_copy: pointers and size must be multiples of 16");
    GS_REQUIRE(blocks >= 1, "stream_copy: blocks >= 1");
    if (bytes == 0) return GS_OK;
    k_stream_copy<<<blocks, 256, 0, (hipStream_t)stream>>>((const copy_f4*)src, (copy_f4*)dst, bytes / 16);
    return check_launch("stream_copy");
}

int gs_accumulate_grad_stats(const void* uv_grad, int uv_row_stride, const int32_t* rank,
                             const void* xyz_grad, const void* K, int N, void* uv_grad_accum,
                             void* xyz_grad_accum, int32_t* grad_accum_count, void* stream) {
    GS_REQUIRE(uv_row_stride >= 2, "accumulate_grad_stats: uv_row_stride must be >= 2");
    if (N <= 0) return GS_OK;
    k_grad_stats<<<div_up(N, 256), 256, 0, (hipStream_t)stream>>>(
        (const float*)uv_grad, uv_row_stride, rank, (const float*)xyz_grad, (const float*)K, N,
        (float*)uv_grad_accum, (float*)xyz_grad_accum, grad_accum_count);
    return check_launch("accumulate_grad_stats");
}

}  // extern "C"
