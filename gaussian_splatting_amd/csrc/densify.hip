// densify.hip -- adaptive density control behind the rasterizer: the data movement of
// splat_py/trainer.py:114-206 (delete / clone / split of Gaussians) and of
// splat_py/optimizer_manager.py:78-172 (the same surgery on Adam's exp_avg / exp_avg_sq) in ONE pass.
//
// The reference performs a density-control step as ~60 PyTorch calls: a boolean-mask gather of the six
// parameter tensors and their twelve optimizer-state tensors for the delete, again for every clone and
// split source set, torch.cat for every append, twice (clone, then split).  Here the host side
// (gaussian_splatting_amd/densify.py) forms the three per-Gaussian decisions exactly as the reference
// does and turns them into destination indices with prefix sums; this kernel then reads every source
// row once and writes every row of the new set once:
//
//   row i survives unsplit          -> copy parameters and optimizer state to dst_self[i]
//   row i is cloned (trainer.py:123-161)   -> parameters to dst_clone[i], xyz - 0.01 * xyz_grad_accum / count;
//                                      optimizer state of the clone = 0 (optimizer_manager.py:120-160)
//   row i (or its clone) is split (trainer.py:163-206) -> `samples` new Gaussians at
//                                      sample_base + s * P + split_rank: xyz + R(q / |q|) (rand * exp(scale)),
//                                      scale = log(exp(scale) / split_scale_factor), quaternion = q / |q|,
//                                      the rest copied; optimizer state 0.  The row itself is dropped.
//
// New order == the reference's: survivors and clones that are not split, in their order, then for
// s in 0..samples-1 the split rows in their order (Tensor.repeat(samples, 1), trainer.py:167-173).
// Copies and the clone's xyz are bit-identical to the PyTorch lines (IEEE operations only); the split
// sample's xyz / scale / quaternion go through expf / logf / sqrtf and a 3x3 product whose association
// PyTorch does not pin (bmm): parity at 1e-6 relative.
#include "gs_common.h"

namespace gs {

constexpr int DM_MAX_TENSORS = 8;
enum { DM_PLAIN = 0, DM_XYZ = 1, DM_SCALE = 2, DM_QUAT = 3 };

struct DensifyTensors {
    const float* src[DM_MAX_TENSORS];
    float* dst[DM_MAX_TENSORS];
    const float* src_m[DM_MAX_TENSORS];   // exp_avg (or null: no optimizer state for this tensor yet)
    float* dst_m[DM_MAX_TENSORS];
    const float* src_v[DM_MAX_TENSORS];   // exp_avg_sq
    float* dst_v[DM_MAX_TENSORS];
    int width[DM_MAX_TENSORS];            // floats per row
    int kind[DM_MAX_TENSORS];
    long long elem_end[DM_MAX_TENSORS];   // inclusive prefix of N0 * width
    int n;
};

struct DensifyPlan {
    const int* dst_self;      // [N0] destination row, or -1
    const int* dst_clone;     // [N0] destination row of the clone, or -1
    const int* split_self;    // [N0] rank among the split rows if the row itself is split, else -1
    const int* split_clone;   // [N0] same for the row's clone
    const float* xyz;         // [N0,3] source positions, scales, quaternions (for the split transform)
    const float* scale;
    const float* quaternion;
    const float* xyz_grad_accum;   // [N0,3]
    const int* grad_accum_count;   // [N0]
    const float* random;           // [P * samples, 3] uniform samples, row s * P + rank (trainer.py:176)
    int N0, P, samples, sample_base;
    float split_scale_factor;
};

// utils.py:30-57 quaternion_to_rotation_torch on the normalised quaternion (w, x, y, z), row r
__device__ inline void rotation_row(const float* q, int r, float* out) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    if (r == 0) {
        out[0] = 1.0f - 2.0f * (y * y + z * z); out[1] = 2.0f * (x * y - w * z); out[2] = 2.0f * (x * z + w * y);
    } else if (r == 1) {
        out[0] = 2.0f * (x * y + w * z); out[1] = 1.0f - 2.0f * (x * x + z * z); out[2] = 2.0f * (y * z - w * x);
    } else {
        out[0] = 2.0f * (x * z - w * y); out[1] = 2.0f * (y * z + w * x); out[2] = 1.0f - 2.0f * (x * x + y * y);
    }
}

__global__ __launch_bounds__(256) void k_densify_move(DensifyTensors T, DensifyPlan P) {
    const long long total = T.elem_end[T.n - 1];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        int k = 0;
        while (e >= T.elem_end[k]) k++;
        const long long local = e - (k ? T.elem_end[k - 1] : 0);
        const int w = T.width[k];
        const int row = (int)(local / w), col = (int)(local - (long long)row * w);
        const float val = T.src[k][local];
        const int d_self = P.dst_self[row], d_clone = P.dst_clone[row];
        const int s_self = P.split_self[row], s_clone = P.split_clone[row];
        if (d_self >= 0) {
            const size_t o = (size_t)d_self * w + col;
            T.dst[k][o] = val;
            if (T.src_m[k]) {
                T.dst_m[k][o] = T.src_m[k][local];
                T.dst_v[k][o] = T.src_v[k][local];
            }
        }
        if (d_clone < 0 && s_self < 0 && s_clone < 0) continue;
        // the clone's value of this element (trainer.py:126-127)
        float cval = val;
        if (T.kind[k] == DM_XYZ && (d_clone >= 0 || s_clone >= 0)) {
            const float avg = P.xyz_grad_accum[(size_t)row * 3 + col] / (float)P.grad_accum_count[row];
            cval = val - avg * 0.01f;
        }
        if (d_clone >= 0) {
            const size_t o = (size_t)d_clone * w + col;
            T.dst[k][o] = cval;
            if (T.src_m[k]) {
                T.dst_m[k][o] = 0.0f;
                T.dst_v[k][o] = 0.0f;
            }
        }
        for (int which = 0; which < 2; which++) {
            const int rank = which ? s_clone : s_self;
            if (rank < 0) continue;
            const float base = which ? cval : val;
            for (int s = 0; s < P.samples; s++) {
                const int sample_row = s * P.P + rank;
                float out = base;
                if (T.kind[k] != DM_PLAIN) {
                    const float* q = P.quaternion + (size_t)row * 4;
                    const float* sc = P.scale + (size_t)row * 3;
                    if (T.kind[k] == DM_SCALE) {
                        out = __builtin_logf(__builtin_expf(val) / P.split_scale_factor);   // trainer.py:190
                    } else {
                        const float nrm = __builtin_sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                        const float qn[4] = {q[0] / nrm, q[1] / nrm, q[2] / nrm, q[3] / nrm};   // :182
                        if (T.kind[k] == DM_QUAT) {
                            out = qn[col];
                        } else {   // DM_XYZ: mean + R (rand * exp(scale)), trainer.py:176-188
                            const float* rs = P.random + (size_t)sample_row * 3;
                            float R[3];
                            rotation_row(qn, col, R);
                            const float v0 = rs[0] * __builtin_expf(sc[0]), v1 = rs[1] * __builtin_expf(sc[1]),
                                        v2 = rs[2] * __builtin_expf(sc[2]);
                            out = base + ((R[0] * v0 + R[1] * v1) + R[2] * v2);
                        }
                    }
                }
                const size_t o = (size_t)(P.sample_base + sample_row) * w + col;
                T.dst[k][o] = out;
                if (T.src_m[k]) {
                    T.dst_m[k][o] = 0.0f;
                    T.dst_v[k][o] = 0.0f;
                }
            }
        }
    }
}

}  // namespace gs

using namespace gs;

extern "C" int gs_densify_move(int n_tensors, const void* const* src, void* const* dst, const void* const* src_exp_avg,
                               void* const* dst_exp_avg, const void* const* src_exp_avg_sq, void* const* dst_exp_avg_sq,
                               const int32_t* width, const int32_t* kind, int N0, const int32_t* dst_self,
                               const int32_t* dst_clone, const int32_t* split_self, const int32_t* split_clone,
                               const void* xyz, const void* scale, const void* quaternion, const void* xyz_grad_accum,
                               const int32_t* grad_accum_count, const void* random_samples, int n_split, int samples,
                               int sample_base, float split_scale_factor, void* stream) {
    GS_REQUIRE(n_tensors >= 1 && n_tensors <= DM_MAX_TENSORS, "gs_densify_move takes 1..%d tensors", DM_MAX_TENSORS);
    GS_REQUIRE(N0 >= 0 && n_split >= 0 && samples >= 0, "bad sizes");
    if (N0 == 0) return GS_OK;
    DensifyTensors T;
    long long end = 0;
    for (int k = 0; k < n_tensors; k++) {
        GS_REQUIRE(width[k] > 0 && kind[k] >= DM_PLAIN && kind[k] <= DM_QUAT, "bad width / kind for tensor %d", k);
        GS_REQUIRE((kind[k] == DM_PLAIN) || (kind[k] == DM_QUAT ? width[k] == 4 : width[k] == 3),
                   "xyz / scale rows have 3 floats, quaternion rows 4");
        GS_REQUIRE((src_exp_avg[k] == nullptr) == (src_exp_avg_sq[k] == nullptr), "exp_avg and exp_avg_sq go together");
        T.src[k] = (const float*)src[k];
        T.dst[k] = (float*)dst[k];
        T.src_m[k] = (const float*)src_exp_avg[k];
        T.dst_m[k] = (float*)dst_exp_avg[k];
        T.src_v[k] = (const float*)src_exp_avg_sq[k];
        T.dst_v[k] = (float*)dst_exp_avg_sq[k];
        T.width[k] = width[k];
        T.kind[k] = kind[k];
        end += (long long)N0 * width[k];
        T.elem_end[k] = end;
    }
    T.n = n_tensors;
    DensifyPlan P{dst_self, dst_clone, split_self, split_clone, (const float*)xyz, (const float*)scale,
                  (const float*)quaternion, (const float*)xyz_grad_accum, grad_accum_count,
                  (const float*)random_samples, N0, n_split, samples, sample_base, split_scale_factor};
    const long long blocks = (end + 255) / 256;
    const int grid = (int)(blocks < 4096 ? blocks : 4096);
    k_densify_move<<<grid, 256, 0, (hipStream_t)stream>>>(T, P);
    return check_launch("densify_move");
}
