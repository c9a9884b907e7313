// loss.hip -- the training loss of the reference, forward and backward in one launch
// (splat_py/trainer.py:363-374):
//
//     loss = (1 - f) * l1_loss(image, gt) + f * (1 - SSIM(image, gt))
//
// SSIM as torchmetrics 1.2.1 computes it with the reference's settings (trainer.py:24:
// StructuralSimilarityIndexMeasure(data_range=1.0): 11x11 Gaussian window, sigma 1.5, k1 0.01,
// k2 0.03; the SSIM map is averaged over the pixels whose window lies entirely inside the image --
// torchmetrics pads by 5 and then crops by 5 twice).  PyTorch runs this as a 5-map grouped conv2d,
// ~15 elementwise kernels and their autograd mirror; here a 16x16 pixel tile is one workgroup:
//   1. image and target of the tile + 10 pixel halo go to LDS (channel-last in HBM, planar in LDS)
//   2. SSIM and its partial derivatives w.r.t. the window means (mu_x, E[x^2], E[xy]) are formed for
//      the tile + 5 pixel halo (every window that contains a pixel of the tile); the window is
//      separable (g g^T), so each sum is a row pass and a column pass through LDS
//   3. each pixel convolves the three derivative maps over its 11x11 neighbourhood (separably):
//        d sum(SSIM) / dx_p = conv(D_mu)[p] + 2 x_p conv(D_xx)[p] + y_p conv(D_xy)[p]
//      adds the L1 term and writes grad_image; per-tile partial sums of SSIM and |x - y| go to a
//      workspace that a second, single-workgroup launch reduces in a fixed order (deterministic).
#include "gs_common.h"

namespace gs {

constexpr int LT = 16;            // tile edge
constexpr int LH = 5;             // window radius
constexpr int LR = LT + 4 * LH;   // staged region edge (36)
constexpr int LD = LT + 2 * LH;   // derivative region edge (26)

struct LossParams {
    float g[2 * LH + 1];   // normalised 1-D Gaussian
    float c1, c2;
    float coef_l1;         // (1 - f) / (3 H W)
    float coef_ssim;       // -f / (3 (H - 10) (W - 10))
};

__global__ __launch_bounds__(LT * LT) void k_ssim_l1(const float* __restrict__ img,
                                                   const float* __restrict__ tgt, int H, int W,
                                                   LossParams P, float* __restrict__ grad,
                                                   double* __restrict__ partial) {
    __shared__ float s_x[3][LR][LR + 1];
    __shared__ float s_y[3][LR][LR + 1];
    __shared__ float s_d[3][3][LD][LD + 1];   // [D_mu | D_xx | D_xy][channel]
    __shared__ float s_h[5][LR][LD];          // row-pass sums of one channel (x, y, xx, yy, xy)
    __shared__ double s_red[3][LT * LT / GS_WAVE];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;

    for (int idx = tid; idx < LR * LR; idx += LT * LT) {
        const int ry = idx / LR, rx = idx - ry * LR;
        const int gy = y0 - 2 * LH + ry, gx = x0 - 2 * LH + rx;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = ((size_t)gy * W + gx) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            s_x[c][ry][rx] = in ? img[o + c] : 0.0f;
            s_y[c][ry][rx] = in ? tgt[o + c] : 0.0f;
        }
    }
    __syncthreads();

    // The 2-D window is the outer product g g^T: every 11x11 sum is an 11-tap row pass followed by an
    // 11-tap column pass (22 instead of 121 taps per value), one channel at a time through s_h.
    double ssim_sum = 0.0;
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        for (int idx = tid; idx < LR * LD; idx += LT * LT) {
            const int r = idx / LD, dx = idx - r * LD;
            float hx = 0, hy = 0, hxx = 0, hyy = 0, hxy = 0;
#pragma unroll
            for (int j = 0; j <= 2 * LH; j++) {
                const float w = P.g[j];
                const float x = s_x[c][r][dx + j], y = s_y[c][r][dx + j];
                hx += w * x;
                hy += w * y;
                hxx += w * (x * x);
                hyy += w * (y * y);
                hxy += w * (x * y);
            }
            s_h[0][r][dx] = hx;
            s_h[1][r][dx] = hy;
            s_h[2][r][dx] = hxx;
            s_h[3][r][dx] = hyy;
            s_h[4][r][dx] = hxy;
        }
        __syncthreads();
        for (int idx = tid; idx < LD * LD; idx += LT * LT) {
            const int dy = idx / LD, dx = idx - dy * LD;
            const int qy = y0 - LH + dy, qx = x0 - LH + dx;
            // windows entirely inside the image (the double crop of torchmetrics)
            const bool interior = qy >= LH && qy < H - LH && qx >= LH && qx < W - LH;
            const bool own = dy >= LH && dy < LH + LT && dx >= LH && dx < LH + LT;
            float d_mu = 0, d_xx = 0, d_xy = 0;
            if (interior) {
                float mx = 0, my = 0, exx = 0, eyy = 0, exy = 0;
#pragma unroll
                for (int i = 0; i <= 2 * LH; i++) {
                    const float w = P.g[i];
                    mx += w * s_h[0][dy + i][dx];
                    my += w * s_h[1][dy + i][dx];
                    exx += w * s_h[2][dy + i][dx];
                    eyy += w * s_h[3][dy + i][dx];
                    exy += w * s_h[4][dy + i][dx];
                }
                const float mxx = mx * mx, myy = my * my, mxy = mx * my;
                const float a1 = 2 * mxy + P.c1, a2 = 2 * (exy - mxy) + P.c2;
                const float b1 = mxx + myy + P.c1, b2 = (exx - mxx) + (eyy - myy) + P.c2;
                const float rb = 1.0f / (b1 * b2);
                const float S = a1 * a2 * rb;
                // d SSIM / d(mu_x, E[x^2], E[xy]) with sigma_xx = E[x^2] - mu_x^2, sigma_xy = E[xy] - mu_x mu_y
                d_mu = 2 * my * (a2 - a1) * rb - 2 * mx * S * (1.0f / b1 - 1.0f / b2);
                d_xx = -S / b2;
                d_xy = 2 * a1 * rb;
                if (own) ssim_sum += (double)S;
            }
            s_d[0][c][dy][dx] = d_mu;
            s_d[1][c][dy][dx] = d_xx;
            s_d[2][c][dy][dx] = d_xy;
        }
        __syncthreads();
    }

    const int ty = tid / LT, tx = tid - ty * LT;
    const int py = y0 + ty, px = x0 + tx;
    double l1_sum = 0.0, sq_sum = 0.0;
    float (*s_t)[LD][LT] = reinterpret_cast<float (*)[LD][LT]>(&s_h[0][0][0]);   // [3][26][16] row-pass sums
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        for (int idx = tid; idx < LD * LT; idx += LT * LT) {
            const int ry = idx / LT, cx = idx - ry * LT;
            float t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
            for (int j = 0; j <= 2 * LH; j++) {
                const float w = P.g[j];
                t0 += w * s_d[0][c][ry][cx + j];
                t1 += w * s_d[1][c][ry][cx + j];
                t2 += w * s_d[2][c][ry][cx + j];
            }
            s_t[0][ry][cx] = t0;
            s_t[1][ry][cx] = t1;
            s_t[2][ry][cx] = t2;
        }
        __syncthreads();
        if (py < H && px < W) {
            float gm = 0, gxx = 0, gxy = 0;
#pragma unroll
            for (int i = 0; i <= 2 * LH; i++) {
                const float w = P.g[i];
                gm += w * s_t[0][ty + i][tx];
                gxx += w * s_t[1][ty + i][tx];
                gxy += w * s_t[2][ty + i][tx];
            }
            const float x = s_x[c][ty + 2 * LH][tx + 2 * LH], y = s_y[c][ty + 2 * LH][tx + 2 * LH];
            const float diff = x - y;
            l1_sum += (double)__builtin_fabsf(diff);
            sq_sum += (double)(diff * diff);
            if (grad != nullptr) {
                const float sgn = diff > 0 ? 1.0f : (diff < 0 ? -1.0f : 0.0f);
                grad[((size_t)py * W + px) * 3 + c] =
                    P.coef_ssim * (gm + 2 * x * gxx + y * gxy) + P.coef_l1 * sgn;
            }
        }
        __syncthreads();
    }
    // per-tile sums: wave shuffles, then the four wave results in thread order
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        ssim_sum += __shfl_xor(ssim_sum, d);
        l1_sum += __shfl_xor(l1_sum, d);
        sq_sum += __shfl_xor(sq_sum, d);
    }
    if ((tid & 63) == 0) {
        s_red[0][tid >> 6] = ssim_sum;
        s_red[1][tid >> 6] = l1_sum;
        s_red[2][tid >> 6] = sq_sum;
    }
    __syncthreads();
    if (tid < 3) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[3 * b + tid] = s_red[tid][0] + s_red[tid][1] + s_red[tid][2] + s_red[tid][3];
    }
}

// fixed-order reduction of the per-tile sums -> out = (loss, l1, ssim, mse)
__global__ __launch_bounds__(256) void k_loss_finish(const double* __restrict__ partial, int n,
                                                     double inv_l1, double inv_ssim, float ssim_frac,
                                                     float* __restrict__ out) {
    __shared__ double s[3][256];
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += partial[3 * i];
        b += partial[3 * i + 1];
        c += partial[3 * i + 2];
    }
    s[0][threadIdx.x] = a;
    s[1][threadIdx.x] = b;
    s[2][threadIdx.x] = c;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
            s[0][threadIdx.x] += s[0][threadIdx.x + d];
            s[1][threadIdx.x] += s[1][threadIdx.x + d];
            s[2][threadIdx.x] += s[2][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim = (float)(s[0][0] * inv_ssim), l1 = (float)(s[1][0] * inv_l1);
        out[0] = (1.0f - ssim_frac) * l1 + ssim_frac * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
        out[3] = (float)(s[2][0] * inv_l1);   // mse_loss (trainer.py:366-367: psnr = -10 log10 of it)
    }
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_ssim_l1_workspace_bytes(int H, int W) {
    return (size_t)div_up(W > 0 ? W : 1, LT) * div_up(H > 0 ? H : 1, LT) * 3 * sizeof(double);
}

int gs_ssim_l1_loss(const void* image, const void* target, int H, int W, float ssim_frac,
                    void* workspace, void* loss_out, void* grad_image, void* stream) {
    GS_REQUIRE(H > 2 * LH && W > 2 * LH, "ssim_l1_loss: the image must be larger than the 11x11 window");
    hipStream_t s = (hipStream_t)stream;
    LossParams P;
    // torchmetrics _gaussian(): exp(-((i - 5) / sigma)^2 / 2) normalised, in fp32
    float sum = 0.0f;
    for (int i = 0; i <= 2 * LH; i++) {
        const float d = (float)(i - LH) / 1.5f;
        P.g[i] = expf(-(d * d) / 2.0f);
        sum += P.g[i];
    }
    for (int i = 0; i <= 2 * LH; i++) P.g[i] /= sum;
    P.c1 = 0.01f * 0.01f;
    P.c2 = 0.03f * 0.03f;
    const double n_l1 = 3.0 * H * W, n_ssim = 3.0 * (H - 2 * LH) * (W - 2 * LH);
    P.coef_l1 = (float)((1.0 - (double)ssim_frac) / n_l1);
    P.coef_ssim = (float)(-(double)ssim_frac / n_ssim);
    const dim3 grid(div_up(W, LT), div_up(H, LT));
    k_ssim_l1<<<grid, LT * LT, 0, s>>>((const float*)image, (const float*)target, H, W, P,
                                       (float*)grad_image, (double*)workspace);
    k_loss_finish<<<1, 256, 0, s>>>((const double*)workspace, (int)(grid.x * grid.y), 1.0 / n_l1,
                                    1.0 / n_ssim, ssim_frac, (float*)loss_out);
    return check_launch("ssim_l1_loss");
}

}  // extern "C"
