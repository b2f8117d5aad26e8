// The two denoiser-loss terms besides L1 (bfcnn/loss.py:190-247): the per-image RMSE term (rmse_diff, :92-113) and
// (1 - mean tf.image.ssim(gt, prediction, filter_size=7, max_val=255)) (:219-227).  Both need the whole prediction before
// any gradient exists (a per-image square root; 7x7 windows), so bf_train_step runs the head twice when one of them is
// on: pass A writes the prediction and the per-image sums, the kernels below turn them into an additive dL/dprediction,
// pass B is the usual fused head forward + loss + backward with that gradient added to the L1 one.
//
// SSIM as TF 2.13 computes it (image_ops_impl._ssim_per_channel / _ssim_helper): window g = softmax over the 7x7 grid of
// -(i^2 + j^2) / (2 * 1.5^2), VALID windows, per window and channel with x = prediction, y = ground truth:
//   a = g*x, b = g*y, s = g*(x y), q = g*(x^2 + y^2), c1 = (0.01 max)^2, c2 = (0.03 max)^2
//   S = (2ab + c1) / (a^2 + b^2 + c1) * (2s - 2ab + c2) / (q - a^2 - b^2 + c2),   ssim = mean of S over everything.
// q - a^2 - b^2 cancels five digits, so the window sums are carried in fp64 (the matrix of the work is tiny: 49 taps).
// dS/dx[p] = sum over the windows w covering p of g[p - w] * (dS/da[w] + y[p] dS/ds[w] + 2 x[p] dS/dq[w]).
#include "bf_common.h"
#include <math.h>

struct SsimWindow { float g[49]; };

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt, SsimWindow win,
                                                       int B, int H, int W, int C, float c1, float c2, float grad_scale,
                                                       float* __restrict__ dA, float* __restrict__ dS, float* __restrict__ dQ,
                                                       float* __restrict__ partial)
{
    __shared__ double red[4];
    const int Hw = H - 6, Ww = W - 6;
    const int64_t n = (int64_t)B * Hw * Ww * C;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int wx = (int)((i / C) % Ww);
        const int wy = (int)((i / ((int64_t)C * Ww)) % Hw);
        const int b = (int)(i / ((int64_t)C * Ww * Hw));
        double a = 0.0, bb = 0.0, s = 0.0, q = 0.0;
        for (int ky = 0; ky < 7; ++ky) {
            const int64_t row = (((int64_t)b * H + wy + ky) * W + wx) * C + c;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const double x = pred[row + (int64_t)kx * C], y = gt[row + (int64_t)kx * C], g = win.g[ky * 7 + kx];
                a += g * x; bb += g * y; s += g * x * y; q += g * (x * x + y * y);
            }
        }
        const double nl = 2.0 * a * bb + c1, dl = a * a + bb * bb + c1;
        const double nc = 2.0 * s - 2.0 * a * bb + c2, dc = q - a * a - bb * bb + c2;
        const double lum = nl / dl, cs = nc / dc;
        acc += lum * cs;
        dA[i] = (float)(grad_scale * (cs * (2.0 * bb * dl - nl * 2.0 * a) / (dl * dl) + lum * (-2.0 * bb * dc + nc * 2.0 * a) / (dc * dc)));
        dS[i] = (float)(grad_scale * lum * 2.0 / dc);
        dQ[i] = (float)(grad_scale * -lum * nc / (dc * dc));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// per-image RMSE term: coef[b] = mse_multiplier * depth_weight / (B * N * rmse_b), rmse_b = sqrt(sum_b / N + 1e-3);
// scal[0] = mean_b rmse_b (the loss term), scal[1] = sum of the ssim partials.  One workgroup, fixed order.
__global__ __launch_bounds__(256) void loss_extra_prepare_kernel(const float* __restrict__ head_partial, int blocks_per_image, int B,
                                                                 double per_image, float mse_scale, float* __restrict__ coef,
                                                                 const float* __restrict__ ssim_partial, int n_ssim,
                                                                 float* __restrict__ scal)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        double sq = 0.0;
        for (int k = 0; k < blocks_per_image; ++k) sq += (double)head_partial[(size_t)(b * blocks_per_image + k) * 80 + 67];
        const double rm = sqrt(sq / per_image + 1e-3);
        coef[b] = (float)((double)mse_scale / ((double)B * per_image * rm));
        acc += rm;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const double rm_mean = red[0] / (double)B;
    __syncthreads();
    acc = 0.0;
    for (int i = threadIdx.x; i < n_ssim; i += 256) acc += (double)ssim_partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { scal[0] = (float)rm_mean; scal[1] = (float)red[0]; }
}

// dextra[p] = (ssim part, pre-scaled by -ssim_multiplier * depth_weight / windows) - coef[b] * relu'(gt - pred)
__global__ __launch_bounds__(256) void loss_extra_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt, SsimWindow win,
                                                              int B, int H, int W, int C, int use_ssim, const float* __restrict__ dA,
                                                              const float* __restrict__ dS, const float* __restrict__ dQ,
                                                              int use_mse, const float* __restrict__ coef, float hinge, float cutoff_sq,
                                                              float* __restrict__ dextra)
{
    const int Hw = H - 6, Ww = W - 6;
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % W);
        const int y = (int)((i / ((int64_t)C * W)) % H);
        const int b = (int)(i / ((int64_t)C * W * H));
        const float p = pred[i], t = gt[i];
        double g = 0.0;
        if (use_ssim) {
            double ga = 0.0, gs = 0.0, gq = 0.0;
            const int wy0 = max(0, y - 6), wy1 = min(Hw - 1, y), wx0 = max(0, x - 6), wx1 = min(Ww - 1, x);
            for (int wy = wy0; wy <= wy1; ++wy)
                for (int wx = wx0; wx <= wx1; ++wx) {
                    const int64_t m = (((int64_t)b * Hw + wy) * Ww + wx) * C + c;
                    const double wgt = win.g[(y - wy) * 7 + (x - wx)];
                    ga += wgt * dA[m]; gs += wgt * dS[m]; gq += wgt * dQ[m];
                }
            g = ga + (double)t * gs + 2.0 * (double)p * gq;
        }
        if (use_mse) {
            const float e = t - p;
            if (e > hinge && e < cutoff_sq) g -= (double)coef[b] * e;
        }
        dextra[i] = (float)g;
    }
}

// adds the two terms to the losses head_finalize_kernel wrote
__global__ void loss_extra_finalize_kernel(const float* __restrict__ scal, double windows, float mse_multiplier, float ssim_multiplier,
                                           float depth_weight, float* __restrict__ losses)
{
    if (threadIdx.x || blockIdx.x) return;
    const double ssim_loss = ssim_multiplier > 0.f ? 1.0 - (double)scal[1] / windows : 0.0;
    const double add = (mse_multiplier > 0.f ? (double)mse_multiplier * scal[0] : 0.0) + (double)ssim_multiplier * ssim_loss;
    losses[BF_LOSS_SSIM] = (float)ssim_loss;
    losses[BF_LOSS_DENOISER_TOTAL] = (float)((double)losses[BF_LOSS_DENOISER_TOTAL] + add);
    losses[BF_LOSS_TOTAL] = (float)((double)losses[BF_LOSS_TOTAL] + add * depth_weight);
}

static SsimWindow make_window()
{
    SsimWindow w;
    double e[49], sum = 0.0;
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) {
            const double ci = i - 3.0, cj = j - 3.0;
            e[i * 7 + j] = exp(-0.5 * (ci * ci + cj * cj) / (1.5 * 1.5));
            sum += e[i * 7 + j];
        }
    for (int k = 0; k < 49; ++k) w.g[k] = (float)(e[k] / sum);
    return w;
}

int bf_loss_extra_grid(int64_t n) { const int64_t g = (n + 255) / 256; return (int)(g < 4096 ? g : 4096); }

// scratch: maps = 3 * B*(H-6)*(W-6)*C floats, ssim_partial = 4096 floats, coef = B floats, scal = 2 floats
hipError_t bf_launch_loss_extra(const float* pred, const float* gt, int B, int H, int W, int C, const float* head_partial,
                                int blocks_per_image, float hinge, float cutoff, float mse_multiplier, float ssim_multiplier,
                                float depth_weight, float max_val, float* maps, float* ssim_partial, float* coef, float* scal,
                                float* dextra, hipStream_t s)
{
    static const SsimWindow win = make_window();
    const int use_ssim = ssim_multiplier > 0.f, use_mse = mse_multiplier > 0.f;
    const int64_t nwin = (int64_t)B * (H - 6) * (W - 6) * C;
    const int g1 = use_ssim ? bf_loss_extra_grid(nwin) : 0;
    float *dA = maps, *dS = maps + (use_ssim ? nwin : 0), *dQ = maps + (use_ssim ? 2 * nwin : 0);
    if (use_ssim) {
        const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
        hipLaunchKernelGGL(ssim_fwd_kernel, dim3(g1), dim3(256), 0, s, pred, gt, win, B, H, W, C, c1, c2,
                           (float)(-(double)ssim_multiplier * depth_weight / (double)nwin), dA, dS, dQ, ssim_partial);
    }
    hipLaunchKernelGGL(loss_extra_prepare_kernel, dim3(1), dim3(256), 0, s, head_partial, blocks_per_image, B, (double)H * W * C,
                       mse_multiplier * depth_weight, coef, ssim_partial, g1, scal);
    hipLaunchKernelGGL(loss_extra_grad_kernel, dim3(bf_loss_extra_grid((int64_t)B * H * W * C)), dim3(256), 0, s, pred, gt, win, B, H, W,
                       C, use_ssim, dA, dS, dQ, use_mse, coef, hinge, cutoff * cutoff, dextra);
    return hipGetLastError();
}

hipError_t bf_launch_loss_extra_finalize(const float* scal, int B, int H, int W, int C, float mse_multiplier, float ssim_multiplier,
                                         float depth_weight, float* losses, hipStream_t s)
{
    hipLaunchKernelGGL(loss_extra_finalize_kernel, dim3(1), dim3(64), 0, s, scal, (double)B * (H - 6) * (W - 6) * C, mse_multiplier,
                       ssim_multiplier, depth_weight, losses);
    return hipGetLastError();
}
