// Backward primitives of the operator library (bf_op_*): what `unet_laplacian` training (bfcnn/train_loop.py:259-312 for
// the multi-output hydra) needs on top of the forward operators of unet_ops.hip -- weight gradients of
// 1x1 / k x k / depthwise convolutions, LayerNorm / activation / ChannelLearnableMultiplier backward, the adjoints of the
// Laplacian split and of the bilinear resamplers, attention backward, the head's backward, the denoiser loss with its
// gradient, and the regularisers (L1 / L2 / SoftOrthonormalConstraintRegularizer).  Exact fp32, NHWC, stream-ordered, no
// allocation; every reduction goes through per-workgroup partials summed in a fixed order (bitwise reproducible, no float
// atomics).  These kernels are written for correctness and clarity first: training of this network is a parity target
// (tests/test_gpu_unet_train.py against the torch-autograd oracle), not a benchmarked configuration.
#include "bf_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float tp_act_grad(float ref, int act, float alpha, int ref_is_output)
{
    // derivative of the activation at a point given its input (ref_is_output = 0) or, for the sign-preserving ones
    // (relu, leaky relu), its output
    switch (act) {
        case 1: return ref > 0.f ? 1.f : 0.f;
        case 2: return ref > 0.f ? 1.f : alpha;
        case 3: {                                        // exact-erf GELU: needs the input
            (void)ref_is_output;
            const float c = 0.7071067811865476f, x = ref;
            return 0.5f * (1.f + erff(x * c)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
        }
        default: return 1.f;
    }
}

__global__ void tp_act_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ dy, float* __restrict__ dx, int64_t n,
                                  int act, float alpha, int ref_is_output)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = dy[i] * tp_act_grad(ref[i], act, alpha, ref_is_output);
}

inline int tp_grid(int64_t n, int per = 256, int cap = 4096)
{
    const int64_t g = (n + per - 1) / per;
    return (int)(g < 1 ? 1 : (g < cap ? g : cap));
}

// out[j] = scale * sum_r partial[r][j]  (fixed order)
__global__ void tp_reduce_kernel(const float* __restrict__ partial, int nblk, int width, float* __restrict__ out, float scale)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= width) return;
    double s = 0.0;
    for (int r = 0; r < nblk; ++r) s += (double)partial[(size_t)r * width + j];
    out[j] = (float)(s * scale);
}

hipError_t tp_reduce(const float* partial, int nblk, int width, float* out, float scale, hipStream_t s)
{
    hipLaunchKernelGGL(tp_reduce_kernel, dim3((width + 255) / 256), dim3(256), 0, s, partial, nblk, width, out, scale);
    return hipGetLastError();
}

// dW[ci][co] = sum_p x[p][ci] * dy[p][co]: one workgroup per (16 x 16 tile of dW, pixel split); 64 pixels staged per round
__global__ __launch_bounds__(256) void tp_matmul_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int64_t npix, int cin, int cout,
                                                              int tiles_i, int tiles_o, int nsplit)
{
    __shared__ float xs[64][17], ds[64][17];
    int t = blockIdx.x;
    const int to = t % tiles_o; t /= tiles_o;
    const int ti = t % tiles_i;
    const int sp = t / tiles_i;
    const int i = threadIdx.x >> 4, o = threadIdx.x & 15;
    const int64_t per = (npix + nsplit - 1) / nsplit;
    const int64_t p0 = (int64_t)sp * per, p1 = p0 + per < npix ? p0 + per : npix;
    float acc = 0.f;
    for (int64_t p = p0; p < p1; p += 64) {
        // 64 pixels x 16 channels of each operand: thread -> (pixel r = tid / 4, channels 4 * (tid % 4) ..)
        const int r = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 4;
        const int64_t pp = p + r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ti * 16 + c4 + k, co = to * 16 + c4 + k;
            xs[r][c4 + k] = (pp < p1 && ci < cin) ? x[pp * cin + ci] : 0.f;
            ds[r][c4 + k] = (pp < p1 && co < cout) ? dy[pp * cout + co] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int r2 = 0; r2 < 64; ++r2) acc = fmaf(xs[r2][i], ds[r2][o], acc);
        __syncthreads();
    }
    const int ci = ti * 16 + i, co = to * 16 + o;
    if (ci < cin && co < cout) partial[(size_t)sp * cin * cout + (size_t)ci * cout + co] = acc;
}

// depthwise weight gradient: dw[i][j][c] = sum_p x[p + (i,j) - pad][c] * dy[p][c]; workgroup = pixel range, thread = channel
// (channels beyond 256 loop), k*k running sums per thread
template <int K>
__global__ __launch_bounds__(256) void tp_dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int B, int H, int W, int C)
{
    const int64_t npix = (int64_t)B * H * W;
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    constexpr int pad = (K - 1) / 2;
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc[K * K];
#pragma unroll
        for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const int xw = (int)(p % W), yh = (int)((p / W) % H);
            const int64_t b = p / ((int64_t)W * H);
            const float g = dy[p * C + c];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int yy = yh + i - pad;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int xx = xw + j - pad;
                    if (xx < 0 || xx >= W) continue;
                    acc[i * K + j] = fmaf(x[((b * H + yy) * W + xx) * C + c], g, acc[i * K + j]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < K * K; ++t) partial[(size_t)blockIdx.x * K * K * C + (size_t)t * C + c] = acc[t];
    }
}

// LayerNormalization(center=False, scale=True) backward, one wave per pixel (lanes stride the channels):
//   xhat = (x - mu) * inv, g = dy * gamma, dx = inv * (g - mean(g) - xhat * mean(g * xhat)), dgamma += dy * xhat
__global__ __launch_bounds__(256) void tp_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ partial, int64_t npix, int C, float eps)
{
    __shared__ float dg[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_lane = (C + 63) / 64;                 // <= 4 (C <= 256)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < npix; p += (int64_t)gridDim.x * 4) {
        float xv[4], gv[4];
        float s1 = 0.f;
        for (int k = 0; k < per_lane; ++k) {
            const int c = lane + 64 * k;
            xv[k] = c < C ? x[p * C + c] : 0.f;
            s1 += xv[k];
        }
        for (int m = 32; m >= 1; m >>= 1) s1 += __shfl_xor(s1, m);
        const float mu = s1 / C;
        float s2 = 0.f;
        for (int k = 0; k < per_lane; ++k) {
            const int c = lane + 64 * k;
            const float d = c < C ? xv[k] - mu : 0.f;
            s2 += d * d;
        }
        for (int m = 32; m >= 1; m >>= 1) s2 += __shfl_xor(s2, m);
        const float inv = rsqrtf(s2 / C + eps);
        float sg = 0.f, sgx = 0.f;
        for (int k = 0; k < per_lane; ++k) {
            const int c = lane + 64 * k;
            const float xh = c < C ? (xv[k] - mu) * inv : 0.f;
            const float d = c < C ? dy[p * C + c] : 0.f;
            gv[k] = c < C ? d * gamma[c] : 0.f;
            xv[k] = xh;
            sg += gv[k];
            sgx += gv[k] * xh;
            acc[k] += d * xh;
        }
        for (int m = 32; m >= 1; m >>= 1) { sg += __shfl_xor(sg, m); sgx += __shfl_xor(sgx, m); }
        const float mg = sg / C, mgx = sgx / C;
        for (int k = 0; k < per_lane; ++k) {
            const int c = lane + 64 * k;
            if (c < C) dx[p * C + c] = inv * (gv[k] - mg - xv[k] * mgx);
        }
    }
    for (int k = 0; k < per_lane; ++k) dg[wave][lane + 64 * k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) partial[(size_t)blockIdx.x * C + c] = (dg[0][c] + dg[1][c]) + (dg[2][c] + dg[3][c]);
}

// out = res + t * m[c] * s[b]   (res, m, s optional)
__global__ void tp_scale_add_kernel(const float* __restrict__ res, const float* __restrict__ t, const float* __restrict__ m,
                                    const float* __restrict__ s, float* __restrict__ out, int64_t n, int64_t per_sample, int C)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float f = m ? m[i % C] : 1.f;
        if (s) f *= s[i / per_sample];
        out[i] = (res ? res[i] : 0.f) + t[i] * f;
    }
}

// dt = dy * m[c] * s[b] ; dm[c] = sum dy * t * s[b]   (block = pixel range, thread = channel)
__global__ __launch_bounds__(256) void tp_scale_add_bwd_kernel(const float* __restrict__ t, const float* __restrict__ m,
                                                               const float* __restrict__ s, const float* __restrict__ dy,
                                                               float* __restrict__ dt, float* __restrict__ partial, int64_t npix,
                                                               int64_t pix_per_sample, int C)
{
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mc = m ? m[c] : 1.f;
        float acc = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const float sc = s ? s[p / pix_per_sample] : 1.f;
            const float g = dy[p * C + c];
            dt[p * C + c] = g * mc * sc;
            acc = fmaf(g * sc, t[p * C + c], acc);
        }
        partial[(size_t)blockIdx.x * C + c] = acc;
    }
}

// ChannelLearnableMultiplier: m = tanh(relu(1 + w))  ->  dw = dm * (1 - m^2) * [1 + w > 0]
__global__ void tp_multiplier_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dm, float* __restrict__ dw, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = 1.f + w[i], th = tanhf(fmaxf(a, 0.f));
    dw[i] = a > 0.f ? dm[i] * (1.f - th * th) : 0.f;
}

// Laplacian split backward.  forward: smooth = AveragePooling2D(k, 1, same)(x) [divisor = in-bounds taps] or the fixed
// Gaussian depthwise filter (zero padding); lap = x - smooth; down = smooth[:, ::2, ::2].
// gs = -dlap + scatter(ddown) is the gradient with respect to smooth; dx = dlap + smooth^T(gs):
//   average: sum over the in-bounds neighbours q of gs[q] / count(q) ; Gaussian: sum of gs[q] * g[p - q] (symmetric window)
__global__ void tp_smooth_split_bwd_kernel(const float* __restrict__ dlap, const float* __restrict__ ddown,
                                           const float* __restrict__ gauss, float* __restrict__ dx, int B, int H, int W, int C, int k,
                                           int down_stride)
{
    // window of output q: inputs q - pb .. q - pb + k - 1 (TF "same": pad_before = (k - 1) / 2, the extra one after for even k)
    const int pb = (k - 1) / 2, Hd = (H + 1) / 2, Wd = (W + 1) / 2;
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        int64_t p = e / C;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int64_t b = p / H;
        float acc = 0.f;
        for (int i = 0; i < k; ++i) {
            const int qy = y + pb - i;                        // the outputs whose window holds input row y (tap i)
            if (qy < 0 || qy >= H) continue;
            for (int j = 0; j < k; ++j) {
                const int qx = x + pb - j;
                if (qx < 0 || qx >= W) continue;
                float gs = -dlap[((b * H + qy) * W + qx) * C + c];
                if (down_stride == 1) gs += ddown[((b * H + qy) * W + qx) * C + c];
                else if (!(qy & 1) && !(qx & 1)) gs += ddown[((b * Hd + (qy >> 1)) * Wd + (qx >> 1)) * C + c];
                if (gauss) {
                    acc = fmaf(gs, gauss[i * k + j], acc);
                } else {
                    const int cy = min(qy - pb + k - 1, H - 1) - max(qy - pb, 0) + 1, cx = min(qx - pb + k - 1, W - 1) - max(qx - pb, 0) + 1;
                    acc += gs / (float)(cy * cx);
                }
            }
        }
        dx[e] = dlap[e] + acc;
    }
}

// adjoint of UpSampling2D(2): bilinear (half-pixel centres: out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] +
// .25 in[i+1], indices clamped) or nearest.  dx [B,H,W,C] from dy [B,2H,2W,C]
__device__ __forceinline__ int tp_up_taps(int i, int n, int (&o)[4], float (&w)[4])
{
    // outputs that read input i, with their weights (clamped reads at the borders fold onto the border sample)
    int cnt = 0;
    o[cnt] = 2 * i; w[cnt++] = 0.75f + (i == 0 ? 0.25f : 0.f);
    o[cnt] = 2 * i + 1; w[cnt++] = 0.75f + (i == n - 1 ? 0.25f : 0.f);
    if (i + 1 < n) { o[cnt] = 2 * i + 2; w[cnt++] = 0.25f; }
    if (i > 0) { o[cnt] = 2 * i - 1; w[cnt++] = 0.25f; }
    return cnt;
}

__global__ void tp_upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C, int bilinear)
{
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        int64_t p = e / C;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int64_t b = p / H;
        float acc = 0.f;
        if (bilinear) {
            int oy[4], ox[4];
            float wy[4], wx[4];
            const int ny = tp_up_taps(y, H, oy, wy), nx = tp_up_taps(x, W, ox, wx);
            for (int i = 0; i < ny; ++i)
                for (int j = 0; j < nx; ++j) acc = fmaf(wy[i] * wx[j], dy[((b * 2 * H + oy[i]) * 2 * W + ox[j]) * C + c], acc);
        } else {
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) acc += dy[((b * 2 * H + 2 * y + i) * 2 * W + 2 * x + j) * C + c];
        }
        dx[e] = acc;
    }
}

// k x k convolution weight gradient for a few input channels (the network's first convolution): element e of the
// [k][k][cin][cout] kernel per thread, workgroup = pixel range.  x is the raw image (uint8 or float): normalised here as the
// forward does (clip to the value range, / range, - 0.5) when `normalize`.
__global__ __launch_bounds__(256) void tp_conv2d_wgrad_kernel(const void* __restrict__ xin, int x_is_u8, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int B, int H, int W, int cin, int cout,
                                                              int k, int normalize, float vmin, float vmax)
{
    const int nel = k * k * cin * cout, pad = (k - 1) / 2;
    const int64_t npix = (int64_t)B * H * W;
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const int co = e % cout, ci = (e / cout) % cin, j = (e / (cout * cin)) % k, i = e / (cout * cin * k);
        float acc = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const int xw = (int)(p % W), yh = (int)((p / W) % H);
            const int64_t b = p / ((int64_t)W * H);
            const int yy = yh + i - pad, xx = xw + j - pad;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t idx = ((b * H + yy) * W + xx) * cin + ci;
            float v = x_is_u8 ? (float)reinterpret_cast<const unsigned char*>(xin)[idx] : reinterpret_cast<const float*>(xin)[idx];
            if (normalize) v = (fminf(fmaxf(v, vmin), vmax) - vmin) / (vmax - vmin) - 0.5f;
            acc = fmaf(v, dy[p * cout + co], acc);
        }
        partial[(size_t)blockIdx.x * nel + e] = acc;
    }
}

// denoiser head, last stage backward: h [npix, hf] (activated hidden layer), w1 [hf, co]:
//   h1 = h . w1, th = tanh(2 h1), p = 0.51 th, pred = denormalise(clip(p)) ; given dL/dpred:
//   dh1 = dpred * [|p| <= 0.5] * range * 0.51 * 2 * (1 - th^2) ; dh = dh1 . w1^T ; dw1 = sum h (x) dh1
__global__ __launch_bounds__(256) void tp_head_out_bwd_kernel(const float* __restrict__ h, const float* __restrict__ w1,
                                                              const float* __restrict__ dpred, float* __restrict__ dh,
                                                              float* __restrict__ partial, int64_t npix, int hf, int co,
                                                              int denormalize, float vmin, float vmax)
{
    __shared__ float ws[256 * 4];
    __shared__ float red[4][256 * 4];
    for (int i = threadIdx.x; i < hf * co; i += 256) ws[i] = w1[i];
    __syncthreads();
    // thread = hidden channel j (hf <= 256 / waves...): simpler: thread handles whole pixels, accumulates its own dw1 rows
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                // this thread's hidden channel j = threadIdx.x % hf, pixels strided
    const int j = threadIdx.x % hf, lanes = 256 / hf;   // hf divides 256 (32, 64, 128)
    const int sub = threadIdx.x / hf;
    for (int64_t p = (int64_t)blockIdx.x * lanes + sub; p < npix; p += (int64_t)gridDim.x * lanes) {
        // every thread of the pixel's group recomputes h1 (hf x co MACs): cheap next to the traffic
        float h1[4] = {0.f, 0.f, 0.f, 0.f};
        for (int jj = 0; jj < hf; ++jj) {
            const float hv = h[p * hf + jj];
            for (int o = 0; o < co; ++o) h1[o] = fmaf(hv, ws[jj * co + o], h1[o]);
        }
        const float hj = h[p * hf + j];
        float d = 0.f;
        for (int o = 0; o < co; ++o) {
            const float th = tanhf(2.f * h1[o]), pv = 0.51f * th;
            float g = dpred[p * co + o];
            if (denormalize) g *= (pv >= -0.5f && pv <= 0.5f) ? (vmax - vmin) : 0.f;
            const float dh1 = g * 1.02f * (1.f - th * th);
            d = fmaf(dh1, ws[j * co + o], d);
            acc[o] = fmaf(hj, dh1, acc[o]);
        }
        dh[p * hf + j] = d;
    }
    for (int o = 0; o < 4; ++o) red[o][threadIdx.x] = acc[o];
    __syncthreads();
    if ((int)threadIdx.x < hf * co) {
        const int jj = threadIdx.x / co, o = threadIdx.x % co;
        float s = 0.f;
        for (int g = 0; g < lanes; ++g) s += red[o][g * hf + jj];
        partial[(size_t)blockIdx.x * hf * co + threadIdx.x] = s;
    }
}

// per-image partial sums of the loss terms in the row format of head_train_kernel (80 floats per row, columns 64..67:
// sum min(|e|, 255), hinge sum, sum relu(e)^2 capped, sum relu(e; hinge, cutoff^2)^2)
__global__ __launch_bounds__(256) void tp_loss_sums_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           float* __restrict__ partial, int64_t per_image, int blocks_per_image,
                                                           float hinge, float cutoff)
{
    __shared__ float red[4][4];
    const int b = blockIdx.x / blocks_per_image, sub = blockIdx.x % blocks_per_image;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)sub * 256 + threadIdx.x; i < per_image; i += (int64_t)blocks_per_image * 256) {
        const float e = gt[b * per_image + i] - pred[b * per_image + i], ae = fabsf(e);
        s[0] += fminf(ae, 255.0f);
        s[1] += ae > hinge ? fminf(ae, cutoff) : 0.f;
        const float ep = e > 0.f ? fminf(e, 255.0f) : 0.f;
        s[2] += ep * ep;
        const float eh = e > hinge ? fminf(e, cutoff * cutoff) : 0.f;
        s[3] += eh * eh;
    }
    for (int k = 0; k < 4; ++k) {
        float v = s[k];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 80) {
        float v = 0.f;
        if (threadIdx.x >= 64 && threadIdx.x < 68) {
            const int k = threadIdx.x - 64;
            v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        }
        partial[(size_t)blockIdx.x * 80 + threadIdx.x] = v;
    }
}

// dpred = L1 term + dextra ; L1: -sign(e) * dscale where hinge < |e| < cutoff
__global__ void tp_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ dextra,
                                    float* __restrict__ dpred, int64_t n, float hinge, float cutoff, float dscale)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float e = gt[i] - pred[i], ae = fabsf(e);
        float g = 0.f;
        if (ae > hinge && ae < cutoff) g = (e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f)) * dscale;
        if (dextra) g += dextra[i];
        dpred[i] = g;
    }
}

// losses[] slots from the partial rows (one workgroup, fixed order): as head_finalize_kernel
__global__ __launch_bounds__(256) void tp_loss_finalize_kernel(const float* __restrict__ partial, int nblk, int blocks_per_image, int B,
                                                               double numel, double per_image, float mae_multiplier,
                                                               float depth_weight, float* __restrict__ losses)
{
    __shared__ double red[256];
    __shared__ double sums[2];
    for (int col = 0; col < 2; ++col) {
        double s = 0.0;
        for (int r = threadIdx.x; r < nblk; r += 256) s += (double)partial[(size_t)r * 80 + 64 + col];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[col] = red[0];
        __syncthreads();
    }
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        double sq = 0.0;
        for (int k = 0; k < blocks_per_image; ++k) sq += (double)partial[(size_t)(b * blocks_per_image + k) * 80 + 66];
        acc += sqrt(sq / per_image + 1e-3);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mae_loss = mae_multiplier > 0.f ? sums[1] / numel : 0.0;
        losses[BF_LOSS_MAE] = (float)(sums[0] / numel);
        losses[BF_LOSS_MSE] = (float)(red[0] / (double)B);
        losses[BF_LOSS_SSIM] = 0.f;
        losses[BF_LOSS_DENOISER_TOTAL] = (float)(mae_loss * mae_multiplier);
        losses[BF_LOSS_TOTAL] = (float)(mae_loss * mae_multiplier * depth_weight);
        losses[BF_LOSS_REGULARIZATION] = 0.f;
        losses[BF_LOSS_MODEL_TOTAL] = 0.f;
        losses[BF_LOSS_GRAD_NORM] = 0.f;
    }
}

// dot-product attention, one workgroup per (batch, query): P = softmax(q k^T) [* pscale], out = P v.  Training forward
// (stores P) and backward: dV = P'^T dO, dP' = dO V^T, dS = P (scale dP' - sum(scale dP' P)), dQ = dS K, dK = dS^T Q
__global__ __launch_bounds__(256) void tp_attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ v,
                                                               const float* __restrict__ k, const float* __restrict__ pscale,
                                                               float* __restrict__ out, float* __restrict__ P, int T, int A)
{
    extern __shared__ float sm[];                        // [T] scores / probabilities, [A] query
    float* sc = sm;
    float* qv = sm + T;
    const int b = blockIdx.x / T, i = blockIdx.x % T;
    for (int a = threadIdx.x; a < A; a += 256) qv[a] = q[((size_t)b * T + i) * A + a];
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += 256) {
        float s = 0.f;
        for (int a = 0; a < A; ++a) s = fmaf(qv[a], k[((size_t)b * T + j) * A + a], s);
        sc[j] = s;
    }
    __syncthreads();
    __shared__ float redv[256];
    float mx = -3.4e38f;
    for (int j = threadIdx.x; j < T; j += 256) mx = fmaxf(mx, sc[j]);
    redv[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) redv[threadIdx.x] = fmaxf(redv[threadIdx.x], redv[threadIdx.x + st]); __syncthreads(); }
    mx = redv[0];
    __syncthreads();
    float sum = 0.f;
    for (int j = threadIdx.x; j < T; j += 256) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
    redv[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) redv[threadIdx.x] += redv[threadIdx.x + st]; __syncthreads(); }
    const float inv = 1.f / redv[0];
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += 256) {
        const float p = sc[j] * inv;
        if (P) P[((size_t)b * T + i) * T + j] = p;                          // the softmax itself (in front of the dropout scale)
        sc[j] = pscale ? p * pscale[((size_t)b * T + i) * T + j] : p;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += 256) {
        float o = 0.f;
        for (int j = 0; j < T; ++j) o = fmaf(sc[j], v[((size_t)b * T + j) * A + a], o);
        out[((size_t)b * T + i) * A + a] = o;
    }
}

// pass 1 (per query row): dS row and dQ ; dS written over a scratch [B,T,T]
__global__ __launch_bounds__(256) void tp_attention_bwd_rows_kernel(const float* __restrict__ P, const float* __restrict__ pscale,
                                                                    const float* __restrict__ v, const float* __restrict__ k,
                                                                    const float* __restrict__ dout, float* __restrict__ dS,
                                                                    float* __restrict__ dq, int T, int A)
{
    extern __shared__ float sm[];
    float* ds = sm;                                      // [T]
    float* dov = sm + T;                                 // [A]
    __shared__ float redv[256];
    const int b = blockIdx.x / T, i = blockIdx.x % T;
    for (int a = threadIdx.x; a < A; a += 256) dov[a] = dout[((size_t)b * T + i) * A + a];
    __syncthreads();
    float part = 0.f;
    for (int j = threadIdx.x; j < T; j += 256) {
        float dp = 0.f;
        for (int a = 0; a < A; ++a) dp = fmaf(dov[a], v[((size_t)b * T + j) * A + a], dp);
        if (pscale) dp *= pscale[((size_t)b * T + i) * T + j];
        ds[j] = dp;
        part = fmaf(dp, P[((size_t)b * T + i) * T + j], part);
    }
    redv[threadIdx.x] = part;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) redv[threadIdx.x] += redv[threadIdx.x + st]; __syncthreads(); }
    const float dot = redv[0];
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += 256) {
        const float s = P[((size_t)b * T + i) * T + j] * (ds[j] - dot);
        ds[j] = s;
        dS[((size_t)b * T + i) * T + j] = s;
    }
    __syncthreads();
    for (int a = threadIdx.x; a < A; a += 256) {
        float o = 0.f;
        for (int j = 0; j < T; ++j) o = fmaf(ds[j], k[((size_t)b * T + j) * A + a], o);
        dq[((size_t)b * T + i) * A + a] = o;
    }
}

// pass 2 (per key row j): dK_j = sum_i dS_ij Q_i ; dV_j = sum_i P'_ij dO_i
__global__ __launch_bounds__(256) void tp_attention_bwd_cols_kernel(const float* __restrict__ P, const float* __restrict__ pscale,
                                                                    const float* __restrict__ dS, const float* __restrict__ q,
                                                                    const float* __restrict__ dout, float* __restrict__ dk,
                                                                    float* __restrict__ dv, int T, int A)
{
    const int b = blockIdx.x / T, j = blockIdx.x % T;
    for (int a = threadIdx.x; a < 2 * A; a += 256) {
        const int aa = a % A;
        float o = 0.f;
        if (a < A) {
            for (int i = 0; i < T; ++i) o = fmaf(dS[((size_t)b * T + i) * T + j], q[((size_t)b * T + i) * A + aa], o);
            dk[((size_t)b * T + j) * A + aa] = o;
        } else {
            for (int i = 0; i < T; ++i) {
                float p = P[((size_t)b * T + i) * T + j];
                if (pscale) p *= pscale[((size_t)b * T + i) * T + j];
                o = fmaf(p, dout[((size_t)b * T + i) * A + aa], o);
            }
            dv[((size_t)b * T + j) * A + aa] = o;
        }
    }
}

// adjoint of tf.image.resize(bilinear, half-pixel centres) along ONE axis: dx[.., i, ..] = sum over the outputs o whose
// two taps touch i.  n_in -> n_out ; `inner` = elements per index of the axis, `outer` = slabs in front of it
__global__ void tp_resize_bwd_axis_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t outer, int n_in, int n_out,
                                          int64_t inner)
{
    const int64_t n = outer * n_in * inner;
    const float ratio = (float)n_in / (float)n_out;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t in = e % inner;
        const int i = (int)((e / inner) % n_in);
        const int64_t ou = e / (inner * n_in);
        float acc = 0.f;
        for (int o = 0; o < n_out; ++o) {
            const float src = ((float)o + 0.5f) * ratio - 0.5f;
            const float fl = floorf(src);
            const int lo = max((int)fl, 0), hi = min((int)ceilf(src), n_in - 1);
            const float t = src - fl;
            float w = 0.f;
            if (lo == i) w += 1.f - t;
            if (hi == i) w += t;
            if (w != 0.f) acc = fmaf(w, dy[(ou * n_out + o) * inner + in], acc);
        }
        dx[e] = acc;
    }
}

// regularisers: value accumulated into *value (one workgroup, fixed order), gradient added to grad
__global__ __launch_bounds__(256) void tp_reg_elementwise_kernel(const float* __restrict__ w, float* __restrict__ grad, int64_t n, int kind,
                                                                 float coef, float grad_scale, float* __restrict__ value)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float x = w[i];
        if (kind == BF_REG_L1) {
            acc += fabs((double)x);
            if (grad) grad[i] += grad_scale * coef * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
        } else {
            acc += (double)x * (double)x;
            if (grad) grad[i] += grad_scale * coef * 2.f * x;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) value[0] = (float)((double)value[0] + (double)coef * red[0]);
}

// SoftOrthonormalConstraintRegularizer (regularizers.py:283-338) on a 1x1 kernel W [cin][cout]: G = W^T W [cout][cout],
// value = lambda ||G - I||_F^2 + l1 sum|G| + l2 sum G^2 ; D = dvalue/dG = 2 lambda (G - I) + l1 sign(G) + 2 l2 G ;
// dvalue/dW = 2 W D (D symmetric)
// mask_diagonal: SoftOrthogonalConstraintRegularizer (regularizers.py:208-280) -- the same three terms on G with its diagonal zeroed
// (no pull of the norms towards 1)
__global__ void tp_so_gram_kernel(const float* __restrict__ w, float* __restrict__ G, float* __restrict__ D, int cin, int cout, float lam,
                                  float l1, float l2, int mask_diagonal)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= cout * cout) return;
    const int a = e / cout, b = e % cout;
    double s = 0.0;
    for (int c = 0; c < cin; ++c) s += (double)w[c * cout + a] * (double)w[c * cout + b];
    if (mask_diagonal) {
        if (a == b) s = 0.0;
        G[e] = (float)(lam * s * s + l1 * fabs(s) + l2 * s * s);
        D[e] = a == b ? 0.f : (float)(2.0 * lam * s + l1 * (s > 0 ? 1.0 : (s < 0 ? -1.0 : 0.0)) + 2.0 * l2 * s);
        return;
    }
    const double gm = s - (a == b ? 1.0 : 0.0);
    G[e] = (float)(lam * gm * gm + l1 * fabs(s) + l2 * s * s);                                    // this element's share of the value
    D[e] = (float)(2.0 * lam * gm + l1 * (s > 0 ? 1.0 : (s < 0 ? -1.0 : 0.0)) + 2.0 * l2 * s);
}

__global__ __launch_bounds__(256) void tp_so_apply_kernel(const float* __restrict__ w, const float* __restrict__ G, const float* __restrict__ D,
                                                          float* __restrict__ grad, int cin, int cout, float grad_scale,
                                                          float* __restrict__ value)
{
    // one workgroup: the gradient 2 W D element by element, then the value in a fixed order
    __shared__ double red[256];
    for (int e = threadIdx.x; grad && e < cin * cout; e += 256) {
        const int c = e / cout, a = e % cout;
        double s = 0.0;
        for (int b = 0; b < cout; ++b) s += (double)w[c * cout + b] * (double)D[b * cout + a];
        grad[e] += grad_scale * (float)(2.0 * s);
    }
    double acc = 0.0;
    for (int e = threadIdx.x; e < cout * cout; e += 256) acc += (double)G[e];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) value[0] = (float)((double)value[0] + red[0]);
}

__global__ void tp_flip_hw_kernel(const float* __restrict__ w, float* __restrict__ out, int k, int inner)
{
    const int n = k * k * inner;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int in = e % inner, t = e / inner;
    out[(k * k - 1 - t) * inner + in] = w[e];
}

// [a][b] -> [b][a]
__global__ void tp_transpose_kernel(const float* __restrict__ w, float* __restrict__ out, int a, int b)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= a * b) return;
    out[(e % b) * a + e / b] = w[e];
}

// adjoint of MaxPooling2D(2, 2, same) (downsampling.py:56-68): the 2 x 2 windows do not overlap, so every input belongs to one
// window: dx = dy at the window's maximum (the first one in row-major window order on a tie, as TensorFlow's and torch's
// max-pool gradients route it), 0 elsewhere.  Thread = 4 channels of one OUTPUT element.
__global__ __launch_bounds__(256) void tp_maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                              int B, int H, int W, int C)
{
    const int Cv = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const int64_t base = (int64_t)b * H * W * Cv + c;
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
            if (yy >= H || xx >= W) continue;
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[base + ((int64_t)yy * W + xx) * Cv];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > best[j]) { best[j] = v[j]; arg[j] = k; }
        }
        for (int k = 0; k < 4; ++k) {
            const int yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
            if (yy >= H || xx >= W) continue;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = arg[j] == k ? g[j] : 0.f;
            reinterpret_cast<f32x4*>(dx)[base + ((int64_t)yy * W + xx) * Cv] = o;
        }
    }
}

}  // namespace

#define TP_OK() (hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP)

extern "C" int bf_op_act_bwd(const float* ref, const float* dy, float* dx, int64_t n, int act, float alpha, int ref_is_output, void* stream)
{
    if (!ref || !dy || !dx || n <= 0) return BF_EINVAL;
    if (act < 0 || act > 3 || (act == 3 && ref_is_output)) return BF_EUNSUPPORTED;      // GELU is not sign-preserving
    hipLaunchKernelGGL(tp_act_bwd_kernel, dim3(tp_grid(n)), dim3(256), 0, (hipStream_t)stream, ref, dy, dx, n, act, alpha, ref_is_output);
    return TP_OK();
}

extern "C" int bf_op_matmul_wgrad(const float* x, const float* dy, float* dw, int64_t npix, int cin, int cout, float* scratch,
                                  int64_t scratch_floats, void* stream)
{
    if (!x || !dy || !dw || !scratch || npix <= 0 || cin <= 0 || cout <= 0) return BF_EINVAL;
    int nsplit = (int)(npix / 4096);
    nsplit = nsplit < 1 ? 1 : (nsplit > 64 ? 64 : nsplit);
    while (nsplit > 1 && (int64_t)nsplit * cin * cout > scratch_floats) --nsplit;
    if ((int64_t)nsplit * cin * cout > scratch_floats) return BF_EWORKSPACE;
    const int ti = (cin + 15) / 16, to = (cout + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_matmul_wgrad_kernel, dim3(ti * to * nsplit), dim3(256), 0, s, x, dy, scratch, npix, cin, cout, ti, to, nsplit);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    return tp_reduce(scratch, nsplit, cin * cout, dw, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_dwconv_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int C, int k, float* scratch,
                                  int64_t scratch_floats, void* stream)
{
    if (!x || !dy || !dw || !scratch || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    if (k != 1 && k != 3 && k != 5 && k != 7) return BF_EUNSUPPORTED;
    const int64_t npix = (int64_t)B * H * W;
    int grid = tp_grid(npix, 512, 256);
    while (grid > 1 && (int64_t)grid * k * k * C > scratch_floats) --grid;
    if ((int64_t)grid * k * k * C > scratch_floats) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
#define TP_DW(K) hipLaunchKernelGGL(tp_dwconv_wgrad_kernel<K>, dim3(grid), dim3(256), 0, s, x, dy, scratch, B, H, W, C)
    if (k == 1) TP_DW(1); else if (k == 3) TP_DW(3); else if (k == 5) TP_DW(5); else TP_DW(7);
#undef TP_DW
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    return tp_reduce(scratch, grid, k * k * C, dw, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, int64_t npix, int C,
                                   float eps, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!x || !gamma || !dy || !dx || !dgamma || !scratch || npix <= 0) return BF_EINVAL;
    if (C <= 0 || C > 256) return BF_EUNSUPPORTED;
    int grid = tp_grid(npix, 64, 512);
    while (grid > 1 && (int64_t)grid * C > scratch_floats) --grid;
    if ((int64_t)grid * C > scratch_floats) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_layernorm_bwd_kernel, dim3(grid), dim3(256), 0, s, x, gamma, dy, dx, scratch, npix, C, eps);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    return tp_reduce(scratch, grid, C, dgamma, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_scale_add(const float* res, const float* t, const float* m, const float* sample_scale, float* out, int B,
                               int64_t hw, int C, void* stream)
{
    if (!t || !out || B <= 0 || hw <= 0 || C <= 0) return BF_EINVAL;
    const int64_t n = (int64_t)B * hw * C;
    hipLaunchKernelGGL(tp_scale_add_kernel, dim3(tp_grid(n)), dim3(256), 0, (hipStream_t)stream, res, t, m, sample_scale, out, n, hw * C, C);
    return TP_OK();
}

extern "C" int bf_op_scale_add_bwd(const float* t, const float* m, const float* sample_scale, const float* dy, float* dt, float* dm,
                                   int B, int64_t hw, int C, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!t || !dy || !dt || !scratch || B <= 0 || hw <= 0 || C <= 0) return BF_EINVAL;
    const int64_t npix = (int64_t)B * hw;
    int grid = tp_grid(npix, 256, 512);
    while (grid > 1 && (int64_t)grid * C > scratch_floats) --grid;
    if ((int64_t)grid * C > scratch_floats) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_scale_add_bwd_kernel, dim3(grid), dim3(256), 0, s, t, m, sample_scale, dy, dt, scratch, npix, hw, C);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    if (!dm) return BF_OK;
    return tp_reduce(scratch, grid, C, dm, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_multiplier_bwd(const float* w, const float* dm, float* dw, int n, void* stream)
{
    if (!w || !dm || !dw || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tp_multiplier_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, dm, dw, n);
    return TP_OK();
}

// down_stride 2: ddown is the gradient of smooth[:, ::2, ::2]; 1: of smooth itself (conv2d / maxpool down-sampling take the
// full-resolution smooth map).  Averaging: any k <= 7; Gaussian: odd k (symmetric window)
extern "C" int bf_op_smooth_split_bwd_ex(const float* dlap, const float* ddown, const float* gauss, float* dx, int B, int H, int W, int C,
                                         int k, int down_stride, void* stream)
{
    if (!dlap || !ddown || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (down_stride != 1 && down_stride != 2)) return BF_EINVAL;
    if (k < 1 || k > 7) return BF_EUNSUPPORTED;                      // any window, even or odd, averaging or Gaussian (same pad_before as the forward)
    const int64_t n = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(tp_smooth_split_bwd_kernel, dim3(tp_grid(n)), dim3(256), 0, (hipStream_t)stream, dlap, ddown, gauss, dx, B, H, W, C, k,
                       down_stride);
    return TP_OK();
}

extern "C" int bf_op_smooth_split_bwd(const float* dlap, const float* ddown, const float* gauss, float* dx, int B, int H, int W, int C,
                                      int k, void* stream)
{
    return bf_op_smooth_split_bwd_ex(dlap, ddown, gauss, dx, B, H, W, C, k, 2, stream);
}

extern "C" int bf_op_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, int bilinear, void* stream)
{
    if (!dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    const int64_t n = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(tp_upsample2x_bwd_kernel, dim3(tp_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C, bilinear);
    return TP_OK();
}

extern "C" int bf_op_conv2d_wgrad(const void* x, int x_is_u8, const float* dy, float* dw, int B, int H, int W, int cin, int cout, int k,
                                  int normalize, float v_min, float v_max, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!x || !dy || !dw || !scratch || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return BF_EINVAL;
    if (k < 1 || k > 7 || !(k & 1) || (normalize && !(v_max > v_min))) return BF_EUNSUPPORTED;
    const int nel = k * k * cin * cout;
    int grid = tp_grid((int64_t)B * H * W, 1024, 256);
    while (grid > 1 && (int64_t)grid * nel > scratch_floats) --grid;
    if ((int64_t)grid * nel > scratch_floats) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_conv2d_wgrad_kernel, dim3(grid), dim3(256), 0, s, x, x_is_u8, dy, scratch, B, H, W, cin, cout, k, normalize, v_min, v_max);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    return tp_reduce(scratch, grid, nel, dw, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_head_out_bwd(const float* h, const float* w1, const float* dpred, float* dh, float* dw1, int64_t npix, int hf,
                                  int cout, int denormalize, float v_min, float v_max, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!h || !w1 || !dpred || !dh || !dw1 || !scratch || npix <= 0) return BF_EINVAL;
    if ((hf != 32 && hf != 64 && hf != 128) || cout <= 0 || cout > 4) return BF_EUNSUPPORTED;
    int grid = tp_grid(npix, 256 / hf * 16, 512);
    while (grid > 1 && (int64_t)grid * hf * cout > scratch_floats) --grid;
    if ((int64_t)grid * hf * cout > scratch_floats) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_head_out_bwd_kernel, dim3(grid), dim3(256), 0, s, h, w1, dpred, dh, scratch, npix, hf, cout, denormalize, v_min, v_max);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    return tp_reduce(scratch, grid, hf * cout, dw1, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

static int tp_loss_bpi(int64_t per_image) { const int64_t b = (per_image + 4095) / 4096; return (int)(b < 1 ? 1 : (b > 64 ? 64 : b)); }

extern "C" int64_t bf_op_denoiser_loss_scratch_floats(int B, int H, int W, int C)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return -1;
    const int64_t n = (int64_t)B * H * W * C;
    return (int64_t)B * tp_loss_bpi((int64_t)H * W * C) * 80 + n /* dextra */ + 3 * n /* ssim maps */ + 4096 + B + 64 + 64;
}

// denoiser_loss (bfcnn/loss.py:190-247) of one output scale and its gradient: losses[] slots as bf_train_step fills them
// (BF_LOSS_TOTAL = denoiser total * depth_weight; the regularisation slots are zeroed), dpred = d(BF_LOSS_TOTAL)/dpred
extern "C" int bf_op_denoiser_loss(const float* pred, const float* gt, int B, int H, int W, int C, const bf_loss_desc* loss, float* dpred,
                                   float* losses, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!pred || !gt || !loss || !dpred || !losses || !scratch || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    if (loss->struct_size != (int32_t)sizeof(bf_loss_desc)) return BF_EINVAL;
    if (scratch_floats < bf_op_denoiser_loss_scratch_floats(B, H, W, C)) return BF_EWORKSPACE;
    const bool use_ssim = loss->ssim_multiplier > 0.f, use_mse = loss->mse_multiplier > 0.f;
    if (use_ssim && (H < 7 || W < 7)) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t per_image = (int64_t)H * W * C, n = per_image * B;
    const int bpi = tp_loss_bpi(per_image);
    float* partial = scratch;
    float* dextra = partial + (int64_t)B * bpi * 80;
    float* maps = dextra + n;
    float* ssim_partial = maps + 3 * n;
    float* coef = ssim_partial + 4096;
    float* scal = coef + ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(tp_loss_sums_kernel, dim3(B * bpi), dim3(256), 0, s, pred, gt, partial, per_image, bpi, loss->hinge, loss->cutoff);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    if (use_ssim || use_mse) {
        if (bf_launch_loss_extra(pred, gt, B, H, W, C, partial, bpi, loss->hinge, loss->cutoff, use_mse ? loss->mse_multiplier : 0.f,
                                 use_ssim ? loss->ssim_multiplier : 0.f, loss->depth_weight, 255.0f, maps, ssim_partial, coef, scal, dextra,
                                 s) != hipSuccess)
            return BF_EHIP;
    }
    const float dscale = loss->mae_multiplier > 0.f ? (float)((double)loss->mae_multiplier * loss->depth_weight / (double)n) : 0.f;
    hipLaunchKernelGGL(tp_loss_grad_kernel, dim3(tp_grid(n)), dim3(256), 0, s, pred, gt, (use_ssim || use_mse) ? dextra : nullptr, dpred, n,
                       loss->hinge, loss->cutoff, dscale);
    hipLaunchKernelGGL(tp_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, B * bpi, bpi, B, (double)n, (double)per_image,
                       loss->mae_multiplier, loss->depth_weight, losses);
    if (hipGetLastError() != hipSuccess) return BF_EHIP;
    if (use_ssim || use_mse)
        if (bf_launch_loss_extra_finalize(scal, B, H, W, C, use_mse ? loss->mse_multiplier : 0.f, use_ssim ? loss->ssim_multiplier : 0.f,
                                          loss->depth_weight, losses, s) != hipSuccess)
            return BF_EHIP;
    return BF_OK;
}

// training forward of dot-product attention: out = (softmax(q k^T) * pscale) v ; P (softmax, [B,T,T]) kept for the backward
extern "C" int bf_op_attention_train(const float* q, const float* v, const float* k, const float* pscale, float* out, float* P, int B,
                                     int T, int A, void* stream)
{
    if (!q || !v || !k || !out || B <= 0 || T <= 0 || A <= 0) return BF_EINVAL;
    if (T > 4096 || A > 1024) return BF_EUNSUPPORTED;
    hipLaunchKernelGGL(tp_attention_fwd_kernel, dim3(B * T), dim3(256), (T + A) * sizeof(float), (hipStream_t)stream, q, v, k, pscale, out,
                       P, T, A);
    return TP_OK();
}

extern "C" int bf_op_attention_bwd(const float* q, const float* v, const float* k, const float* pscale, const float* P, const float* dout,
                                   float* dq, float* dv, float* dk, float* dS_scratch, int B, int T, int A, void* stream)
{
    if (!q || !v || !k || !P || !dout || !dq || !dv || !dk || !dS_scratch || B <= 0 || T <= 0 || A <= 0) return BF_EINVAL;
    if (T > 4096 || A > 1024) return BF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tp_attention_bwd_rows_kernel, dim3(B * T), dim3(256), (T + A) * sizeof(float), s, P, pscale, v, k, dout, dS_scratch, dq, T, A);
    hipLaunchKernelGGL(tp_attention_bwd_cols_kernel, dim3(B * T), dim3(256), 0, s, P, pscale, dS_scratch, q, dout, dk, dv, T, A);
    return TP_OK();
}

// adjoint of bf_op_resize_bilinear ([B,H,W,C] -> [B,oh,ow,C]): dx [B,H,W,C] from dy [B,oh,ow,C]; scratch: B*H*ow*C floats
extern "C" int bf_op_resize_bilinear_bwd(const float* dy, float* dx, int B, int H, int W, int C, int oh, int ow, float* scratch, void* stream)
{
    if (!dy || !dx || !scratch || B <= 0 || H <= 0 || W <= 0 || C <= 0 || oh <= 0 || ow <= 0) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n1 = (int64_t)B * H * ow * C, n2 = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(tp_resize_bwd_axis_kernel, dim3(tp_grid(n1)), dim3(256), 0, s, dy, scratch, (int64_t)B, H, oh, (int64_t)ow * C);
    hipLaunchKernelGGL(tp_resize_bwd_axis_kernel, dim3(tp_grid(n2)), dim3(256), 0, s, scratch, dx, (int64_t)B * H, W, ow, (int64_t)C);
    return TP_OK();
}

// value[0] += coef * sum |w| (L1) or coef * sum w^2 (L2) ; grad += grad_scale * d/dw
extern "C" int bf_op_reg_elementwise(const float* w, float* grad, int64_t n, int kind, float coef, float grad_scale, float* value, void* stream)
{
    if (!w || !value || n <= 0) return BF_EINVAL;              // grad may be NULL: the value alone
    if (kind != BF_REG_L1 && kind != BF_REG_L2) return BF_EUNSUPPORTED;
    hipLaunchKernelGGL(tp_reg_elementwise_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w, grad, n, kind, coef, grad_scale, value);
    return TP_OK();
}

// SoftOrthonormalConstraintRegularizer on a 1x1 kernel [cin][cout]; scratch: 2 * cout * cout floats
extern "C" int bf_op_reg_soft_orthogonal_ex(const float* w, float* grad, int cin, int cout, float lambda, float l1, float l2, float grad_scale,
                                            float* value, float* scratch, int mask_diagonal, void* stream)
{
    if (!w || !value || !scratch || cin <= 0 || cout <= 0) return BF_EINVAL;                // grad may be NULL: the value alone
    hipStream_t s = (hipStream_t)stream;
    float* G = scratch;
    float* D = scratch + (size_t)cout * cout;
    hipLaunchKernelGGL(tp_so_gram_kernel, dim3((cout * cout + 255) / 256), dim3(256), 0, s, w, G, D, cin, cout, lambda, l1, l2, mask_diagonal);
    hipLaunchKernelGGL(tp_so_apply_kernel, dim3(1), dim3(256), 0, s, w, G, D, grad, cin, cout, grad_scale, value);
    return TP_OK();
}

extern "C" int bf_op_reg_soft_orthonormal(const float* w, float* grad, int cin, int cout, float lambda, float l1, float l2, float grad_scale,
                                          float* value, float* scratch, void* stream)
{
    if (!grad) return BF_EINVAL;
    return bf_op_reg_soft_orthogonal_ex(w, grad, cin, cout, lambda, l1, l2, grad_scale, value, scratch, 0, stream);
}

// spatially flipped copy of a [k][k][inner] kernel (depthwise data gradient = depthwise convolution with it)
extern "C" int bf_op_flip_hw(const float* w, float* out, int k, int inner, void* stream)
{
    if (!w || !out || k <= 0 || inner <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tp_flip_hw_kernel, dim3((k * k * inner + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, k, inner);
    return TP_OK();
}

// [a][b] -> [b][a] (1x1 data gradient = 1x1 convolution with the transposed kernel)
extern "C" int bf_op_transpose2d(const float* w, float* out, int a, int b, void* stream)
{
    if (!w || !out || a <= 0 || b <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tp_transpose_kernel, dim3((a * b + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, a, b);
    return TP_OK();
}

extern "C" int bf_op_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream)
{
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return BF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(tp_maxpool2_bwd_kernel, dim3(tp_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, B, H, W, C);
    return TP_OK();
}
