// Training-mode operators of the resnet builder outside the 16-filter 3x3 engine (bfcnn/backbone_resnet.py:36-298,
// backbone_blocks.py:163-246): BatchNormalization with batch statistics (forward + backward) for any channel count that
// divides 256, the squeeze-style channel gate of `add_gates` (backbone_blocks.py:199-208: mean over the image -> Dense relu ->
// Dense hard_sigmoid -> Multiply) forward + backward, and the layout helpers that let depthwise-with-multiplier and grouped
// convolutions reuse the plain operators (channel repeat / group sum, block-diagonal expand / extract).
// Exact fp32 tensors, fp64 statistics, fixed summation order (bitwise reproducible), stream-ordered, no allocation.
// Written for correctness: these networks' training is a parity row (tests/test_gpu_resnet_generic_train.py), not a benchmark.
#include "bf_common.h"
#include <math.h>

namespace {

constexpr int CS_GRID = 128;          // row chunks per sample (or per tensor) of the column sums

// partial[(s * gridDim.x + blockIdx.x) * 2C + {c, C + c}] = sum over the chunk's rows of a[r][c] and a[r][c] * b[r][c]
// rows of sample s = blockIdx.y: [s * rows, (s + 1) * rows)
__global__ __launch_bounds__(256) void tg_colsum2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t rows, int C,
                                                         double* __restrict__ partial)
{
    __shared__ double r1[256], r2[256];
    const int tid = threadIdx.x, c = tid % C, rr = tid / C, R = 256 / C;
    const int64_t base = (int64_t)blockIdx.y * rows;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = lo + rr; r < hi; r += R) {
        const double av = a[(base + r) * C + c], bv = b[(base + r) * C + c];
        s1 += av;
        s2 += av * bv;
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    if (tid < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < R; ++k) { t1 += r1[k * C + tid]; t2 += r2[k * C + tid]; }
        double* p = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
        p[tid] = t1;
        p[C + tid] = t2;
    }
}

// BatchNorm forward finalisation (keras BatchNormalization(center=False), training=True; fused-kernel moving variance)
__global__ __launch_bounds__(256) void tg_bn_fwd_finalize_kernel(const double* __restrict__ partial, int nblk, double count, int C,
                                                                const float* __restrict__ gamma, float eps, float momentum,
                                                                float* moving_mean, float* moving_var, float* save, float* coef)
{
    const int c = threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nblk; ++k) { s1 += partial[(int64_t)k * 2 * C + c]; s2 += partial[(int64_t)k * 2 * C + C + c]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps), g = gamma[c];
    save[c] = (float)mean;
    save[C + c] = (float)inv;
    coef[c] = (float)(g * inv);                    // y = coef[c] * x + coef[C + c]
    coef[C + c] = (float)(-g * inv * mean);
    if (moving_mean) {
        const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
        moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
        moving_var[c] = (float)((double)moving_var[c] * momentum + unbiased * (1.0 - (double)momentum));
    }
}

// dgamma = sum dy * xhat ; dx = k1 dy + k2 x + k3 (train_ops.hip bn_bwd_finalize_kernel, any C)
__global__ __launch_bounds__(256) void tg_bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, double count, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ save,
                                                                float* coef, float* dgamma)
{
    const int c = threadIdx.x;
    if (c >= C) return;
    double sdy = 0.0, sdyx = 0.0;
    for (int k = 0; k < nblk; ++k) { sdy += partial[(int64_t)k * 2 * C + c]; sdyx += partial[(int64_t)k * 2 * C + C + c]; }
    const double mean = save[c], inv = save[C + c], g = gamma[c];
    const double sdyh = (sdyx - mean * sdy) * inv;            // sum dy * xhat
    dgamma[c] = (float)sdyh;
    const double mdy = sdy / count, mdyh = sdyh / count;
    coef[c] = (float)(g * inv);
    coef[C + c] = (float)(-g * inv * inv * mdyh);
    coef[2 * C + c] = (float)(-g * inv * mdy + g * inv * inv * mean * mdyh);
}

__device__ __forceinline__ float tg_act(float v, int act, float alpha)
{
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : alpha * v;
    return v;
}

// out = act(k1[c] * a + k2[c] * b + k3[c])   (b, k2, k3 optional)
__global__ __launch_bounds__(256) void tg_lincomb_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ k1, const float* __restrict__ k2,
                                                         const float* __restrict__ k3, float* __restrict__ out, int64_t n, int C, int act,
                                                         float alpha)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = k1[c] * a[i];
        if (b) v = fmaf(k2[c], b[i], v);
        if (k3) v += k3[c];
        out[i] = tg_act(v, act, alpha);
    }
}

// ---- channel gate (backbone_blocks.py:199-208) and its relatives ----------------------------------------------------------
// one workgroup per sample: m = mean over the image of the selector tensor (Cs channels) ; h = act0(m W0 + b0) ;
// p = h W1 + b1 ; g = F(p).  act0: 1 relu, 2 leaky relu (alpha0).  F (mode): 0 hard_sigmoid(p) = clip(0.2 p + 0.5, 0, 1) (the gate;
// squeeze_and_excite_block's hard version), 1 sigmoid(p) (squeeze_and_excite_block), 2 hard_sigmoid(2.5 - relu(p)) (its
// learn_to_turn_off form; selector_block GLOBAL / HARD), 3 sigmoid(2.5 - relu(p)) (selector_block GLOBAL / SOFT).
// save: [B][Cs] mean | [B][C8] h | [B][C] p | [B][C] g
__device__ __forceinline__ float tg_gate_final(float p, int mode)
{
    if (mode == 4) return fmaxf(p, 0.f);                       // plain relu (the selector's LOCAL map, in front of its up-sampling)
    if (mode >= 2) p = 2.5f - fmaxf(p, 0.f);
    if (mode == 1 || mode == 3) return 1.f / (1.f + expf(-p));
    return fminf(fmaxf(0.2f * p + 0.5f, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void tg_gate_dense_kernel(const double* __restrict__ partial, int nblk, double hw, int Cs, int C, int C8,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w1, const float* __restrict__ b1, int act0,
                                                            float alpha0, int mode, int B, float* __restrict__ save,
                                                            const float* __restrict__ direct, float* __restrict__ gout)
{
    __shared__ float m[256], h[64];
    const int b = blockIdx.x, t = threadIdx.x;
    float* s_mean = save + (int64_t)b * Cs;
    float* s_h = save + (int64_t)B * Cs + (int64_t)b * C8;
    float* s_p = save + (int64_t)B * (Cs + C8) + (int64_t)b * C;
    float* s_g = save + (int64_t)B * (Cs + C8 + C) + (int64_t)b * C;
    if (t < Cs) {
        if (direct) {
            m[t] = direct[(int64_t)b * Cs + t];
        } else {
            double s = 0.0;
            for (int k = 0; k < nblk; ++k) s += partial[((int64_t)b * nblk + k) * 2 * Cs + t];
            m[t] = (float)(s / hw);
        }
        if (save) s_mean[t] = m[t];
    }
    __syncthreads();
    if (t < C8) {
        float a = b0 ? b0[t] : 0.f;
        for (int c = 0; c < Cs; ++c) a = fmaf(m[c], w0[c * C8 + t], a);
        h[t] = act0 == 2 ? (a > 0.f ? a : alpha0 * a) : fmaxf(a, 0.f);
        if (save) s_h[t] = h[t];
    }
    __syncthreads();
    if (t < C) {
        float p = b1 ? b1[t] : 0.f;
        for (int j = 0; j < C8; ++j) p = fmaf(h[j], w1[j * C + t], p);
        const float gv = tg_gate_final(p, mode);
        if (save) { s_p[t] = p; s_g[t] = gv; }
        if (gout) gout[(int64_t)b * C + t] = gv;
    }
}

// out = x * g[b][c] [+ x2 * (1 - g[b][c])] [+ add_bc[b][c] * add_scale] [+ res]
__global__ __launch_bounds__(256) void tg_gate_mul_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ x2, const float* __restrict__ add_bc, float add_scale,
                                                          const float* __restrict__ res, float* __restrict__ out, int64_t hwC, int C,
                                                          int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hwC;
        const int c = (int)(i % C);
        const float gv = g[b * C + c];
        float v = x[i] * gv;
        if (x2) v = fmaf(x2[i], 1.f - gv, v);
        if (add_bc) v = fmaf(add_bc[b * C + c], add_scale, v);
        if (res) v += res[i];
        out[i] = v;
    }
}

// selector_block with a per-pixel selector map u >= 0 (custom_layers_selector.py:316-330): s = F(2.5 - u), out = x1 s + x2 (1 - s)
__global__ __launch_bounds__(256) void tg_selector_mix_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                              const float* __restrict__ u, float* __restrict__ out, int64_t n, int soft)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = 2.5f - u[i];
        const float sv = soft ? 1.f / (1.f + expf(-p)) : fminf(fmaxf(0.2f * p + 0.5f, 0.f), 1.f);
        out[i] = fmaf(x1[i], sv, x2[i] * (1.f - sv));
    }
}

// AveragePooling2D(pool, strides, padding="same"), any pool / stride: the divisor is the number of taps inside the image
__global__ __launch_bounds__(256) void tg_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                                                         int ph, int pw, int sh, int sw, int OH, int OW, int pt, int pl)
{
    const int64_t n = (int64_t)B * OH * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const int y0 = max(oy * sh - pt, 0), y1 = min(oy * sh - pt + ph, H);
        const int x0 = max(ox * sw - pl, 0), x1 = min(ox * sw - pl + pw, W);
        float a = 0.f;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) a += in[(((int64_t)b * H + y) * W + x) * C + c];
        out[i] = a / (float)((y1 - y0) * (x1 - x0));
    }
}

// single workgroup, samples in order (fixed summation order): dg[b][c] = sum_hw dy * x (from the partials) ->
// dp = dg * 0.2 [|p| < 2.5] ; dW1 += h^T dp ; dh = dp W1^T * [h > 0] ; dW0 += m^T dh ; dmean[b] = dh W0^T
__global__ __launch_bounds__(256) void tg_gate_dense_bwd_kernel(const double* __restrict__ partial, int nblk, int C, int C8, int B,
                                                                const float* __restrict__ w0, const float* __restrict__ w1,
                                                                const float* __restrict__ save, float* __restrict__ dw0,
                                                                float* __restrict__ dw1, float* __restrict__ dmean)
{
    __shared__ float dp[256], dh[64], m[256], h[64];
    const int t = threadIdx.x;
    // accumulators: thread t owns dw1[:, t] (t < C) and dw0[t, :] (t < C)
    float a1[32], a0[32];                      // C8 <= 32
#pragma unroll
    for (int j = 0; j < 32; ++j) { a1[j] = 0.f; a0[j] = 0.f; }
    for (int b = 0; b < B; ++b) {
        const float* s_mean = save + (int64_t)b * C;
        const float* s_h = save + (int64_t)B * C + (int64_t)b * C8;
        const float* s_p = save + (int64_t)B * (C + C8) + (int64_t)b * C;
        if (t < C) {
            double s = 0.0;
            for (int k = 0; k < nblk; ++k) s += partial[((int64_t)b * nblk + k) * 2 * C + C + t];
            const float p = s_p[t];
            dp[t] = (p > -2.5f && p < 2.5f) ? 0.2f * (float)s : 0.f;
            m[t] = s_mean[t];
        }
        if (t < C8) h[t] = s_h[t];
        __syncthreads();
        if (t < C8) {
            float a = 0.f;
            for (int c = 0; c < C; ++c) a = fmaf(dp[c], w1[t * C + c], a);
            dh[t] = h[t] > 0.f ? a : 0.f;
        }
        __syncthreads();
        if (t < C) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (j < C8) { a1[j] = fmaf(h[j], dp[t], a1[j]); a0[j] = fmaf(m[t], dh[j], a0[j]); }
            float d = 0.f;
            for (int j = 0; j < C8; ++j) d = fmaf(dh[j], w0[t * C8 + j], d);
            dmean[(int64_t)b * C + t] = d;
        }
        __syncthreads();
    }
    if (t < C) {
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < C8) { dw1[j * C + t] = a1[j]; dw0[t * C8 + j] = a0[j]; }
    }
}

// out[r][c * m + j] = x[r][c]
__global__ __launch_bounds__(256) void tg_repeat_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n_out, int C, int m)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (C * m);
        const int cm = (int)(i % (C * m));
        out[i] = x[r * C + cm / m];
    }
}

// out[r][c] = sum_j x[r][c * m + j]
__global__ __launch_bounds__(256) void tg_group_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n_out, int C, int m)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i % C);
        float s = 0.f;
        for (int j = 0; j < m; ++j) s += x[(r * C + c) * m + j];
        out[i] = s;
    }
}

// keras Conv2D(groups = g) kernel [cin / g][cout] <-> block-diagonal dense [cin][cout]
__global__ __launch_bounds__(256) void tg_group_expand_kernel(const float* __restrict__ w, float* __restrict__ dense, int cin, int cout, int g,
                                                              int extract)
{
    const int ci_g = cin / g, co_g = cout / g;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cin * cout; i += gridDim.x * 256) {
        const int ci = i / cout, co = i % cout;
        const bool on = ci / ci_g == co / co_g;
        if (extract) {
            if (on) const_cast<float*>(w)[(ci % ci_g) * cout + co] = dense[i];
        } else {
            dense[i] = on ? w[(ci % ci_g) * cout + co] : 0.f;
        }
    }
}

inline int tg_grid(int64_t n)
{
    const int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g < 4096 ? g : 4096));
}

inline bool tg_ok_c(int C) { return C > 0 && C <= 256 && 256 % C == 0; }

}  // namespace

extern "C" int64_t bf_op_bn_train_scratch_floats(int channels) { return (int64_t)CS_GRID * 2 * channels * 2 + 3 * channels + 64; }

extern "C" int bf_op_bn_train_fwd(const float* x, const float* gamma, float* y, float* save, float* moving_mean, float* moving_var,
                                  int64_t npix, int C, float eps, float momentum, int act, float alpha, float* scratch,
                                  int64_t scratch_floats, void* stream)
{
    if (!x || !gamma || !y || !save || !scratch || npix <= 0) return BF_EINVAL;
    if (!tg_ok_c(C)) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_bn_train_scratch_floats(C) || (uintptr_t)scratch % 8) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(scratch);
    float* coef = scratch + (int64_t)CS_GRID * 2 * C * 2;
    hipLaunchKernelGGL(tg_colsum2_kernel, dim3(CS_GRID, 1), dim3(256), 0, s, x, x, npix, C, partial);
    hipLaunchKernelGGL(tg_bn_fwd_finalize_kernel, dim3(1), dim3(256), 0, s, partial, CS_GRID, (double)npix, C, gamma, eps, momentum,
                       moving_mean, moving_var, save, coef);
    hipLaunchKernelGGL(tg_lincomb_kernel, dim3(tg_grid(npix * C)), dim3(256), 0, s, x, (const float*)nullptr, coef, (const float*)nullptr,
                       coef + C, y, npix * C, C, act, alpha);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_bn_train_bwd(const float* x, const float* gamma, const float* save, const float* dy, float* dx, float* dgamma,
                                  int64_t npix, int C, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!x || !gamma || !save || !dy || !dx || !dgamma || !scratch || npix <= 0) return BF_EINVAL;
    if (!tg_ok_c(C)) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_bn_train_scratch_floats(C) || (uintptr_t)scratch % 8) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(scratch);
    float* coef = scratch + (int64_t)CS_GRID * 2 * C * 2;
    hipLaunchKernelGGL(tg_colsum2_kernel, dim3(CS_GRID, 1), dim3(256), 0, s, dy, x, npix, C, partial);
    hipLaunchKernelGGL(tg_bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, s, partial, CS_GRID, (double)npix, C, gamma, save, coef, dgamma);
    hipLaunchKernelGGL(tg_lincomb_kernel, dim3(tg_grid(npix * C)), dim3(256), 0, s, dy, x, coef, coef + C, coef + 2 * C, dx, npix * C, C, 0,
                       0.f);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int64_t bf_op_gate_save_floats(int batch, int channels, int squeeze) { return (int64_t)batch * (3 * channels + squeeze); }
extern "C" int64_t bf_op_channel_gate_save_floats(int batch, int sel_channels, int channels, int squeeze)
{
    return (int64_t)batch * (sel_channels + squeeze + 2 * channels);
}
extern "C" int64_t bf_op_gate_scratch_floats(int batch, int channels)
{
    return (int64_t)batch * CS_GRID * 2 * channels * 2 + (int64_t)batch * channels + 64;
}

extern "C" int bf_op_channel_gate_ex(const float* sel, const float* x, const float* x2, const float* res, float* out, const float* w0,
                                     const float* b0, const float* w1, const float* b1, float* save, int B, int64_t hw, int Cs, int C,
                                     int C8, int act0, float alpha0, int mode, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!sel || !x || !w0 || !w1 || !out || !save || !scratch || B <= 0 || hw <= 0) return BF_EINVAL;
    if (!tg_ok_c(Cs) || C <= 0 || C > 256 || C8 <= 0 || C8 > 64 || (act0 != 1 && act0 != 2) || mode < 0 || mode > 3) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_gate_scratch_floats(B, Cs) || (uintptr_t)scratch % 8) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(tg_colsum2_kernel, dim3(CS_GRID, B), dim3(256), 0, s, sel, sel, hw, Cs, partial);
    hipLaunchKernelGGL(tg_gate_dense_kernel, dim3(B), dim3(256), 0, s, partial, CS_GRID, (double)hw, Cs, C, C8, w0, b0, w1, b1, act0, alpha0,
                       mode, B, save, (const float*)nullptr, (float*)nullptr);
    const float* g = save + (int64_t)B * (Cs + C8 + C);
    hipLaunchKernelGGL(tg_gate_mul_kernel, dim3(tg_grid((int64_t)B * hw * C)), dim3(256), 0, s, x, g, x2, (const float*)nullptr, 0.f, res, out,
                       hw * C, C, (int64_t)B * hw * C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_gate_fwd(const float* x, const float* w0, const float* w1, const float* res, float* out, float* save, int B,
                              int64_t hw, int C, int C8, float* scratch, int64_t scratch_floats, void* stream)
{
    if (C8 > 32) return BF_EUNSUPPORTED;                  // (the backward keeps C8 accumulators per thread)
    return bf_op_channel_gate_ex(x, x, nullptr, res, out, w0, nullptr, w1, nullptr, save, B, hw, C, C, C8, 1, 0.f, 0, scratch, scratch_floats,
                                 stream);
}

// out[r] = F(act0(in[r] W0 + b0) W1 + b1) for every row r of in [n][Cs]: two small dense layers (the selector's LOCAL 1x1
// convolutions on the pooled map; F as in bf_op_channel_gate_ex, plus mode 4 = relu)
extern "C" int bf_op_dense2(const float* in, const float* w0, const float* b0, const float* w1, const float* b1, float* out, int64_t n,
                            int Cs, int C, int C8, int act0, float alpha0, int mode, void* stream)
{
    if (!in || !w0 || !w1 || !out || n <= 0 || n > 0x7fffffff) return BF_EINVAL;
    if (Cs <= 0 || Cs > 256 || C <= 0 || C > 256 || C8 <= 0 || C8 > 64 || (act0 != 1 && act0 != 2) || mode < 0 || mode > 4) return BF_EUNSUPPORTED;
    hipLaunchKernelGGL(tg_gate_dense_kernel, dim3((int)n), dim3(256), 0, (hipStream_t)stream, (const double*)nullptr, 0, 1.0, Cs, C, C8, w0, b0,
                       w1, b1, act0, alpha0, mode, (int)n, (float*)nullptr, in, out);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_selector_mix(const float* x1, const float* x2, const float* u, float* out, int64_t n, int soft, void* stream)
{
    if (!x1 || !x2 || !u || !out || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_selector_mix_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x1, x2, u, out, n, soft);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_avgpool_same(const float* in, float* out, int B, int H, int W, int C, int pool_h, int pool_w, int stride_h,
                                  int stride_w, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || pool_h <= 0 || pool_w <= 0 || stride_h <= 0 || stride_w <= 0) return BF_EINVAL;
    const int OH = (H + stride_h - 1) / stride_h, OW = (W + stride_w - 1) / stride_w;
    int th = (OH - 1) * stride_h + pool_h - H, tw = (OW - 1) * stride_w + pool_w - W;
    if (th < 0) th = 0;
    if (tw < 0) tw = 0;
    hipLaunchKernelGGL(tg_avgpool_kernel, dim3(tg_grid((int64_t)B * OH * OW * C)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C,
                       pool_h, pool_w, stride_h, stride_w, OH, OW, th / 2, tw / 2);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// dy: gradient at the gate's output (x * g); dx: gradient at x through both uses (the multiply and the mean)
extern "C" int bf_op_gate_bwd(const float* x, const float* w0, const float* w1, const float* save, const float* dy, float* dx, float* dw0,
                              float* dw1, int B, int64_t hw, int C, int C8, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!x || !w0 || !w1 || !save || !dy || !dx || !dw0 || !dw1 || !scratch || B <= 0 || hw <= 0) return BF_EINVAL;
    if (!tg_ok_c(C) || C8 <= 0 || C8 > 32) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_gate_scratch_floats(B, C) || (uintptr_t)scratch % 8) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(scratch);
    float* dmean = scratch + (int64_t)B * CS_GRID * 2 * C * 2;
    hipLaunchKernelGGL(tg_colsum2_kernel, dim3(CS_GRID, B), dim3(256), 0, s, dy, x, hw, C, partial);
    hipLaunchKernelGGL(tg_gate_dense_bwd_kernel, dim3(1), dim3(256), 0, s, partial, CS_GRID, C, C8, B, w0, w1, save, dw0, dw1, dmean);
    const float* g = save + (int64_t)B * (2 * C + C8);
    hipLaunchKernelGGL(tg_gate_mul_kernel, dim3(tg_grid((int64_t)B * hw * C)), dim3(256), 0, s, dy, g, (const float*)nullptr, dmean,
                       (float)(1.0 / (double)hw), (const float*)nullptr, dx, hw * C, C, (int64_t)B * hw * C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_channel_repeat(const float* x, float* out, int64_t npix, int C, int m, void* stream)
{
    if (!x || !out || npix <= 0 || C <= 0 || m <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_repeat_kernel, dim3(tg_grid(npix * C * m)), dim3(256), 0, (hipStream_t)stream, x, out, npix * C * m, C, m);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_channel_group_sum(const float* x, float* out, int64_t npix, int C, int m, void* stream)
{
    if (!x || !out || npix <= 0 || C <= 0 || m <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_group_sum_kernel, dim3(tg_grid(npix * C)), dim3(256), 0, (hipStream_t)stream, x, out, npix * C, C, m);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// extract = 0: dense[cin][cout] <- block-diagonal expansion of w[cin / groups][cout]; extract = 1: w <- the diagonal blocks of dense
extern "C" int bf_op_group_kernel(float* w, float* dense, int cin, int cout, int groups, int extract, void* stream)
{
    if (!w || !dense || cin <= 0 || cout <= 0 || groups <= 0 || cin % groups || cout % groups) return BF_EINVAL;
    hipLaunchKernelGGL(tg_group_expand_kernel, dim3(tg_grid((int64_t)cin * cout)), dim3(256), 0, (hipStream_t)stream, w, dense, cin, cout,
                       groups, extract);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// build_normalize_model / build_denormalize_model as standalone layers (bfcnn/model.py:364-430, utilities.py:435-461):
// inverse = 0: out = (clip(x, lo, hi) - lo) / (hi - lo) - 0.5 ; inverse = 1: out = (clip(x, -0.5, 0.5) + 0.5) * (hi - lo) + lo
__global__ __launch_bounds__(256) void tg_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float lo,
                                                           float hi, int inverse)
{
    const float range = hi - lo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        out[i] = inverse ? (fminf(fmaxf(v, -0.5f), 0.5f) + 0.5f) * range + lo : (fminf(fmaxf(v, lo), hi) - lo) / range - 0.5f;
    }
}

extern "C" int bf_op_normalize(const float* x, float* out, int64_t n, float v_min, float v_max, int inverse, void* stream)
{
    if (!x || !out || n <= 0 || !(v_max > v_min)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_normalize_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, v_min, v_max, inverse);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// AdditiveAttentionGate's last stage (bfcnn/custom_layers.py:826-832) + the Add behind it: out = enc * sigmoid(4 o) + up, and
// its backward: denc = dy * s, do = dy * enc * 4 s (1 - s)  (the gradient with respect to `up` through the Add is dy itself)
__global__ __launch_bounds__(256) void tg_sigmoid_gate_kernel(const float* __restrict__ enc, const float* __restrict__ o,
                                                              const float* __restrict__ up, float* __restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sg = 1.f / (1.f + expf(-4.f * o[i]));
        out[i] = fmaf(enc[i], sg, up ? up[i] : 0.f);
    }
}

__global__ __launch_bounds__(256) void tg_sigmoid_gate_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ o,
                                                                  const float* __restrict__ dy, float* __restrict__ denc,
                                                                  float* __restrict__ dout_o, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sg = 1.f / (1.f + expf(-4.f * o[i]));
        const float d = dy[i];
        denc[i] = d * sg;
        dout_o[i] = d * enc[i] * 4.f * sg * (1.f - sg);
    }
}

extern "C" int bf_op_sigmoid_gate(const float* enc, const float* o, const float* up, float* out, int64_t n, void* stream)
{
    if (!enc || !o || !out || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_sigmoid_gate_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, enc, o, up, out, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_sigmoid_gate_bwd(const float* enc, const float* o, const float* dy, float* denc, float* dout_o, int64_t n, void* stream)
{
    if (!enc || !o || !dy || !denc || !dout_o || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_sigmoid_gate_bwd_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, enc, o, dy, denc, dout_o, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// out[r][0:Ca | Ca:Ca+Cb | ...] = a[r] | b[r] | c[r]  (keras Concatenate on the channel axis; b, c may be NULL with Cb = Cc = 0)
__global__ __launch_bounds__(256) void tg_concat_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                        float* __restrict__ out, int64_t rows, int Ca, int Cb, int Cc)
{
    const int Ct = Ca + Cb + Cc;
    const int64_t n = rows * Ct;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Ct;
        const int k = (int)(i % Ct);
        out[i] = k < Ca ? a[r * Ca + k] : (k < Ca + Cb ? b[r * Cb + k - Ca] : c[r * Cc + k - Ca - Cb]);
    }
}

// mean[b][c] = mean over the hw positions of x[b][.][c] (tf.reduce_mean(x, axis=[1, 2])), broadcast to out[b][rows_out][c]
__global__ __launch_bounds__(256) void tg_mean_bcast_kernel(const double* __restrict__ partial, int nblk, double hw, int C, float* __restrict__ out,
                                                            int64_t rows_out)
{
    __shared__ float m[256];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < C) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += partial[((int64_t)b * nblk + k) * 2 * C + t];
        m[t] = (float)(s / hw);
    }
    __syncthreads();
    for (int64_t i = t; i < rows_out * C; i += 256) out[(int64_t)b * rows_out * C + i] = m[i % C];
}

extern "C" int bf_op_concat_channels(const float* a, const float* b, const float* c, float* out, int64_t rows, int Ca, int Cb, int Cc, void* stream)
{
    if (!a || !out || rows <= 0 || Ca <= 0 || Cb < 0 || Cc < 0 || (Cb > 0 && !b) || (Cc > 0 && !c)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_concat_kernel, dim3(tg_grid(rows * (Ca + Cb + Cc))), dim3(256), 0, (hipStream_t)stream, a, b, c, out, rows, Ca, Cb, Cc);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_channel_mean_broadcast(const float* x, float* out, int B, int64_t hw, int C, int64_t rows_out, float* scratch,
                                            int64_t scratch_floats, void* stream)
{
    if (!x || !out || !scratch || B <= 0 || hw <= 0 || rows_out <= 0) return BF_EINVAL;
    if (!tg_ok_c(C)) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_gate_scratch_floats(B, C) || (uintptr_t)scratch % 8) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* partial = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL(tg_colsum2_kernel, dim3(CS_GRID, B), dim3(256), 0, s, x, x, hw, C, partial);
    hipLaunchKernelGGL(tg_mean_bcast_kernel, dim3(B), dim3(256), 0, s, partial, CS_GRID, (double)hw, C, out, rows_out);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}


// ------------------------------------------------------------------------------------------
// ChannelwiseMultiplier / Multiplier (bfcnn/custom_layers.py:1028-1160): x * relu(w0 + w1), w0 trainable ([C] or [1]), w1 a
// constant.  The factor as a [C] vector (what bf_op_scale_add multiplies by), and the gradient of w0 from the gradient of that
// vector (bf_op_scale_add_bwd's dm): dw0[c] = dm[c] [w0[c] + w1 > 0], summed over the channels for the scalar form.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_relu_shift_kernel(const float* __restrict__ w0, int nw, float w1, float* __restrict__ m, int C)
{
    for (int c = threadIdx.x; c < C; c += 256) m[c] = fmaxf(w0[nw == 1 ? 0 : c] + w1, 0.f);
}

__global__ __launch_bounds__(256) void tg_relu_shift_bwd_kernel(const float* __restrict__ w0, int nw, float w1, const float* __restrict__ dm,
                                                                float* __restrict__ dw0, int C)
{
    if (nw != 1) {
        for (int c = threadIdx.x; c < C; c += 256) dw0[c] = w0[c] + w1 > 0.f ? dm[c] : 0.f;
        return;
    }
    __shared__ double red[256];
    double s = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) s += (double)dm[c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) dw0[0] = w0[0] + w1 > 0.f ? (float)red[0] : 0.f;
}

// the same factor with the layers' default activation "linear": m[c] = w0[c or 0] + w1
__global__ __launch_bounds__(256) void tg_shift_kernel(const float* __restrict__ w0, int nw, float w1, float* __restrict__ m, int C)
{
    for (int c = threadIdx.x; c < C; c += 256) m[c] = w0[nw == 1 ? 0 : c] + w1;
}

extern "C" int bf_op_linear_shift(const float* w0, int nw, float w1, float* m, int C, void* stream)
{
    if (!w0 || !m || C <= 0 || (nw != 1 && nw != C)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_shift_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w0, nw, w1, m, C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_relu_shift(const float* w0, int nw, float w1, float* m, int C, void* stream)
{
    if (!w0 || !m || C <= 0 || (nw != 1 && nw != C)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_relu_shift_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w0, nw, w1, m, C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_relu_shift_bwd(const float* w0, int nw, float w1, const float* dm, float* dw0, int C, void* stream)
{
    if (!w0 || !dm || !dw0 || C <= 0 || (nw != 1 && nw != C)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_relu_shift_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w0, nw, w1, dm, dw0, C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// selector_block in training (bfcnn/custom_layers_selector.py:81-330): the adjoints of bf_op_selector_mix, bf_op_avgpool_same and
// bf_op_dense2 (bias-free, act0 = leaky ReLU, final ReLU: the selector's two layers), and the channel slice that undoes
// bf_op_concat_channels.  The up-sampling's adjoint is bf_op_resize_bilinear_bwd; scale_type GLOBAL is the same chain with one
// pooling window over the whole image.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_selector_mix_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                                  const float* __restrict__ u, const float* __restrict__ dy,
                                                                  float* __restrict__ dx1, float* __restrict__ dx2, float* __restrict__ du,
                                                                  int64_t n, int soft)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = 2.5f - u[i];
        float sv, dsdu;
        if (soft) {
            sv = 1.f / (1.f + expf(-p));
            dsdu = -sv * (1.f - sv);
        } else {
            const float lin = 0.2f * p + 0.5f;
            sv = fminf(fmaxf(lin, 0.f), 1.f);
            dsdu = (lin >= 0.f && lin <= 1.f) ? -0.2f : 0.f;           // clip_by_value passes the gradient on its closed interval
        }
        const float g = dy[i];
        dx1[i] = g * sv;
        dx2[i] = g * (1.f - sv);
        du[i] = g * (x1[i] - x2[i]) * dsdu;
    }
}

// gather form: an input element collects dp / (taps of the window inside the image) from every window that covers it
__global__ __launch_bounds__(256) void tg_avgpool_bwd_kernel(const float* __restrict__ dp, float* __restrict__ dx, int B, int H, int W, int C,
                                                             int ph, int pw, int sh, int sw, int OH, int OW, int pt, int pl, int accumulate)
{
    const int64_t n = (int64_t)B * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int oy0 = y + pt - ph + 1, ox0 = x + pl - pw + 1;
        oy0 = oy0 <= 0 ? 0 : (oy0 + sh - 1) / sh;
        ox0 = ox0 <= 0 ? 0 : (ox0 + sw - 1) / sw;
        const int oy1 = min((y + pt) / sh, OH - 1), ox1 = min((x + pl) / sw, OW - 1);
        float a = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const int ny = min(oy * sh - pt + ph, H) - max(oy * sh - pt, 0);
            for (int ox = ox0; ox <= ox1; ++ox) {
                const int nx = min(ox * sw - pl + pw, W) - max(ox * sw - pl, 0);
                a += dp[(((int64_t)b * OH + oy) * OW + ox) * C + c] / (float)(ny * nx);
            }
        }
        dx[i] = accumulate ? dx[i] + a : a;
    }
}

// one workgroup per row r: h_pre = p W0, h = leaky(h_pre), u_pre = h W1; g = du [u_pre > 0]; dh = (g W1^T) leaky'(h_pre); dp = dh W0^T.
// g, dh and h are kept for the weight gradients (tg_dense2_wgrad_kernel)
__global__ __launch_bounds__(256) void tg_dense2_bwd_kernel(const float* __restrict__ p, const float* __restrict__ w0, const float* __restrict__ w1,
                                                            const float* __restrict__ du, int Cs, int C, int C8, float alpha0,
                                                            float* __restrict__ dp, float* __restrict__ G, float* __restrict__ DH,
                                                            float* __restrict__ Hs)
{
    __shared__ float ps[256], hp[64], hh[64], gs[256], dh[64];
    const int r = blockIdx.x, t = threadIdx.x;
    if (t < Cs) ps[t] = p[(int64_t)r * Cs + t];
    __syncthreads();
    if (t < C8) {
        float a = 0.f;
        for (int i = 0; i < Cs; ++i) a = fmaf(ps[i], w0[i * C8 + t], a);
        hp[t] = a;
        hh[t] = a > 0.f ? a : alpha0 * a;
        Hs[(int64_t)r * C8 + t] = hh[t];
    }
    __syncthreads();
    if (t < C) {
        float a = 0.f;
        for (int j = 0; j < C8; ++j) a = fmaf(hh[j], w1[j * C + t], a);
        const float g = a > 0.f ? du[(int64_t)r * C + t] : 0.f;
        gs[t] = g;
        G[(int64_t)r * C + t] = g;
    }
    __syncthreads();
    if (t < C8) {
        float a = 0.f;
        for (int k = 0; k < C; ++k) a = fmaf(gs[k], w1[t * C + k], a);
        a *= hp[t] > 0.f ? 1.f : alpha0;
        dh[t] = a;
        DH[(int64_t)r * C8 + t] = a;
    }
    __syncthreads();
    if (t < Cs) {
        float a = 0.f;
        for (int j = 0; j < C8; ++j) a = fmaf(dh[j], w0[t * C8 + j], a);
        dp[(int64_t)r * Cs + t] = a;
    }
}

// dW0[i][j] = sum_r p[r][i] DH[r][j] ; dW1[j][k] = sum_r Hs[r][j] G[r][k]: one thread per element, rows in order (fixed summation order)
__global__ __launch_bounds__(256) void tg_dense2_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ G, const float* __restrict__ DH,
                                                              const float* __restrict__ Hs, int64_t n, int Cs, int C, int C8,
                                                              float* __restrict__ dw0, float* __restrict__ dw1)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < Cs * C8) {
        const int i = e / C8, j = e % C8;
        double a = 0.0;
        for (int64_t r = 0; r < n; ++r) a += (double)p[r * Cs + i] * (double)DH[r * C8 + j];
        dw0[e] = (float)a;
    } else if (e < Cs * C8 + C8 * C) {
        const int f = e - Cs * C8, j = f / C, k = f % C;
        double a = 0.0;
        for (int64_t r = 0; r < n; ++r) a += (double)Hs[r * C8 + j] * (double)G[r * C + k];
        dw1[f] = (float)a;
    }
}

__global__ __launch_bounds__(256) void tg_slice_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int Csrc,
                                                                int off, int C)
{
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dst[i] = src[(i / C) * Csrc + off + i % C];
}

extern "C" int bf_op_selector_mix_bwd(const float* x1, const float* x2, const float* u, const float* dy, float* dx1, float* dx2, float* du,
                                      int64_t n, int soft, void* stream)
{
    if (!x1 || !x2 || !u || !dy || !dx1 || !dx2 || !du || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_selector_mix_bwd_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x1, x2, u, dy, dx1, dx2, du, n, soft);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_avgpool_same_bwd(const float* dp, float* dx, int B, int H, int W, int C, int pool_h, int pool_w, int stride_h,
                                      int stride_w, int accumulate, void* stream)
{
    if (!dp || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || pool_h <= 0 || pool_w <= 0 || stride_h <= 0 || stride_w <= 0) return BF_EINVAL;
    const int OH = (H + stride_h - 1) / stride_h, OW = (W + stride_w - 1) / stride_w;
    int th = (OH - 1) * stride_h + pool_h - H, tw = (OW - 1) * stride_w + pool_w - W;
    if (th < 0) th = 0;
    if (tw < 0) tw = 0;
    hipLaunchKernelGGL(tg_avgpool_bwd_kernel, dim3(tg_grid((int64_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dp, dx, B, H, W, C,
                       pool_h, pool_w, stride_h, stride_w, OH, OW, th / 2, tw / 2, accumulate);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int64_t bf_op_dense2_bwd_scratch_floats(int64_t n, int C, int C8) { return n > 0 ? n * (C + 2 * (int64_t)C8) : -1; }

extern "C" int bf_op_dense2_bwd(const float* in, const float* w0, const float* w1, const float* dout, float* din, float* dw0, float* dw1,
                                int64_t n, int Cs, int C, int C8, float alpha0, float* scratch, int64_t scratch_floats, void* stream)
{
    if (!in || !w0 || !w1 || !dout || !din || !dw0 || !dw1 || !scratch || n <= 0 || n > 0x7fffffff) return BF_EINVAL;
    if (Cs <= 0 || Cs > 256 || C <= 0 || C > 256 || C8 <= 0 || C8 > 64) return BF_EUNSUPPORTED;
    if (scratch_floats < bf_op_dense2_bwd_scratch_floats(n, C, C8)) return BF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* G = scratch;
    float* DH = G + n * C;
    float* Hs = DH + n * C8;
    hipLaunchKernelGGL(tg_dense2_bwd_kernel, dim3((int)n), dim3(256), 0, s, in, w0, w1, dout, Cs, C, C8, alpha0, din, G, DH, Hs);
    hipLaunchKernelGGL(tg_dense2_wgrad_kernel, dim3((Cs * C8 + C8 * C + 255) / 256), dim3(256), 0, s, in, G, DH, Hs, n, Cs, C, C8, dw0, dw1);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_slice_channels(const float* src, float* dst, int64_t rows, int src_channels, int offset, int channels, void* stream)
{
    if (!src || !dst || rows <= 0 || channels <= 0 || offset < 0 || offset + channels > src_channels) return BF_EINVAL;
    hipLaunchKernelGGL(tg_slice_channels_kernel, dim3(tg_grid(rows * channels)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, src_channels,
                       offset, channels);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// add_concat_input (bfcnn/backbone_resnet.py:277-279): Concatenate([backbone features, backbone input]) in front of the head.  The
// backbone input is the NORMALISED image (model.py:100-102), zero-padded as pad_to_power_of_2 pads the raw image (value 0 -> its
// normalised value).  out [B,H,W,Cout] = feat [C] | normalise(x) [cin] | zeros up to Cout (the channel count the head's matrix
// kernel takes; the head's first kernel carries zero rows there).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_concat_input_kernel(const float* __restrict__ feat, const void* __restrict__ x, int x_is_u8,
                                                              float* __restrict__ out, int B, int H, int W, int Hs, int Ws, int C, int cin,
                                                              int Cout, float v_min, float v_max)
{
    const int64_t n = (int64_t)B * H * W * Cout;
    const float range = v_max - v_min;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cout);
        const int64_t px = i / Cout;
        float v = 0.f;
        if (c < C) {
            v = feat[px * C + c];
        } else if (c < C + cin) {
            const int xx = (int)(px % W);
            const int yy = (int)((px / W) % H);
            const int64_t b = px / ((int64_t)W * H);
            float raw = 0.f;
            if (yy < Hs && xx < Ws) {
                const int64_t si = ((b * Hs + yy) * Ws + xx) * cin + (c - C);
                raw = x_is_u8 ? (float)reinterpret_cast<const uint8_t*>(x)[si] : reinterpret_cast<const float*>(x)[si];
            }
            v = (fminf(fmaxf(raw, v_min), v_max) - v_min) / range - 0.5f;
        }
        out[i] = v;
    }
}

extern "C" int bf_op_concat_input(const float* feat, const void* x, int x_is_u8, float* out, int B, int H, int W, int Hs, int Ws, int C,
                                  int cin, int Cout, float v_min, float v_max, void* stream)
{
    if (!feat || !x || !out || B <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || Hs > H || Ws > W || C <= 0 || cin <= 0 || Cout < C + cin)
        return BF_EINVAL;
    if (!(v_max > v_min)) return BF_EINVAL;
    hipLaunchKernelGGL(tg_concat_input_kernel, dim3(tg_grid((int64_t)B * H * W * Cout)), dim3(256), 0, (hipStream_t)stream, feat, x, x_is_u8, out,
                       B, H, W, Hs, Ws, C, cin, Cout, v_min, v_max);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// selector_block's optional pre-filters on the selector layer (bfcnn/custom_layers_selector.py:160-185; utilities.py:566-620):
//   local_normalization: (x - m) / sqrt(pool((x - m)^2) + eps), m = AveragePooling2D(pool, strides 1, same)(x): the two poolings are
//     bf_op_avgpool_same, the two element-wise stages bf_op_center_scale (var NULL: (x - m)^2; else (x - m) / sqrt(var + eps));
//   global_normalization is bf_op_bn_train_fwd per sample with gamma = 1 (same formula, biased variance);
//   lowpass / highpass: x (1 - tanh(a x)^b) / x tanh(a x)^b, b a small positive integer (the reference passes 4.0).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_center_scale_kernel(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ var,
                                                              float* __restrict__ out, int64_t n, float eps)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = x[i] - m[i];
        out[i] = var ? d / sqrtf(var[i] + eps) : d * d;
    }
}

__global__ __launch_bounds__(256) void tg_pass_filter_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float a, int b, int highpass)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i], t = tanhf(a * v);
        float f = 1.f;
        for (int k = 0; k < b; ++k) f *= t;
        out[i] = (highpass ? f : 1.f - f) * v;
    }
}

extern "C" int bf_op_center_scale(const float* x, const float* mean, const float* var, float* out, int64_t n, float eps, void* stream)
{
    if (!x || !mean || !out || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_center_scale_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, mean, var, out, n, eps);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_pass_filter(const float* x, float* out, int64_t n, float a, int b, int highpass, void* stream)
{
    if (!x || !out || n <= 0 || b < 1 || b > 16) return BF_EINVAL;
    hipLaunchKernelGGL(tg_pass_filter_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, a, b, highpass);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// adjoints of the selector's pre-filters (training): bf_op_pass_filter_bwd; local_normalization y = d r, d = x - m, r = (v + eps)^-1/2,
// m = pool(x), v = pool(d^2): bf_op_center_scale_bwd gives dd = dy r and dv = -0.5 dy d r^3; with t = pool^T(dv) (bf_op_avgpool_same_bwd)
// bf_op_center_sq_bwd forms dd + 2 d t, and dx = that - pool^T(that).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tg_pass_filter_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                                 int64_t n, float a, int b, int highpass)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i], t = tanhf(a * v);
        float fm1 = 1.f;                                   // t^(b-1)
        for (int k = 0; k < b - 1; ++k) fm1 *= t;
        const float f = fm1 * t, df = (float)b * fm1 * a * (1.f - t * t);
        dx[i] = dy[i] * (highpass ? f + v * df : (1.f - f) - v * df);
    }
}

__global__ __launch_bounds__(256) void tg_center_scale_bwd_kernel(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ var,
                                                                  const float* __restrict__ dy, float* __restrict__ dd, float* __restrict__ dv,
                                                                  int64_t n, float eps)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = x[i] - m[i], r = 1.f / sqrtf(var[i] + eps), g = dy[i];
        dd[i] = g * r;
        dv[i] = -0.5f * g * d * r * r * r;
    }
}

__global__ __launch_bounds__(256) void tg_center_sq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ t,
                                                               const float* __restrict__ dd, float* __restrict__ out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = fmaf(2.f * (x[i] - m[i]), t[i], dd[i]);
}

extern "C" int bf_op_pass_filter_bwd(const float* x, const float* dy, float* dx, int64_t n, float a, int b, int highpass, void* stream)
{
    if (!x || !dy || !dx || n <= 0 || b < 1 || b > 16) return BF_EINVAL;
    hipLaunchKernelGGL(tg_pass_filter_bwd_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, a, b, highpass);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_center_scale_bwd(const float* x, const float* mean, const float* var, const float* dy, float* dd, float* dv, int64_t n,
                                      float eps, void* stream)
{
    if (!x || !mean || !var || !dy || !dd || !dv || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_center_scale_bwd_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, mean, var, dy, dd, dv, n, eps);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_center_sq_bwd(const float* x, const float* mean, const float* t, const float* dd, float* out, int64_t n, void* stream)
{
    if (!x || !mean || !t || !dd || !out || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(tg_center_sq_bwd_kernel, dim3(tg_grid(n)), dim3(256), 0, (hipStream_t)stream, x, mean, t, dd, out, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
