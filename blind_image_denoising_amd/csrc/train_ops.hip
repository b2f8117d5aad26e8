// Small reductions / elementwise kernels around the convolutions: batch-norm statistics,
// batch-norm backward, deterministic partial reductions, regularisers, Adam.
// All reductions use a fixed summation order (no float atomics) so a training step is bitwise
// reproducible run to run.
#include "bf_common.h"

// ------------------------------------------------------------------------------------------
// out[i] = scale * sum_{r<nblk} partial[r*width + i]
// ------------------------------------------------------------------------------------------
// 64 columns x 16 row stripes per workgroup (one thread per column walking all rows alone needed 40 us for the 512 x 2304
// partials of a weight gradient: 9 workgroups, 512 dependent-ish loads each).  Fixed summation order: stripe s adds rows
// s, s+16, ... ; the 16 stripe sums are then added in order.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, int nblk, int width,
                                                               float* __restrict__ out, float scale)
{
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    if (i < width) {
        int r = stripe;
        for (; r + 16 < nblk; r += 32) {
            s0 += partial[(size_t)r * width + i];
            s1 += partial[(size_t)(r + 16) * width + i];
        }
        if (r < nblk) s0 += partial[(size_t)r * width + i];
    }
    red[stripe][col] = s0 + s1;
    __syncthreads();
    if (stripe == 0 && i < width) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][col];
        out[i] = a * scale;
    }
}

hipError_t bf_launch_reduce_partials(const float* partial, int nblk, int width, float* out, float scale, hipStream_t s)
{
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((width + 63) / 64), dim3(1024), 0, s, partial, nblk, width, out, scale);
    return hipGetLastError();
}

// the weight-gradient partials of ALL convolutions of the blocks in one launch (blockIdx.y = block * nconv + j): slot k holds
// [nblk][2304]; its sum goes to out + block * p_stride + (j == 0 ? 0 : 2304 + (j - 1) * unit).  Same summation order as above.
__global__ __launch_bounds__(1024) void reduce_wgrad_slots_kernel(const float* __restrict__ slots, int64_t slot_floats, int nblk,
                                                                  float* __restrict__ out, int64_t p_stride, int nconv, int unit)
{
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    const int blk = blockIdx.y / nconv, j = blockIdx.y % nconv;
    const float* partial = slots + (int64_t)blockIdx.y * slot_floats;
    float s0 = 0.f, s1 = 0.f;
    int r = stripe;
    for (; r + 16 < nblk; r += 32) {
        s0 += partial[(size_t)r * 2304 + i];
        s1 += partial[(size_t)(r + 16) * 2304 + i];
    }
    if (r < nblk) s0 += partial[(size_t)r * 2304 + i];
    red[stripe][col] = s0 + s1;
    __syncthreads();
    if (stripe == 0) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][col];
        out[(int64_t)blk * p_stride + (j == 0 ? 0 : 2304 + (int64_t)(j - 1) * unit) + i] = a;
    }
}

hipError_t bf_launch_reduce_wgrad_slots(const float* slots, int64_t slot_floats, int nblk, float* out, int64_t p_stride, int layers,
                                        int nconv, int unit, hipStream_t s)
{
    hipLaunchKernelGGL(reduce_wgrad_slots_kernel, dim3(2304 / 64, layers * nconv), dim3(1024), 0, s, slots, slot_floats, nblk, out,
                       p_stride, nconv, unit);
    return hipGetLastError();
}

__global__ void zero_kernel(float* p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

hipError_t bf_launch_zero(float* p, int64_t n, hipStream_t s)
{
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(zero_kernel, dim3((int)(g < 1024 ? g : 1024)), dim3(256), 0, s, p, n);
    return hipGetLastError();
}

// y += a * x (gradient accumulation over gpu_batches_per_step micro-batches, bfcnn/train_loop.py:296-310); a = 0 with
// overwrite: y = x
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, int overwrite, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = overwrite ? x[i] : fmaf(a, x[i], y[i]);
}

extern "C" int bf_op_axpy(float* y, const float* x, float a, int overwrite, int64_t n, void* stream)
{
    if (!y || !x || n <= 0) return BF_EINVAL;
    int64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(axpy_kernel, dim3((int)(g < 1024 ? g : 1024)), dim3(256), 0, (hipStream_t)stream, y, x, a, overwrite, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// training-mode BatchNormalization(scale=True, center=False, momentum, epsilon)
// (bfcnn/backbone_resnet.py:129-135; keras fused semantics):
//   mean, biased var over (N,H,W); y = gamma*(x-mean)*rsqrt(var+eps)
//   moving_mean = moving_mean*m + mean*(1-m); moving_var = moving_var*m + var*n/(n-1)*(1-m)
// partial = [nblk][32] (sum[16], sumsq[16]) from the conv epilogue.  Outputs the folded
// scale/shift for the apply pass and mean/inv for the backward pass.
// ------------------------------------------------------------------------------------------
// stage 1 of the [nblk][32] reductions: 64 workgroups, each sums its share of the rows in fp64 (fixed order) -> [64][32]
constexpr int BN_STAGE1 = 64;
__global__ __launch_bounds__(1024) void reduce_rows32_kernel(const float* __restrict__ partial, int nblk, double* __restrict__ out)
{
    __shared__ double red[32][32];
    const int ch = threadIdx.x & 31, stripe = threadIdx.x >> 5;
    double s = 0.0;
    for (int r = blockIdx.x * 32 + stripe; r < nblk; r += BN_STAGE1 * 32) s += (double)partial[(size_t)r * 32 + ch];
    red[stripe][ch] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        double a = 0.0;
        for (int k = 0; k < 32; ++k) a += red[k][threadIdx.x];
        out[blockIdx.x * 32 + threadIdx.x] = a;
    }
}

template <typename PT>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const PT* __restrict__ partial, int nblk, double count,
                                                          const float* __restrict__ gamma, float* moving_mean,
                                                          float* moving_var, float eps, float momentum,
                                                          float* scale, float* shift, float* mean_inv)
{
    // 32 stripes x 32 columns: one workgroup of 1024 threads (with 8 stripes this tiny kernel took 138 us per
    // BatchNorm at 4096 partial rows -- 10 % of a training step); fixed summation order, fp64
    __shared__ double red[32][32];
    const int ch = threadIdx.x & 31, stripe = threadIdx.x >> 5;
    double s = 0.0;
    for (int r = stripe; r < nblk; r += 32) s += (double)partial[(size_t)r * 32 + ch];
    red[stripe][ch] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 32; ++k) { s1 += red[k][c]; s2 += red[k][16 + c]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + (double)eps);
        const double g = gamma ? (double)gamma[c] : 1.0;
        scale[c] = (float)(g * inv);
        shift[c] = (float)(-g * inv * mean);
        mean_inv[c] = (float)mean;
        mean_inv[16 + c] = (float)inv;
        const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
        moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
        moving_var[c] = (float)((double)moving_var[c] * momentum + unbiased * (1.0 - (double)momentum));
    }
}

hipError_t bf_launch_bn_finalize(const float* partial, int nblk, double count, const float* gamma, float* moving_mean,
                                 float* moving_var, float eps, float momentum, float* scale, float* shift,
                                 float* mean_inv, double* stage1, hipStream_t s)
{
    if (nblk > 512 && stage1) {      // 4096 tile partials at B=32, 256x256: one workgroup alone needed 40 us
        hipLaunchKernelGGL(reduce_rows32_kernel, dim3(BN_STAGE1), dim3(1024), 0, s, partial, nblk, stage1);
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(1), dim3(1024), 0, s, stage1, BN_STAGE1, count, gamma, moving_mean,
                           moving_var, eps, momentum, scale, shift, mean_inv);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(1), dim3(1024), 0, s, partial, nblk, count, gamma, moving_mean, moving_var,
                           eps, momentum, scale, shift, mean_inv);
    }
    return hipGetLastError();
}

// y = x + scale*c + shift   (BN apply + residual Add, bfcnn/backbone_blocks.py:242)
__global__ __launch_bounds__(256) void affine_add_kernel(const float4* __restrict__ x, const float4* __restrict__ c,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float4* __restrict__ y, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3) * 4;
        const float4 xv = x[i], cv = c[i];
        float4 o;
        o.x = xv.x + fmaf(scale[q], cv.x, shift[q]);
        o.y = xv.y + fmaf(scale[q + 1], cv.y, shift[q + 1]);
        o.z = xv.z + fmaf(scale[q + 2], cv.z, shift[q + 2]);
        o.w = xv.w + fmaf(scale[q + 3], cv.w, shift[q + 3]);
        y[i] = o;
    }
}

static inline int stream_grid(int64_t n4)
{
    int64_t g = (n4 + 255) / 256;
    return (int)(g < 4096 ? g : 4096);
}

hipError_t bf_launch_affine_add(const float* x, const float* c, const float* scale, const float* shift, float* y,
                                int64_t npix, hipStream_t s)
{
    const int64_t n4 = npix * 4;
    hipLaunchKernelGGL(affine_add_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (const float4*)x, (const float4*)c, scale,
                       shift, (float4*)y, n4);
    return hipGetLastError();
}

// y = [relu](scale*c + shift): BatchNorm apply + activation of a block's middle convolution (backbone_blocks.py:191-196)
__global__ __launch_bounds__(256) void affine_act_kernel(const float4* __restrict__ c, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float4* __restrict__ y, int relu, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3) * 4;
        const float4 cv = c[i];
        float4 o;
        o.x = fmaf(scale[q], cv.x, shift[q]);
        o.y = fmaf(scale[q + 1], cv.y, shift[q + 1]);
        o.z = fmaf(scale[q + 2], cv.z, shift[q + 2]);
        o.w = fmaf(scale[q + 3], cv.w, shift[q + 3]);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y[i] = o;
    }
}

hipError_t bf_launch_affine_act(const float* c, const float* scale, const float* shift, float* y, int relu, int64_t npix,
                                hipStream_t s)
{
    const int64_t n4 = npix * 4;
    hipLaunchKernelGGL(affine_act_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (const float4*)c, scale, shift, (float4*)y, relu, n4);
    return hipGetLastError();
}

// BN backward, pass 1: partial[blk][32] = (sum dy[16], sum dy*c[16]) over the block's pixels.
// (sum dy*xhat follows in the finalize: xhat = (c-mean)*inv.)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float4* __restrict__ dy, const float4* __restrict__ c,
                                                            float* __restrict__ partial, int64_t n4)
{
    __shared__ float red[4][32];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    // gridDim*256 is a multiple of 4, so a thread always sees the same channel quad
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 d = dy[i], cv = c[i];
        a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
        b.x = fmaf(d.x, cv.x, b.x); b.y = fmaf(d.y, cv.y, b.y); b.z = fmaf(d.z, cv.z, b.z); b.w = fmaf(d.w, cv.w, b.w);
    }
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) v[k] += __shfl_xor(v[k], m);   // lanes sharing (lane & 3)
    }
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wave][lane * 4 + k] = v[k];
            red[wave][16 + lane * 4 + k] = v[4 + k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 32)
        partial[(size_t)blockIdx.x * 32 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

hipError_t bf_launch_bn_bwd_reduce(const float* dy, const float* c, float* partial, int64_t npix, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(256), 0, s, (const float4*)dy, (const float4*)c, partial, npix * 4);
    return hipGetLastError();
}

// pass 2: dgamma = sum dy*xhat ; dc = k1*dy + k2*c + k3 with
//   k1 = gamma*inv, k2 = -gamma*inv^2*mean(dy*xhat), k3 = -gamma*inv*mean(dy) + gamma*inv^2*mean*mean(dy*xhat)
template <typename PT>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const PT* __restrict__ partial, int nblk, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_inv,
                                                              float* coef, float* dgamma)
{
    // 32 stripes x 32 columns: one workgroup of 1024 threads (with 8 stripes this tiny kernel took 138 us per
    // BatchNorm at 4096 partial rows -- 10 % of a training step); fixed summation order, fp64
    __shared__ double red[32][32];
    const int ch = threadIdx.x & 31, stripe = threadIdx.x >> 5;
    double s = 0.0;
    for (int r = stripe; r < nblk; r += 32) s += (double)partial[(size_t)r * 32 + ch];
    red[stripe][ch] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        double sdy = 0.0, sdyc = 0.0;
        for (int k = 0; k < 32; ++k) { sdy += red[k][c]; sdyc += red[k][16 + c]; }
        const double mean = mean_inv[c], inv = mean_inv[16 + c], g = gamma[c];
        const double sdyx = (sdyc - mean * sdy) * inv;          // sum dy*xhat
        dgamma[c] = (float)sdyx;
        const double mdy = sdy / count, mdyx = sdyx / count;
        coef[c] = (float)(g * inv);
        coef[16 + c] = (float)(-g * inv * inv * mdyx);
        coef[32 + c] = (float)(-g * inv * mdy + g * inv * inv * mean * mdyx);
    }
}

hipError_t bf_launch_bn_bwd_finalize(const float* partial, int nblk, double count, const float* gamma,
                                     const float* mean_inv, float* coef, float* dgamma, double* stage1, hipStream_t s)
{
    if (nblk > 512 && stage1) {
        hipLaunchKernelGGL(reduce_rows32_kernel, dim3(BN_STAGE1), dim3(1024), 0, s, partial, nblk, stage1);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(1), dim3(1024), 0, s, stage1, BN_STAGE1, count, gamma, mean_inv, coef,
                           dgamma);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3(1), dim3(1024), 0, s, partial, nblk, count, gamma, mean_inv, coef, dgamma);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float4* __restrict__ dy, const float4* __restrict__ c,
                                                           const float* __restrict__ coef, float4* __restrict__ dc, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3) * 4;
        const float4 d = dy[i], cv = c[i];
        float4 o;
        o.x = fmaf(coef[q], d.x, fmaf(coef[16 + q], cv.x, coef[32 + q]));
        o.y = fmaf(coef[q + 1], d.y, fmaf(coef[17 + q], cv.y, coef[33 + q]));
        o.z = fmaf(coef[q + 2], d.z, fmaf(coef[18 + q], cv.z, coef[34 + q]));
        o.w = fmaf(coef[q + 3], d.w, fmaf(coef[19 + q], cv.w, coef[35 + q]));
        dc[i] = o;
    }
}

hipError_t bf_launch_bn_bwd_apply(const float* dy, const float* c, const float* coef, float* dc, int64_t npix, hipStream_t s)
{
    const int64_t n4 = npix * 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, s, (const float4*)dy, (const float4*)c, coef,
                       (float4*)dc, n4);
    return hipGetLastError();
}
