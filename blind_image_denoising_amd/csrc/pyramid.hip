// Pyramid down/up-sampling (bfcnn/pyramid.py, bfcnn/upsampling.py, bfcnn/downsampling.py,
// bfcnn/utilities.py:642-672).  Pure HBM-bound NHWC kernels: one thread per output element,
// consecutive threads on consecutive channels/pixels so every wave access is contiguous;
// float4 path when channels % 4 == 0.
#include "bf_common.h"

// AveragePooling2D(pool=(kh,kw), strides=2, padding="same") (pyramid.py:266-270, 374-378):
// TF SAME pad split (extra at bottom/right), divisor = number of in-bounds taps.
template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool_s2_same_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W,
                                                              int C, int kh, int kw, int OH, int OW, int pt, int pl)
{
    const int Cv = C / V;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const int y0 = oy * 2 - pt, x0 = ox * 2 - pl;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        int cnt = 0;
        for (int ky = 0; ky < kh; ++ky) {
            const int y = y0 + ky;
            if (y < 0 || y >= H) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int x = x0 + kx;
                if (x < 0 || x >= W) continue;
                const T val = in[(((int64_t)b * H + y) * W + x) * Cv + c];
                const float* pv = reinterpret_cast<const float*>(&val);
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] += pv[v];
                ++cnt;
            }
        }
        T o;
        float* po = reinterpret_cast<float*>(&o);
        const float d = (float)cnt;
#pragma unroll
        for (int v = 0; v < V; ++v) po[v] = acc[v] / d;
        out[i] = o;
    }
}

static inline int grid_for(int64_t n)
{
    int64_t g = (n + 255) / 256;
    return (int)(g < 8192 ? (g < 1 ? 1 : g) : 8192);
}

static inline void same_pad(int n, int k, int s, int* out, int* before)
{
    *out = (n + s - 1) / s;
    int total = (*out - 1) * s + k - n;
    if (total < 0) total = 0;
    *before = total / 2;
}

extern "C" int bf_avgpool_s2_same(const float* in, float* out, int B, int H, int W, int C, int kh, int kw, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0) return BF_EINVAL;
    int OH, OW, pt, pl;
    same_pad(H, kh, 2, &OH, &pt);
    same_pad(W, kw, 2, &OW, &pl);
    hipStream_t s = (hipStream_t)stream;
    if (C % 4 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        const int64_t n = (int64_t)B * OH * OW * (C / 4);
        hipLaunchKernelGGL((avgpool_s2_same_kernel<float4, 4>), dim3(grid_for(n)), dim3(256), 0, s, (const float4*)in,
                           (float4*)out, B, H, W, C, kh, kw, OH, OW, pt, pl);
    } else {
        const int64_t n = (int64_t)B * OH * OW * C;
        hipLaunchKernelGGL((avgpool_s2_same_kernel<float, 1>), dim3(grid_for(n)), dim3(256), 0, s, in, out, B, H, W, C, kh, kw,
                           OH, OW, pt, pl);
    }
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// tf.nn.avg_pool2d(2x2, s2, VALID) + clip[0,255] + tf.round (utilities.py:655-663, the GT pyramid
// of train_loop.py:239-247, 273-274)
__global__ __launch_bounds__(256) void avgpool2_valid_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                             int W, int C, int OH, int OW, int clip_values, int round_values)
{
    const int64_t n = (int64_t)B * OH * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const float* p = in + (((int64_t)b * H + oy * 2) * W + ox * 2) * C + c;
        float v = ((p[0] + p[C]) + (p[(int64_t)W * C] + p[(int64_t)W * C + C])) * 0.25f;
        if (clip_values) v = fminf(fmaxf(v, 0.f), 255.f);
        if (round_values) v = rintf(v);
        out[i] = v;
    }
}

extern "C" int bf_avgpool2_valid(const float* in, float* out, int B, int H, int W, int C, int clip_values, int round_values,
                                 void* stream)
{
    if (!in || !out || B <= 0 || H < 2 || W < 2 || C <= 0) return BF_EINVAL;
    const int OH = H / 2, OW = W / 2;
    const int64_t n = (int64_t)B * OH * OW * C;
    hipLaunchKernelGGL(avgpool2_valid_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C, OH, OW,
                       clip_values, round_values);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// UpSampling2D(2, bilinear) = tf.image.resize half-pixel centres, edge clamp: per axis
// out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1] (pyramid.py:319-325,380-382,
// 434-436); nearest = pixel replication (upsampling.py:65,105).  out = alpha*up(in) + beta*other
// fuses the Laplacian split (x - up(down), pyramid.py:383) and merge (up(acc) + level, :437).
template <typename T, int V>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ in, const T* __restrict__ other, T* __restrict__ out,
                                                         int B, int H, int W, int C, int bilinear, float alpha, float beta)
{
    const int Cv = C / V, OH = 2 * H, OW = 2 * W;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const int iy = oy >> 1, ix = ox >> 1;
        float r[V];
        const T* base = in + (int64_t)b * H * W * Cv + c;
        if (bilinear) {
            const int y1 = (oy & 1) ? min(iy + 1, H - 1) : max(iy - 1, 0);
            const int x1 = (ox & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
            const T v00 = base[((int64_t)iy * W + ix) * Cv], v01 = base[((int64_t)iy * W + x1) * Cv];
            const T v10 = base[((int64_t)y1 * W + ix) * Cv], v11 = base[((int64_t)y1 * W + x1) * Cv];
            const float *p00 = (const float*)&v00, *p01 = (const float*)&v01, *p10 = (const float*)&v10, *p11 = (const float*)&v11;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                // rows first (as the separable resize does), then columns
                const float top = 0.75f * p00[v] + 0.25f * p10[v];
                const float top1 = 0.75f * p01[v] + 0.25f * p11[v];
                r[v] = 0.75f * top + 0.25f * top1;
            }
        } else {
            const T v00 = base[((int64_t)iy * W + ix) * Cv];
            const float* p00 = (const float*)&v00;
#pragma unroll
            for (int v = 0; v < V; ++v) r[v] = p00[v];
        }
        T o;
        float* po = (float*)&o;
        if (other) {
            const T ov = other[i];
            const float* pp = (const float*)&ov;
#pragma unroll
            for (int v = 0; v < V; ++v) po[v] = alpha * r[v] + beta * pp[v];
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) po[v] = alpha * r[v];
        }
        out[i] = o;
    }
}

extern "C" int bf_upsample2x(const float* in, const float* other, float* out, int B, int H, int W, int C, int bilinear,
                             float alpha, float beta, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool al = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!other || (uintptr_t)other % 16 == 0);
    if (C % 4 == 0 && al) {
        const int64_t n = (int64_t)B * 4 * H * W * (C / 4);
        hipLaunchKernelGGL((upsample2x_kernel<float4, 4>), dim3(grid_for(n)), dim3(256), 0, s, (const float4*)in,
                           (const float4*)other, (float4*)out, B, H, W, C, bilinear, alpha, beta);
    } else {
        const int64_t n = (int64_t)B * 4 * H * W * C;
        hipLaunchKernelGGL((upsample2x_kernel<float, 1>), dim3(grid_for(n)), dim3(256), 0, s, in, other, out, B, H, W, C,
                           bilinear, alpha, beta);
    }
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// x[:, ::2, ::2, :] (downsampling.py:61)
__global__ __launch_bounds__(256) void strided_slice2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                             int W, int C, int OH, int OW)
{
    const int64_t n = (int64_t)B * OH * OW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        out[i] = in[(((int64_t)b * H + oy * 2) * W + ox * 2) * C + c];
    }
}

extern "C" int bf_strided_slice2(const float* in, float* out, int B, int H, int W, int C, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return BF_EINVAL;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * C;
    hipLaunchKernelGGL(strided_slice2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C, OH, OW);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
