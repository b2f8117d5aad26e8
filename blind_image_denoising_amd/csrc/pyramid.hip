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

// Row-oriented form for channel counts that are not a multiple of 4 (the colour pyramids: C = 3, C = 1).  One workgroup
// = one output row x 256 consecutive output floats: the kh input rows are summed vertically while they are loaded
// (flat, fully coalesced row segments, no per-element pixel arithmetic) into an LDS line of column sums, then every
// output adds its kw taps from LDS.  The element-per-thread kernel above issues kh*kw scalar loads and two 64-bit
// divisions per output: 136 us for a [32,512,512,3] level (0.9 TB/s); this form takes 93 us.  (Staging 4 output rows
// per workgroup and summing all kh*kw taps from LDS measured slower: 127 us.)
constexpr int AP_LINE = 1024;
__global__ __launch_bounds__(256) void avgpool_s2_same_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                                  int C, int kh, int kw, int OH, int OW, int pt, int pl)
{
    __shared__ float colsum[AP_LINE];
    const int oy = blockIdx.y % OH, b = blockIdx.y / OH;
    const int row_floats = OW * C;
    const int xo0 = blockIdx.x * 256;                              // first flat output of this workgroup
    const int xo1 = min(xo0 + 256, row_floats) - 1;                // last one
    const int px_lo = xo0 / C, px_hi = xo1 / C;
    const int xin0 = 2 * px_lo - pl;                               // first input pixel of the line (may be < 0)
    const int nfl = (2 * (px_hi - px_lo) + kw) * C;                // floats of the line
    const int jmin = max(0, -xin0) * C, jmax = min(nfl, (W - xin0) * C);
    const int y0 = oy * 2 - pt;
    const int ylo = max(y0, 0), yhi = min(y0 + kh, H);             // valid input rows [ylo, yhi)
    const float* base = in + ((int64_t)b * H * W + xin0) * C;      // + y*W*C + j
    for (int j = threadIdx.x; j < nfl; j += 256) {
        float a = 0.f;
        if (j >= jmin && j < jmax)
            for (int y = ylo; y < yhi; ++y) a += base[(int64_t)y * W * C + j];
        colsum[j] = a;
    }
    __syncthreads();
    const int xo = xo0 + threadIdx.x;
    if (xo <= xo1) {
        const int ox = xo / C, c = xo - ox * C;
        const int x0 = 2 * ox - pl;
        const int klo = max(0, -x0), khi = min(kw, W - x0);        // valid taps [klo, khi)
        const float* p = colsum + (x0 - xin0) * C + c;
        float a = 0.f;
        for (int kx = klo; kx < khi; ++kx) a += p[kx * C];
        out[((int64_t)b * OH + oy) * row_floats + xo] = a / (float)((yhi - ylo) * (khi - klo));
    }
}

// The same, RO output rows per workgroup: every thread walks down its columns of the line keeping the last KH input rows in
// registers (two new rows per output row instead of KH): 19 row loads for 8 output rows where the kernel above issues 40, and
// eight times fewer workgroups.  Same summation order (rows top to bottom, then the kw taps): identical results.
template <int KH>
__global__ __launch_bounds__(256) void avgpool_s2_same_band_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                                  int C, int kw, int OH, int OW, int pt, int pl, int RO)
{
    constexpr int NJ = AP_LINE / 256;
    __shared__ float colsum[AP_LINE];
    const int nbands = (OH + RO - 1) / RO;
    const int oy0 = (blockIdx.y % nbands) * RO, b = blockIdx.y / nbands;
    const int row_floats = OW * C;
    const int xo0 = blockIdx.x * 256;
    const int xo1 = min(xo0 + 256, row_floats) - 1;
    const int px_lo = xo0 / C, px_hi = xo1 / C;
    const int xin0 = 2 * px_lo - pl;
    const int nfl = (2 * (px_hi - px_lo) + kw) * C;
    const int jmin = max(0, -xin0) * C, jmax = min(nfl, (W - xin0) * C);
    const float* base = in + ((int64_t)b * H * W + xin0) * C;
    float rows[NJ][KH];
    bool live[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int j = threadIdx.x + u * 256;
        live[u] = j >= jmin && j < jmax;
    }
    auto load_row = [&](const int y, const int u) -> float {
        return (live[u] && y >= 0 && y < H) ? base[(int64_t)y * W * C + threadIdx.x + u * 256] : 0.f;
    };
    const int xo = xo0 + threadIdx.x;
    const int ox = xo / C, c = xo - ox * C;
    const int x0 = 2 * ox - pl;
    const int klo = max(0, -x0), khi = min(kw, W - x0);
    const int oy_end = min(oy0 + RO, OH);
    // first output row: all KH rows; then shift by two and load the two new ones
#pragma unroll
    for (int u = 0; u < NJ; ++u)
#pragma unroll
        for (int k = 0; k < KH; ++k) rows[u][k] = load_row(2 * oy0 - pt + k, u);
    for (int oy = oy0; oy < oy_end; ++oy) {
        const int y0 = 2 * oy - pt;
        const int ylo = max(y0, 0), yhi = min(y0 + KH, H);
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int j = threadIdx.x + u * 256;
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < KH; ++k) a += rows[u][k];          // rows outside the image hold 0: same sum as the valid ones alone
            if (j < nfl) colsum[j] = a;
        }
        __syncthreads();
        if (xo <= xo1) {
            const float* p = colsum + (x0 - xin0) * C + c;
            float a = 0.f;
            for (int kx = klo; kx < khi; ++kx) a += p[kx * C];
            out[((int64_t)b * OH + oy) * row_floats + xo] = a / (float)((yhi - ylo) * (khi - klo));
        }
        if (oy + 1 < oy_end) {
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
#pragma unroll
                for (int k = 0; k + 2 < KH; ++k) rows[u][k] = rows[u][k + 2];
                if (KH >= 2) rows[u][KH - 2] = load_row(y0 + KH, u);
                rows[u][KH - 1] = load_row(y0 + KH + 1, u);
            }
        }
        __syncthreads();
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
    } else if ((2 * (256 / C + 1) + kw) * C <= AP_LINE && (int64_t)B * OH <= 65535 && (kh == 3 || kh == 5 || kh == 7)) {
        const int RO = 8;
        const dim3 grid((OW * C + 255) / 256, B * ((OH + RO - 1) / RO));
        if (kh == 3) hipLaunchKernelGGL(avgpool_s2_same_band_kernel<3>, grid, dim3(256), 0, s, in, out, H, W, C, kw, OH, OW, pt, pl, RO);
        else if (kh == 5) hipLaunchKernelGGL(avgpool_s2_same_band_kernel<5>, grid, dim3(256), 0, s, in, out, H, W, C, kw, OH, OW, pt, pl, RO);
        else hipLaunchKernelGGL(avgpool_s2_same_band_kernel<7>, grid, dim3(256), 0, s, in, out, H, W, C, kw, OH, OW, pt, pl, RO);
    } else if ((2 * (256 / C + 1) + kw) * C <= AP_LINE && (int64_t)B * OH <= 65535) {
        hipLaunchKernelGGL(avgpool_s2_same_rows_kernel, dim3((OW * C + 255) / 256, B * OH), dim3(256), 0, s, in, out, H, W, C, kh, kw,
                           OH, OW, pt, pl);
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
// The bilinear x2 value of one output element and the combination with `other`, written with explicit fused multiply-adds so
// that every kernel that forms them (element-per-thread, row-oriented, fused Laplacian split) produces the SAME bits: hipcc picks
// its own contractions per kernel otherwise.  rows first (as the separable resize does), then columns.
__device__ __forceinline__ float bf_bilinear_tap(const float v00, const float v01, const float m0, const float m1)
{
    const float a = __builtin_fmaf(0.75f, v00, 0.25f * m0);
    const float b = __builtin_fmaf(0.75f, v01, 0.25f * m1);
    return __builtin_fmaf(0.75f, a, 0.25f * b);
}
__device__ __forceinline__ float bf_axpby(const float alpha, const float up, const float beta, const float other)
{
    return __builtin_fmaf(alpha, up, beta * other);
}

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
            for (int v = 0; v < V; ++v) r[v] = bf_bilinear_tap(p00[v], p01[v], p10[v], p11[v]);
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
            for (int v = 0; v < V; ++v) po[v] = bf_axpby(alpha, r[v], beta, pp[v]);
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) po[v] = alpha * r[v];
        }
        out[i] = o;
    }
}

// Row-oriented form for C % 4 != 0: one workgroup = one INPUT row x 256 consecutive floats of the two output rows it
// produces; the three input rows it needs are addressed with row-uniform (scalar) bases, each thread does one 32-bit
// division, 6 input loads, 2 `other` loads and 2 stores for two outputs (the element-per-thread form: two 64-bit
// divisions, 4 + 1 loads and 1 store for one output; 105 us -> 64 us for a [32,512,512,3] level).
__global__ __launch_bounds__(256) void upsample2x_rows_kernel(const float* __restrict__ in, const float* __restrict__ other,
                                                              float* __restrict__ out, int H, int W, int C, int bilinear,
                                                              float alpha, float beta)
{
    const int iy = blockIdx.y % H, b = blockIdx.y / H;
    const int OW = 2 * W, row_floats = OW * C;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    if (xo >= row_floats) return;
    const int ox = xo / C, c = xo - ox * C;
    const int ix = ox >> 1;
    const float* r0 = in + ((int64_t)b * H + iy) * W * C + c;                     // this row
    float top_a, top_b;                                                             // results of output rows 2iy, 2iy+1
    if (bilinear) {
        const float* rm = in + ((int64_t)b * H + max(iy - 1, 0)) * W * C + c;      // row above (clamped)
        const float* rp = in + ((int64_t)b * H + min(iy + 1, H - 1)) * W * C + c;  // row below
        const int x1 = (ox & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
        const float v00 = r0[ix * C], v01 = r0[x1 * C];
        const float m0 = rm[ix * C], m1 = rm[x1 * C], p0 = rp[ix * C], p1 = rp[x1 * C];
        top_a = bf_bilinear_tap(v00, v01, m0, m1);
        top_b = bf_bilinear_tap(v00, v01, p0, p1);
    } else {
        top_a = top_b = r0[ix * C];
    }
    const int64_t o = ((int64_t)b * 2 * H + 2 * iy) * row_floats + xo;
    if (other) {
        out[o] = bf_axpby(alpha, top_a, beta, other[o]);
        out[o + row_floats] = bf_axpby(alpha, top_b, beta, other[o + row_floats]);
    } else {
        out[o] = alpha * top_a;
        out[o + row_floats] = alpha * top_b;
    }
}

static int launch_upsample2x_band(const float* in, const float* other, float* out, int B, int H, int W, int C, float alpha, float beta,
                                  hipStream_t s);
static int g_up_band = 1;                              // 0: upsample2x_rows_kernel (A/B, bitwise comparison in the tests)
extern "C" int bf_debug_set_upsample_band(int on) { g_up_band = on ? 1 : 0; return BF_OK; }

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
    } else if (bilinear && g_up_band && launch_upsample2x_band(in, other, out, B, H, W, C, alpha, beta, s)) {
        // (row-walking 16-byte form, defined below)
    } else if ((int64_t)B * H <= 65535 && (int64_t)2 * W * C < ((int64_t)1 << 30)) {
        hipLaunchKernelGGL(upsample2x_rows_kernel, dim3((2 * W * C + 255) / 256, B * H), dim3(256), 0, s, in, other, out, H, W, C,
                           bilinear, alpha, beta);
    } else {
        const int64_t n = (int64_t)B * 4 * H * W * C;
        hipLaunchKernelGGL((upsample2x_kernel<float, 1>), dim3(grid_for(n)), dim3(256), 0, s, in, other, out, B, H, W, C,
                           bilinear, alpha, beta);
    }
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// One level of the Laplacian split in ONE kernel (bfcnn/pyramid.py:374-385): down = AveragePooling2D(k, strides 2, same)(x) and
// lap = x - UpSampling2D(2, bilinear)(down).  The two-kernel form (bf_avgpool_s2_same, then bf_upsample2x with other = x) reads x
// twice and down once: 3.5 n floats per level where this reads x once and never reads down back: 2.25 n (x in, lap out, down out).
// Same arithmetic, operation by operation, as those two kernels (rows summed top to bottom, then the kw taps, one division by the
// number of valid taps; rows first, then columns in the resize): bitwise the same results.
//
// A workgroup (128 threads) owns DP down-pixels x DR down-rows of one image and walks down its band one down-row per step:
//   * thread t carries a sliding window of x rows 2d-2 .. 2d-PT+KH-1 for the four consecutive floats f0 + 4t .. of the x line
//     the chunk needs (16-byte loads and stores; the next step's two rows are requested before this step's arithmetic);
//   * step d: column sums of the KH pooling rows -> LDS; the down row d (one float per thread, halo pixel either side) from kw
//     LDS taps -> a ring of four down rows in LDS (+ global, where owned); then the two lap rows of down-row d-1, whose x values
//     are still in the window and whose bilinear taps are the ring's rows d-2, d-1, d.
// Needs even H and W (the reference's Laplacian levels have them: up(down) must match x), W * C % 4 == 0 and 16-byte aligned
// tensors; kh in {3, 5, 7}.
// ------------------------------------------------------------------------------------------
constexpr int LS_NT = 128, LS_XLINE = 4 * LS_NT, LS_DLINE = 2 * LS_NT;
template <int KH>
__global__ __launch_bounds__(LS_NT) void lap_split_kernel(const float* __restrict__ in, float* __restrict__ down, float* __restrict__ lap,
                                                          int H, int W, int C, int kw, int OH, int OW, int pl, int DP, int DR,
                                                          int nchunks, int nbands)
{
    constexpr int PT = (KH - 2) / 2;                    // SAME padding above the first row for an even H
    constexpr int WIN = KH - PT + 2;                    // window rows: x rows 2d-2 .. 2d-PT+KH-1
    __shared__ __attribute__((aligned(16))) float colsum[LS_XLINE];
    __shared__ float dring[4][LS_DLINE];
    const int t = threadIdx.x;
    int bx = blockIdx.x;
    const int ch = bx % nchunks; bx /= nchunks;
    const int band = bx % nbands;
    const int b = bx / nbands;
    const int pd0 = ch * DP, pd1 = min(pd0 + DP, OW);               // owned down pixels
    const int d0 = band * DR, d1 = min(d0 + DR, OH);                // owned down rows
    const int hp0 = max(pd0 - 1, 0), hp1 = min(pd1 + 1, OW);        // with the halo pixel either side
    const int row_x = W * C, row_d = OW * C;
    const int xin0 = 2 * hp0 - pl;
    const int f0 = (xin0 * C) >= 0 ? (xin0 * C) / 4 * 4 : -((-(xin0 * C) + 3) / 4 * 4);      // floor to a multiple of 4
    const int fx = f0 + 4 * t;                                      // this thread's first x float (of the row)
    const bool live = fx >= 0 && fx + 4 <= row_x && fx < (2 * (hp1 - 1) - pl + kw) * C;
    const bool owned = fx >= 2 * pd0 * C && fx < 2 * pd1 * C;       // its lap outputs belong to this chunk (float4 granular)
    const float* xb = in + (int64_t)b * H * row_x + fx;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_row = [&](const int y) -> f32x4 {
        return (live && y >= 0 && y < H) ? *reinterpret_cast<const f32x4*>(xb + (int64_t)y * row_x) : zero4;
    };
    const int ds = max(d0 - 1, 0), de = min(d1, OH - 1);
    f32x4 win[WIN];
#pragma unroll
    for (int k = 0; k < WIN; ++k) win[k] = load_row(2 * ds - 2 + k);
    // down floats of this thread: u = t and t + 128 of the halo-extended down line
    const int nd = (hp1 - hp0) * C;
    // bilinear taps of this thread's four x floats (column part: fixed for the band)
    int i0[4], i1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = owned ? fx + e : 2 * pd0 * C;
        const int ox = f / C, c = f - ox * C, ix = ox >> 1;
        const int x1 = (ox & 1) ? min(ix + 1, OW - 1) : max(ix - 1, 0);
        i0[e] = (ix - hp0) * C + c;
        i1[e] = (x1 - hp0) * C + c;
    }
    auto emit = [&](const int i, const f32x4 xa, const f32x4 xbb) {       // lap rows 2i, 2i+1 (x values xa, xbb)
        const float* dm = dring[max(i - 1, 0) & 3];
        const float* dc = dring[i & 3];
        const float* dp = dring[min(i + 1, OH - 1) & 3];
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v00 = dc[i0[e]], v01 = dc[i1[e]];
            const float m0 = dm[i0[e]], m1 = dm[i1[e]], p0 = dp[i0[e]], p1 = dp[i1[e]];
            ra[e] = bf_axpby(-1.0f, bf_bilinear_tap(v00, v01, m0, m1), 1.0f, xa[e]);
            rb[e] = bf_axpby(-1.0f, bf_bilinear_tap(v00, v01, p0, p1), 1.0f, xbb[e]);
        }
        float* o = lap + ((int64_t)b * H + 2 * i) * row_x + fx;
        *reinterpret_cast<f32x4*>(o) = ra;
        *reinterpret_cast<f32x4*>(o + row_x) = rb;
    };
    for (int d = ds; d <= de; ++d) {
        // the two rows the NEXT step adds to the window: requested now, used at the end of the step
        f32x4 nx0 = zero4, nx1 = zero4;
        if (d < de) {
            nx0 = load_row(2 * d - 2 + WIN);
            nx1 = load_row(2 * d - 2 + WIN + 1);
        }
        f32x4 a = zero4;
#pragma unroll
        for (int k = 0; k < KH; ++k) a += win[2 - PT + k];                 // rows outside the image hold 0
        *reinterpret_cast<f32x4*>(colsum + 4 * t) = a;
        __syncthreads();
        const int y0 = 2 * d - PT;
        const int nrows = min(y0 + KH, H) - max(y0, 0);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int u = t + uu * LS_NT;
            if (u < nd) {
                const int p = hp0 + u / C, c = u - (u / C) * C;
                const int x0 = 2 * p - pl;
                const int klo = max(0, -x0), khi = min(kw, W - x0);
                const float* q = colsum + (x0 * C + c - f0);
                float sum = 0.f;
                for (int kx = klo; kx < khi; ++kx) sum += q[kx * C];
                const float v = sum / (float)(nrows * (khi - klo));
                dring[d & 3][u] = v;
                if (d >= d0 && d < d1 && p >= pd0 && p < pd1) down[((int64_t)b * OH + d) * row_d + p * C + c] = v;
            }
        }
        __syncthreads();
        if (owned && d - 1 >= d0) emit(d - 1, win[0], win[1]);
        if (owned && d == OH - 1 && d < d1) emit(d, win[2], win[3]);       // last image row: its lower neighbour is itself
#pragma unroll
        for (int k = 0; k + 2 < WIN; ++k) win[k] = win[k + 2];
        win[WIN - 2] = nx0;
        win[WIN - 1] = nx1;
    }
}

// returns BF_EUNSUPPORTED (nothing launched) for shapes the fused kernel does not take: the caller runs the two-kernel form
extern "C" int bf_laplacian_split(const float* in, float* down, float* lap, int B, int H, int W, int C, int kh, int kw, void* stream)
{
    if (!in || !down || !lap || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0) return BF_EINVAL;
    if ((H & 1) || (W & 1) || (W * C) % 4 || !(kh == 3 || kh == 5 || kh == 7) || kw < 2 || kw > 15) return BF_EUNSUPPORTED;
    if (((uintptr_t)in | (uintptr_t)lap) % 16) return BF_EUNSUPPORTED;
    // channel counts that are a multiple of 4 keep the element-per-thread float4 kernels (another summation order: the promise of
    // bitwise-equal results would not hold); the colour pyramids of the reference are C = 3 and C = 1
    if (C % 4 == 0) return BF_EUNSUPPORTED;
    int OH, OW, pt, pl;
    same_pad(H, kh, 2, &OH, &pt);
    same_pad(W, kw, 2, &OW, &pl);
    if (pt != (kh - 2) / 2) return BF_EUNSUPPORTED;
    // down pixels per chunk: the x line (2 (DP + 2) + kw) C + 3 floats fits 4 per thread, the down line (DP + 2) C fits 2 per thread
    int dpmax = (LS_XLINE / C - kw - 1) / 2 - 2;
    if ((LS_DLINE / C) - 2 < dpmax) dpmax = LS_DLINE / C - 2;
    if (dpmax < 2) return BF_EUNSUPPORTED;
    int nchunks = (OW + dpmax - 1) / dpmax;
    int DP = (OW + nchunks - 1) / nchunks;
    if ((DP * C) % 2) ++DP;                              // 2 DP C % 4 == 0: a thread's four floats never straddle two chunks
    if (DP > dpmax) { DP = dpmax - (((dpmax * C) % 2) ? 1 : 0); }
    if (DP < 1 || (2 * DP * C) % 4) return BF_EUNSUPPORTED;
    nchunks = (OW + DP - 1) / DP;
    // down rows per band: enough workgroups to fill the chip several times over, bands tall enough to amortise the 2 + KH halo rows
    int DR = 32;
    while (DR > 8 && (int64_t)B * nchunks * ((OH + DR - 1) / DR) < 2048) DR /= 2;
    const int nbands = (OH + DR - 1) / DR;
    const int64_t grid = (int64_t)B * nchunks * nbands;
    if (grid > 0x7fffffff) return BF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (kh == 3) hipLaunchKernelGGL(lap_split_kernel<3>, dim3((unsigned)grid), dim3(LS_NT), 0, s, in, down, lap, H, W, C, kw, OH, OW, pl, DP, DR, nchunks, nbands);
    else if (kh == 5) hipLaunchKernelGGL(lap_split_kernel<5>, dim3((unsigned)grid), dim3(LS_NT), 0, s, in, down, lap, H, W, C, kw, OH, OW, pl, DP, DR, nchunks, nbands);
    else hipLaunchKernelGGL(lap_split_kernel<7>, dim3((unsigned)grid), dim3(LS_NT), 0, s, in, down, lap, H, W, C, kw, OH, OW, pl, DP, DR, nchunks, nbands);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// out = alpha * UpSampling2D(2, bilinear)(in) + beta * other in the row-walking 16-byte form of lap_split_kernel (the merge of the
// inverse Laplacian pyramid, pyramid.py:429-437, and every other up-sample + combine on C % 4 != 0 maps): a workgroup owns DP input
// pixels x DR input rows, keeps a ring of four input rows in LDS (one float per thread and row, halo pixel either side) and emits the
// two output rows of input row d-1 per step as 16-byte loads of `other` and 16-byte stores.  upsample2x_rows_kernel does the same
// arithmetic (bf_bilinear_tap / bf_axpby: bitwise the same results) with 4-byte accesses: 4.1 TB/s on a [32,512,512,3] level.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LS_NT) void upsample2x_band_kernel(const float* __restrict__ in, const float* __restrict__ other,
                                                                float* __restrict__ out, int H, int W, int C, float alpha, float beta,
                                                                int DP, int DR, int nchunks, int nbands)
{
    __shared__ float dring[4][LS_DLINE];
    const int t = threadIdx.x;
    int bx = blockIdx.x;
    const int ch = bx % nchunks; bx /= nchunks;
    const int band = bx % nbands;
    const int b = bx / nbands;
    const int pd0 = ch * DP, pd1 = min(pd0 + DP, W);                // owned input pixels
    const int d0 = band * DR, d1 = min(d0 + DR, H);                 // owned input rows
    const int hp0 = max(pd0 - 1, 0), hp1 = min(pd1 + 1, W);
    const int row_o = 2 * W * C, row_i = W * C;
    const int nd = (hp1 - hp0) * C;
    const int fx = 2 * pd0 * C + 4 * t;                             // this thread's first output float of a row
    const bool owned = fx < 2 * pd1 * C;
    int i0[4], i1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = owned ? fx + e : 2 * pd0 * C;
        const int ox = f / C, c = f - ox * C, ix = ox >> 1;
        const int x1 = (ox & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
        i0[e] = (ix - hp0) * C + c;
        i1[e] = (x1 - hp0) * C + c;
    }
    const float* ib = in + (int64_t)b * H * row_i + hp0 * C;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto emit = [&](const int i, const f32x4 oa, const f32x4 ob) {
        const float* dm = dring[max(i - 1, 0) & 3];
        const float* dc = dring[i & 3];
        const float* dp = dring[min(i + 1, H - 1) & 3];
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v00 = dc[i0[e]], v01 = dc[i1[e]];
            const float ta = bf_bilinear_tap(v00, v01, dm[i0[e]], dm[i1[e]]), tb = bf_bilinear_tap(v00, v01, dp[i0[e]], dp[i1[e]]);
            ra[e] = other ? bf_axpby(alpha, ta, beta, oa[e]) : alpha * ta;
            rb[e] = other ? bf_axpby(alpha, tb, beta, ob[e]) : alpha * tb;
        }
        float* o = out + ((int64_t)b * 2 * H + 2 * i) * row_o + fx;
        *reinterpret_cast<f32x4*>(o) = ra;
        *reinterpret_cast<f32x4*>(o + row_o) = rb;
    };
    const int ds = max(d0 - 1, 0), de = min(d1, H - 1);
    for (int d = ds; d <= de; ++d) {
        // `other` rows of input row d-1 (and of d on the image's last row): requested before the ring is touched
        f32x4 oa = zero4, ob = zero4, oc = zero4, od = zero4;
        const bool e1 = owned && d - 1 >= d0, e2 = owned && d == H - 1 && d < d1;
        if (other && e1) {
            const float* o = other + ((int64_t)b * 2 * H + 2 * (d - 1)) * row_o + fx;
            oa = *reinterpret_cast<const f32x4*>(o);
            ob = *reinterpret_cast<const f32x4*>(o + row_o);
        }
        if (other && e2) {
            const float* o = other + ((int64_t)b * 2 * H + 2 * d) * row_o + fx;
            oc = *reinterpret_cast<const f32x4*>(o);
            od = *reinterpret_cast<const f32x4*>(o + row_o);
        }
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int u = t + uu * LS_NT;
            if (u < nd) dring[d & 3][u] = ib[(int64_t)d * row_i + u];
        }
        __syncthreads();
        if (e1) emit(d - 1, oa, ob);
        if (e2) emit(d, oc, od);
    }
}

// 1 when the band kernel took the call (bilinear, C % 4 != 0, 2 W C % 4 == 0, aligned tensors), 0 when the caller must use another form
static int launch_upsample2x_band(const float* in, const float* other, float* out, int B, int H, int W, int C, float alpha, float beta,
                                  hipStream_t s)
{
    if (C % 4 == 0 || (2 * W * C) % 4 || (((uintptr_t)out | (uintptr_t)other) % 16)) return 0;
    int dpmax = LS_DLINE / C - 2;                       // input line (DP + 2) C floats: two per thread
    if (LS_XLINE / (2 * C) < dpmax) dpmax = LS_XLINE / (2 * C);      // output line 2 DP C floats: four per thread
    if (dpmax < 2) return 0;
    int nchunks = (W + dpmax - 1) / dpmax;
    int DP = (W + nchunks - 1) / nchunks;
    if ((DP * C) % 2) ++DP;                             // 2 DP C % 4 == 0
    if (DP > dpmax) DP = dpmax - (((dpmax * C) % 2) ? 1 : 0);
    if (DP < 1 || (2 * DP * C) % 4) return 0;
    nchunks = (W + DP - 1) / DP;
    int DR = 32;
    while (DR > 8 && (int64_t)B * nchunks * ((H + DR - 1) / DR) < 2048) DR /= 2;
    const int nbands = (H + DR - 1) / DR;
    const int64_t grid = (int64_t)B * nchunks * nbands;
    if (grid > 0x7fffffff) return 0;
    hipLaunchKernelGGL(upsample2x_band_kernel, dim3((unsigned)grid), dim3(LS_NT), 0, s, in, other, out, H, W, C, alpha, beta, DP, DR,
                       nchunks, nbands);
    return 1;
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
