// The layers either side of the 3x3 C16 stack: input normalisation + base convolution, and the
// denoiser head + denormalisation + uint8 store.  Both are HBM-bound (64 B/pixel of fp32
// activations against <2 % of the FLOPs), so they are plain VALU kernels with 16-byte accesses.
#include "bf_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------
// base convolution: [cast] -> [virtual pad_to_power_of_2] -> normalise -> conv k x k, Cin -> 16
//   cast / pad : bfcnn/module_denoiser.py:53-56, bfcnn/utilities.py:736-751
//   normalise  : bfcnn/model.py:100-102 -> bfcnn/utilities.py:449-461  clip(x)/(max-min) - 0.5
//   conv       : bfcnn/backbone_resnet.py:137-147,258-262 (no BN on the base layer)
// Source image [B,Hs,Ws,Cin] (u8 or f32); output [B,H,W,16] with H >= Hs, W >= Ws.  Pixels in
// the padded band carry the value 0 (=> normalised -0.5), pixels outside HxW are conv zeros.
// ------------------------------------------------------------------------------------------
// One workgroup = one 16x16 output tile.  The (16+K-1)^2 input patch is normalised once into LDS
// (3 B/px of HBM reads instead of K*K byte loads per output pixel); the K*K*Cin*16 weights are
// wave-uniform and reach the FMAs through scalar loads; each lane writes its pixel's 16 channels
// as four 16-byte stores.
constexpr int BC_T = 16;

template <int CIN, int K, bool U8>
__global__ __launch_bounds__(256) void base_conv_kernel(BaseConvArgs a)
{
    constexpr int R = K / 2, IT = BC_T + 2 * R;
    __shared__ float tile[IT * IT * CIN];
    const float* __restrict__ wg = a.w;
    const int tiles_x = (a.W + BC_T - 1) / BC_T, tiles_y = (a.H + BC_T - 1) / BC_T;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * BC_T, x0 = tx * BC_T;
    if (a.status && blockIdx.x == 0 && threadIdx.x == 0) *a.status = 0;       // first kernel of a forward
    const float range = a.v_max - a.v_min;       // true division: (x - min) / (max - min) - 0.5 is exact for mid-grey
    {
        // all loads of a thread first, then the stores (a rolled load -> store loop pays one memory round trip per element)
        constexpr int NE = (IT * IT * CIN + 255) / 256;
        float raw[NE];
        bool inimg[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int n = threadIdx.x + i * 256;
            const int ci = n % CIN, px = (n / CIN) % IT, row = n / (CIN * IT);
            const int gy = y0 - R + row, gx = x0 - R + px;
            inimg[i] = n < IT * IT * CIN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            raw[i] = 0.f;                                       // pad_to_power_of_2 band: value 0
            if (inimg[i] && gy < a.Hs && gx < a.Ws) {
                const int64_t si = (((int64_t)b * a.Hs + gy) * a.Ws + gx) * CIN + ci;
                raw[i] = U8 ? (float)reinterpret_cast<const uint8_t*>(a.in)[si] : reinterpret_cast<const float*>(a.in)[si];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int n = threadIdx.x + i * 256;
            // conv zero padding outside H x W
            const float v = inimg[i] ? (fminf(fmaxf(raw[i], a.v_min), a.v_max) - a.v_min) / range - 0.5f : 0.f;
            if (n < IT * IT * CIN) tile[n] = v;
        }
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int y = y0 + ly, x = x0 + lx;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float v = tile[((ly + ky) * IT + lx + kx) * CIN + ci];
                const float* wr = wg + ((ky * K + kx) * CIN + ci) * 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
            }
    if (y < a.H && x < a.W) {
        if (a.act_relu) {
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaxf(acc[c], 0.f);
        }
        if (a.out_split) {
            // split-planar: 4 planes [H][W][8 x f16] per image = hi(c0..7), hi(c8..15), lo(c0..7), lo(c8..15)
            const int64_t hw = (int64_t)a.H * a.W;
            char* base = reinterpret_cast<char*>(a.out) + (int64_t)b * hw * 64 + ((int64_t)y * a.W + x) * 16;
            h8 hi[2], lo[2];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const _Float16 h = (_Float16)acc[c];
                hi[c >> 3][c & 7] = h;
                lo[c >> 3][c & 7] = (_Float16)(acc[c] - (float)h);
            }
            *reinterpret_cast<h8*>(base) = hi[0];
            *reinterpret_cast<h8*>(base + hw * 16) = hi[1];
            if (a.out_split == 2) {                                  // compact: fp8 lo planes of 8 bytes per pixel
                char* lob = reinterpret_cast<char*>(a.out) + (int64_t)b * hw * 64 + hw * 32 + ((int64_t)y * a.W + x) * 8;
                *reinterpret_cast<bf_u2*>(lob) = bf_h3c_encode8(lo[0]);
                *reinterpret_cast<bf_u2*>(lob + hw * 8) = bf_h3c_encode8(lo[1]);
                return;
            }
            *reinterpret_cast<h8*>(base + hw * 32) = lo[0];
            *reinterpret_cast<h8*>(base + hw * 48) = lo[1];
            return;
        }
        float4* o = reinterpret_cast<float4*>(a.out + (((int64_t)b * a.H + y) * a.W + x) * 16);
        o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        o[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
        o[3] = make_float4(acc[12], acc[13], acc[14], acc[15]);
    }
}

template <int CIN, int K>
static hipError_t launch_base(const BaseConvArgs& a, hipStream_t s)
{
    const int grid = a.B * ((a.H + BC_T - 1) / BC_T) * ((a.W + BC_T - 1) / BC_T);
    if (a.in_is_u8) hipLaunchKernelGGL((base_conv_kernel<CIN, K, true>), dim3(grid), dim3(256), 0, s, a);
    else            hipLaunchKernelGGL((base_conv_kernel<CIN, K, false>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

static int g_base_rows = 1;                           // 0: the vector kernel everywhere, 2: the row kernel wherever it can run (A/B, tests)
void bf_set_base_conv_rows(int on) { g_base_rows = on == 2 ? 2 : (on ? 1 : 0); }
// The row kernel pays ~11 row steps of latency per workgroup before it streams (three ring rows + at least eight rows of a band) and
// works on 256-column chunks: below ~8 192 rows of chunks per forward, or with chunks less than 60 % full, the tile kernel is ahead
// (tools/exp/base_select.py, resnet 1x6 per call: 1 x 256^2 57 us for 68, 8 x 256^2 151 for 158, 16 x 256^2 237 for 239, 32 x 256^2 439
// for 420, 64 x 64^2 138 for 154).
static bool base_rows_preferred(const BaseConvArgs& a)
{
    const int64_t nchunks = (a.W + 255) / 256;
    return (int64_t)a.B * a.H * nchunks >= 8192 && (int64_t)a.W * 5 >= nchunks * 256 * 3;
}

hipError_t bf_launch_base_conv(const BaseConvArgs& a, hipStream_t s)
{
    // the metric's configuration (u8 in, 3x3x3 -> 16, split-planar out): row-streaming matrix-core kernel (base_rows.hip)
    if (g_base_rows && bf_base_conv_rows_supports(a) && (g_base_rows == 2 || base_rows_preferred(a))) return bf_launch_base_conv_rows(a, s);
#define BF_BASE(C, KK) if (a.cin == C && a.k == KK) return launch_base<C, KK>(a, s);
    BF_BASE(3, 3) BF_BASE(3, 5) BF_BASE(3, 7) BF_BASE(3, 1)
    BF_BASE(1, 3) BF_BASE(1, 5) BF_BASE(1, 7) BF_BASE(1, 1)
#undef BF_BASE
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// base convolution weight gradient  dW[ky,kx,ci,co] = sum xn[pix + tap][ci] * dy[pix][co]
// (tape.gradient of bfcnn/train_loop.py:302-304 for the base kernel).  Persistent workgroups,
// LDS-staged tiles, one thread per (tap,ci,co) output, fixed-order partial reduction.
// ------------------------------------------------------------------------------------------
constexpr int BW_TH = 16, BW_TW = 32;

template <int CIN, int K>
__global__ __launch_bounds__(512) void base_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dy,
                                                         float* __restrict__ partial, int B, int H, int W,
                                                         float v_min, float v_max, int tiles_x, int tiles_y, int ntiles)
{
    // thread = (row slice s = tid / 32 of the 16-row tile, tap-channel combination j = tid % 32 [+ 32 i]); it keeps the 16
    // output-channel accumulators of its combinations: per pixel ONE read of x and four 16-byte reads of dy (the same address
    // for the 32 lanes of a slice: a broadcast) feed 16 FMAs.  The first form (one thread per output, two LDS reads per FMA)
    // was LDS-bound: 155 us at 32 x 256 x 256.
    constexpr int R = K / 2, IH = BW_TH + 2 * R, IW = BW_TW + 2 * R;
    constexpr int NCOMB = K * K * CIN, PERC = (NCOMB + 31) / 32, NOUT = NCOMB * 16;
    static_assert(BW_TH == 16, "one row slice per 32 threads");
    __shared__ float tx_[IH * IW * CIN];
    __shared__ __attribute__((aligned(16))) float td[BW_TH * BW_TW * 16];
    const int tid = threadIdx.x, slice = tid >> 5, j0 = tid & 31;
    const float range = v_max - v_min;
    f32x4 acc[PERC][4];
    int xoff[PERC];
#pragma unroll
    for (int i = 0; i < PERC; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int jj = j0 + 32 * i;
        if (jj >= NCOMB) jj = 0;
        const int ci = jj % CIN, tap = jj / CIN;
        xoff[i] = ((tap / K) * IW + (tap % K)) * CIN + ci;
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tt = t;
        const int txi = tt % tiles_x; tt /= tiles_x;
        const int tyi = tt % tiles_y;
        const int b = tt / tiles_y;
        const int y0 = tyi * BW_TH, x0 = txi * BW_TW;
        for (int n = tid; n < IH * IW * CIN; n += 512) {
            const int ci = n % CIN, px = (n / CIN) % IW, row = n / (CIN * IW);
            const int gy = y0 - R + row, gx = x0 - R + px;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const float raw = in[(((int64_t)b * H + gy) * W + gx) * CIN + ci];
                v = (fminf(fmaxf(raw, v_min), v_max) - v_min) / range - 0.5f;
            }
            tx_[n] = v;
        }
        for (int n = tid; n < BW_TH * BW_TW * 4; n += 512) {
            const int row = n / (BW_TW * 4), rem = n - row * (BW_TW * 4);
            const int gy = y0 + row, gx = x0 + (rem >> 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W)
                v = *reinterpret_cast<const float4*>(dy + (((int64_t)b * H + gy) * W + gx) * 16 + (rem & 3) * 4);
            *reinterpret_cast<float4*>(td + n * 4) = v;
        }
        __syncthreads();
        for (int c = 0; c < BW_TW; ++c) {
            const int pb = (slice * IW + c) * CIN, db = (slice * BW_TW + c) * 16;
            f32x4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = *reinterpret_cast<const f32x4*>(td + db + 4 * q);
#pragma unroll
            for (int i = 0; i < PERC; ++i) {
                const float xv = tx_[pb + xoff[i]];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][q] += d[q] * xv;
            }
        }
        __syncthreads();
    }
    // 16 row slices -> one partial per workgroup (fixed order), through the dy tile's LDS ([16 slices][<= 32 * 16] per round)
    float* red = td;
#pragma unroll
    for (int i = 0; i < PERC; ++i) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(red + (slice * 32 + j0) * 16 + 4 * q) = acc[i][q];
        __syncthreads();
        const int jj = (tid >> 4) + 32 * i, co = tid & 15;            // 512 threads = 32 combinations x 16 channels
        if (jj < NCOMB) {
            float a = 0.f;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) a += red[(sl * 32 + (tid >> 4)) * 16 + co];
            partial[(size_t)blockIdx.x * NOUT + jj * 16 + co] = a;
        }
    }
}

int bf_base_wgrad_grid(int B, int H, int W)
{
    const int ntiles = B * ((H + BW_TH - 1) / BW_TH) * ((W + BW_TW - 1) / BW_TW);
    return ntiles < 512 ? ntiles : 512;
}

template <int CIN, int K>
static hipError_t launch_base_wgrad(const float* in, const float* dy, float* partial, float* dw, int B, int H, int W,
                                    float v_min, float v_max, hipStream_t s)
{
    const int tiles_x = (W + BW_TW - 1) / BW_TW, tiles_y = (H + BW_TH - 1) / BW_TH;
    const int ntiles = B * tiles_x * tiles_y, grid = bf_base_wgrad_grid(B, H, W);
    hipLaunchKernelGGL((base_wgrad_kernel<CIN, K>), dim3(grid), dim3(512), 0, s, in, dy, partial, B, H, W, v_min, v_max,
                       tiles_x, tiles_y, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return bf_launch_reduce_partials(partial, grid, K * K * CIN * 16, dw, 1.0f, s);
}

hipError_t bf_launch_base_wgrad(const float* in, const float* dy, float* partial, float* dw, int B, int H, int W,
                                int cin, int k, float v_min, float v_max, hipStream_t s)
{
#define BF_BW(C, KK) if (cin == C && k == KK) return launch_base_wgrad<C, KK>(in, dy, partial, dw, B, H, W, v_min, v_max, s);
    BF_BW(3, 3) BF_BW(3, 5) BF_BW(3, 7) BF_BW(3, 1)
    BF_BW(1, 3) BF_BW(1, 5) BF_BW(1, 7) BF_BW(1, 1)
#undef BF_BW
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// denoiser head (bfcnn/model.py:297-342): 1x1 16->hf (activation) -> 1x1 hf->cout -> tanh(2x)*0.51
// then denormalise (model.py:136-139 / utilities.py:435-443), remove_padding (crop to Ho x Wo,
// utilities.py:755-764), tf.round (half-to-even) and the uint8 cast (module_denoiser.py:71-73).
// With a linear first 1x1 the two matrices are pre-multiplied at pack time (wh[16][4]).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf_act(float x, int act, float alpha)
{
    if (act == BF_ACT_RELU) return fmaxf(x, 0.f);
    if (act == BF_ACT_LEAKY_RELU) return x > 0.f ? x : alpha * x;
    return x;
}

template <bool U8>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a)
{
    __shared__ float w0s[16 * 64];
    __shared__ float w1s[64 * 4];
    __shared__ float whs[64];
    const bool fused = a.wh != nullptr;
    if (fused) {
        for (int i = threadIdx.x; i < 64; i += 256) whs[i] = a.wh[i];
    } else {
        for (int i = threadIdx.x; i < 16 * a.hf; i += 256) w0s[i] = a.w0[i];
        for (int i = threadIdx.x; i < a.hf * a.cout; i += 256) w1s[i] = a.w1[i];
    }
    __syncthreads();
    const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
    for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (int64_t)gridDim.x * 256) {
        const int x = (int)(pix % a.Wo);
        const int64_t t = pix / a.Wo;
        const int y = (int)(t % a.Ho);
        const int b = (int)(t / a.Ho);
        float f[16];
        if (a.feat_split) {
            const int64_t hw = (int64_t)a.H * a.W;
            const char* base = reinterpret_cast<const char*>(a.feat) + (int64_t)b * hw * 64 + ((int64_t)y * a.W + x) * 16;
            const h8 hi0 = *reinterpret_cast<const h8*>(base), hi1 = *reinterpret_cast<const h8*>(base + hw * 16);
            h8 lo0, lo1;
            if (a.feat_split == 2) {
                const char* lob = reinterpret_cast<const char*>(a.feat) + (int64_t)b * hw * 64 + hw * 32 + ((int64_t)y * a.W + x) * 8;
                lo0 = bf_h3c_decode8(*reinterpret_cast<const bf_u2*>(lob));
                lo1 = bf_h3c_decode8(*reinterpret_cast<const bf_u2*>(lob + hw * 8));
            } else {
                lo0 = *reinterpret_cast<const h8*>(base + hw * 32);
                lo1 = *reinterpret_cast<const h8*>(base + hw * 48);
            }
            bool finite = true;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f[c] = (float)hi0[c] + (float)lo0[c];
                f[8 + c] = (float)hi1[c] + (float)lo1[c];
                finite = finite && fabsf(f[c]) <= 3.0e38f && fabsf(f[8 + c]) <= 3.0e38f;      // false for inf and NaN
            }
            // an activation left the f16 range somewhere in the split-f16 blocks (inf / NaN propagate to here)
            if (!finite && a.status) atomicOr(a.status, BF_STATUS_F16_RANGE);
        } else {
            const float4* fp = reinterpret_cast<const float4*>(a.feat + (((int64_t)b * a.H + y) * a.W + x) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = fp[i];
                f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
            }
        }
        float h1[4] = {0.f, 0.f, 0.f, 0.f};
        if (fused) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int o = 0; o < 4; ++o) h1[o] = fmaf(f[c], whs[c * 4 + o], h1[o]);
        } else {
            for (int j = 0; j < a.hf; ++j) {
                float h = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) h = fmaf(f[c], w0s[c * a.hf + j], h);
                h = bf_act(h, a.act, a.leaky_alpha);
                for (int o = 0; o < a.cout; ++o) h1[o] = fmaf(h, w1s[j * a.cout + o], h1[o]);
            }
        }
        for (int o = 0; o < a.cout; ++o) {
            float v = tanhf(2.0f * h1[o]) * 0.51f;
            if (a.denormalize) v = (fminf(fmaxf(v, -0.5f), 0.5f) + 0.5f) * (a.v_max - a.v_min) + a.v_min;
            if (U8) {
                const float r = fminf(fmaxf(rintf(v), 0.f), 255.f);     // rintf = round-half-even
                reinterpret_cast<uint8_t*>(a.out)[pix * a.cout + o] = (uint8_t)r;
            } else {
                reinterpret_cast<float*>(a.out)[pix * a.cout + o] = v;
            }
        }
    }
}

hipError_t bf_launch_head(const HeadArgs& a, hipStream_t s)
{
    const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
    int64_t g = (npix + 255) / 256;
    const int grid = (int)(g < 8192 ? g : 8192);
    if (a.out_is_u8) hipLaunchKernelGGL(head_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else             hipLaunchKernelGGL(head_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// training head: forward + L1 loss (bfcnn/loss.py:40-65,190-247) + backward to the features.
// Linear head only: with h1 = feat.Wh every head weight gradient follows from
// M[c][o] = sum_pix feat[c]*dh1[o]  (dW1 = W0^T M, dW0 = M W1^T), so a thread only carries
// 16*cout accumulators.  One workgroup handles pixels of ONE image (per-image sum of squares
// for the rmse metric, loss.py:92-113).  partial row layout (80 floats):
//   [0,64) M[c*4+o] | 64 sum|e| | 65 sum relu(|e|,hinge,cutoff) | 66 sum relu(e,0,255)^2 | 67 sum relu(e,hinge,cutoff^2)^2 | 68.. unused
// ------------------------------------------------------------------------------------------
constexpr int HT_BLOCKS_PER_IMAGE_MAX = 64;

__global__ __launch_bounds__(256) void head_train_kernel(HeadTrainArgs a, int blocks_per_image)
{
    __shared__ float whs[64];
    __shared__ float red[4][80];
    if (threadIdx.x < 64) whs[threadIdx.x] = a.wh[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.x / blocks_per_image, sub = blockIdx.x % blocks_per_image;
    const int hw = a.H * a.W;
    float M[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) M[i] = 0.f;
    float s_abs = 0.f, s_hinge = 0.f, s_sq = 0.f, s_sqh = 0.f;
    for (int pi = sub * 256 + threadIdx.x; pi < hw; pi += blocks_per_image * 256) {
        const int64_t pix = (int64_t)b * hw + pi;
        const float4* fp = reinterpret_cast<const float4*>(a.feat + pix * 16);
        float f[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 v = fp[i];
            f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
        }
        float dh1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o < a.cout) {
                float h = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) h = fmaf(f[c], whs[c * 4 + o], h);
                const float th = tanhf(2.0f * h);
                const float pv = th * 0.51f;
                float pred = pv, dpred_dp = 1.0f;
                if (a.denormalize) {
                    pred = (fminf(fmaxf(pv, -0.5f), 0.5f) + 0.5f) * (a.v_max - a.v_min) + a.v_min;
                    dpred_dp = (pv >= -0.5f && pv <= 0.5f) ? (a.v_max - a.v_min) : 0.f;
                }
                if (a.pred) a.pred[pix * a.cout + o] = pred;
                const float e = a.gt[pix * a.cout + o] - pred;
                const float ae = fabsf(e);
                s_abs += fminf(ae, 255.0f);                  // mae_actual: hinge 0, cutoff 255
                s_hinge += ae > a.hinge ? fminf(ae, a.cutoff) : 0.f;
                const float ep = e > 0.f ? fminf(e, 255.0f) : 0.f;   // rmse_diff: relu on the signed error
                s_sq += ep * ep;
                const float eh = e > a.hinge ? fminf(e, a.cutoff * a.cutoff) : 0.f;   // the RMSE loss term's own thresholds
                s_sqh += eh * eh;
                float dpred = 0.f;
                if (ae > a.hinge && ae < a.cutoff) dpred = (e > 0.f ? -1.f : (e < 0.f ? 1.f : 0.f)) * a.dscale;
                if (a.dextra) dpred += a.dextra[pix * a.cout + o];
                dh1[o] = dpred * dpred_dp * (0.51f * 2.0f) * (1.0f - th * th);
            }
        }
        float df[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                d = fmaf(dh1[o], whs[c * 4 + o], d);
                M[c * 4 + o] = fmaf(f[c], dh1[o], M[c * 4 + o]);
            }
            df[c] = d * a.dfeat_scale;
        }
        float4* dp = reinterpret_cast<float4*>(a.dfeat + pix * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) dp[i] = make_float4(df[4 * i], df[4 * i + 1], df[4 * i + 2], df[4 * i + 3]);
    }
    // wave reduction then cross-wave through LDS (fixed order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        float v = M[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][i] = v;
    }
    float sv[4] = {s_abs, s_hinge, s_sq, s_sqh};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = sv[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][64 + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 80) {
        const int i = threadIdx.x;
        a.partial[(size_t)blockIdx.x * 80 + i] = i < 68 ? (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]) : 0.f;
    }
}

int bf_head_train_grid(int B, int H, int W)
{
    int bpi = (H * W + 1023) / 1024;
    if (bpi > HT_BLOCKS_PER_IMAGE_MAX) bpi = HT_BLOCKS_PER_IMAGE_MAX;
    if (bpi < 1) bpi = 1;
    return B * bpi;
}

hipError_t bf_launch_head_train(const HeadTrainArgs& a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(head_train_kernel, dim3(grid), dim3(256), 0, s, a, grid / a.B);
    return hipGetLastError();
}
