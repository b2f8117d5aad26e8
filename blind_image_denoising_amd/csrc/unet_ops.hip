// Operators of the `unet_laplacian` backbone (bfcnn/backbone_unet_laplacian.py, bfcnn/custom_layers.py ConvNextBlock /
// ConvolutionalSelfAttention / ChannelLearnableMultiplier / GaussianFilter, bfcnn/upsampling.py, bfcnn/downsampling.py,
// bfcnn/model.py denoiser heads) for gfx950.  fp32 NHWC, inference.
//
//   * the 1x1 convolutions (92 % of the FLOPs) run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32
//     products, fp32 accumulate): M = 16 output channels, N = 16 pixels, K = 4 input channels per instruction.  A lane
//     fetches 4 consecutive channels of its pixel with one 16-byte load; the K order inside a 16-channel chunk is
//     permuted (k = 4q + j in step j) and the weights are pre-packed in the same order, so neither operand is shuffled.
//   * the ConvNext MLP (1x1 C->4C, activation, 1x1 4C->C, channel multiplier, + skip) is ONE kernel: the accumulator
//     layout of the first GEMM (lane (q, n), register r <-> channel 16t + 4q + r of pixel n) is a legal B operand of the
//     second with the weights packed to match, so the 4C-wide intermediate never leaves the registers.
//   * depthwise k x k + LayerNorm, LayerNorm + activation, the Laplacian split, resize and the heads are HBM-bound
//     vector kernels: C/4 lanes per pixel, 16-byte accesses, channel reductions with DPP shuffles inside a wave.
#include "bf_common.h"
#include "unet_h3_core.h"
#include <math.h>

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define UO_RING 4     // weight tiles requested ahead of their use

static int uo_grid(int64_t n, int per_block, int cap = 256 * 32)
{
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// activation codes of the C ABI (BF_ACT_*): 0 linear, 1 relu, 2 leaky relu (alpha), 3 gelu (erf form), 4 tanh
template <int ACT>
__device__ __forceinline__ float uo_act(float v, float alpha)
{
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return v > 0.f ? v : alpha * v;
    if (ACT == 3) return bf_gelu(v);
    if (ACT == 4) return tanhf(v);
    return v;
}
__device__ __forceinline__ float uo_act_rt(float v, int act, float alpha)
{
    switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : alpha * v;
    case 3: return bf_gelu(v);
    case 4: return tanhf(v);
    default: return v;
    }
}

// ------------------------------------------------------------------------------------------
// weight packing for the MFMA GEMMs: w [cin][cout] (HWIO of a 1x1 kernel) -> [cin/16][cout/16][64 lanes][4]
// element (c, t, lane = 16q + m, j) = w[16c + 4q + j][16t + m]
// ------------------------------------------------------------------------------------------
__global__ void uo_pack_pointwise_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int cout)
{
    const int n = cin * cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int j = i & 3, lane = (i >> 2) & 63;
        const int ct = i >> 8;
        const int T = cout >> 4;
        const int t = ct % T, c = ct / T;
        const int q = lane >> 4, m = lane & 15;
        wp[i] = w[(16 * c + 4 * q + j) * cout + 16 * t + m];
    }
}

extern "C" int bf_op_pack_pointwise(const float* w, float* wp, int cin, int cout, void* stream)
{
    if (!w || !wp || cin <= 0 || cout <= 0 || cin % 16 || cout % 16) return BF_EINVAL;
    hipLaunchKernelGGL(uo_pack_pointwise_kernel, dim3(uo_grid((int64_t)cin * cout, 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                       cin, cout);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// 1x1 convolution: out[p][co] = res[p][co] + mult[co] * act(sum_ci in[p][ci] w[ci][co])
// one wave = NP groups of 16 pixels, all output channels; 4 waves per workgroup
// ------------------------------------------------------------------------------------------
// mode 0: out = res + mult * act(acc)                      (1x1 convolution, activation, channel multiplier, residual Add)
// mode 1: out = act(acc + res)                             (AdditiveAttentionGate: leaky_relu(conv_x(x) + conv_y(y)), custom_layers.py:823)
// mode 2: out = res * sigmoid(4 * mult * acc) + add        (the gate applied to the encoder feature, :824-832, + the decoder Add)
// mode 3: out = act(acc + mult) + res                      (convolution with a folded BatchNorm: mult = per-channel shift)
template <int CIN, int COUT, int NP, int ACT>
__global__ __launch_bounds__(256, 2) void uo_pointwise_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const float* __restrict__ wp, const float* __restrict__ mult,
                                                           const float* __restrict__ res, int64_t npix, float alpha, int mode,
                                                           const float* __restrict__ add)
{
    constexpr int KC = CIN / 16, T = COUT / 16;
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        // opaque per iteration: the weights do not depend on g, and hipcc would otherwise hoist ALL their loads out of
        // this loop and hold every tile in registers (spills)
        const float* wpo = wp;
        asm volatile("" : "+s"(wpo));
        const f32x4* wv = reinterpret_cast<const f32x4*>(wpo) + lane;
        const float* src[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = p0 + 16 * i + n;
            p = p < npix ? p : npix - 1;                          // tail: clamp the read, predicate the store
            src[i] = in + p * CIN + 4 * q;
        }
        f32x4 acc[T][NP];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // mode 0 with a residual (the attention block's output projection 32 -> 128 reads four times the bytes it multiplies): the
        // residual is requested BEFORE the matrix work instead of tile by tile in the epilogue (each of those loads waited a full
        // memory round trip with two workgroups per CU: 188 us for 604 MB).  Where the tiles fit the registers (T * NP <= 16).
        constexpr bool PRERES = T * NP <= 16;
        f32x4 rs[PRERES ? T : 1][PRERES ? NP : 1];
        const bool pre_res = PRERES && mode == 0 && res != nullptr;
        if (pre_res) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int64_t p = p0 + 16 * i + n;
                p = p < npix ? p : npix - 1;
#pragma unroll
                for (int t = 0; t < T; ++t) rs[PRERES ? t : 0][PRERES ? i : 0] = *reinterpret_cast<const f32x4*>(res + p * COUT + 16 * t + 4 * q);
            }
        }
        // weight tiles stream through a ring of D registers (requested D steps ahead), the pixel chunks through a
        // double buffer; a scheduling barrier per step keeps hipcc from hoisting every load to the top (it spills)
        constexpr int S = KC * T, D = UO_RING;
        f32x4 ring[D], bcur[NP], bnext[NP];
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < S) ring[d] = wv[d * 64];
#pragma unroll
        for (int i = 0; i < NP; ++i) bcur[i] = *reinterpret_cast<const f32x4*>(src[i]);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            if (c + 1 < KC) {
#pragma unroll
                for (int i = 0; i < NP; ++i) bnext[i] = *reinterpret_cast<const f32x4*>(src[i] + 16 * (c + 1));
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int st = c * T + t;
                const f32x4 a = ring[st % D];
                if (st + D < S) ring[st % D] = wv[(st + D) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[t][i] = MFMA4(a[j], bcur[i][j], acc[t][i]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) bcur[i] = bnext[i];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = p0 + 16 * i + n;
            if (p >= npix) continue;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = bf_acc_ready(acc[t][i]);
                const int co = 16 * t + 4 * q;
                if (mode == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = uo_act<ACT>(v[r], alpha);
                    if (mult) v *= *reinterpret_cast<const f32x4*>(mult + co);
                    if (pre_res) v += rs[PRERES ? t : 0][PRERES ? i : 0];
                    else if (res) v += *reinterpret_cast<const f32x4*>(res + p * COUT + co);
                } else if (mode == 1) {
                    if (res) v += *reinterpret_cast<const f32x4*>(res + p * COUT + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = uo_act<ACT>(v[r], alpha);
                } else if (mode == 3) {
                    if (mult) v += *reinterpret_cast<const f32x4*>(mult + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = uo_act<ACT>(v[r], alpha);
                    if (res) v += *reinterpret_cast<const f32x4*>(res + p * COUT + co);
                } else {
                    if (mult) v *= *reinterpret_cast<const f32x4*>(mult + co);
                    const f32x4 e = *reinterpret_cast<const f32x4*>(res + p * COUT + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = e[r] / (1.f + __expf(-4.f * v[r]));
                    if (add) v += *reinterpret_cast<const f32x4*>(add + p * COUT + co);
                }
                *reinterpret_cast<f32x4*>(out + p * COUT + co) = v;
            }
        }
    }
}

template <int CIN, int COUT, int NP>
static hipError_t uo_launch_pointwise(const float* in, float* out, const float* wp, const float* mult, const float* res, int64_t npix,
                                      int act, float alpha, int mode, const float* add, hipStream_t s)
{
    const int grid = uo_grid(npix, 4 * 16 * NP, 256 * 8);
#define UO_PW(A) hipLaunchKernelGGL((uo_pointwise_kernel<CIN, COUT, NP, A>), dim3(grid), dim3(256), 0, s, in, out, wp, mult, res, npix, alpha, mode, add)
    switch (act) {
    case 0: UO_PW(0); break;
    case 1: UO_PW(1); break;
    case 2: UO_PW(2); break;
    case 3: UO_PW(3); break;
    default: return hipErrorInvalidValue;
    }
#undef UO_PW
    return hipGetLastError();
}

static int uo_pointwise_dispatch(const float* in, float* out, const float* wp, const float* mult, const float* res, int64_t npix,
                                 int cin, int cout, int act, float alpha, int mode, const float* add, hipStream_t s)
{
    hipError_t e = hipErrorInvalidValue;
#define UO_CASE(CI, CO, NP) if (cin == CI && cout == CO) e = uo_launch_pointwise<CI, CO, NP>(in, out, wp, mult, res, npix, act, alpha, mode, add, s)
    UO_CASE(32, 32, 4); UO_CASE(32, 64, 2); UO_CASE(32, 128, 2); UO_CASE(64, 32, 2); UO_CASE(64, 64, 4); UO_CASE(64, 128, 4);
    UO_CASE(128, 32, 4); UO_CASE(128, 64, 4); UO_CASE(128, 128, 4);
    UO_CASE(128, 96, 2);                                   // query | key | value of one attention block in one pass
    UO_CASE(128, 256, 2); UO_CASE(256, 128, 4); UO_CASE(256, 32, 4); UO_CASE(32, 256, 2);      // 4-level models (256 channels)
    UO_CASE(64, 256, 2); UO_CASE(256, 64, 4);              // the 64-channel ConvNext MLP as two convolutions (training: unet_train.py)
#undef UO_CASE
    if (e == hipErrorInvalidValue) return BF_EUNSUPPORTED;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_pointwise(const float* in, float* out, const float* wp, const float* mult, const float* res, int64_t npix,
                               int cin, int cout, int act, float alpha, void* stream)
{
    if (!in || !out || !wp || npix <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)wp | (uintptr_t)mult | (uintptr_t)res) % 16) return BF_EINVAL;
    return uo_pointwise_dispatch(in, out, wp, mult, res, npix, cin, cout, act, alpha, 0, nullptr, (hipStream_t)stream);
}

extern "C" int bf_op_pointwise_ex(const float* in, float* out, const float* wp, const float* mult, const float* res, const float* add,
                                  int64_t npix, int cin, int cout, int act, float alpha, int mode, void* stream)
{
    if (!in || !out || !wp || npix <= 0 || mode < 0 || mode > 3 || (mode == 2 && !res)) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)wp | (uintptr_t)mult | (uintptr_t)res | (uintptr_t)add) % 16) return BF_EINVAL;
    return uo_pointwise_dispatch(in, out, wp, mult, res, npix, cin, cout, act, alpha, mode, add, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Conv2D kh x kw, stride s, padding="same" (TF pad split: extra at the bottom / right), use_bias=False, Cin -> Cout with
// Cin, Cout in {32, 64, 128}: implicit GEMM over the taps with the 1x1 machinery above (the 2x2 stride-2 "conv2d"
// downsample, downsampling.py:45-55; the 3x3 convolutions behind upsample_bilinear_conv2d / upsample_nearest_conv2d,
// upsampling.py:52-72).  wp = the taps' [Cin][Cout] matrices, each packed by bf_op_pack_pointwise, tap-major.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int NP, int ACT>
__global__ __launch_bounds__(256, 2) void uo_conv2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const float* __restrict__ wp, const float* __restrict__ res,
                                                        const float* __restrict__ bias, int B, int H, int W, int OH, int OW, int kh,
                                                        int kw, int stride, int pt, int pl, float alpha)
{
    constexpr int KC = CIN / 16, T = COUT / 16;
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t npix = (int64_t)B * OH * OW;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        int oy[NP], ox[NP];
        int64_t ib[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = p0 + 16 * i + n;
            p = p < npix ? p : npix - 1;
            ox[i] = (int)(p % OW);
            oy[i] = (int)((p / OW) % OH);
            ib[i] = (p / ((int64_t)OW * OH)) * H * W;
        }
        f32x4 acc[T][NP];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int tap = 0; tap < kh * kw; ++tap) {
            const int ky = tap / kw, kx = tap - ky * kw;
            const float* wpo = wp + (int64_t)tap * CIN * COUT;
            asm volatile("" : "+s"(wpo));                          // see uo_pointwise_kernel
            const f32x4* wv = reinterpret_cast<const f32x4*>(wpo) + lane;
            const float* src[NP];
            float ok[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int yy = oy[i] * stride + ky - pt, xx = ox[i] * stride + kx - pl;
                ok[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.f : 0.f;
                src[i] = in + (ib[i] + (int64_t)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * CIN + 4 * q;
            }
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                f32x4 b[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) b[i] = *reinterpret_cast<const f32x4*>(src[i] + 16 * c) * ok[i];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const f32x4 a = wv[(c * T + t) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[t][i] = MFMA4(a[j], b[i][j], acc[t][i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = p0 + 16 * i + n;
            if (p >= npix) continue;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = bf_acc_ready(acc[t][i]);
                if (bias) v += *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = uo_act<ACT>(v[r], alpha);
                if (res) v += *reinterpret_cast<const f32x4*>(res + p * COUT + 16 * t + 4 * q);
                *reinterpret_cast<f32x4*>(out + p * COUT + 16 * t + 4 * q) = v;
            }
        }
    }
}

extern "C" int bf_op_conv2d(const float* in, float* out, const float* wp, const float* res, const float* bias, int B, int H, int W,
                            int cin, int cout, int kh, int kw, int stride, int act, float alpha, void* stream)
{
    if (!in || !out || !wp || B <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)wp | (uintptr_t)res | (uintptr_t)bias) % 16) return BF_EINVAL;
    if (act < 0 || act > 3) return BF_EUNSUPPORTED;
    const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
    const int th = max((OH - 1) * stride + kh - H, 0), tw = max((OW - 1) * stride + kw - W, 0);
    const int pt = th / 2, pl = tw / 2;
    const int64_t npix = (int64_t)B * OH * OW;
    hipStream_t s = (hipStream_t)stream;
    bool ok = false;
#define UO_CV_A(CI, CO, NPP, A)                                                                                                \
    hipLaunchKernelGGL((uo_conv2d_kernel<CI, CO, NPP, A>), dim3(uo_grid(npix, 4 * 16 * NPP, 256 * 8)), dim3(256), 0, s, in, out, wp, res, bias, B,  \
                       H, W, OH, OW, kh, kw, stride, pt, pl, alpha)
#define UO_CV(CI, CO, NPP)                                                                                                     \
    if (cin == CI && cout == CO) {                                                                                             \
        ok = true;                                                                                                             \
        switch (act) {                                                                                                         \
        case 0: UO_CV_A(CI, CO, NPP, 0); break;                                                                                \
        case 1: UO_CV_A(CI, CO, NPP, 1); break;                                                                                \
        case 2: UO_CV_A(CI, CO, NPP, 2); break;                                                                                \
        default: UO_CV_A(CI, CO, NPP, 3); break;                                                                               \
        }                                                                                                                      \
    }
    UO_CV(32, 32, 4) UO_CV(32, 64, 4) UO_CV(64, 32, 4) UO_CV(64, 64, 4) UO_CV(64, 128, 2) UO_CV(128, 64, 4) UO_CV(128, 128, 2) UO_CV(256, 128, 2)
    UO_CV(128, 256, 1)                     // conv2d down-sampling into the 256-channel attention level of the four-level models
#undef UO_CV
#undef UO_CV_A
    if (!ok) return BF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// DepthwiseConv2D k x k with depth_multiplier m (output channel c * m + j, keras order), SAME zero padding, + per-channel
// bias (a folded BatchNorm shift) + activation: the depthwise middle convolution of the shipped resnet config
// (backbone_resnet.py:165-176; block_depthwise).  w [k][k][C][m]; thread = 4 consecutive OUTPUT channels of one pixel.
__global__ __launch_bounds__(256) void uo_dwconv_mult_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const float* __restrict__ w, const float* __restrict__ bias, int B, int H,
                                                            int W, int C, int m, int k, int act, float alpha)
{
    const int CO = C * m, Cv = CO / 4, R = (k - 1) / 2;
    const int64_t n = (int64_t)B * H * W * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % Cv) * 4;
        int64_t t = i / Cv;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int64_t b = t / H;
        f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + co) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y + ky - R;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x + kx - R;
                if (xx < 0 || xx >= W) continue;
                const float* px = in + ((b * H + yy) * W + xx) * C;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (ky * k + kx) * CO + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += wv[j] * px[(co + j) / m];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = uo_act_rt(acc[j], act, alpha);
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
}

extern "C" int bf_op_dwconv_mult(const float* in, float* out, const float* w, const float* bias, int B, int H, int W, int C, int m,
                                 int k, int act, float alpha, void* stream)
{
    if (!in || !out || !w || B <= 0 || H <= 0 || W <= 0 || C <= 0 || m <= 0 || k <= 0 || (C * m) % 4) return BF_EINVAL;
    if (((uintptr_t)out | (uintptr_t)w | (uintptr_t)bias) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * H * W * (C * m / 4);
    hipLaunchKernelGGL(uo_dwconv_mult_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, w, bias, B, H, W, C, m,
                       k, act, alpha);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// Depthwise k x k with depth multiplier M (+ bias1, act1) followed by a 1x1 convolution HID = CIN * M -> COUT (+ bias2, act2,
// + res) in ONE kernel: the bottleneck tail of the shipped resnet config (depthwise 3x3 x4 -> BN -> ReLU -> grouped 1x1 ->
// BN -> Add, backbone_resnet.py:149-178; both BatchNorms folded, the grouped kernel packed block-diagonal).  The HID-wide
// tensor (4x the block's traffic) never reaches memory: a workgroup stages the CIN-channel input tile of 8 x 32 pixels
// (+ halo) and the depthwise weights in LDS; lane (q, n) of a wave computes hidden channels 16c + 4q + j of pixel n --
// exactly the B operand of step j of chunk c of the fp32 16x16x4 MFMA GEMM (weights packed by bf_op_pack_pointwise).
constexpr int UO_DP_TH = 8, UO_DP_TW = 32;
template <int CIN, int M, int COUT, int K>
__global__ __launch_bounds__(256) void uo_dwmult_pw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ wd, const float* __restrict__ bias1, int act1,
                                                          float alpha1, const float* __restrict__ wp, const float* __restrict__ bias2,
                                                          int act2, float alpha2, const float* __restrict__ res, int H, int W)
{
    constexpr int HID = CIN * M, KC = HID / 16, T = COUT / 16, RAD = (K - 1) / 2, IH = UO_DP_TH + K - 1, IW = UO_DP_TW + K - 1;
    static_assert(M == 1 || M == 2 || M == 4, "depth multiplier");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds;                                   // [IH][IW][CIN]
    float* wl = lds + IH * IW * CIN;                     // [K*K][CIN][M]
    const int x0 = blockIdx.x * UO_DP_TW, y0 = blockIdx.y * UO_DP_TH;
    const int64_t img = (int64_t)blockIdx.z * H * W;
    {
        constexpr int NE = (IH * IW * (CIN / 4) + 255) / 256;      // loads first, stores after (see uo_first_conv_tile_kernel)
        f32x4 rv[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            const int c4 = e % (CIN / 4), px = (e / (CIN / 4)) % IW, py = e / ((CIN / 4) * IW);
            const int yy = y0 + py - RAD, xx = x0 + px - RAD;
            rv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < IH * IW * (CIN / 4) && yy >= 0 && yy < H && xx >= 0 && xx < W)
                rv[i] = *reinterpret_cast<const f32x4*>(in + (img + (int64_t)yy * W + xx) * CIN + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            if (e < IH * IW * (CIN / 4)) reinterpret_cast<f32x4*>(tile)[e] = rv[i];
        }
    }
    for (int e = threadIdx.x; e < K * K * HID; e += 256) wl[e] = wd[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
    const f32x4* wv = reinterpret_cast<const f32x4*>(wp) + lane;
    for (int gi = wave; gi < UO_DP_TH * (UO_DP_TW / 16); gi += 4) {
        const int ry = gi / (UO_DP_TW / 16), cx = (gi % (UO_DP_TW / 16)) * 16 + n;
        const float* base = tile + (ry * IW + cx) * CIN;
        f32x4 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int h0 = 16 * c + 4 * q;               // the lane's 4 hidden channels of this chunk
            f32x4 hv = bias1 ? *reinterpret_cast<const f32x4*>(bias1 + h0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* px = base + (ky * IW + kx) * CIN;
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + (ky * K + kx) * HID + h0);   // [tap][ic][mi] == [tap][h]
                    if (M == 4) {
                        hv += w4 * px[h0 / 4];
                    } else if (M == 2) {
                        const float a = px[h0 / 2], b2 = px[h0 / 2 + 1];
                        hv += w4 * (f32x4){a, a, b2, b2};
                    } else {
                        hv += w4 * *reinterpret_cast<const f32x4*>(px + h0);
                    }
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = uo_act_rt(hv[j], act1, alpha1);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 a = wv[(c * T + t) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = MFMA4(a[j], hv[j], acc[t]);
            }
        }
        const int gy = y0 + ry, gx = x0 + cx;
        if (gy < H && gx < W) {
            const int64_t p = img + (int64_t)gy * W + gx;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = bf_acc_ready(acc[t]);
                const int co = 16 * t + 4 * q;
                if (bias2) v += *reinterpret_cast<const f32x4*>(bias2 + co);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = uo_act_rt(v[r], act2, alpha2);
                if (res) v += *reinterpret_cast<const f32x4*>(res + p * COUT + co);
                *reinterpret_cast<f32x4*>(out + p * COUT + co) = v;
            }
        }
    }
}

extern "C" int bf_op_dwmult_pointwise(const float* in, float* out, const float* wd, const float* bias1, int act1, float alpha1,
                                      const float* wp, const float* bias2, int act2, float alpha2, const float* res, int B, int H,
                                      int W, int cin, int m, int k, int cout, void* stream)
{
    if (!in || !out || !wd || !wp || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)wp | (uintptr_t)bias1 | (uintptr_t)bias2 | (uintptr_t)res) % 16) return BF_EINVAL;
    if (B > 65535 || (H + UO_DP_TH - 1) / UO_DP_TH > 65535) return BF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((W + UO_DP_TW - 1) / UO_DP_TW, (H + UO_DP_TH - 1) / UO_DP_TH, B);
    bool ok = false;
#define UO_DP(CI, MM, CO, KK)                                                                                                  \
    if (cin == CI && m == MM && cout == CO && k == KK) {                                                                       \
        constexpr int LDSB = ((UO_DP_TH + KK - 1) * (UO_DP_TW + KK - 1) * CI + KK * KK * CI * MM) * 4;                          \
        if (bf_set_max_lds(reinterpret_cast<const void*>(uo_dwmult_pw_kernel<CI, MM, CO, KK>), LDSB) != hipSuccess) return BF_EHIP;      \
        hipLaunchKernelGGL((uo_dwmult_pw_kernel<CI, MM, CO, KK>), grid, dim3(256), LDSB, s, in, out, wd, bias1, act1, alpha1, wp,  \
                           bias2, act2, alpha2, res, H, W);                                                                    \
        ok = true;                                                                                                             \
    }
    UO_DP(32, 4, 32, 3) UO_DP(32, 2, 32, 3) UO_DP(64, 2, 64, 3) UO_DP(32, 4, 64, 3) UO_DP(32, 1, 32, 3) UO_DP(64, 1, 64, 3)
#undef UO_DP
    if (!ok) return BF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// MaxPooling2D(pool 2x2, strides 2, padding="same") (downsampling.py:56-58): out-of-image taps are ignored
__global__ __launch_bounds__(256) void uo_maxpool2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C)
{
    const int Cv = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const f32x4* img = reinterpret_cast<const f32x4*>(in) + (int64_t)b * H * W * Cv + c;
        const int y1 = min(2 * oy + 1, H - 1), x1 = min(2 * ox + 1, W - 1);      // a clamped tap repeats an in-image one
        const f32x4 v00 = img[((int64_t)2 * oy * W + 2 * ox) * Cv], v01 = img[((int64_t)2 * oy * W + x1) * Cv];
        const f32x4 v10 = img[((int64_t)y1 * W + 2 * ox) * Cv], v11 = img[((int64_t)y1 * W + x1) * Cv];
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = fmaxf(fmaxf(v00[j], v01[j]), fmaxf(v10[j], v11[j]));
        reinterpret_cast<f32x4*>(out)[i] = r;
    }
}

extern "C" int bf_op_maxpool2(const float* in, float* out, int B, int H, int W, int C, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(uo_maxpool2_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// ConvNext MLP in one kernel: out = skip + mult * (act(in . w1) . w2)     in: [npix][C] (LayerNorm output), w1 [C][4C],
// w2 [4C][C] (both packed by bf_op_pack_pointwise)
// ------------------------------------------------------------------------------------------
template <int C, int NP, int ACT>
__global__ __launch_bounds__(256, 2) void uo_convnext_mlp_kernel(const float* __restrict__ in, const float* __restrict__ skip,
                                                              float* __restrict__ out, const float* __restrict__ w1p,
                                                              const float* __restrict__ w2p, const float* __restrict__ mult,
                                                              int64_t npix, float alpha)
{
    constexpr int KC = C / 16, T1 = 4 * C / 16, T2 = C / 16;
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        const float *w1o = w1p, *w2o = w2p;                       // opaque per iteration (see uo_pointwise_kernel)
        asm volatile("" : "+s"(w1o), "+s"(w2o));
        const f32x4* w1v = reinterpret_cast<const f32x4*>(w1o) + lane;
        const f32x4* w2v = reinterpret_cast<const f32x4*>(w2o) + lane;
        const float* src[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = p0 + 16 * i + n;
            p = p < npix ? p : npix - 1;
            src[i] = in + p * C + 4 * q;
        }
        f32x4 h[T1][NP];
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) h[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int D = UO_RING;
        {
            constexpr int S = KC * T1;
            f32x4 ring[D], bcur[NP], bnext[NP];
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (d < S) ring[d] = w1v[d * 64];
#pragma unroll
            for (int i = 0; i < NP; ++i) bcur[i] = *reinterpret_cast<const f32x4*>(src[i]);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) bnext[i] = *reinterpret_cast<const f32x4*>(src[i] + 16 * (c + 1));
                }
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    const int st = c * T1 + t;
                    const f32x4 a = ring[st % D];
                    if (st + D < S) ring[st % D] = w1v[(st + D) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < NP; ++i) h[t][i] = MFMA4(a[j], bcur[i][j], h[t][i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < NP; ++i) bcur[i] = bnext[i];
            }
        }
        f32x4 acc[T2][NP];
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // second GEMM: chunk c of K = hidden tile c; its register r of lane (q, n) is hidden channel 16c + 4q + r
        {
            constexpr int S = T1 * T2;
            f32x4 ring[D];
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (d < S) ring[d] = w2v[d * 64];
#pragma unroll
            for (int c = 0; c < T1; ++c) {
                f32x4 b[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    b[i] = bf_acc_ready(h[c][i]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[i][r] = uo_act<ACT>(b[i][r], alpha);
                }
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    const int st = c * T2 + t;
                    const f32x4 a = ring[st % D];
                    if (st + D < S) ring[st % D] = w2v[(st + D) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[t][i] = MFMA4(a[j], b[i][j], acc[t][i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = p0 + 16 * i + n;
            if (p >= npix) continue;
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                f32x4 v = bf_acc_ready(acc[t][i]);
                const int co = 16 * t + 4 * q;
                if (mult) v *= *reinterpret_cast<const f32x4*>(mult + co);
                if (skip) v += *reinterpret_cast<const f32x4*>(skip + p * C + co);
                *reinterpret_cast<f32x4*>(out + p * C + co) = v;
            }
        }
    }
}

template <int C, int NP>
static hipError_t uo_launch_mlp(const float* in, const float* skip, float* out, const float* w1p, const float* w2p, const float* mult,
                                int64_t npix, int act, float alpha, hipStream_t s)
{
    const int grid = uo_grid(npix, 4 * 16 * NP, 256 * 8);
#define UO_MLP(A) hipLaunchKernelGGL((uo_convnext_mlp_kernel<C, NP, A>), dim3(grid), dim3(256), 0, s, in, skip, out, w1p, w2p, mult, npix, alpha)
    switch (act) {
    case 0: UO_MLP(0); break;
    case 1: UO_MLP(1); break;
    case 2: UO_MLP(2); break;
    case 3: UO_MLP(3); break;
    default: return hipErrorInvalidValue;
    }
#undef UO_MLP
    return hipGetLastError();
}

extern "C" int bf_op_convnext_mlp(const float* in, const float* skip, float* out, const float* w1p, const float* w2p,
                                  const float* mult, int64_t npix, int C, int act, float alpha, void* stream)
{
    if (!in || !out || !w1p || !w2p || npix <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)w1p | (uintptr_t)w2p | (uintptr_t)mult | (uintptr_t)skip) % 16) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (C == 32) e = uo_launch_mlp<32, 4>(in, skip, out, w1p, w2p, mult, npix, act, alpha, s);
    else if (C == 64) e = uo_launch_mlp<64, 2>(in, skip, out, w1p, w2p, mult, npix, act, alpha, s);
    else if (C == 128) e = uo_launch_mlp<128, 1>(in, skip, out, w1p, w2p, mult, npix, act, alpha, s);
    else return BF_EUNSUPPORTED;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// depthwise k x k (zero SAME padding) -> [LayerNorm(center=False) * gamma] -> [activation]
// thread = 4 channels of one pixel, C/4 consecutive lanes = one pixel; w [k][k][C] or NULL (k = 0: no convolution)
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float uo_dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
// sum over the LPP = C / 4 consecutive lanes of a pixel, result in all of them: butterfly on DPP (quad_perm xor 1, xor 2;
// then the half-row / row mirrors, which act as xor 4 / xor 8 once the lower lanes agree); 32 lanes: one cross-row shuffle
template <int LPP>
__device__ __forceinline__ float uo_pixel_sum(float v)
{
    v = uo_dpp_add<0xB1>(v);                      // quad_perm [1,0,3,2]
    v = uo_dpp_add<0x4E>(v);                      // quad_perm [2,3,0,1]
    if (LPP >= 8) v = uo_dpp_add<0x141>(v);       // row_half_mirror
    if (LPP >= 16) v = uo_dpp_add<0x140>(v);      // row_mirror
    if (LPP >= 32) v += __shfl_xor(v, 16, 64);
    if (LPP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <int C, int K>
__global__ __launch_bounds__(256) void uo_dwconv_ln_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           const float* __restrict__ w, const float* __restrict__ gamma, int B, int H,
                                                           int W, float eps, int act, float alpha)
{
    constexpr int LPP = C / 4, PPB = 256 / LPP;        // pixels per workgroup pass
    const int cl = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const int c0 = 4 * cl;
    f32x4 wk[K * K > 0 ? K * K : 1];
    if (K > 0) {
#pragma unroll
        for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(w + i * C + c0);
    }
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + c0);
    const int64_t npix = (int64_t)B * H * W;
    const int64_t nsteps = (npix + PPB - 1) / PPB;
    for (int64_t st = blockIdx.x; st < nsteps; st += gridDim.x) {
        const int64_t p = st * PPB + pl;
        const bool live = p < npix;
        const int64_t pc = live ? p : npix - 1;
        const int x = (int)(pc % W);
        const int y = (int)((pc / W) % H);
        const float* img = in + (pc - (int64_t)y * W - x) * C + c0;      // image base + channel
        f32x4 v;
        if (K > 0) {
            v = (f32x4){0.f, 0.f, 0.f, 0.f};
            constexpr int R = K / 2;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = y + ky - R;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int xx = x + kx - R;
                    if (xx < 0 || xx >= W) continue;
                    v += wk[ky * K + kx] * *reinterpret_cast<const f32x4*>(img + ((int64_t)yy * W + xx) * C);
                }
            }
        } else {
            v = *reinterpret_cast<const f32x4*>(img + ((int64_t)y * W + x) * C);
        }
        if (gamma) {
            const float mean = uo_pixel_sum<LPP>(v[0] + v[1] + v[2] + v[3]) * (1.f / C);
            const f32x4 d = v - mean;
            const float var = uo_pixel_sum<LPP>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
            v = d * (gm * rsqrtf(var + eps));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = uo_act_rt(v[r], act, alpha);
        if (live) *reinterpret_cast<f32x4*>(out + p * C + c0) = v;
    }
}

// Column-walking form for k >= 3: a thread owns 4 channels of one image column and walks down R output rows; every input
// row it loads (k 16-byte loads) is multiplied into the k output rows it touches, which are held as k rotating
// accumulators: k (R + k - 1) / R loads per output instead of k^2 (5x5, R = 16: 6.25 instead of 25; 3.4 ms -> see DESIGN).
#ifndef UO_DWR_PD
#define UO_DWR_PD 2
#endif
template <int C, int K, int R>
__global__ __launch_bounds__(256) void uo_dwconv_ln_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                const float* __restrict__ w, const float* __restrict__ gamma, int H,
                                                                int W, float eps, int act, float alpha)
{
    constexpr int LPP = C / 4, PPB = 256 / LPP, RAD = K / 2;
    const int cl = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const int c0 = 4 * cl;
    const int x = blockIdx.x * PPB + pl;
    const bool xlive = x < W;
    const int y0 = blockIdx.y * R;
    const int64_t img = (int64_t)blockIdx.z * H * W;
    f32x4 wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(w + i * C + c0);
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + c0);
    // column offsets of the k taps (clamped; out-of-image taps are zeroed by the mask)
    int xo[K];
    float xm[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int xx = x + kx - RAD;
        xm[kx] = (xx >= 0 && xx < W) ? 1.f : 0.f;
        xo[kx] = min(max(xx, 0), W - 1) * C + c0;
    }
    f32x4 acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int yend = min(y0 + R, H);
    // the k loads of input row yi + 1 are in flight while row yi is multiplied (rows outside the image load row 0 / H-1
    // and are masked: the loop body has no divergent memory path)
    auto load_row = [&](int yi, f32x4 (&v)[K]) {
        const float* row = in + (img + (int64_t)min(max(yi, 0), H - 1) * W) * C;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(row + xo[kx]);
    };
    // The k loads of an input row are requested PD rows ahead of the row that is multiplied (round 4; one row ahead before: a row step
    // is ~160 vector instructions, a memory round trip is several times that, and the 100 weight registers leave two waves per SIMD to
    // cover it -- the kernel ran at the latency, 358 us for 1.07 GB at 64 channels).  Rows outside the image load row 0 / H-1 and are
    // masked, the requests are unconditional: the loop body has no divergent memory path.
    constexpr int PD = UO_DWR_PD;
    f32x4 ring[PD][K];
#pragma unroll
    for (int d = 0; d < PD; ++d) load_row(y0 - RAD + d, ring[d]);
    const int nrows = yend + RAD - (y0 - RAD);
    for (int base = 0; base < nrows; base += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int yi = y0 - RAD + base + d;
            f32x4 v[K];
            const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) v[kx] = ring[d][kx] * (xm[kx] * ym);
            load_row(yi + PD, ring[d]);
            // input row yi is tap row ky of output row yi - ky + RAD = accumulator ky
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[ky] += wk[ky * K + kx] * v[kx];
            const int yo = yi - RAD;                       // acc[K-1] is complete
            if (yo >= y0 && yo < yend) {                   // (the last pass of PD rows may run past the strip)
                f32x4 r = acc[K - 1];
                if (gamma) {
                    const float mean = uo_pixel_sum<LPP>(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                    const f32x4 dd = r - mean;
                    const float var = uo_pixel_sum<LPP>(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2] + dd[3] * dd[3]) * (1.f / C);
                    r = dd * (gm * rsqrtf(var + eps));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = uo_act_rt(r[j], act, alpha);
                if (xlive) *reinterpret_cast<f32x4*>(out + ((img + (int64_t)yo * W) + x) * C + c0) = r;
            }
#pragma unroll
            for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}

extern "C" int bf_op_dwconv_ln(const float* in, float* out, const float* w, const float* ln_gamma, int B, int H, int W, int C, int k,
                               float eps, int act, float alpha, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || (k > 0 && !w)) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)w | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)B * H * W;
    bool ok = false;
    constexpr int RROWS = 32;          // rows per workgroup: k - 1 halo rows are re-read and re-multiplied per strip
                                       // (64 channels, 5x5, 32 x 256 x 256: 16 rows 376 us, 32 rows 350 us)
    // small inputs: 8-row strips put four times the workgroups on the chip (one 512 x 512 image, level 1: 27 us with 128 workgroups
    // of 32 rows; tools/exp/unet_b1_prof.sh)
    constexpr int RSMALL = 8;
#define UO_DWR(CC, KK)                                                                                                        \
    if (C == CC && k == KK && B <= 65535 && (H + RSMALL - 1) / RSMALL <= 65535) {                                             \
        constexpr int PPB = 256 / (CC / 4);                                                                                   \
        if ((int64_t)B * ((W + PPB - 1) / PPB) * ((H + RROWS - 1) / RROWS) < 512)                                             \
            hipLaunchKernelGGL((uo_dwconv_ln_rows_kernel<CC, KK, RSMALL>), dim3((W + PPB - 1) / PPB, (H + RSMALL - 1) / RSMALL, B), \
                               dim3(256), 0, s, in, out, w, ln_gamma, H, W, eps, act, alpha);                                  \
        else                                                                                                                  \
            hipLaunchKernelGGL((uo_dwconv_ln_rows_kernel<CC, KK, RROWS>), dim3((W + PPB - 1) / PPB, (H + RROWS - 1) / RROWS, B), \
                               dim3(256), 0, s, in, out, w, ln_gamma, H, W, eps, act, alpha);                                  \
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;                                                              \
    }
    UO_DWR(32, 3) UO_DWR(32, 5) UO_DWR(64, 3) UO_DWR(64, 5) UO_DWR(128, 3) UO_DWR(128, 5)
#undef UO_DWR
#define UO_DW(CC, KK)                                                                                                         \
    if (C == CC && k == KK) {                                                                                                 \
        hipLaunchKernelGGL((uo_dwconv_ln_kernel<CC, KK>), dim3(uo_grid(npix, 256 / (CC / 4))), dim3(256), 0, s, in, out, w,   \
                           ln_gamma, B, H, W, eps, act, alpha);                                                               \
        ok = true;                                                                                                            \
    }
    UO_DW(32, 0) UO_DW(32, 1) UO_DW(32, 3) UO_DW(32, 5) UO_DW(64, 0) UO_DW(64, 1) UO_DW(64, 3) UO_DW(64, 5)
    UO_DW(128, 0) UO_DW(128, 1) UO_DW(128, 3) UO_DW(128, 5) UO_DW(256, 0)
    UO_DW(256, 1) UO_DW(256, 3) UO_DW(256, 5)            // a 256-channel ConvNext level (the deepest level of a four-level model)
#undef UO_DW
    if (!ok) return BF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// Laplacian split between the levels (backbone_unet_laplacian.py:366-386): smooth = AveragePooling2D(k, strides 1, same)
// (divisor = in-bounds taps) or GaussianFilter (fixed kernel, zero padding); lap = x - smooth at full resolution;
// down = smooth[:, ::2, ::2, :] (the "strides" downsample reads nothing else).  gauss: [k][k] or NULL (= average)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uo_smooth_split_kernel(const float* __restrict__ in, float* __restrict__ lap,
                                                              float* __restrict__ down, const float* __restrict__ gauss, int B, int H,
                                                              int W, int C, int k, int down_stride)
{
    const int Cv = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2, R = (k - 1) / 2;    // TF SAME: the extra pad of an even k is after
    const int64_t n = (int64_t)B * H * W * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const f32x4* img = reinterpret_cast<const f32x4*>(in) + (int64_t)b * H * W * Cv + c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y + ky - R;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x + kx - R;
                if (xx < 0 || xx >= W) continue;
                const f32x4 v = img[((int64_t)yy * W + xx) * Cv];
                if (gauss) acc += gauss[ky * k + kx] * v;
                else acc += v;
                ++cnt;
            }
        }
        if (!gauss) acc = acc / (float)cnt;
        const f32x4 ctr = img[((int64_t)y * W + x) * Cv];
        reinterpret_cast<f32x4*>(lap)[i] = ctr - acc;
        if (down_stride == 1) reinterpret_cast<f32x4*>(down)[i] = acc;
        else if (!((x | y) & 1)) reinterpret_cast<f32x4*>(down)[(((int64_t)b * OH + (y >> 1)) * OW + (x >> 1)) * Cv + c] = acc;
    }
}

// Column-walking form with the level's output normalisation fused in front (backbone_unet_laplacian.py:355-386):
//   y = act(LayerNorm(in) * gamma)  (gamma NULL: y = act(in));  smooth = AveragePooling2D / GaussianFilter k x k of y;
//   lap = y - smooth;  down = smooth[:, ::2, ::2, :].
// y is never written: every thread normalises the k pixels of the input row it loads (C/4 lanes per pixel, DPP
// reductions) and adds them to k rotating accumulators, as uo_dwconv_ln_rows_kernel does.
template <int C, int K, int R>
__global__ __launch_bounds__(256) void uo_norm_smooth_split_rows_kernel(const float* __restrict__ in, const float* __restrict__ gamma,
                                                                        float eps, int act, float alpha, const float* __restrict__ gauss,
                                                                        float* __restrict__ lap, float* __restrict__ down, int H, int W)
{
    constexpr int LPP = C / 4, PPB = 256 / LPP, RAD = K / 2;
    const int cl = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const int c0 = 4 * cl;
    const int x = blockIdx.x * PPB + pl;
    const bool xlive = x < W;
    const int y0 = blockIdx.y * R;
    const int64_t img = (int64_t)blockIdx.z * H * W;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + c0);
    float g[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) g[i] = gauss ? gauss[i] : 1.f;
    int xo[K];
    float xm[K];
    int cntx = 0;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const int xx = x + kx - RAD;
        const bool ok = xx >= 0 && xx < W;
        xm[kx] = ok ? 1.f : 0.f;
        cntx += ok ? 1 : 0;
        xo[kx] = min(max(xx, 0), W - 1) * C + c0;
    }
    f32x4 acc[K], ctr[RAD + 1];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j <= RAD; ++j) ctr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int yend = min(y0 + R, H);
    auto load_row = [&](int yi, f32x4 (&v)[K]) {
        const float* row = in + (img + (int64_t)min(max(yi, 0), H - 1) * W) * C;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(row + xo[kx]);
    };
    f32x4 vn[K];
    load_row(y0 - RAD, vn);
    for (int yi = y0 - RAD; yi < yend + RAD; ++yi) {
        f32x4 v[K];
        const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            f32x4 t = vn[kx];
            if (gamma) {
                const float mean = uo_pixel_sum<LPP>(t[0] + t[1] + t[2] + t[3]) * (1.f / C);
                const f32x4 d = t - mean;
                const float var = uo_pixel_sum<LPP>(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                t = d * (gm * rsqrtf(var + eps));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = uo_act_rt(t[j], act, alpha);
            v[kx] = t * (xm[kx] * ym);
        }
        load_row(yi + 1, vn);
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[ky] += g[ky * K + kx] * v[kx];
        // centre pixel of output row yi: needed RAD rows later
#pragma unroll
        for (int j = RAD; j > 0; --j) ctr[j] = ctr[j - 1];
        ctr[0] = v[RAD];
        const int yo = yi - RAD;
        if (yo >= y0) {
            f32x4 sm = acc[K - 1];
            if (!gauss) {
                const int cnty = min(yo + RAD, H - 1) - max(yo - RAD, 0) + 1;
                sm = sm / (float)(cntx * cnty);
            }
            if (xlive) {
                *reinterpret_cast<f32x4*>(lap + ((img + (int64_t)yo * W) + x) * C + c0) = ctr[RAD] - sm;
                if (!((x | yo) & 1))
                    *reinterpret_cast<f32x4*>(down + (((int64_t)blockIdx.z * OH + (yo >> 1)) * OW + (x >> 1)) * C + c0) = sm;
            }
        }
#pragma unroll
        for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

extern "C" int bf_op_norm_smooth_split(const float* in, const float* ln_gamma, float eps, int act, float alpha, const float* gauss,
                                       float* lap, float* down, int B, int H, int W, int C, int k, void* stream)
{
    if (!in || !lap || !down || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)lap | (uintptr_t)down | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    constexpr int RROWS = 16;
    if (B > 65535 || (H + RROWS - 1) / RROWS > 65535) return BF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define UO_NS(CC, KK)                                                                                                          \
    if (C == CC && k == KK) {                                                                                                  \
        constexpr int PPB = 256 / (CC / 4);                                                                                    \
        hipLaunchKernelGGL((uo_norm_smooth_split_rows_kernel<CC, KK, RROWS>), dim3((W + PPB - 1) / PPB, (H + RROWS - 1) / RROWS, B), \
                           dim3(256), 0, s, in, ln_gamma, eps, act, alpha, gauss, lap, down, H, W);                            \
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;                                                               \
    }
    UO_NS(32, 3) UO_NS(64, 3) UO_NS(128, 3) UO_NS(32, 5) UO_NS(64, 5) UO_NS(128, 5)
#undef UO_NS
    return BF_EUNSUPPORTED;
}

extern "C" int bf_op_smooth_split(const float* in, float* lap, float* down, const float* gauss, int B, int H, int W, int C, int k,
                                  int down_stride, void* stream)
{
    if (!in || !lap || !down || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || k <= 0) return BF_EINVAL;
    if (down_stride != 1 && down_stride != 2) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)lap | (uintptr_t)down) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(uo_smooth_split_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, lap, down, gauss, B, H, W,
                       C, k, down_stride);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// out = other + act(UpSampling2D(2, bilinear)(in))   (decoder: Add(skip, activation(1x1(upsample(x)))) with the linear 1x1
// commuted in front of the linear resize, as upsampling.py:80-90 does itself for a linear activation)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uo_upsample_act_add_kernel(const float* __restrict__ in, const float* __restrict__ other,
                                                                  float* __restrict__ out, int B, int H, int W, int C, int act,
                                                                  float alpha)
{
    const int Cv = C / 4, OH = 2 * H, OW = 2 * W;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const int iy = oy >> 1, ix = ox >> 1;
        const int y1 = (oy & 1) ? min(iy + 1, H - 1) : max(iy - 1, 0);
        const int x1 = (ox & 1) ? min(ix + 1, W - 1) : max(ix - 1, 0);
        const f32x4* base = reinterpret_cast<const f32x4*>(in) + (int64_t)b * H * W * Cv + c;
        const f32x4 v00 = base[((int64_t)iy * W + ix) * Cv], v01 = base[((int64_t)iy * W + x1) * Cv];
        const f32x4 v10 = base[((int64_t)y1 * W + ix) * Cv], v11 = base[((int64_t)y1 * W + x1) * Cv];
        f32x4 r = 0.75f * (0.75f * v00 + 0.25f * v10) + 0.25f * (0.75f * v01 + 0.25f * v11);
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = uo_act_rt(r[j], act, alpha);
        if (other) r += reinterpret_cast<const f32x4*>(other)[i];
        reinterpret_cast<f32x4*>(out)[i] = r;
    }
}

extern "C" int bf_op_upsample_act_add(const float* in, const float* other, float* out, int B, int H, int W, int C, int act,
                                      float alpha, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)other | (uintptr_t)out) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(uo_upsample_act_add_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, other, out, B, H, W, C,
                       act, alpha);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// tf.image.resize(BILINEAR, antialias=False): half-pixel centres (custom_layers.py:1328-1334, 1351-1357)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uo_resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                                 int W, int C, int OH, int OW, float sy, float sx)
{
    const int Cv = C / 4;
    const int64_t n = (int64_t)B * OH * OW * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv);
        int64_t t = i / Cv;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int b = (int)(t / OH);
        const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        const float fly = floorf(fy), flx = floorf(fx);
        const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), H - 1);
        const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), W - 1);
        const float ty = fy - fly, tx = fx - flx;
        const f32x4* base = reinterpret_cast<const f32x4*>(in) + (int64_t)b * H * W * Cv + c;
        const f32x4 v00 = base[((int64_t)y0 * W + x0) * Cv], v01 = base[((int64_t)y0 * W + x1) * Cv];
        const f32x4 v10 = base[((int64_t)y1 * W + x0) * Cv], v11 = base[((int64_t)y1 * W + x1) * Cv];
        const f32x4 top = v00 + (v01 - v00) * tx, bot = v10 + (v11 - v10) * tx;
        reinterpret_cast<f32x4*>(out)[i] = top + (bot - top) * ty;
    }
}

extern "C" int bf_op_resize_bilinear(const float* in, float* out, int B, int H, int W, int C, int OH, int OW, void* stream)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || C <= 0 || C % 4) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out) % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(uo_resize_bilinear_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// keras.layers.Attention(use_scale=False, score_mode="dot"): out = softmax(q k^T) v per image; q, k, v [B][T][A],
// A = 32.  One workgroup = 64 queries of one image (one per thread, online softmax); K and V of the image sit in LDS
// and every key / value row is an LDS broadcast.  (256 tokens x 32 channels per image: 8 MFLOP, negligible.)
// ------------------------------------------------------------------------------------------
constexpr int UO_ATT_A = 32;
constexpr int UO_ATT_WAVES = 4;        // waves per workgroup; a wave owns 16 queries of one sequence
// softmax(q k^T) v on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, bitwise an fmaf chain), no LDS:
// a wave walks the keys of its sequence 16 at a time with the running-max form of the softmax.
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T, contraction over the 32 channels in the order c = 8 kq + s (step s of 8,
//     lane group kq): A = lane (key m, kq) holds K[m][8kq .. 8kq+7], B = lane (query n, kq) holds Q[n][8kq .. 8kq+7] - two
//     16-byte loads each; accumulator: lane (g, n), register r = score of key 4g + r with query n.
//   O^T (32 channels x 16 queries) += V_tile^T . P^T: those four registers ARE the B operand of step r when the contraction
//     runs over the keys in the order k(g, r) = 4g + r; A = lane (channel m, g) holds V[4g + r][16 t + m].
//   A query's 16 scores of a tile sit in 4 registers x 4 lane groups: max / sum = in-lane + two cross-row shuffles.
__global__ __launch_bounds__(64 * UO_ATT_WAVES) void uo_attention_kernel(const float* __restrict__ q, const float* __restrict__ v,
                                                                 const float* __restrict__ k, float* __restrict__ out, int T,
                                                                 int tiles_per_seq, int64_t total_tiles, int ld)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int64_t tile = (int64_t)blockIdx.x * UO_ATT_WAVES + (threadIdx.x >> 6);
    if (tile >= total_tiles) return;                      // whole waves leave; nothing below synchronises across waves
    const int64_t b = tile / tiles_per_seq;
    const int q0 = (int)(tile % tiles_per_seq) * 16;
    const float* qb = q + b * T * ld;                       // ld: row pitch of q / k / v in floats (32, or 96 when the three
    const float* kb = k + b * T * ld;                       // projections come interleaved out of one 1x1 convolution)
    const float* vb = v + b * T * ld;
    float qf[8];
    {
        const float* qp = qb + (int64_t)min(q0 + n, T - 1) * ld + 8 * g;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(qp), t1 = *reinterpret_cast<const f32x4*>(qp + 4);
        qf[0] = t0[0]; qf[1] = t0[1]; qf[2] = t0[2]; qf[3] = t0[3]; qf[4] = t1[0]; qf[5] = t1[1]; qf[6] = t1[2]; qf[7] = t1[3];
    }
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float mrun = -INFINITY, lrun = 0.f;
    for (int k0 = 0; k0 < T; k0 += 16) {
        float kf[8];
        {
            const float* kp = kb + (int64_t)min(k0 + n, T - 1) * ld + 8 * g;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(kp), t1 = *reinterpret_cast<const f32x4*>(kp + 4);
            kf[0] = t0[0]; kf[1] = t0[1]; kf[2] = t0[2]; kf[3] = t0[3]; kf[4] = t1[0]; kf[5] = t1[1]; kf[6] = t1[2]; kf[7] = t1[3];
        }
        float vf[2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vp = vb + (int64_t)min(k0 + 4 * g + r, T - 1) * ld + n;
            vf[0][r] = vp[0]; vf[1][r] = vp[16];
        }
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) sc = MFMA4(kf[s2], qf[s2], sc);
        sc = bf_acc_ready(sc);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (k0 + 4 * g + r >= T) sc[r] = -INFINITY;          // keys past the end of the sequence
        float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);                      // finite: every tile holds at least one real key
        const float corr = __expf(mrun - mnew);
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __expf(sc[r] - mnew);
        float ps = (p[0] + p[1]) + (p[2] + p[3]);
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        lrun = lrun * corr + ps;
        mrun = mnew;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o[t] *= corr;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[t] = MFMA4(vf[t][r], p[r], o[t]);
        }
    }
    if (q0 + n >= T) return;
    const float inv = 1.f / lrun;
    float* op = out + (b * T + q0 + n) * UO_ATT_A + 4 * g;
#pragma unroll
    for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(op + 16 * t) = bf_acc_ready(o[t]) * inv;
}

extern "C" int bf_op_attention_ld(const float* q, const float* v, const float* k, float* out, int B, int T, int A, int ld,
                                  void* stream)
{
    if (!q || !v || !k || !out || B <= 0 || T <= 0 || ld < A || ld % 4) return BF_EINVAL;
    if (A != UO_ATT_A) return BF_EUNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)v | (uintptr_t)k | (uintptr_t)out) % 16) return BF_EINVAL;
    const int tiles = (T + 15) / 16;
    const int64_t total = (int64_t)B * tiles;
    const int64_t grid = (total + UO_ATT_WAVES - 1) / UO_ATT_WAVES;
    if (grid > 0x7fffffff) return BF_EUNSUPPORTED;
    hipLaunchKernelGGL(uo_attention_kernel, dim3((unsigned)grid), dim3(64 * UO_ATT_WAVES), 0, (hipStream_t)stream, q, v, k, out, T,
                       tiles, total, ld);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_op_attention(const float* q, const float* v, const float* k, float* out, int B, int T, int A, void* stream)
{
    return bf_op_attention_ld(q, v, k, out, B, T, A, A, stream);
}

// ------------------------------------------------------------------------------------------
// first convolution: k x k, Cin (<= 4) -> COUT on the normalised image, SAME zero padding, activation.
// The source image [B,Hs,Ws,cin] (u8 or f32, 0..255) is virtually zero-padded to [H,W] BEFORE normalisation
// (pad_to_power_of_2, utilities.py:736-751: padded pixels normalise to -0.5); outside [H,W] the convolution pads with 0.
// thread = one pixel, all output channels; weights broadcast from LDS (general k / cin / cout; the 5x5 3 -> 32 case of the
// unet_laplacian builder goes to uo_first_conv_tile_kernel below).
// ------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void uo_first_conv_kernel(const void* __restrict__ in, int in_is_u8, float* __restrict__ out,
                                                            const float* __restrict__ w, int B, int Hs, int Ws, int H, int W, int cin,
                                                            int k, int normalize, float v_min, float v_max, int act, float alpha)
{
    extern __shared__ __attribute__((aligned(16))) float wl[];     // [k][k][cin][COUT]
    const int nw = k * k * cin * COUT;
    for (int i = threadIdx.x; i < nw; i += 256) wl[i] = w[i];
    __syncthreads();
    const int64_t npix = (int64_t)B * H * W;
    const int R = k / 2;
    const float range = v_max - v_min;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
        const int x = (int)(p % W);
        const int y = (int)((p / W) % H);
        const int b = (int)(p / ((int64_t)W * H));
        float acc[COUT];
#pragma unroll
        for (int i = 0; i < COUT; ++i) acc[i] = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int yy = y + ky - R;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int xx = x + kx - R;
                if (xx < 0 || xx >= W) continue;
                const bool inside = yy < Hs && xx < Ws;
                const int64_t o = (((int64_t)b * Hs + yy) * Ws + xx) * cin;
                for (int ci = 0; ci < cin; ++ci) {
                    float v = 0.f;
                    if (inside) v = in_is_u8 ? (float)reinterpret_cast<const unsigned char*>(in)[o + ci]
                                             : reinterpret_cast<const float*>(in)[o + ci];
                    if (normalize) v = (fminf(fmaxf(v, v_min), v_max) - v_min) / range - 0.5f;
                    const f32x4* wr = reinterpret_cast<const f32x4*>(wl + ((ky * k + kx) * cin + ci) * COUT);
#pragma unroll
                    for (int i = 0; i < COUT / 4; ++i) {
                        const f32x4 ww = wr[i];
                        acc[4 * i] += v * ww[0]; acc[4 * i + 1] += v * ww[1]; acc[4 * i + 2] += v * ww[2]; acc[4 * i + 3] += v * ww[3];
                    }
                }
            }
        }
        f32x4* op = reinterpret_cast<f32x4*>(out + p * COUT);
#pragma unroll
        for (int i = 0; i < COUT / 4; ++i)
            op[i] = (f32x4){uo_act_rt(acc[4 * i], act, alpha), uo_act_rt(acc[4 * i + 1], act, alpha),
                            uo_act_rt(acc[4 * i + 2], act, alpha), uo_act_rt(acc[4 * i + 3], act, alpha)};
    }
}

// Matrix-core form for the 5x5, 3 -> 32 case (the one the unet_laplacian builder emits): a workgroup stages the
// normalised input tile of 8 x 64 output pixels (+ halo 2) in LDS once; K = 75 patch elements (tap-major, channel-minor)
// padded to 76 = 19 steps of the fp32 16x16x4 MFMA; lane (q, n) gathers element k = 4j + q of pixel n's patch with a
// 4-byte LDS read, the 19 x 2 weight operands stay in registers.  (Gathering the patch straight from global memory with
// byte loads was slower than the VALU kernel: 1.66 vs 1.09 ms on [32,512,512,3] uint8.)
constexpr int UO_FC_TH = 8, UO_FC_TW = 64;
template <int COUT>
__global__ __launch_bounds__(256) void uo_first_conv_tile_kernel(const void* __restrict__ in, int in_is_u8, float* __restrict__ out,
                                                                 const float* __restrict__ w, int Hs, int Ws, int H, int W,
                                                                 int normalize, float v_min, float v_max, int act, float alpha)
{
    constexpr int KS = 5, CIN = 3, KT = KS * KS * CIN, NJ = (KT + 3) / 4, T = COUT / 16, RAD = KS / 2;
    constexpr int IH = UO_FC_TH + 2 * RAD, IW = UO_FC_TW + 2 * RAD;
    __shared__ float tile[IH * IW * CIN];
    const int x0 = blockIdx.x * UO_FC_TW, y0 = blockIdx.y * UO_FC_TH;
    const int64_t b = blockIdx.z;
    const float range = v_max - v_min;
    // all loads of a thread are issued before the first is consumed (a rolled load -> store loop costs one memory round
    // trip per element)
    {
        constexpr int NE = (IH * IW * CIN + 255) / 256;
        float rv[NE];
        bool inpad[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            const int ci = e % CIN, px = (e / CIN) % IW, py = e / (CIN * IW);
            const int yy = y0 + py - RAD, xx = x0 + px - RAD;
            inpad[i] = e < IH * IW * CIN && yy >= 0 && yy < H && xx >= 0 && xx < W;      // inside the (virtually padded) image
            rv[i] = 0.f;
            if (inpad[i] && yy < Hs && xx < Ws) {
                const int64_t o = ((b * Hs + yy) * Ws + xx) * CIN + ci;
                rv[i] = in_is_u8 ? (float)reinterpret_cast<const unsigned char*>(in)[o] : reinterpret_cast<const float*>(in)[o];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            float v = rv[i];
            if (inpad[i] && normalize) v = (fminf(fmaxf(v, v_min), v_max) - v_min) / range - 0.5f;
            if (!inpad[i]) v = 0.f;
            if (e < IH * IW * CIN) tile[e] = v;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
    float a[NJ][T];
    int koff[NJ];                                      // LDS offset of patch element k = 4j + q relative to the pixel
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = 4 * j + q;
        const bool real = k < KT;
        const int tap = real ? k / CIN : 0, ci = real ? k - tap * CIN : 0;
        koff[j] = ((tap / KS) * IW + (tap % KS)) * CIN + ci;
#pragma unroll
        for (int t = 0; t < T; ++t) a[j][t] = real ? w[k * COUT + 16 * t + n] : 0.f;
    }
    __syncthreads();
    // 8 rows x 4 column groups of 16 pixels = 32 groups, 8 per wave
    for (int gi = wave; gi < UO_FC_TH * (UO_FC_TW / 16); gi += 4) {
        const int ry = gi / (UO_FC_TW / 16), cx = (gi % (UO_FC_TW / 16)) * 16 + n;
        const float* base = tile + (ry * IW + cx) * CIN;
        f32x4 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bv[j] = base[koff[j]];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = MFMA4(a[j][t], bv[j], acc[t]);
        const int gy = y0 + ry, gx = x0 + cx;
        if (gy < H && gx < W) {
            float* op = out + ((b * H + gy) * (int64_t)W + gx) * COUT + 4 * q;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4 v = bf_acc_ready(acc[t]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = uo_act_rt(v[r], act, alpha);
                *reinterpret_cast<f32x4*>(op + 16 * t) = v;
            }
        }
    }
}

extern "C" int bf_op_first_conv(const void* in, int in_is_u8, float* out, const float* w, int B, int Hs, int Ws, int H, int W, int cin,
                                int cout, int k, int normalize, float v_min, float v_max, int act, float alpha, void* stream)
{
    if (!in || !out || !w || B <= 0 || Hs <= 0 || Ws <= 0 || H < Hs || W < Ws || cin <= 0 || k <= 0 || !(k & 1)) return BF_EINVAL;
    if (normalize && !(v_max > v_min)) return BF_EINVAL;
    if ((uintptr_t)out % 16) return BF_EINVAL;
    const size_t lds = (size_t)k * k * cin * cout * sizeof(float);
    if (lds > 64 * 1024) return BF_EUNSUPPORTED;
    const int64_t npix = (int64_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const int grid = uo_grid(npix, 256);
    if (cout == 32 && k == 5 && cin == 3 && B <= 65535 && (H + UO_FC_TH - 1) / UO_FC_TH <= 65535) {
        hipLaunchKernelGGL((uo_first_conv_tile_kernel<32>), dim3((W + UO_FC_TW - 1) / UO_FC_TW, (H + UO_FC_TH - 1) / UO_FC_TH, B),
                           dim3(256), 0, s, in, in_is_u8, out, w, Hs, Ws, H, W, normalize, v_min, v_max, act, alpha);
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
    }
    if (cout == 32)
        hipLaunchKernelGGL((uo_first_conv_kernel<32>), dim3(grid), dim3(256), lds, s, in, in_is_u8, out, w, B, Hs, Ws, H, W, cin, k,
                           normalize, v_min, v_max, act, alpha);
    else if (cout == 64)
        hipLaunchKernelGGL((uo_first_conv_kernel<64>), dim3(grid), dim3(256), lds, s, in, in_is_u8, out, w, B, Hs, Ws, H, W, cin, k,
                           normalize, v_min, v_max, act, alpha);
    else if (cout == 16)
        hipLaunchKernelGGL((uo_first_conv_kernel<16>), dim3(grid), dim3(256), lds, s, in, in_is_u8, out, w, B, Hs, Ws, H, W, cin, k,
                           normalize, v_min, v_max, act, alpha);
    else return BF_EUNSUPPORTED;
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// last convolution of a denoiser head (model.py:321-342): 1x1 hf -> cout (<= 4), tanh(2x) * 0.51, [denormalise:
// (clip(y, -.5, .5) + .5) * (v_max - v_min) + v_min], [round half to even, clip, uint8], crop to [Ho,Wo]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uo_head_out_kernel(const float* __restrict__ in, const float* __restrict__ w, void* __restrict__ out,
                                                          int out_is_u8, int B, int H, int W, int Ho, int Wo, int hf, int cout,
                                                          int denormalize, float v_min, float v_max, int* __restrict__ status)
{
    __shared__ float wl[4 * 256];
    for (int i = threadIdx.x; i < hf * cout; i += 256) wl[i] = w[i];
    __syncthreads();
    const int64_t n = (int64_t)B * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wo);
        const int y = (int)((i / Wo) % Ho);
        const int b = (int)(i / ((int64_t)Wo * Ho));
        const float* src = in + (((int64_t)b * H + y) * W + x) * hf;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < hf; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                for (int o = 0; o < cout; ++o) acc[o] += v[j] * wl[(c + j) * cout + o];
        }
        for (int o = 0; o < cout; ++o) {
            // an inf / NaN anywhere upstream (f16 range left inside a split-f16 kernel) reaches this sum; tanh would hide it
            if (status && !(fabsf(acc[o]) <= 3.0e38f)) atomicOr(status, BF_STATUS_F16_RANGE);
            float r = tanhf(2.f * acc[o]) * 0.51f;
            if (denormalize) r = (fminf(fmaxf(r, -0.5f), 0.5f) + 0.5f) * (v_max - v_min) + v_min;
            if (out_is_u8) reinterpret_cast<unsigned char*>(out)[i * cout + o] = (unsigned char)fminf(fmaxf(rintf(r), 0.f), 255.f);
            else reinterpret_cast<float*>(out)[i * cout + o] = r;
        }
    }
}

extern "C" int bf_op_head_out(const float* in, const float* w, void* out, int out_is_u8, int B, int H, int W, int Ho, int Wo, int hf,
                              int cout, int denormalize, float v_min, float v_max, int* status, void* stream)
{
    if (!in || !w || !out || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Ho > H || Wo > W) return BF_EINVAL;
    if (hf <= 0 || hf % 4 || hf > 256 || cout <= 0 || cout > 4) return BF_EUNSUPPORTED;
    if ((uintptr_t)in % 16) return BF_EINVAL;
    const int64_t n = (int64_t)B * Ho * Wo;
    hipLaunchKernelGGL(uo_head_out_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, w, out, out_is_u8, B, H, W, Ho,
                       Wo, hf, cout, denormalize, v_min, v_max, status);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// whole denoiser head on one feature map, one kernel (model.py:297-342 + the backbone's output LayerNorm,
// backbone_unet_laplacian.py:547-551): [LayerNorm(x) * gamma] -> 1x1 C -> 32, activation -> 1x1 32 -> cout (<= 4) ->
// tanh(2x) * 0.51 -> [denormalise] -> [round, uint8], cropped to [Ho,Wo].  The 32-wide hidden layer stays in the
// accumulators: lane (q, n) holds hidden channels 16t + 4q + r of pixel n, multiplies them with its rows of the last
// kernel and the four q-lanes of a pixel are summed with two cross-row shuffles.
// ------------------------------------------------------------------------------------------
// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48): the sum over the four lanes (q = 0..3) that hold one pixel in the
// matrix-core layout, on the vector ALU (v_permlane16_swap / v_permlane32_swap, gfx950) instead of two ds_bpermute round trips
// through the LDS crossbar: with two waves per SIMD nothing hides those (the head kernel has 12 of them per 16-pixel group).
// swap(a, b): rows 1, 3 (16-lane groups) of a <-> rows 0, 2 of b; with a = b = v: a = [r0 r0 r2 r2], b = [r1 r1 r3 r3].
__device__ __forceinline__ float uo_sum_q(float v)
{
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);          // rows 0, 1: r0 + r1 ; rows 2, 3: r2 + r3
    a = __builtin_bit_cast(unsigned, s); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // lanes 32..63 of a <-> lanes 0..31 of b
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

template <int CIN>
__global__ __launch_bounds__(256, 2) void uo_head_fused_kernel(const float* __restrict__ in, const float* __restrict__ gamma, float eps,
                                                               const float* __restrict__ w0p, int act, float alpha,
                                                               const float* __restrict__ w1, void* __restrict__ out, int out_is_u8,
                                                               int B, int H, int W, int Ho, int Wo, int cout, int denormalize,
                                                               float v_min, float v_max, int* __restrict__ status)
{
    constexpr int KC = CIN / 16, T = 2, NP = 2, HF = 32;
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t npix = (int64_t)B * Ho * Wo;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    // rows 16t + 4q + r of the last kernel [HF][cout]
    float wl[T][4][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < 4; ++o) wl[t][r][o] = o < cout ? w1[(16 * t + 4 * q + r) * cout + o] : 0.f;
    f32x4 gm[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) gm[c] = gamma ? *reinterpret_cast<const f32x4*>(gamma + 16 * c + 4 * q) : (f32x4){1.f, 1.f, 1.f, 1.f};
    // the raw pixels of the NEXT group are requested before the current one is normalised and multiplied (CIN <= 64: the
    // extra registers fit), as in the split-f16 MLP kernel
    constexpr bool PF = CIN <= 64;
    f32x4 raw[NP][KC];
    auto load_raw = [&](int64_t gg) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = gg * 16 * NP + 16 * i + n;
            p = p < npix ? p : npix - 1;
            const int x = (int)(p % Wo);
            const int y = (int)((p / Wo) % Ho);
            const int64_t bi = p / ((int64_t)Wo * Ho);
            const float* src = in + ((bi * H + y) * W + x) * CIN + 4 * q;
#pragma unroll
            for (int c = 0; c < KC; ++c) raw[i][c] = *reinterpret_cast<const f32x4*>(src + 16 * c);
        }
    };
    if (PF && wave < ngroups) load_raw(wave);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        const float* wpo = w0p;
        asm volatile("" : "+s"(wpo));                              // see uo_pointwise_kernel
        const f32x4* wv = reinterpret_cast<const f32x4*>(wpo) + lane;
        if (!PF) load_raw(g);
        f32x4 b[NP][KC];
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c) b[i][c] = raw[i][c];
        if (PF) {
            __builtin_amdgcn_sched_barrier(0);
            if (g + nwaves < ngroups) load_raw(g + nwaves);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) sum += b[i][c][0] + b[i][c][1] + b[i][c][2] + b[i][c][3];
            if (gamma) {
                sum = uo_sum_q(sum);
                const float mean = sum * (1.f / CIN);
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    b[i][c] = b[i][c] - mean;
                    sq += b[i][c][0] * b[i][c][0] + b[i][c][1] * b[i][c][1] + b[i][c][2] * b[i][c][2] + b[i][c][3] * b[i][c][3];
                }
                sq = uo_sum_q(sq);
                const float rs = rsqrtf(sq * (1.f / CIN) + eps);
#pragma unroll
                for (int c = 0; c < KC; ++c) b[i][c] = b[i][c] * (gm[c] * rs);
            }
        }
        f32x4 acc[T][NP];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 a = wv[(c * T + t) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[t][i] = MFMA4(a[j], b[i][c][j], acc[t][i]);
            }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 hv = bf_acc_ready(acc[t][i]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hh = uo_act_rt(hv[r], act, alpha);
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] += hh * wl[t][r][k];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = uo_sum_q(o[k]);
            }
            // every lane holds the pixel's sums now: lane group q finishes output channel q (one tanh per lane instead of
            // cout in a quarter of the lanes; one store instruction per group).  tanh(2x) = 1 - 2 / (exp(4x) + 1), the form
            // the resnet head uses (fused_h3.hip): absolute error ~1e-7, 1e-5 of an output grey level.
            const int64_t p = p0 + 16 * i + n;
            const float ok = q == 0 ? o[0] : (q == 1 ? o[1] : (q == 2 ? o[2] : o[3]));
            // an inf / NaN anywhere upstream (f16 range left inside a split-f16 kernel) reaches this sum; tanh would hide it
            if (status && !(fabsf(ok) <= 3.0e38f)) atomicOr(status, BF_STATUS_F16_RANGE);
            float r = (1.0f - 2.0f / (__expf(4.0f * ok) + 1.0f)) * 0.51f;
            if (denormalize) r = (fminf(fmaxf(r, -0.5f), 0.5f) + 0.5f) * (v_max - v_min) + v_min;
            if (q < cout && p < npix) {
                if (out_is_u8) reinterpret_cast<unsigned char*>(out)[p * cout + q] = (unsigned char)fminf(fmaxf(rintf(r), 0.f), 255.f);
                else reinterpret_cast<float*>(out)[p * cout + q] = r;
            }
        }
    }
    (void)HF;
}

extern "C" int bf_op_head_fused(const float* in, const float* ln_gamma, float eps, const float* w0p, int act, float alpha,
                                const float* w1, void* out, int out_is_u8, int B, int H, int W, int Ho, int Wo, int cin, int hf,
                                int cout, int denormalize, float v_min, float v_max, int* status, void* stream)
{
    if (!in || !w0p || !w1 || !out || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Ho > H || Wo > W) return BF_EINVAL;
    if (hf != 32 || cout <= 0 || cout > 4) return BF_EUNSUPPORTED;
    if (((uintptr_t)in | (uintptr_t)w0p | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    const int64_t npix = (int64_t)B * Ho * Wo;
    const int grid = uo_grid(npix, 4 * 32, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
#define UO_HEAD(CC)                                                                                                            \
    hipLaunchKernelGGL((uo_head_fused_kernel<CC>), dim3(grid), dim3(256), 0, s, in, ln_gamma, eps, w0p, act, alpha, w1, out,    \
                       out_is_u8, B, H, W, Ho, Wo, cout, denormalize, v_min, v_max, status)
    if (cin == 32) UO_HEAD(32);
    else if (cin == 64) UO_HEAD(64);
    else if (cin == 128) UO_HEAD(128);
    else if (cin == 256) UO_HEAD(256);
    else return BF_EUNSUPPORTED;
#undef UO_HEAD
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// The same head with its first 1x1 (CIN -> 32) on the f16 matrix cores with split-f16 operands (three products, fp32 accumulation: the
// arithmetic of unet_h3.hip) for CIN = 32 / 64: 6 / 12 MFMAs of 16 cycles per 16 pixels where the fp32 form above issues 16 / 32 of 32
// cycles (433 us per batch of 32 x 512 x 512 at 32 channels: the fp32 matrix chain and the epilogue's vector work did not overlap).
// Lane (q, n) loads channels 32c + 8q .. + 7 of pixel n (the B fragment of K chunk c); the weight fragments are built once per wave from
// the fp32 operand bf_op_pack_pointwise wrote (element W0[16 c16 + 4 q' + j][16 t + m] at ((c16 * 2 + t) * 64 + 16 q' + m) * 4 + j),
// scaled by a power of two that puts the largest weight in [2^13, 2^14).
// NO = output channels the per-pixel epilogue carries (3 for the colour models: a quarter of its multiplies and one of its four lane sums less than 4)
template <int CIN, int NO>
__global__ __launch_bounds__(256, 2) void uo_head_fused_h3_kernel(const float* __restrict__ in, const float* __restrict__ gamma, float eps,
                                                                  const float* __restrict__ w0p, int act, float alpha,
                                                                  const float* __restrict__ w1, void* __restrict__ out, int out_is_u8,
                                                                  int B, int H, int W, int Ho, int Wo, int cout, int denormalize,
                                                                  float v_min, float v_max, int* __restrict__ status)
{
    constexpr int KC = CIN / 32, T = 2, NP = 2;
    __shared__ float red[256];
    float mx = 0.f;
    for (int i = threadIdx.x; i < CIN * 32; i += 256) mx = fmaxf(mx, fabsf(w0p[i]));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    mx = red[0];
    float scale = 1.f;
    if (mx > 0.f && isfinite(mx)) {
        int ex;
        (void)frexpf(mx, &ex);
        scale = ldexpf(1.f, 14 - max(-100, min(100, ex)));
    }
    const float inv = 1.f / scale;
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    uh8 wh[KC][T], wlo[KC][T];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x4 a0, a1;
            const int c16 = 2 * c + (q >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0[i] = w0p[((c16 * T + t) * 64 + 16 * (2 * (q & 1)) + n) * 4 + i] * scale;          // k = 32c + 8q + i
                a1[i] = w0p[((c16 * T + t) * 64 + 16 * (2 * (q & 1) + 1) + n) * 4 + i] * scale;      // k = 32c + 8q + 4 + i
            }
            uh_split8(a0, a1, wh[c][t], wlo[c][t]);
        }
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t npix = (int64_t)B * Ho * Wo;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    float wl[T][4][NO];                                            // rows 16t + 4q + r of the last kernel [32][cout]
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int o = 0; o < NO; ++o) wl[t][r][o] = o < cout ? w1[(16 * t + 4 * q + r) * cout + o] : 0.f;
    f32x4 gm[KC][2];
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            gm[c][u] = gamma ? *reinterpret_cast<const f32x4*>(gamma + 32 * c + 8 * q + 4 * u) : (f32x4){1.f, 1.f, 1.f, 1.f};
    f32x4 raw[NP][KC][2];
    auto load_raw = [&](int64_t gg) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = gg * 16 * NP + 16 * i + n;
            p = p < npix ? p : npix - 1;
            const int x = (int)(p % Wo);
            const int y = (int)((p / Wo) % Ho);
            const int64_t bi = p / ((int64_t)Wo * Ho);
            const float* src = in + ((bi * H + y) * W + x) * CIN + 8 * q;
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u) raw[i][c][u] = *reinterpret_cast<const f32x4*>(src + 32 * c + 4 * u);
        }
    };
    if (wave < ngroups) load_raw(wave);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        f32x4 b[NP][KC][2];
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u) b[i][c][u] = raw[i][c][u];
        __builtin_amdgcn_sched_barrier(0);
        {
            const int64_t gn = g + nwaves;
            load_raw(gn < ngroups ? gn : g);                       // unconditional: no branch around the loads
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[T][NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (gamma) {
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u) sum += b[i][c][u][0] + b[i][c][u][1] + b[i][c][u][2] + b[i][c][u][3];
                sum = uo_sum_q(sum);
                const float mean = sum * (1.f / CIN);
                float sq = 0.f;
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        b[i][c][u] = b[i][c][u] - mean;
                        sq += b[i][c][u][0] * b[i][c][u][0] + b[i][c][u][1] * b[i][c][u][1] + b[i][c][u][2] * b[i][c][u][2] + b[i][c][u][3] * b[i][c][u][3];
                    }
                sq = uo_sum_q(sq);
                const float rs = rsqrtf(sq * (1.f / CIN) + eps);
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u) b[i][c][u] = b[i][c][u] * (gm[c][u] * rs);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                uh8 xh, xl;
                uh_split8(b[i][c][0], b[i][c][1], xh, xl);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[t][i] = UH_MFMA_REAL(wh[c][t], xh, acc[t][i]);
                    acc[t][i] = UH_MFMA_REAL(wlo[c][t], xh, acc[t][i]);
                    acc[t][i] = UH_MFMA_REAL(wh[c][t], xl, acc[t][i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};                   // o[NO ..] stay 0
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const f32x4 hv = bf_acc_ready(acc[t][i]) * inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hh = uo_act_rt(hv[r], act, alpha);
#pragma unroll
                    for (int k = 0; k < NO; ++k) o[k] += hh * wl[t][r][k];
                }
            }
#pragma unroll
            for (int k = 0; k < NO; ++k) o[k] = uo_sum_q(o[k]);
            const int64_t p = p0 + 16 * i + n;
            const float ok = q == 0 ? o[0] : (q == 1 ? o[1] : (q == 2 ? o[2] : o[3]));
            if (status && !(fabsf(ok) <= 3.0e38f)) atomicOr(status, BF_STATUS_F16_RANGE);
            float r = (1.0f - 2.0f / (__expf(4.0f * ok) + 1.0f)) * 0.51f;
            if (denormalize) r = (fminf(fmaxf(r, -0.5f), 0.5f) + 0.5f) * (v_max - v_min) + v_min;
            if (q < cout && p < npix) {
                if (out_is_u8) reinterpret_cast<unsigned char*>(out)[p * cout + q] = (unsigned char)fminf(fmaxf(rintf(r), 0.f), 255.f);
                else reinterpret_cast<float*>(out)[p * cout + q] = r;
            }
        }
    }
}

extern "C" int bf_op_head_fused_h3(const float* in, const float* ln_gamma, float eps, const float* w0p, int act, float alpha,
                                   const float* w1, void* out, int out_is_u8, int B, int H, int W, int Ho, int Wo, int cin, int hf,
                                   int cout, int denormalize, float v_min, float v_max, int* status, void* stream)
{
    if (!in || !w0p || !w1 || !out || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || Ho > H || Wo > W) return BF_EINVAL;
    if (hf != 32 || cout <= 0 || cout > 4 || (cin != 32 && cin != 64)) return BF_EUNSUPPORTED;
    if (((uintptr_t)in | (uintptr_t)w0p | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    const int64_t npix = (int64_t)B * Ho * Wo;
    const int grid = uo_grid(npix, 4 * 32, 256 * 4);               // persistent: the weight fragments are built once per wave
    hipStream_t s = (hipStream_t)stream;
#define UO_HEAD_H3(CC, NN)                                                                                                        \
    hipLaunchKernelGGL((uo_head_fused_h3_kernel<CC, NN>), dim3(grid), dim3(256), 0, s, in, ln_gamma, eps, w0p, act, alpha, w1, out, out_is_u8, \
                       B, H, W, Ho, Wo, cout, denormalize, v_min, v_max, status)
    if (cin == 32) { if (cout <= 3) UO_HEAD_H3(32, 3); else UO_HEAD_H3(32, 4); }
    else { if (cout <= 3) UO_HEAD_H3(64, 3); else UO_HEAD_H3(64, 4); }
#undef UO_HEAD_H3
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// Conv2DTranspose k x k, stride s, padding "same", no bias (upsample_type "conv2d_transpose", bfcnn/upsampling.py:37-48;
// utilities.py:200-202): out [B, H*s, W*s, cout], kernel [k, k, cout, cin] (keras layout).  It is the transpose of the
// stride-s SAME convolution that maps [H*s, W*s] to [H, W] (pad_before = max(k - s, 0) / 2):
//   out[y, x, co] = sum over (iy, i), (ix, j) with iy*s + i - pb = y, ix*s + j - pb = x of in[iy, ix, :] . K[i, j, co, :]
// One thread per output element; an upsampling primitive of configurations nobody ships, not a hot kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uo_conv2d_transpose_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                  float* __restrict__ out, int B, int H, int W, int cin, int cout,
                                                                  int k, int s, int act, float alpha)
{
    const int Ho = H * s, Wo = W * s, pb = (k > s ? k - s : 0) / 2;
    const int64_t n = (int64_t)B * Ho * Wo * cout;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int co = (int)(e % cout);
        int64_t p = e / cout;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float acc = 0.f;
        for (int i = 0; i < k; ++i) {
            const int ty = y + pb - i;
            if (ty < 0 || ty % s) continue;
            const int iy = ty / s;
            if (iy >= H) continue;
            for (int j = 0; j < k; ++j) {
                const int tx = x + pb - j;
                if (tx < 0 || tx % s) continue;
                const int ix = tx / s;
                if (ix >= W) continue;
                const float* src = in + (((int64_t)b * H + iy) * W + ix) * cin;
                const float* kw = w + ((int64_t)(i * k + j) * cout + co) * cin;
                for (int c = 0; c < cin; ++c) acc = fmaf(src[c], kw[c], acc);
            }
        }
        out[e] = uo_act_rt(acc, act, alpha);
    }
}

extern "C" int bf_op_conv2d_transpose(const float* in, const float* w, float* out, int B, int H, int W, int cin, int cout, int k,
                                      int stride, int act, float alpha, void* stream)
{
    if (!in || !w || !out || B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0) return BF_EINVAL;
    if (k <= 0 || k > 16 || stride <= 0 || stride > 8) return BF_EUNSUPPORTED;
    const int64_t n = (int64_t)B * H * stride * W * stride * cout;
    hipLaunchKernelGGL(uo_conv2d_transpose_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, in, w, out, B, H, W, cin,
                       cout, k, stride, act, alpha);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// 32-bit fill on the stream (status words, accumulators): the host library allocates, the engine initialises
__global__ void uo_fill32_kernel(int* __restrict__ p, int value, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = value;
}

extern "C" int bf_op_fill32(void* p, int value, int64_t n, void* stream)
{
    if (!p || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(uo_fill32_kernel, dim3(uo_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (int*)p, value, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// mult[c] = tanh(relu(1 + w[c]))   (ChannelLearnableMultiplier, custom_layers.py:304-306)
__global__ void uo_channel_multiplier_kernel(const float* __restrict__ w, float* __restrict__ m, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) m[i] = tanhf(fmaxf(1.f + w[i], 0.f));
}

extern "C" int bf_op_channel_multiplier(const float* w, float* mult, int n, void* stream)
{
    if (!w || !mult || n <= 0) return BF_EINVAL;
    hipLaunchKernelGGL(uo_channel_multiplier_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, mult, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
