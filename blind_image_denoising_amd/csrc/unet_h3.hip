// ConvNext MLP of the unet_laplacian backbone on the f16 matrix cores with split-f16 ("f16x3") arithmetic:
//   out = skip + mult * (act(x . W1) . W2)        x [npix][C] fp32 (LayerNorm output), W1 [C][4C], W2 [4C][C]
// (bfcnn/custom_layers.py:990-1008 conv_2 -> activation -> conv_3 -> ChannelLearnableMultiplier; the Add of
// backbone_unet_laplacian.py:351-354.)  Same arithmetic as the resnet blocks (fused_h3.hip): every fp32
// operand v is carried as hi = f16(v), lo = f16(v - hi) (22 mantissa bits), a product is w_hi x_hi + w_lo x_hi + w_hi x_lo
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation; the weights are pre-scaled by a power of two so that their lo parts
// stay normal f16 numbers.  3 MFMAs of 16 cycles contract K = 32 where the fp32 path (unet_ops.hip) needs 8 MFMAs of 32
// cycles: 5.3x fewer matrix-pipe cycles, which makes the kernel HBM-bound (x + skip + out = 12 C bytes per pixel).
//
// Data flow per wave and 16 pixels (N = pixel, M = output channel of a tile):
//   GEMM1  B = x^T: lane (q, n) loads channels 32c + 8q .. +7 of pixel n (two 16-byte loads) and splits them;
//          A = W1 fragments from LDS.  Two hidden tiles (2c2, 2c2+1) are produced at a time:
//          lane (q, n), register r of tile t = hidden channel 16t + 4q + r of pixel n.
//   GEMM2  those 8 values of a lane, activated and split, ARE a legal B fragment of the next GEMM for the K order
//          k(q, i) = 32 c2 + 16 (i / 4) + 4 q + (i % 4); W2 is packed in that order.  The 4C-wide hidden tensor never
//          exists outside a pair of accumulators.
#include "unet_h3_core.h"
#ifndef UH_NP32
#define UH_NP32 2                  // 16-pixel groups a wave of the C = 32 MLP kernels carries through the two GEMMs at a time
#endif
#ifndef UH_NP64
#define UH_NP64 2
#endif

// ------------------------------------------------------------------------------------------
// packing: [W1 fragments | W2 fragments] as f16, then {1/s1, 1/s2} as fp32.
// fragment f = (chunk * tiles + tile) * 2 + (0 hi | 1 lo); element (f * 64 + lane) * 8 + i, lane = 16 q + m
//   W1: value W1[32 chunk + 8 q + i][16 tile + m] * s1
//   W2: value W2[32 chunk + 16 (i / 4) + 4 q + (i % 4)][16 tile + m] * s2
// s = power of two with max |w| s in [2^13, 2^14)
// ------------------------------------------------------------------------------------------
__device__ float uh_block_scale(const float* __restrict__ w, int n, float* red)
{
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    const float mx = red[0];
    __syncthreads();
    if (!(mx > 0.f) || !isfinite(mx)) return 1.f;
    int ex;
    (void)frexpf(mx, &ex);                                           // mx = f 2^ex, f in [0.5, 1)
    ex = max(-100, min(100, ex));
    return ldexpf(1.f, 14 - ex);
}

// chain = 1 (C = 32): W1 in the K order of W2, k(q, i) = 16 (i / 4) + 4 q + (i % 4) -- the order in which a lane of the OUTPUT layout
// holds a pixel's 32 channels (tile 0 registers, tile 1 registers), so that a block's output is the next block's B fragment as it stands
// (uh_chain32_kernel)
__global__ __launch_bounds__(256) void uh_pack_mlp_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                          _Float16* __restrict__ dst, int C, int chain = 0)
{
    __shared__ float red[256];
    const int H = 4 * C;
    const float s1 = uh_block_scale(w1, C * H, red);
    const float s2 = uh_block_scale(w2, C * H, red);
    const int n = C * H;                                             // values per matrix; 2 n halves per matrix
    for (int e = threadIdx.x; e < 2 * n; e += 256) {
        const int i = e & 7, lane = (e >> 3) & 63, f = e >> 9;
        const int hl = f & 1, ct = f >> 1;
        const int q = lane >> 4, m = lane & 15;
        {
            const int T = H / 16, t = ct % T, c = ct / T;
            const int k1 = chain ? 16 * (i >> 2) + 4 * q + (i & 3) : 32 * c + 8 * q + i;
            const float v = w1[k1 * H + 16 * t + m] * s1;
            const _Float16 hi = (_Float16)v;
            dst[e] = hl ? (_Float16)(v - (float)hi) : hi;
        }
        {
            const int T = C / 16, t = ct % T, c = ct / T;
            const float v = w2[(32 * c + 16 * (i >> 2) + 4 * q + (i & 3)) * C + 16 * t + m] * s2;
            const _Float16 hi = (_Float16)v;
            dst[2 * n + e] = hl ? (_Float16)(v - (float)hi) : hi;
        }
    }
    if (threadIdx.x == 0) {
        float* aux = reinterpret_cast<float*>(dst + 4 * n);
        aux[0] = 1.f / s1;
        aux[1] = 1.f / s2;
        aux[2] = 0.f;
        aux[3] = 0.f;
    }
}

extern "C" int64_t bf_op_mlp_h3_pack_bytes(int C)
{
    if (C != 32 && C != 64) return -1;
    return (int64_t)32 * C * C + 16;
}

extern "C" int bf_op_pack_mlp_h3(const float* w1, const float* w2, void* packed, int C, void* stream)
{
    if (!w1 || !w2 || !packed) return BF_EINVAL;
    if (C != 32 && C != 64) return BF_EUNSUPPORTED;
    if ((uintptr_t)packed % 16) return BF_EINVAL;
    hipLaunchKernelGGL(uh_pack_mlp_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w1, w2, (_Float16*)packed, C, 0);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// the operand of bf_op_convnext_chain32_h3 (32 channels; same size as bf_op_pack_mlp_h3's)
extern "C" int bf_op_pack_mlp_h3_chain(const float* w1, const float* w2, void* packed, int C, void* stream)
{
    if (!w1 || !w2 || !packed) return BF_EINVAL;
    if (C != 32) return BF_EUNSUPPORTED;
    if ((uintptr_t)packed % 16) return BF_EINVAL;
    hipLaunchKernelGGL(uh_pack_mlp_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w1, w2, (_Float16*)packed, C, 1);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}


// PRE = 1: the whole ConvNextBlock with a 1x1 depthwise convolution (the decoder blocks, decoder_kernel_size 1): `in` is
// the block input x; t = LayerNorm(x * dw) * gamma is formed in registers (the 4 lanes (q, n) of a pixel hold all its
// channels: two cross-row shuffles per reduction) and never written; skip = x.
// PRE = 2 (C = 32; round 4): the FIRST decoder block of a level with the node in front of it formed while the pixels are loaded:
// x = in + act_up(UpSampling2D(2, bilinear)(low)) (in = the encoder's skip map [B, OH, OW, C], low = the 1x1-projected map of the level
// below [B, OH / 2, OW / 2, C]; backbone_unet_laplacian.py:438-568, upsampling.py:80-90) -- what uo_upsample_act_add_kernel wrote and this
// kernel read back (2.1 GB of a batch-32 forward and one launch).  The residual needs x in the OUTPUT lane layout (tile t, lane q:
// channels 16 t + 4 q ..): it is fetched from the lanes that hold it in the input layout (lane 2 t + (q >> 1), half q & 1) with
// ds_bpermute instead of being re-read.
template <int C, int NP, int ACT, int NT, int PRE>
__global__ __launch_bounds__(NT, 512 / NT) void uh_mlp_kernel(const float* __restrict__ in, const float* __restrict__ skip, float* __restrict__ out,
                                                    const void* __restrict__ packed, const float* __restrict__ mult, int64_t npix,
                                                    float alpha, const float* __restrict__ dw, const float* __restrict__ gamma,
                                                    float eps, const float* __restrict__ low = nullptr, int OH = 0, int OW = 0, int act_up = 0,
                                                    float alpha_up = 0.f)
{
    static_assert(PRE != 2 || C == 32, "the fused up-sampling form is built for 32 channels");
    constexpr int KC1 = C / 32, T1 = 4 * C / 16, KC2 = 4 * C / 32, T2 = C / 16;
    constexpr int W1_BYTES = KC1 * T1 * 2 * 1024, W2_BYTES = KC2 * T2 * 2 * 1024;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < (W1_BYTES + W2_BYTES) / 16; i += NT) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W1_BYTES + W2_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    // per-channel output factor mult / s2 of the lane's 4 channels of every output tile
    f32x4 m4[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        m4[t] = (f32x4){inv2, inv2, inv2, inv2};
        if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
    }
    f32x4 dwv[KC1][2], gmv[KC1][2];
    if (PRE) {
#pragma unroll
        for (int c = 0; c < KC1; ++c)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dwv[c][u] = *reinterpret_cast<const f32x4*>(dw + 32 * c + 8 * q + 4 * u);
                gmv[c][u] = gamma ? *reinterpret_cast<const f32x4*>(gamma + 32 * c + 8 * q + 4 * u) : (f32x4){1.f, 1.f, 1.f, 1.f};
            }
    }
    // raw input of the NEXT group is requested as soon as the current one has been converted: its latency hides behind
    // the matrix work of the current group instead of sitting in front of it
    f32x4 xr[NP][KC1][2];
    f32x4 tap[PRE == 2 ? NP : 1][PRE == 2 ? 4 : 1][2];          // PRE = 2: the four low-resolution taps of every pixel (C = 32: one chunk)
    auto load_raw = [&](int64_t gg) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = gg * 16 * NP + 16 * i + n;
            p = p < npix ? p : npix - 1;                          // tail: clamp the read, predicate the store
            const float* src = in + p * C + 8 * q;
#pragma unroll
            for (int c = 0; c < KC1; ++c)
#pragma unroll
                for (int u = 0; u < 2; ++u) xr[i][c][u] = *reinterpret_cast<const f32x4*>(src + 32 * c + 4 * u);
            if (PRE == 2) {
                // output pixel (b, oy, ox) -> its 2 x 2 low-resolution taps (tf.image.resize half-pixel centres at factor 2: weights
                // 0.75 / 0.25, indices clamped to the map: uo_upsample_act_add_kernel's expression)
                const int LH = OH >> 1, LW = OW >> 1;
                const int hw = OH * OW;
                const int b = (int)(p / hw), r = (int)(p - (int64_t)b * hw);
                const int oy = r / OW, ox = r - oy * OW;
                const int iy = oy >> 1, ix = ox >> 1;
                const int y1 = (oy & 1) ? min(iy + 1, LH - 1) : max(iy - 1, 0);
                const int x1 = (ox & 1) ? min(ix + 1, LW - 1) : max(ix - 1, 0);
                const float* lb = low + (int64_t)b * LH * LW * C + 8 * q;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    tap[i][0][u] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)iy * LW + ix) * C + 4 * u);
                    tap[i][1][u] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)iy * LW + x1) * C + 4 * u);
                    tap[i][2][u] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)y1 * LW + ix) * C + 4 * u);
                    tap[i][3][u] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)y1 * LW + x1) * C + 4 * u);
                }
            }
        }
    };
    if (wave < ngroups) load_raw(wave);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        // opaque per iteration: the LDS-resident weights do not depend on g; without this hipcc hoists every fragment
        // read out of the loop and keeps all of them in registers
        int wl = lane * 16;
        asm volatile("" : "+v"(wl));
        const char* w1l = lds + wl;
        const char* w2l = lds + W1_BYTES + wl;
        uh8 xh[KC1][NP], xl[KC1][NP];
        // XSKIP: the residual IS the block input, already in registers in the INPUT lane layout: the lanes of the output layout fetch it
        // from there (ds_bpermute) instead of reading x a second time -- by the counters that second read missed L2 half the time (FETCH
        // 1.6 GB for a 1.07 GB tensor: the 32-channel decoder block moved 3.2 GB where 2.15 are needed).  PRE = 2 always; PRE = 1 for 32
        // channels (the 64-channel instance has no 32 registers to keep the copy in)
        constexpr bool XSKIP = PRE == 2 || (PRE == 1 && C == 32);
        f32x4 xv[XSKIP ? NP : 1][2];                             // the block input x (input lane layout), kept for the residual
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (PRE == 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 r = 0.75f * (0.75f * tap[i][0][u] + 0.25f * tap[i][2][u]) + 0.25f * (0.75f * tap[i][1][u] + 0.25f * tap[i][3][u]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = act_up == 1 ? fmaxf(r[j], 0.f) : (act_up == 2 ? (r[j] > 0.f ? r[j] : alpha_up * r[j]) : r[j]);
                    xr[i][0][u] += r;
                }
            }
            if (XSKIP) {
#pragma unroll
                for (int u = 0; u < 2; ++u) xv[i][u] = xr[i][0][u];
            }
            if (PRE) {
                f32x4 v[KC1][2];
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < KC1; ++c)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        v[c][u] = xr[i][c][u] * dwv[c][u];
                        sum += v[c][u][0] + v[c][u][1] + v[c][u][2] + v[c][u][3];
                    }
                if (gamma) {
                    sum = UH_SUM_Q(sum);
                    const float mean = sum * (1.f / C);
                    float sq = 0.f;
#pragma unroll
                    for (int c = 0; c < KC1; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            v[c][u] = v[c][u] - mean;
                            sq += v[c][u][0] * v[c][u][0] + v[c][u][1] * v[c][u][1] + v[c][u][2] * v[c][u][2] + v[c][u][3] * v[c][u][3];
                        }
                    sq = UH_SUM_Q(sq);
                    const float rs = rsqrtf(sq * (1.f / C) + eps);
#pragma unroll
                    for (int c = 0; c < KC1; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) v[c][u] = v[c][u] * (gmv[c][u] * rs);
                }
#pragma unroll
                for (int c = 0; c < KC1; ++c) uh_split8(v[c][0], v[c][1], xh[c][i], xl[c][i]);
            } else {
#pragma unroll
                for (int c = 0; c < KC1; ++c) uh_split8(xr[i][c][0], xr[i][c][1], xh[c][i], xl[c][i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g + nwaves < ngroups) load_raw(g + nwaves);
        f32x4 sk[T2][NP];
        if (XSKIP) {
            // output tile t, lane (q, n): channels 16 t + 4 q .. + 3 = half (q & 1) of input-layout lane (2 t + (q >> 1), n)
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    const int srcl = 16 * (2 * t + (q >> 1)) + n;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a0 = __shfl(xv[i][0][j], srcl, 64), a1 = __shfl(xv[i][1][j], srcl, 64);
                        sk[t][i][j] = (q & 1) ? a1 : a0;
                    }
                }
        } else if (skip) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int64_t p = p0 + 16 * i + n;
                p = p < npix ? p : npix - 1;
#pragma unroll
                for (int t = 0; t < T2; ++t) sk[t][i] = *reinterpret_cast<const f32x4*>(skip + p * C + 16 * t + 4 * q);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc2[T2][NP];
        uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = p0 + 16 * i + n;
            if (p >= npix) continue;
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                f32x4 v = bf_acc_ready(acc2[t][i]) * m4[t];
                const int co = 16 * t + 4 * q;
                if (XSKIP || skip) v += sk[t][i];
                *reinterpret_cast<f32x4*>(out + p * C + co) = v;
            }
        }
    }
}

template <int C, int NP, int NT, int PRE>
static hipError_t uh_launch(const float* in, const float* skip, float* out, const void* packed, const float* mult, int64_t npix, int act,
                            float alpha, const float* dw, const float* gamma, float eps, hipStream_t s, const float* low = nullptr, int OH = 0,
                            int OW = 0, int act_up = 0, float alpha_up = 0.f)
{
    constexpr int LDS = 32 * C * C;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    const int wpb = NT / 64;
    int64_t grid = (ngroups + wpb - 1) / wpb;
    const int cap = LDS > 64 * 1024 ? 256 : 256 * 3;             // persistent: every workgroup loads the weights once
    if (grid > cap) grid = cap;
#define UH_LAUNCH(A)                                                                                                          \
    {                                                                                                                         \
        {                                                                                                                     \
            const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(uh_mlp_kernel<C, NP, A, NT, PRE>), LDS);         \
            if (e != hipSuccess) return e;                                                                                    \
        }                                                                                                                     \
        hipLaunchKernelGGL((uh_mlp_kernel<C, NP, A, NT, PRE>), dim3((int)grid), dim3(NT), LDS, s, in, skip, out, packed, mult, npix,   \
                           alpha, dw, gamma, eps, low, OH, OW, act_up, alpha_up);                                             \
    }
    switch (act) {
    case 0: UH_LAUNCH(0) break;
    case 1: UH_LAUNCH(1) break;
    case 2: UH_LAUNCH(2) break;
    case 3: UH_LAUNCH(3) break;
    default: return hipErrorInvalidValue;
    }
#undef UH_LAUNCH
    return hipGetLastError();
}

extern "C" int bf_op_convnext_mlp_h3(const float* in, const float* skip, float* out, const void* packed, const float* mult,
                                     int64_t npix, int C, int act, float alpha, void* stream)
{
    if (!in || !out || !packed || npix <= 0) return BF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)mult | (uintptr_t)skip) % 16) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (C == 32) e = uh_launch<32, UH_NP32, 256, 0>(in, skip, out, packed, mult, npix, act, alpha, nullptr, nullptr, 0.f, s);
    else if (C == 64) e = uh_launch<64, UH_NP64, 512, 0>(in, skip, out, packed, mult, npix, act, alpha, nullptr, nullptr, 0.f, s);
    else return BF_EUNSUPPORTED;
    if (e == hipErrorInvalidValue) return BF_EINVAL;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

// the first decoder block of a level, its input formed on load: x = enc + act_up(UpSampling2D(2, bilinear)(low)), out = x + ConvNextBlock(x)
// (1x1 depthwise, C = 32).  enc [B, OH, OW, 32], low [B, OH / 2, OW / 2, 32]; act_up: 0 linear, 1 relu, 2 leaky relu (alpha_up)
extern "C" int bf_op_convnext_block1_up_h3(const float* enc, const float* low, float* out, const float* dw, const float* ln_gamma, float eps,
                                           const void* packed, const float* mult, int B, int OH, int OW, int C, int act, float alpha,
                                           int act_up, float alpha_up, void* stream)
{
    if (!enc || !low || !out || !dw || !packed || B <= 0 || OH <= 0 || OW <= 0) return BF_EINVAL;
    if (C != 32 || (OH & 1) || (OW & 1) || act_up < 0 || act_up > 2 || (int64_t)B * OH * OW >= ((int64_t)1 << 31)) return BF_EUNSUPPORTED;
    if (((uintptr_t)enc | (uintptr_t)low | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)mult | (uintptr_t)dw | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    const hipError_t e = uh_launch<32, UH_NP32, 256, 2>(enc, nullptr, out, packed, mult, (int64_t)B * OH * OW, act, alpha, dw, ln_gamma, eps,
                                                        (hipStream_t)stream, low, OH, OW, act_up, alpha_up);
    if (e == hipErrorInvalidValue) return BF_EINVAL;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

// whole ConvNextBlock with a 1x1 depthwise convolution + the residual Add:
//   out = x + mult * (act(LayerNorm(x * dw) * gamma . W1) . W2)       dw [C] (DepthwiseConv2D 1x1 kernel), gamma [C] or NULL
extern "C" int bf_op_convnext_block1_h3(const float* x, float* out, const float* dw, const float* ln_gamma, float eps,
                                        const void* packed, const float* mult, int64_t npix, int C, int act, float alpha, void* stream)
{
    if (!x || !out || !dw || !packed || npix <= 0) return BF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)mult | (uintptr_t)dw | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (C == 32) e = uh_launch<32, UH_NP32, 256, 1>(x, x, out, packed, mult, npix, act, alpha, dw, ln_gamma, eps, s);
    else if (C == 64) e = uh_launch<64, UH_NP64, 512, 1>(x, x, out, packed, mult, npix, act, alpha, dw, ln_gamma, eps, s);
    else return BF_EUNSUPPORTED;
    if (e == hipErrorInvalidValue) return BF_EINVAL;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

