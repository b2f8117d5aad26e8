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
#ifndef UH_LN_PERMLANE
#define UH_LN_PERMLANE 1           // LayerNorm sums over the four lanes of a pixel: 1 = v_permlane swaps on the vector ALU, 0 = ds_bpermute
#endif
#if UH_LN_PERMLANE
#define UH_SUM_Q(v) uh_sum_q(v)
#else
__device__ __forceinline__ float uh_sum_q_shfl(float v)
{
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
#define UH_SUM_Q(v) uh_sum_q_shfl(v)
#endif
#ifndef UH_CHAIN_NT
#define UH_CHAIN_NT 1024           // threads of uh_chain32_kernel's workgroup (one per CU: its LDS holds up to 96 KB of weights); 122 registers
#endif
#ifndef UH_CHAIN_NT_UP
#define UH_CHAIN_NT_UP 512         // ... of the instance that forms the level's node on load (64 more registers for the taps)
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

// ------------------------------------------------------------------------------------------
// A CHAIN of decoder ConvNext blocks in one kernel (round 4).  With decoder_kernel_size 1 (configs/unet_laplacian_v5.json and the trained
// v5.6 archive) a decoder block is pixel-wise -- x -> x + mult * W2 act(W1 LayerNorm(x * dw)) -- so the `width` blocks of a level need
// no neighbour and no round trip through memory between them: the 32-channel level 0 of a batch-32 512 x 512 forward read and wrote its
// 1.07 GB map three times (three launches, 0.84 + 2 x 0.53 ms).  Here a wave carries its pixels through all NB blocks:
//   * the pixel's 32 channels live in the OUTPUT lane layout from the load on (lane (q, n): channels 16 t + 4 q .. + 3 of tile t = 0, 1):
//     per-channel work (depthwise scale, LayerNorm, multiplier, residual) does not care about the order, the accumulators of a block ARE
//     the next block's input, and the first GEMM of every block takes them as its B fragment because W1 is packed in that K order
//     (bf_op_pack_mlp_h3_chain) -- no shuffle, no second read for the residual;
//   * the weights of all NB blocks sit in LDS (NB x 32 KB, one workgroup of NT threads per CU);
//   * UP: the node in front of the first block, enc + act_up(bilinear x2 of low), is formed from five loads per pixel slice as in
//     uh_mlp_kernel<.., PRE = 2>.
// ------------------------------------------------------------------------------------------
struct UhChainArgs {
    const float* in;                 // [npix][32]: the block input, or the encoder's skip map (UP)
    const float* low;                // UP: [B][OH / 2][OW / 2][32]
    float* out;
    int64_t npix;
    int OH, OW, act_up;
    float alpha_up, alpha, eps;
    const void* packed[3];           // bf_op_pack_mlp_h3_chain
    const float* dw[3];              // [32] 1x1 depthwise kernels
    const float* gamma[3];           // [32] LayerNorm gammas, or NULL (no LayerNorm)
    const float* mult[3];            // [32] channel multipliers, or NULL
};

template <int ACT, int NB, bool UP, int NT>
__global__ __launch_bounds__(NT, 1) void uh_chain32_kernel(const UhChainArgs a)
{
    constexpr int C = 32, NP = 2, T2 = 2, W_BYTES = 32 * C * C, VEC_OFF = NB * W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* vec = reinterpret_cast<float*>(lds + VEC_OFF);          // per block: dw[32] | gamma[32] | mult / s2 [32]
    float inv1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int4* src = reinterpret_cast<const int4*>(a.packed[b]);
        int4* dstv = reinterpret_cast<int4*>(lds + b * W_BYTES);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += NT) dstv[i] = src[i];
        const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.packed[b]) + W_BYTES);
        inv1[b] = aux[0];
        if (threadIdx.x < 32) {
            vec[b * 96 + threadIdx.x] = a.dw[b][threadIdx.x];
            vec[b * 96 + 32 + threadIdx.x] = a.gamma[b] ? a.gamma[b][threadIdx.x] : 1.f;
            vec[b * 96 + 64 + threadIdx.x] = aux[1] * (a.mult[b] ? a.mult[b][threadIdx.x] : 1.f);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (NT / 64);
    const int64_t ngroups = (a.npix + 16 * NP - 1) / (16 * NP);
    f32x4 xr[NP][T2];
    f32x4 tap[UP ? NP : 1][UP ? 4 : 1][T2];
    auto load_raw = [&](int64_t gg) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int64_t p = gg * 16 * NP + 16 * i + n;
            p = p < a.npix ? p : a.npix - 1;                      // tail: clamp the read, predicate the store
            const float* src = a.in + p * C + 4 * q;
#pragma unroll
            for (int t = 0; t < T2; ++t) xr[i][t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
            if (UP) {
                const int LH = a.OH >> 1, LW = a.OW >> 1;
                const int hw = a.OH * a.OW;
                const int b = (int)(p / hw), r = (int)(p - (int64_t)b * hw);
                const int oy = r / a.OW, ox = r - oy * a.OW;
                const int iy = oy >> 1, ix = ox >> 1;
                const int y1 = (oy & 1) ? min(iy + 1, LH - 1) : max(iy - 1, 0);
                const int x1 = (ox & 1) ? min(ix + 1, LW - 1) : max(ix - 1, 0);
                const float* lb = a.low + (int64_t)b * LH * LW * C + 4 * q;
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    tap[i][0][t] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)iy * LW + ix) * C + 16 * t);
                    tap[i][1][t] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)iy * LW + x1) * C + 16 * t);
                    tap[i][2][t] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)y1 * LW + ix) * C + 16 * t);
                    tap[i][3][t] = *reinterpret_cast<const f32x4*>(lb + ((int64_t)y1 * LW + x1) * C + 16 * t);
                }
            }
        }
    };
    if (wave < ngroups) load_raw(wave);
    for (int64_t g = wave; g < ngroups; g += nwaves) {
        const int64_t p0 = g * 16 * NP;
        f32x4 x[NP][T2];
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                x[i][t] = xr[i][t];
                if (UP) {
                    f32x4 r = 0.75f * (0.75f * tap[i][0][t] + 0.25f * tap[i][2][t]) + 0.25f * (0.75f * tap[i][1][t] + 0.25f * tap[i][3][t]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[j] = a.act_up == 1 ? fmaxf(r[j], 0.f) : (a.act_up == 2 ? (r[j] > 0.f ? r[j] : a.alpha_up * r[j]) : r[j]);
                    x[i][t] += r;
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int64_t gn = g + nwaves;
            load_raw(gn < ngroups ? gn : g);                      // unconditional (a branch around loads drains the queue where it joins)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            // opaque per block and iteration: without it hipcc hoists every fragment read out of the loops and keeps them in registers
            int wl = lane * 16 + b * W_BYTES;
            asm volatile("" : "+v"(wl));
            const char* w1l = lds + wl;
            const char* w2l = w1l + 16 * C * C;
            const float* vb = vec + b * 96 + 4 * q;
            uh8 xh[1][NP], xl[1][NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                f32x4 v[T2];
                float sum = 0.f;
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    v[t] = x[i][t] * *reinterpret_cast<const f32x4*>(vb + 16 * t);
                    sum += v[t][0] + v[t][1] + v[t][2] + v[t][3];
                }
                if (a.gamma[b]) {
                    sum = UH_SUM_Q(sum);
                    const float mean = sum * (1.f / C);
                    float sq = 0.f;
#pragma unroll
                    for (int t = 0; t < T2; ++t) {
                        v[t] = v[t] - mean;
                        sq += v[t][0] * v[t][0] + v[t][1] * v[t][1] + v[t][2] * v[t][2] + v[t][3] * v[t][3];
                    }
                    sq = UH_SUM_Q(sq);
                    const float rs = rsqrtf(sq * (1.f / C) + a.eps);
#pragma unroll
                    for (int t = 0; t < T2; ++t) v[t] = v[t] * (*reinterpret_cast<const f32x4*>(vb + 32 + 16 * t) * rs);
                }
                uh_split8(v[0], v[1], xh[0][i], xl[0][i]);
            }
            f32x4 acc2[T2][NP];
            uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1[b], a.alpha, acc2);
#pragma unroll
            for (int i = 0; i < NP; ++i)
#pragma unroll
                for (int t = 0; t < T2; ++t) x[i][t] = bf_acc_ready(acc2[t][i]) * *reinterpret_cast<const f32x4*>(vb + 64 + 16 * t) + x[i][t];
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int64_t p = p0 + 16 * i + n;
            if (p >= a.npix) continue;
#pragma unroll
            for (int t = 0; t < T2; ++t) *reinterpret_cast<f32x4*>(a.out + p * C + 16 * t + 4 * q) = x[i][t];
        }
    }
}

// nblocks (1..3) pixel-wise ConvNext blocks (1x1 depthwise) of 32 channels in one kernel; low != NULL: the first block's input is
// x + act_up(UpSampling2D(2, "bilinear")(low)) (x = the encoder's skip map [B, OH, OW, 32], low [B, OH / 2, OW / 2, 32]).  packed[b] from
// bf_op_pack_mlp_h3_chain; dw[b] [32]; gamma[b] / mult[b] [32] or NULL.  out may alias x.
extern "C" int bf_op_convnext_chain32_h3(const float* x, const float* low, float* out, int nblocks, const void* const* packed,
                                         const float* const* dw, const float* const* gamma, const float* const* mult, float eps, int B,
                                         int OH, int OW, int act, float alpha, int act_up, float alpha_up, void* stream)
{
    if (!x || !out || !packed || !dw || !gamma || !mult || nblocks < 1 || nblocks > 3 || B <= 0 || OH <= 0 || OW <= 0) return BF_EINVAL;
    if ((int64_t)B * OH * OW >= ((int64_t)1 << 31) || act < 0 || act > 3) return BF_EUNSUPPORTED;
    if (low && ((OH & 1) || (OW & 1) || act_up < 0 || act_up > 2)) return BF_EUNSUPPORTED;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    UhChainArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.low = low; a.out = out; a.npix = (int64_t)B * OH * OW; a.OH = OH; a.OW = OW; a.act_up = act_up; a.alpha_up = alpha_up;
    a.alpha = alpha; a.eps = eps;
    uintptr_t al = (uintptr_t)x | (uintptr_t)low | (uintptr_t)out;
    for (int b = 0; b < nblocks; ++b) {
        if (!packed[b] || !dw[b]) return BF_EINVAL;
        a.packed[b] = packed[b]; a.dw[b] = dw[b]; a.gamma[b] = gamma[b]; a.mult[b] = mult[b];
        al |= (uintptr_t)packed[b];
    }
    if (al % 16) return BF_EINVAL;
    const int lds = nblocks * 32 * 32 * 32 + nblocks * 96 * 4;
    const int64_t ngroups = (a.npix + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
#define UH_CHAIN(A, N_, U_)                                                                                                   \
    {                                                                                                                         \
        constexpr int NT = U_ ? UH_CHAIN_NT_UP : UH_CHAIN_NT;                                                                 \
        int64_t grid = (ngroups + NT / 64 - 1) / (NT / 64);                                                                   \
        if (grid > 256) grid = 256;                                /* persistent: one workgroup per CU, the weights loaded once */ \
        if (bf_set_max_lds(reinterpret_cast<const void*>(uh_chain32_kernel<A, N_, U_, NT>), lds) != hipSuccess) return BF_EHIP; \
        hipLaunchKernelGGL((uh_chain32_kernel<A, N_, U_, NT>), dim3((int)grid), dim3(NT), lds, s, a);                         \
    }
#define UH_CHAIN_N(A)                                                                                                         \
    if (low) { if (nblocks == 1) UH_CHAIN(A, 1, true) else if (nblocks == 2) UH_CHAIN(A, 2, true) else UH_CHAIN(A, 3, true) }   \
    else { if (nblocks == 1) UH_CHAIN(A, 1, false) else if (nblocks == 2) UH_CHAIN(A, 2, false) else UH_CHAIN(A, 3, false) }
    switch (act) {
    case 0: UH_CHAIN_N(0) break;
    case 1: UH_CHAIN_N(1) break;
    case 2: UH_CHAIN_N(2) break;
    default: UH_CHAIN_N(3) break;
    }
#undef UH_CHAIN_N
#undef UH_CHAIN
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
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

