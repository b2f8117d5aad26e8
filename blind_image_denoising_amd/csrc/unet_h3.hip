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
#include "bf_common.h"
#include <math.h>
#include <string.h>

#ifndef UH_ROLE_ABLATE
#define UH_ROLE_ABLATE 0   // diagnostic builds of uh_enc32s_kernel: 1 consumers idle, 2 producers idle
#endif

typedef _Float16 uh8 __attribute__((ext_vector_type(8)));
typedef _Float16 uh4 __attribute__((ext_vector_type(4)));
typedef _Float16 uh2 __attribute__((ext_vector_type(2)));
#define UH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#ifndef UH_MLP_VALU_PER_MFMA
#define UH_MLP_VALU_PER_MFMA 3
#endif

// v - float(one half of the packed f16 pair hh) in one instruction (v_fma_mix_f32)
__device__ __forceinline__ float uh_sub_half(const float v, const unsigned hh, const bool high)
{
    float r;
    if (high) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    return r;
}

// 8 fp32 values -> hi / lo f16 fragments
__device__ __forceinline__ void uh_split8(const f32x4 a, const f32x4 b, uh8& hi, uh8& lo)
{
    const uh4 ha = __builtin_convertvector(a, uh4), hb = __builtin_convertvector(b, uh4);
    const unsigned p0 = __builtin_bit_cast(unsigned, (uh2){ha[0], ha[1]}), p1 = __builtin_bit_cast(unsigned, (uh2){ha[2], ha[3]});
    const unsigned p2 = __builtin_bit_cast(unsigned, (uh2){hb[0], hb[1]}), p3 = __builtin_bit_cast(unsigned, (uh2){hb[2], hb[3]});
    const f32x4 da = {uh_sub_half(a[0], p0, false), uh_sub_half(a[1], p0, true), uh_sub_half(a[2], p1, false), uh_sub_half(a[3], p1, true)};
    const f32x4 db = {uh_sub_half(b[0], p2, false), uh_sub_half(b[1], p2, true), uh_sub_half(b[2], p3, false), uh_sub_half(b[3], p3, true)};
    const uh4 la = __builtin_convertvector(da, uh4), lb = __builtin_convertvector(db, uh4);
    hi = (uh8){ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
    lo = (uh8){la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
}

// sum over the 8 consecutive lanes of a pixel in the depthwise layout (DPP butterfly: quad_perm xor 1, xor 2, half-row mirror)
template <int CTRL>
__device__ __forceinline__ float uh_dpp_add(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float uh_pixel_sum8(float v)
{
    v = uh_dpp_add<0xB1>(v);
    v = uh_dpp_add<0x4E>(v);
    return uh_dpp_add<0x141>(v);
}

template <int ACT>
__device__ __forceinline__ float uh_act(float v, float alpha)
{
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return fmaxf(v, alpha * v);                      // leaky relu, 0 <= alpha <= 1
    if (ACT == 3) {
        // exact (erf) GELU = v * Phi(v) without erff's two-branch polynomial (it doubled the time of the MLP kernels):
        // Phi(|v|) = 1 - h, Phi(-|v|) = h, h = 0.5 (a1 t + .. + a5 t^5) exp(-v^2 / 2), t = 1 / (1 + p |v| / sqrt 2)
        // (Abramowitz-Stegun 7.1.26).  Max |error| 4.2e-7 over [-12, 12] in fp32, the same as 0.5 v (1 + erff(v / sqrt 2)).
        const float t = __builtin_amdgcn_rcpf(fmaf(0.231641888f, fabsf(v), 1.f));
        float poly = fmaf(0.5307027145f, t, -0.7265760135f);
        poly = fmaf(poly, t, 0.7107068705f);
        poly = fmaf(poly, t, -0.142248368f);
        poly = fmaf(poly, t, 0.127414796f) * t;
        const float h = poly * __builtin_amdgcn_exp2f(v * v * -0.72134752044f);
        return v * (v >= 0.f ? 1.f - h : h);
    }
    return v;
}

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

__global__ __launch_bounds__(256) void uh_pack_mlp_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                          _Float16* __restrict__ dst, int C)
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
            const float v = w1[(32 * c + 8 * q + i) * H + 16 * t + m] * s1;
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
    hipLaunchKernelGGL(uh_pack_mlp_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w1, w2, (_Float16*)packed, C);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// the two GEMMs of the MLP on one wave's NP groups of 16 pixels: xh / xl = split B fragments of the input (K chunk c of 32
// channels), w1l / w2l = the lane's byte address inside the LDS fragment arrays, acc2 = C / 16 output tiles
template <int C, int NP, int ACT>
__device__ __forceinline__ void uh_mlp_core(const uh8 (&xh)[C / 32][NP], const uh8 (&xl)[C / 32][NP], const char* w1l, const char* w2l,
                                            const float inv1, const float alpha, f32x4 (&acc2)[C / 16][NP])
{
    constexpr int KC1 = C / 32, T1 = 4 * C / 16, KC2 = 4 * C / 32, T2 = C / 16;
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
        for (int i = 0; i < NP; ++i) acc2[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // GEMM1 of hidden tiles 2 c2, 2 c2 + 1 into h
    auto gemm1 = [&](const int c2, f32x4 (&h)[2][NP]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < NP; ++i) h[u][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < KC1; ++c) {
                const int f = (c * T1 + 2 * c2 + u) * 2;
                const uh8 ah = *reinterpret_cast<const uh8*>(w1l + f * 1024);
                const uh8 al = *reinterpret_cast<const uh8*>(w1l + (f + 1) * 1024);
#pragma unroll
                for (int i = 0; i < NP; ++i) h[u][i] = UH_MFMA(ah, xh[c][i], h[u][i]);
#pragma unroll
                for (int i = 0; i < NP; ++i) h[u][i] = UH_MFMA(al, xh[c][i], h[u][i]);
#pragma unroll
                for (int i = 0; i < NP; ++i) h[u][i] = UH_MFMA(ah, xl[c][i], h[u][i]);
            }
        }
    };
    // activation + split of chunk c2 (the lane's 8 hidden values are its B fragment), then GEMM2 with K chunk c2
    auto finish = [&](const int c2, f32x4 (&h)[2][NP]) {
        uh8 bh[NP], bl[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            // straight-line code from the MFMAs to here: hipcc pads the MFMA -> VALU hazard itself (bf_acc_ready is for reads
            // behind branches; its volatile s_nop would also pin the schedule)
            f32x4 v0 = h[0][i] * inv1, v1 = h[1][i] * inv1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = uh_act<ACT>(v0[r], alpha);
                v1[r] = uh_act<ACT>(v1[r], alpha);
            }
            uh_split8(v0, v1, bh[i], bl[i]);
        }
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            const int f = (c2 * T2 + t) * 2;
            const uh8 ah = *reinterpret_cast<const uh8*>(w2l + f * 1024);
            const uh8 al = *reinterpret_cast<const uh8*>(w2l + (f + 1) * 1024);
#pragma unroll
            for (int i = 0; i < NP; ++i) acc2[t][i] = UH_MFMA(ah, bh[i], acc2[t][i]);
#pragma unroll
            for (int i = 0; i < NP; ++i) acc2[t][i] = UH_MFMA(al, bh[i], acc2[t][i]);
#pragma unroll
            for (int i = 0; i < NP; ++i) acc2[t][i] = UH_MFMA(ah, bl[i], acc2[t][i]);
        }
    };
    // software pipeline: the matrix instructions of GEMM1 (chunk c2 + 1) are in the instruction stream before the
    // vector-ALU work of chunk c2 (scale, activation, hi/lo split) that depends on the PREVIOUS GEMM1, and the scheduler is
    // asked to interleave them (1 MFMA : UH_MLP_VALU_PER_MFMA VALU) so that a wave that is alone on its SIMD (the consumer
    // waves of uh_enc32s_kernel) keeps both pipes busy
    f32x4 hA[2][NP], hB[2][NP];
    gemm1(0, hA);
#pragma unroll
    for (int c2 = 0; c2 < KC2; ++c2) {
        if (c2 + 1 < KC2) {
            if (c2 & 1) gemm1(c2 + 1, hA);
            else gemm1(c2 + 1, hB);
        }
        if (c2 & 1) finish(c2, hB);
        else finish(c2, hA);
#pragma unroll
        for (int g = 0; g < 6 * KC1 * NP + 3 * T2 * NP; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, UH_MLP_VALU_PER_MFMA, 0);    // VALU
        }
    }
}

// PRE = 1: the whole ConvNextBlock with a 1x1 depthwise convolution (the decoder blocks, decoder_kernel_size 1): `in` is
// the block input x; t = LayerNorm(x * dw) * gamma is formed in registers (the 4 lanes (q, n) of a pixel hold all its
// channels: two cross-row shuffles per reduction) and never written; skip = x.
template <int C, int NP, int ACT, int NT, int PRE>
__global__ __launch_bounds__(NT, 512 / NT) void uh_mlp_kernel(const float* __restrict__ in, const float* __restrict__ skip, float* __restrict__ out,
                                                    const void* __restrict__ packed, const float* __restrict__ mult, int64_t npix,
                                                    float alpha, const float* __restrict__ dw, const float* __restrict__ gamma,
                                                    float eps)
{
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
#pragma unroll
        for (int i = 0; i < NP; ++i) {
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
                    sum += __shfl_xor(sum, 16, 64);
                    sum += __shfl_xor(sum, 32, 64);
                    const float mean = sum * (1.f / C);
                    float sq = 0.f;
#pragma unroll
                    for (int c = 0; c < KC1; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            v[c][u] = v[c][u] - mean;
                            sq += v[c][u][0] * v[c][u][0] + v[c][u][1] * v[c][u][1] + v[c][u][2] * v[c][u][2] + v[c][u][3] * v[c][u][3];
                        }
                    sq += __shfl_xor(sq, 16, 64);
                    sq += __shfl_xor(sq, 32, 64);
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
        if (skip) {
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
                if (skip) v += sk[t][i];
                *reinterpret_cast<f32x4*>(out + p * C + co) = v;
            }
        }
    }
}

template <int C, int NP, int NT, int PRE>
static hipError_t uh_launch(const float* in, const float* skip, float* out, const void* packed, const float* mult, int64_t npix, int act,
                            float alpha, const float* dw, const float* gamma, float eps, hipStream_t s)
{
    constexpr int LDS = 32 * C * C;
    const int64_t ngroups = (npix + 16 * NP - 1) / (16 * NP);
    const int wpb = NT / 64;
    int64_t grid = (ngroups + wpb - 1) / wpb;
    const int cap = LDS > 64 * 1024 ? 256 : 256 * 3;             // persistent: every workgroup loads the weights once
    if (grid > cap) grid = cap;
    static bool attr_done[4] = {false, false, false, false};
#define UH_LAUNCH(A)                                                                                                          \
    {                                                                                                                         \
        if (!attr_done[A]) {                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(uh_mlp_kernel<C, NP, A, NT, PRE>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS);                              \
            if (e != hipSuccess) return e;                                                                                    \
            attr_done[A] = true;                                                                                              \
        }                                                                                                                     \
        hipLaunchKernelGGL((uh_mlp_kernel<C, NP, A, NT, PRE>), dim3((int)grid), dim3(NT), LDS, s, in, skip, out, packed, mult, npix,   \
                           alpha, dw, gamma, eps);                                                                            \
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
    if (C == 32) e = uh_launch<32, 2, 256, 0>(in, skip, out, packed, mult, npix, act, alpha, nullptr, nullptr, 0.f, s);
    else if (C == 64) e = uh_launch<64, 2, 512, 0>(in, skip, out, packed, mult, npix, act, alpha, nullptr, nullptr, 0.f, s);
    else return BF_EUNSUPPORTED;
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
    if (C == 32) e = uh_launch<32, 2, 256, 1>(x, x, out, packed, mult, npix, act, alpha, dw, ln_gamma, eps, s);
    else if (C == 64) e = uh_launch<64, 2, 512, 1>(x, x, out, packed, mult, npix, act, alpha, dw, ln_gamma, eps, s);
    else return BF_EUNSUPPORTED;
    if (e == hipErrorInvalidValue) return BF_EINVAL;
    return e == hipSuccess ? BF_OK : BF_EHIP;
}

// ------------------------------------------------------------------------------------------
// Whole ENCODER ConvNextBlock (k x k depthwise, 32 channels) + residual Add in one kernel:
//   out = x + mult * (act(LayerNorm(dw_kxk(x)) * gamma . W1) . W2)
// A wave owns an 8-pixel-wide column strip and walks down it as uo_dwconv_ln_rows_kernel does (4 channels per lane,
// 8 lanes per pixel, k rotating accumulators, DPP LayerNorm).  Every 8 finished rows (64 pixels) are handed over through a
// wave-private LDS buffer to the matrix-core layout (lane (q, n): channels 8q..8q+7 of pixel n; a 16-pixel group = 2 rows
// x 8 columns) and go through the split-f16 MLP; the skip comes from the rows just read (cache hits).  The LayerNorm
// output and the hidden layer never reach HBM: x is read once (+ halo), out written once.
// ------------------------------------------------------------------------------------------
constexpr int UH_ENC_ROWS = 16;        // rows per tile
constexpr int UH_ROWBUF_F4 = 96;       // 16-byte elements of a strip row incl. halo (12 pixels x 8)
constexpr int UH_STG_PITCH = 36;       // floats per staged pixel (32 + 4: the 16 lanes of a fragment read spread over the banks)
template <int K, int ACT>
__global__ __launch_bounds__(256, 2) void uh_enc32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                          const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                          int W, float alpha)
{
    constexpr int C = 32, NP = 4, RAD = K / 2, T2 = 2, RB = 8;
    constexpr int W_BYTES = 32 * C * C;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += 256) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;            // matrix-core layout
    const int cl = lane & 7, pl = lane >> 3;           // depthwise layout: channels 4cl..4cl+3 of strip column pl
    float* stg = reinterpret_cast<float*>(lds + W_BYTES) + wave * (RB * 8 * UH_STG_PITCH);
    f32x4 m4[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
        m4[t] = (f32x4){inv2, inv2, inv2, inv2};
        if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
    }
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const int ty = (int)((tile / tiles_x) % tiles_y);
        const int64_t img = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
        const int x0 = tx * 32 + wave * 8, y0 = ty * UH_ENC_ROWS;
        if (x0 >= W) continue;                           // wave-uniform; no workgroup barrier inside the tile loop
        // ---- depthwise column walk state
        const int xd = x0 + pl;
        int xo[K];
        float xm[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int xx = xd + kx - RAD;
            xm[kx] = (xx >= 0 && xx < W) ? 1.f : 0.f;
            xo[kx] = min(max(xx, 0), W - 1) * C + 4 * cl;
        }
        auto load_row = [&](int yi, f32x4 (&v)[K]) {
            const float* row = x + (img + (int64_t)min(max(yi, 0), H - 1) * W) * C;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) v[kx] = *reinterpret_cast<const f32x4*>(row + xo[kx]);
        };
        f32x4 acc[K], vn[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        load_row(y0 - RAD, vn);
        int yi = y0 - RAD;
        for (int batch = 0; batch < UH_ENC_ROWS / RB; ++batch) {
            const int yb = y0 + batch * RB;              // first output row of the batch
            if (yb >= H) break;
            {
                // depthwise weights: reloaded per batch (opaque pointer) so that they are not live across the matrix phase
                const float* wp = dww;
                asm volatile("" : "+s"(wp));
                f32x4 wk[K * K];
#pragma unroll
                for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(wp + i * C + 4 * cl);
                for (; yi < yb + RB + RAD; ++yi) {
                    f32x4 v[K];
                    const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) v[kx] = vn[kx] * (xm[kx] * ym);
                    load_row(yi + 1, vn);
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) acc[ky] += wk[ky * K + kx] * v[kx];
                    const int yo = yi - RAD;
                    if (yo >= yb) {
                        f32x4 r = acc[K - 1];
                        if (gamma) {
                            const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                            const f32x4 d = r - mean;
                            const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                            r = d * (gm * rsqrtf(var + eps));
                        }
                        *reinterpret_cast<f32x4*>(stg + ((yo - yb) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                    }
#pragma unroll
                    for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
                    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            // ---- hand-over inside the wave: the staged rows are read by other lanes than wrote them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int wl = lane * 16;
            asm volatile("" : "+v"(wl));
            const char* w1l = lds + wl;
            const char* w2l = lds + 16 * C * C + wl;
            uh8 xh[1][NP], xl[1][NP];
            f32x4 sk[T2][NP];
            int64_t pix[NP];
            bool ok[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int s = 16 * i + n;                 // staged pixel: row 2i + n / 8, column n % 8
                const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                const int py = yb + 2 * i + (n >> 3), px = x0 + (n & 7);
                ok[i] = py < H && px < W;
                pix[i] = img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
#pragma unroll
                for (int t = 0; t < T2; ++t) sk[t][i] = *reinterpret_cast<const f32x4*>(x + pix[i] * C + 16 * t + 4 * q);
            }
            f32x4 acc2[T2][NP];
            uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (!ok[i]) continue;
#pragma unroll
                for (int t = 0; t < T2; ++t)
                    *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
            }
            // the next batch overwrites the staging buffer
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// workgroup barrier that publishes LDS writes (staging) but leaves vector-memory operations in flight: hipcc puts
// "s_waitcnt vmcnt(0)" in front of every barrier it can see, which here would drain the producers' row ring and make the
// consumers wait for their stores to reach memory at every step (measured: producers alone 606 us, consumers alone 789 us,
// together 1076 us with __syncthreads()).  The wait goes through the builtin so that hipcc's own bookkeeping sees it.
__device__ __forceinline__ void uh_step_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0); vmcnt / expcnt untouched (gfx9 encoding)
    asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// Wave-specialised form of the kernel above: waves 0-3 of a 512-thread workgroup only PRODUCE (depthwise walk + LayerNorm
// of their 8-pixel strip into a staging buffer), waves 4-7 only CONSUME (split-f16 MLP, skip, store) the batch of 8 rows
// the producers finished one step earlier; one workgroup barrier per step flips the double-buffered staging area.
// A SIMD hosts one wave of each kind, so the vector-ALU stream of the depthwise phase and the matrix stream of the MLP
// overlap instead of alternating inside one wave (ablations of uh_enc32_kernel: data movement 696 us + depthwise 265 us
// + matrix 280 us, additive), and the depthwise weights stay in registers.
// ------------------------------------------------------------------------------------------
template <int K, int ACT>
__global__ __launch_bounds__(512, 1) void uh_enc32s_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           const float* __restrict__ dww, const float* __restrict__ gamma, float eps,
                                                           const void* __restrict__ packed, const float* __restrict__ mult, int B, int H,
                                                           int W, float alpha)
{
    constexpr int C = 32, NP = 4, RAD = K / 2, T2 = 2, RB = 8;
    constexpr int W_BYTES = 32 * C * C, STG_FLOATS = RB * 8 * UH_STG_PITCH;       // one strip of one batch
    extern __shared__ __attribute__((aligned(16))) char lds[];
    {
        const int4* src = reinterpret_cast<const int4*>(packed);
        int4* dstv = reinterpret_cast<int4*>(lds);
        for (int i = threadIdx.x; i < W_BYTES / 16; i += 512) dstv[i] = src[i];
    }
    const float* aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + W_BYTES);
    const float inv1 = aux[0], inv2 = aux[1];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave < 4;
    const int strip = wave & 3;
    float* stg_base = reinterpret_cast<float*>(lds + W_BYTES) + strip * STG_FLOATS;     // + (step & 1) * 4 * STG_FLOATS
    const int tiles_x = (W + 31) / 32, tiles_y = (H + UH_ENC_ROWS - 1) / UH_ENC_ROWS;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    const int64_t my_tiles = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t nsteps = 2 * my_tiles;                   // batches of RB rows; UH_ENC_ROWS / RB = 2 per tile
    static_assert(UH_ENC_ROWS == 2 * RB, "two batches per tile");

    // The two roles are two separate loops (wave-uniform branch) with the same number of barrier arrivals per wave: as one
    // loop hipcc kept the producers' 100 weight registers AND the consumers' matrix state live together (spills).
    if (producer) {
        // ---- depthwise layout: channels 4cl..4cl+3 of strip column pl
        const int cl = lane & 7, pl = lane >> 3;
        // A strip row (8 + 2 RAD pixels x 32 channels, contiguous in memory) is fetched by the whole wave with one 16-byte
        // and one 8-byte load per lane, PD rows ahead of its use (register ring with static slots: the row loop is
        // unrolled by PD), passed through a wave-private LDS row buffer and read back as the k taps of every lane.
        // A SIMD hosts ONE producer wave, so nothing else hides its load latency, and what counts is UNIQUE bytes in
        // flight: with the k taps loaded directly one row ahead (5 KB per wave in flight, 1.5 KB of it unique) every row
        // cost a memory round trip (~0.9 us) and the kernel was no faster than uh_enc32_kernel.
        constexpr int PD = 4, ROWF4 = (8 + 2 * RAD) * 8, HALVES = 2 * (ROWF4 - 64);
        static_assert((RB + 2 * RAD) % PD == 0 || K == 3, "row groups");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        struct RowRegs { f32x4 a; f32x2 b; };
        f32x4 wk[K * K], gm = {1.f, 1.f, 1.f, 1.f}, acc[K];
        RowRegs ring[PD];
        int go0 = 0, go1 = 0;
        float gm0 = 0.f, gm1 = 0.f;
        int yi = 0, py0 = 0;
        int64_t pimg = 0;
        bool plive = false;
        const int h1 = lane & (HALVES - 1);                  // half-element (8 bytes) of the row tail this lane fetches
        f32x4* rowbuf = reinterpret_cast<f32x4*>(lds + W_BYTES + 2 * 4 * STG_FLOATS * 4) + strip * (2 * UH_ROWBUF_F4);
#pragma unroll
        for (int i = 0; i < K * K; ++i) wk[i] = *reinterpret_cast<const f32x4*>(dww + i * C + 4 * cl);
        if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + 4 * cl);
        auto issue = [&](int yy, RowRegs& r) {
            const float* row = x + (pimg + (int64_t)min(max(yy, 0), H - 1) * W) * C;
            r.a = *reinterpret_cast<const f32x4*>(row + go0);
            r.b = *reinterpret_cast<const f32x2*>(row + go1);
        };
        for (int64_t step = 0; step <= nsteps; ++step) {
            if (step < nsteps) {
                const int batch = (int)(step & 1);
                float* stg = stg_base + (step & 1) * 4 * STG_FLOATS;
                if (batch == 0) {                           // new tile: coordinates, column masks, first rows
                    const int64_t tile = blockIdx.x + (step >> 1) * gridDim.x;
                    const int tx = (int)(tile % tiles_x);
                    const int ty = (int)((tile / tiles_x) % tiles_y);
                    pimg = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
                    const int x0 = tx * 32 + strip * 8;
                    py0 = ty * UH_ENC_ROWS;
                    plive = x0 < W;
                    const int xg0 = x0 - RAD + (lane >> 3), xg1 = x0 - RAD + 8 + (h1 >> 4);
                    gm0 = (xg0 >= 0 && xg0 < W) ? 1.f : 0.f;
                    gm1 = (xg1 >= 0 && xg1 < W) ? 1.f : 0.f;
                    go0 = min(max(xg0, 0), W - 1) * C + 4 * cl;
                    go1 = min(max(xg1, 0), W - 1) * C + 2 * (h1 & 15);
#pragma unroll
                    for (int j = 0; j < K; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    yi = py0 - RAD;
                    if (plive) {
#pragma unroll
                        for (int d = 0; d < PD; ++d) issue(yi + d, ring[d]);
                    }
                }
                const int yb = py0 + batch * RB;
                if (plive && yb < H && !(UH_ROLE_ABLATE & 2)) {
                    auto row_body = [&](RowRegs& slot, f32x4* rb) {
                        // row yi: registers -> row buffer (column mask applied here), then request row yi + PD
                        rb[lane] = slot.a * gm0;
                        if (lane < HALVES) reinterpret_cast<f32x2*>(rb + 64)[lane] = slot.b * gm1;
                        issue(yi + PD, slot);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const float ym = (yi >= 0 && yi < H) ? 1.f : 0.f;
                        f32x4 v[K];
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) v[kx] = rb[(pl + kx) * 8 + cl] * ym;
#pragma unroll
                        for (int ky = 0; ky < K; ++ky)
#pragma unroll
                            for (int kx = 0; kx < K; ++kx) acc[ky] += wk[ky * K + kx] * v[kx];
                        const int yo = yi - RAD;
                        if (yo >= yb) {
                            f32x4 r = acc[K - 1];
                            if (gamma) {
                                const float mean = uh_pixel_sum8(r[0] + r[1] + r[2] + r[3]) * (1.f / C);
                                const f32x4 d = r - mean;
                                const float var = uh_pixel_sum8(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / C);
                                r = d * (gm * rsqrtf(var + eps));
                            }
                            *reinterpret_cast<f32x4*>(stg + ((yo - yb) * 8 + pl) * UH_STG_PITCH + 4 * cl) = r;
                        }
#pragma unroll
                        for (int j = K - 1; j > 0; --j) acc[j] = acc[j - 1];
                        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        ++yi;
                    };
                    // rows per batch: RB + 2 RAD, then RB; K = 3 (10 rows) ends its first batch on a half group
                    while (yi < yb + RB + RAD) {
                        row_body(ring[0], rowbuf);
                        row_body(ring[1], rowbuf + UH_ROWBUF_F4);
                        if (K == 3 && yi >= yb + RB + RAD) {          // rotate the ring by two so that slot 0 is the next row
                            const RowRegs t0 = ring[0], t1 = ring[1];
                            ring[0] = ring[2]; ring[1] = ring[3]; ring[2] = t0; ring[3] = t1;
                            break;
                        }
                        row_body(ring[2], rowbuf);
                        row_body(ring[3], rowbuf + UH_ROWBUF_F4);
                    }
                }
            }
            uh_step_barrier();
        }
    } else {
        // ---- matrix-core layout
        const int q = lane >> 4, n = lane & 15;
        f32x4 m4[T2];
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            m4[t] = (f32x4){inv2, inv2, inv2, inv2};
            if (mult) m4[t] *= *reinterpret_cast<const f32x4*>(mult + 16 * t + 4 * q);
        }
        for (int64_t step = 0; step <= nsteps; ++step) {
            if (step >= 1) {
            const int64_t cs = step - 1;                    // the batch the producers finished in the previous step
            const int batch = (int)(cs & 1);
            const float* stg = stg_base + (cs & 1) * 4 * STG_FLOATS;
            const int64_t tile = blockIdx.x + (cs >> 1) * gridDim.x;
            const int tx = (int)(tile % tiles_x);
            const int ty = (int)((tile / tiles_x) % tiles_y);
            const int64_t img = (tile / ((int64_t)tiles_x * tiles_y)) * H * W;
            const int x0 = tx * 32 + strip * 8, yb = ty * UH_ENC_ROWS + batch * RB;
            if (x0 < W && yb < H && !(UH_ROLE_ABLATE & 1)) {
                int wl = lane * 16;
                asm volatile("" : "+v"(wl));
                const char* w1l = lds + wl;
                const char* w2l = lds + 16 * C * C + wl;
                uh8 xh[1][NP], xl[1][NP];
                f32x4 sk[T2][NP];
                int64_t pix[NP];
                bool ok[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int s = 16 * i + n;             // staged pixel: row 2i + n / 8, column n % 8
                    const float* sp = stg + s * UH_STG_PITCH + 8 * q;
                    uh_split8(*reinterpret_cast<const f32x4*>(sp), *reinterpret_cast<const f32x4*>(sp + 4), xh[0][i], xl[0][i]);
                    const int py = yb + 2 * i + (n >> 3), px = x0 + (n & 7);
                    ok[i] = py < H && px < W;
                    pix[i] = img + (int64_t)min(py, H - 1) * W + min(px, W - 1);
#pragma unroll
                    for (int t = 0; t < T2; ++t) sk[t][i] = *reinterpret_cast<const f32x4*>(x + pix[i] * C + 16 * t + 4 * q);
                }
                f32x4 acc2[T2][NP];
                uh_mlp_core<C, NP, ACT>(xh, xl, w1l, w2l, inv1, alpha, acc2);
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    if (!ok[i]) continue;
#pragma unroll
                    for (int t = 0; t < T2; ++t)
                        *reinterpret_cast<f32x4*>(out + pix[i] * C + 16 * t + 4 * q) = bf_acc_ready(acc2[t][i]) * m4[t] + sk[t][i];
                }
            }
            }
            uh_step_barrier();
        }
    }
}

static int g_uh_enc_variant = 1;     // 1 wave-specialised (default), 0 one kind of wave (A/B, tests)
extern "C" int bf_op_set_variant(const char* key, int value)
{
    if (key && !strcmp(key, "enc32")) { g_uh_enc_variant = value ? 1 : 0; return BF_OK; }
    return BF_EINVAL;
}

extern "C" int bf_op_convnext_block_h3(const float* x, float* out, const float* dw, int k, const float* ln_gamma, float eps,
                                       const void* packed, const float* mult, int B, int H, int W, int C, int act, float alpha,
                                       void* stream)
{
    if (!x || !out || !dw || !packed || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)packed | (uintptr_t)mult | (uintptr_t)dw | (uintptr_t)ln_gamma) % 16) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    if (C != 32 || (k != 3 && k != 5) || act < 0 || act > 3) return BF_EUNSUPPORTED;
    if (x == out) return BF_EINVAL;                      // neighbouring strips read the halo of this one
    hipStream_t s = (hipStream_t)stream;
    const int64_t ntiles = (int64_t)B * ((H + UH_ENC_ROWS - 1) / UH_ENC_ROWS) * ((W + 31) / 32);
    if (g_uh_enc_variant == 1) {
        constexpr int LDS_S = 32 * 32 * 32 + 2 * 4 * 8 * 8 * UH_STG_PITCH * 4 + 4 * 2 * UH_ROWBUF_F4 * 16;
        const int grid_s = (int)(ntiles < 256 ? ntiles : 256);
        static bool attr_s[2][4] = {};
#define UH_ENCS(KK, A)                                                                                                         \
    {                                                                                                                          \
        if (!attr_s[KK == 5][A]) {                                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(uh_enc32s_kernel<KK, A>),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS_S) != hipSuccess)                           \
                return BF_EHIP;                                                                                                \
            attr_s[KK == 5][A] = true;                                                                                         \
        }                                                                                                                      \
        hipLaunchKernelGGL((uh_enc32s_kernel<KK, A>), dim3(grid_s), dim3(512), LDS_S, s, x, out, dw, ln_gamma, eps, packed, mult, B, H,  \
                           W, alpha);                                                                                          \
    }
#define UH_ENCS_K(KK)                                                                                                          \
    switch (act) {                                                                                                             \
    case 0: UH_ENCS(KK, 0) break;                                                                                              \
    case 1: UH_ENCS(KK, 1) break;                                                                                              \
    case 2: UH_ENCS(KK, 2) break;                                                                                              \
    default: UH_ENCS(KK, 3) break;                                                                                             \
    }
        if (k == 5) { UH_ENCS_K(5) } else { UH_ENCS_K(3) }
#undef UH_ENCS_K
#undef UH_ENCS
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
    }
    constexpr int LDS = 32 * 32 * 32 + 4 * 8 * 8 * UH_STG_PITCH * 4;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);
    static bool attr_done[2][4] = {};
#define UH_ENC(KK, A)                                                                                                          \
    {                                                                                                                          \
        if (!attr_done[KK == 5][A]) {                                                                                          \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(uh_enc32_kernel<KK, A>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    LDS) != hipSuccess)                                                                        \
                return BF_EHIP;                                                                                                \
            attr_done[KK == 5][A] = true;                                                                                      \
        }                                                                                                                      \
        hipLaunchKernelGGL((uh_enc32_kernel<KK, A>), dim3(grid), dim3(256), LDS, s, x, out, dw, ln_gamma, eps, packed, mult, B, H, W, \
                           alpha);                                                                                             \
    }
#define UH_ENC_K(KK)                                                                                                           \
    switch (act) {                                                                                                             \
    case 0: UH_ENC(KK, 0) break;                                                                                               \
    case 1: UH_ENC(KK, 1) break;                                                                                               \
    case 2: UH_ENC(KK, 2) break;                                                                                               \
    default: UH_ENC(KK, 3) break;                                                                                              \
    }
    if (k == 5) { UH_ENC_K(5) } else { UH_ENC_K(3) }
#undef UH_ENC_K
#undef UH_ENC
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
