// Shared device code of the split-f16 ConvNext kernels (unet_h3.hip: MLP / decoder block kernels and weight packing;
// unet_h3_enc.hip: fused encoder block kernels).  Two translation units so that they compile side by side.
#pragma once
#include "bf_common.h"
#include <math.h>
#include <string.h>

#ifndef UH_ROLE_ABLATE
#define UH_ROLE_ABLATE 0   // diagnostic builds of uh_enc32s / uh_enc32u_kernel (timing only, results wrong): 1 consumers idle, 2 producers
                           // idle; consumers: 4 no skip loads / output stores, 8 no scale / activation / split of the hidden layer,
                           // 16 no MFMAs (the operand reads stay)
#endif

typedef _Float16 uh8 __attribute__((ext_vector_type(8)));
typedef _Float16 uh4 __attribute__((ext_vector_type(4)));
typedef _Float16 uh2 __attribute__((ext_vector_type(2)));
#define UH_MFMA_REAL(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define UH_MFMA(a, b, c) uh_mfma_maybe((a), (b), (c))
#ifndef UH_MLP_VALU_PER_MFMA
#define UH_MLP_VALU_PER_MFMA 3
#endif

__device__ __forceinline__ f32x4 uh_mfma_maybe(const uh8 a, const uh8 b, f32x4 c)
{
#if UH_ROLE_ABLATE & 16
    c[0] += (float)a[0] + (float)b[0];
    return c;
#else
    return UH_MFMA_REAL(a, b, c);
#endif
}

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

// v + (v of lane ^ 16) + (v of lane ^ 32) + (v of lane ^ 48): the sum over the four lanes (q = 0..3) that hold one pixel in the matrix-core
// layouts, on the vector ALU (v_permlane16_swap / v_permlane32_swap, gfx950) instead of two ds_bpermute round trips through the LDS
// crossbar (unet_ops.hip: uo_sum_q).  swap(a, b): rows 1, 3 (16-lane groups) of a <-> rows 0, 2 of b.
__device__ __forceinline__ float uh_sum_q(float v)
{
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float s = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);          // rows 0, 1: r0 + r1 ; rows 2, 3: r2 + r3
    a = __builtin_bit_cast(unsigned, s); b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // lanes 32..63 of a <-> lanes 0..31 of b
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}

template <int ACT>
__device__ __forceinline__ float uh_act(float v, float alpha)
{
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return fmaxf(v, alpha * v);                      // leaky relu, 0 <= alpha <= 1
    if (ACT == 3) {
        return bf_gelu(v);                                         // bf_common.h: erf form, one transcendental, no branch
    }
    return v;
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
#if UH_ROLE_ABLATE & 8
            bh[i] = __builtin_bit_cast(uh8, h[0][i]);
            bl[i] = __builtin_bit_cast(uh8, h[1][i]);
#else
            f32x4 v0 = h[0][i] * inv1, v1 = h[1][i] * inv1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] = uh_act<ACT>(v0[r], alpha);
                v1[r] = uh_act<ACT>(v1[r], alpha);
            }
            uh_split8(v0, v1, bh[i], bl[i]);
#endif
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
