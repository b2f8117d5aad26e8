// uh_chain32_kernel: the pixel-wise decoder blocks of a 32-channel unet_laplacian level (and the node in front of them) in one kernel.
// A translation unit of its own: its 24 template instances compile beside unet_h3.hip's instead of behind them.
#include "unet_h3_core.h"
#ifndef UH_CHAIN_NT
#define UH_CHAIN_NT 1024           // threads of uh_chain32_kernel's workgroup (one per CU: its LDS holds up to 96 KB of weights); 122 registers
#endif
#ifndef UH_CHAIN_NT_UP
#define UH_CHAIN_NT_UP 512         // ... of the instance that forms the level's node on load (64 more registers for the taps)
#endif

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
    case 0: return BF_EUNSUPPORTED;                                 // a linear MLP: no instance (the one-block kernels run it)
    case 1: UH_CHAIN_N(1) break;
    case 2: UH_CHAIN_N(2) break;
    default: UH_CHAIN_N(3) break;
    }
#undef UH_CHAIN_N
#undef UH_CHAIN
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

