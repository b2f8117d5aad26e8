// First convolution of the unet_laplacian backbone (Conv2D 5x5, 3 -> 32 on the normalised image,
// backbone_unet_laplacian.py:296-309) on the f16 matrix cores with split-f16 operands - the arithmetic of unet_h3.hip:
// pixel value and weight each as hi + lo f16, three products, fp32 accumulation.  Same tiling as uo_first_conv_tile_kernel
// (unet_ops.hip: 8 x 64 output pixels per workgroup, the 12 x 68 x 3 input patch normalised once into LDS); the 75-long
// contraction takes 3 x 3 MFMAs of 16 cycles per output tile instead of 19 of 32 cycles, which moves the kernel from the
// fp32 matrix pipe (0.59 ms at 32 x 512 x 512) to the 128 B/px it writes.
#include "unet_h3_core.h"

//
// Round 4: the kernel size is a template parameter (3, 5, 7): the base convolution of the resnet configs is the same operator with
// kernel_size 7 (resnet_color_1x6_bn_32x128x32_1x3x1.json: 147-long contraction = 5 chunks of 32; the fp32 kernel took 829 us per
// batch of 64 x 256 x 256, more than two of that model's blocks).
constexpr int UF_TH = 8, UF_TW = 64, UF_CIN = 3, UF_COUT = 32;
typedef unsigned uf_u4 __attribute__((ext_vector_type(4)));
#ifndef UF_GRID
// workgroups of the persistent launch (k = 7 only: two per CU at 256 registers; the next tile's patch is requested a tile ahead: 418 -> 298 us
// at 64 x 256 x 256).  k = 3, 5 run uf_first_conv_tile_kernel, one tile per workgroup: at k = 5 it needs 128 registers (four waves per
// SIMD) where this kernel's loop form needs 188 (two), and measures 460 us against 522 us persistent / 667 us for this code with one tile per
// workgroup (tools/exp/first_conv_ab.sh).  Note for whoever tunes the persistent form further: gfx950 counts loads and stores in one queue
// (vmcnt) and the stores of a tile are conditional (edge tiles), so hipcc waits for the prefetched patch with vmcnt(0) -- every tile waits
// for its own stores to land; unrolling the eight pixel groups to make the store count static took 331 / 503 registers.
#define UF_GRID(k) ((int64_t)512)
#endif

// One tile per workgroup (the form of rounds 2-3 with the kernel size as a template parameter): the patch loads are the first thing a
// workgroup issues, and the weight preparation runs behind them.  k = 3, 5 take this kernel: for k = 5 it measures 460 us at 32 x 512 x 512
// against 522 us for the persistent kernel below and 667 us for that kernel launched with one tile per workgroup (same box, alternating:
// tools/exp/first_conv_ab.sh) -- a regression the persistent form brought in mid-round and a same-box comparison found.
#ifndef UF_TILE_ABLATE
#define UF_TILE_ABLATE 0      // timing builds (results wrong): 1 gathers without bank conflicts, 2 no stores, 4 no MFMAs
#endif
template <int UF_KS>
__global__ __launch_bounds__(256) void uf_first_conv_tile_kernel(const void* __restrict__ in, int in_is_u8, float* __restrict__ out,
                                                            const float* __restrict__ w, int Hs, int Ws, int H, int W, int normalize,
                                                            float v_min, float v_max, int act, float alpha)
{
    constexpr int UF_KT = UF_KS * UF_KS * UF_CIN, UF_NS = (UF_KT + 31) / 32;
    constexpr int UF_IH = UF_TH + UF_KS - 1, UF_IW = UF_TW + UF_KS - 1, UF_NE = UF_IH * UF_IW * UF_CIN;
    __shared__ unsigned tile[UF_NE];                   // f16 hi in the low half, f16 lo in the high half
    __shared__ float red[256];
    const int x0 = blockIdx.x * UF_TW, y0 = blockIdx.y * UF_TH;
    const int64_t b = blockIdx.z;
    const float range = v_max - v_min;
    {
        constexpr int NE = (UF_NE + 255) / 256;
        float rv[NE];
        bool inpad[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            const int ci = e % UF_CIN, px = (e / UF_CIN) % UF_IW, py = e / (UF_CIN * UF_IW);
            const int yy = y0 + py - UF_KS / 2, xx = x0 + px - UF_KS / 2;
            inpad[i] = e < UF_NE && yy >= 0 && yy < H && xx >= 0 && xx < W;      // inside the (virtually padded) image
            rv[i] = 0.f;
            if (inpad[i] && yy < Hs && xx < Ws) {
                const int64_t o = ((b * Hs + yy) * Ws + xx) * UF_CIN + ci;
                rv[i] = in_is_u8 ? (float)reinterpret_cast<const unsigned char*>(in)[o] : reinterpret_cast<const float*>(in)[o];
            }
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            float v = rv[i];
            if (inpad[i] && normalize) v = (fminf(fmaxf(v, v_min), v_max) - v_min) / range - 0.5f;
            if (!inpad[i]) v = 0.f;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            if (e < UF_NE) tile[e] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
    }
    // power-of-two scale that puts the largest weight in [2^13, 2^14): the lo parts stay normal f16 numbers
    float mx = 0.f;
    for (int i = threadIdx.x; i < UF_KT * UF_COUT; i += 256) mx = fmaxf(mx, fabsf(w[i]));
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
    // contraction index k = 32 s + 8 q + i  (tap = k / 3, channel = k % 3; k >= 75: zero weight)
    uh8 wh[UF_NS][2], wl[UF_NS][2];
    int koff[UF_NS][8];
#pragma unroll
    for (int s = 0; s < UF_NS; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 32 * s + 8 * q + i;
            const bool real = k < UF_KT;
            const int tap = real ? k / UF_CIN : 0, ci = real ? k - tap * UF_CIN : 0;
            koff[s][i] = ((tap / UF_KS) * UF_IW + (tap % UF_KS)) * UF_CIN + ci;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 a0, a1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k0 = 32 * s + 8 * q + i, k1 = k0 + 4;
                a0[i] = k0 < UF_KT ? w[k0 * UF_COUT + 16 * t + n] * scale : 0.f;
                a1[i] = k1 < UF_KT ? w[k1 * UF_COUT + 16 * t + n] * scale : 0.f;
            }
            uh_split8(a0, a1, wh[s][t], wl[s][t]);
        }
    }
    // 8 rows x 4 column groups of 16 pixels = 32 groups, 8 per wave
    for (int gi = wave; gi < UF_TH * (UF_TW / 16); gi += 4) {
        const int ry = gi / (UF_TW / 16), cx = (gi % (UF_TW / 16)) * 16 + n;
        const unsigned* base = tile + (ry * UF_IW + cx) * UF_CIN;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < UF_NS; ++s) {
            unsigned e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (UF_TILE_ABLATE & 1) ? base[8 * s + i] : base[koff[s][i]];
            uf_u4 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ph[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);     // the two hi halves
                pl[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);     // the two lo halves
            }
            const uh8 xh = __builtin_bit_cast(uh8, ph), xl = __builtin_bit_cast(uh8, pl);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (UF_TILE_ABLATE & 4) { acc[t][0] += (float)xh[0] + (float)xl[1] + (float)wh[s][t][2] + (float)wl[s][t][3]; } else {
                acc[t] = UH_MFMA(wh[s][t], xh, acc[t]);
                acc[t] = UH_MFMA(wl[s][t], xh, acc[t]);
                acc[t] = UH_MFMA(wh[s][t], xl, acc[t]); }
            }
        }
        const int gy = y0 + ry, gx = x0 + cx;
        if (gy < H && gx < W) {
            float* op = out + ((b * H + gy) * (int64_t)W + gx) * UF_COUT + 4 * q;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 v = bf_acc_ready(acc[t]) * inv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (act == 1) v[r] = fmaxf(v[r], 0.f);
                    else if (act == 2) v[r] = fmaxf(v[r], alpha * v[r]);
                    else if (act == 3) v[r] = uh_act<3>(v[r], 0.f);
                }
                if (!(UF_TILE_ABLATE & 2) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(op + 16 * t) = v;
            }
        }
    }
}


// Persistent form (k = 7: 80 weight registers per lane are worth preparing once; 418 -> 298 us at 64 x 256 x 256).
template <int UF_KS>
__global__ __launch_bounds__(256) void uf_first_conv_kernel(const void* __restrict__ in, int in_is_u8, float* __restrict__ out,
                                                            const float* __restrict__ w, int Hs, int Ws, int H, int W, int normalize,
                                                            float v_min, float v_max, int act, float alpha, int ntiles)
{
    constexpr int UF_KT = UF_KS * UF_KS * UF_CIN, UF_NS = (UF_KT + 31) / 32;
    constexpr int UF_IH = UF_TH + UF_KS - 1, UF_IW = UF_TW + UF_KS - 1, UF_NE = UF_IH * UF_IW * UF_CIN;
    __shared__ unsigned tile[UF_NE];                   // f16 hi in the low half, f16 lo in the high half
    __shared__ float red[256];
    const float range = v_max - v_min;
    // the first tile's patch is requested BEFORE the weight preparation below, which then runs while those loads are in flight (the round's
    // first persistent form prepared the weights first: the one-tile-per-workgroup launch of k = 5 went from 463 to 655 us)
    const int tiles_x = (W + UF_TW - 1) / UF_TW, tiles_y = (H + UF_TH - 1) / UF_TH;
    constexpr int NE = (UF_NE + 255) / 256;
    float rv[NE];
    int pyk[NE], pxk[NE];                              // the element's row / column inside the patch (-UF_KS / 2 applied); same for every tile
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = min((int)threadIdx.x + i * 256, UF_NE - 1);
        pxk[i] = (e / UF_CIN) % UF_IW - UF_KS / 2;
        pyk[i] = e / (UF_CIN * UF_IW) - UF_KS / 2;
    }
    const int ci = (threadIdx.x % UF_CIN);             // 256 % 3 = 1: the channel of element threadIdx.x + 256 i is (ci + i) % 3
    auto request = [&](const int tid) {
        const int x0 = (tid % tiles_x) * UF_TW, y0 = ((tid / tiles_x) % tiles_y) * UF_TH;
        const int64_t b = tid / (tiles_x * tiles_y);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int yy = min(max(y0 + pyk[i], 0), Hs - 1), xx = min(max(x0 + pxk[i], 0), Ws - 1);
            const int64_t o = ((b * Hs + yy) * Ws + xx) * UF_CIN + (ci + i) % UF_CIN;
            rv[i] = in_is_u8 ? (float)reinterpret_cast<const unsigned char*>(in)[o] : reinterpret_cast<const float*>(in)[o];
        }
    };
    if ((int)blockIdx.x < ntiles) request(blockIdx.x);
    // power-of-two scale that puts the largest weight in [2^13, 2^14): the lo parts stay normal f16 numbers
    float mx = 0.f;
    for (int i = threadIdx.x; i < UF_KT * UF_COUT; i += 256) mx = fmaxf(mx, fabsf(w[i]));
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, n = lane & 15;
    // contraction index k = 32 s + 8 q + i  (tap = k / 3, channel = k % 3; k >= UF_KT: zero weight)
    uh8 wh[UF_NS][2], wl[UF_NS][2];
    int koff[UF_NS][8];
#pragma unroll
    for (int s = 0; s < UF_NS; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 32 * s + 8 * q + i;
            const bool real = k < UF_KT;
            const int tap = real ? k / UF_CIN : 0, ci = real ? k - tap * UF_CIN : 0;
            koff[s][i] = ((tap / UF_KS) * UF_IW + (tap % UF_KS)) * UF_CIN + ci;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 a0, a1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k0 = 32 * s + 8 * q + i, k1 = k0 + 4;
                a0[i] = k0 < UF_KT ? w[k0 * UF_COUT + 16 * t + n] * scale : 0.f;
                a1[i] = k1 < UF_KT ? w[k1 * UF_COUT + 16 * t + n] * scale : 0.f;
            }
            uh_split8(a0, a1, wh[s][t], wl[s][t]);
        }
    }
    // The workgroup prepares its weight fragments ONCE (a max reduction over the kernel + 16 * UF_NS scalar loads per lane: as much L2
    // traffic per 8 x 64 tile as the tile's output when every tile was a workgroup of its own) and then walks tiles; the raw values of the
    // NEXT tile's patch are requested before the current tile is multiplied (unconditional loads from clamped addresses, masked when they
    // are converted: with two or three resident workgroups per CU nothing else covers the load -> barrier -> multiply chain of a tile).
    for (int tid = blockIdx.x; tid < ntiles; tid += gridDim.x) {
        const int x0 = (tid % tiles_x) * UF_TW, y0 = ((tid / tiles_x) % tiles_y) * UF_TH;
        const int64_t b = tid / (tiles_x * tiles_y);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * 256;
            const int yy = y0 + pyk[i], xx = x0 + pxk[i];
            const bool inpad = yy >= 0 && yy < H && xx >= 0 && xx < W;           // inside the (virtually padded) image
            float v = (inpad && yy < Hs && xx < Ws) ? rv[i] : 0.f;               // the pad band between the source and [H, W] is value 0
            if (inpad && normalize) v = (fminf(fmaxf(v, v_min), v_max) - v_min) / range - 0.5f;
            if (!inpad) v = 0.f;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            if (e < UF_NE) tile[e] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
        __syncthreads();
        if ((int)gridDim.x < ntiles) {                   // kernel-uniform: a launch with one tile per workgroup has nothing to request
            const int nxt = tid + (int)gridDim.x;
            request(nxt < ntiles ? nxt : tid);
        }
        // 8 rows x 4 column groups of 16 pixels = 32 groups, 8 per wave
        for (int gi = wave; gi < UF_TH * (UF_TW / 16); gi += 4) {
            const int ry = gi / (UF_TW / 16), cx = (gi % (UF_TW / 16)) * 16 + n;
            const unsigned* base = tile + (ry * UF_IW + cx) * UF_CIN;
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < UF_NS; ++s) {
                unsigned e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = base[koff[s][i]];
                uf_u4 ph, pl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ph[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x05040100u);     // the two hi halves
                    pl[j] = __builtin_amdgcn_perm(e[2 * j + 1], e[2 * j], 0x07060302u);     // the two lo halves
                }
                const uh8 xh = __builtin_bit_cast(uh8, ph), xl = __builtin_bit_cast(uh8, pl);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[t] = UH_MFMA(wh[s][t], xh, acc[t]);
                    acc[t] = UH_MFMA(wl[s][t], xh, acc[t]);
                    acc[t] = UH_MFMA(wh[s][t], xl, acc[t]);
                }
            }
            const int gy = y0 + ry, gx = x0 + cx;
            if (gy < H && gx < W) {
                float* op = out + ((b * H + gy) * (int64_t)W + gx) * UF_COUT + 4 * q;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 v = bf_acc_ready(acc[t]) * inv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (act == 1) v[r] = fmaxf(v[r], 0.f);
                        else if (act == 2) v[r] = fmaxf(v[r], alpha * v[r]);
                        else if (act == 3) v[r] = uh_act<3>(v[r], 0.f);
                    }
                    *reinterpret_cast<f32x4*>(op + 16 * t) = v;
                }
            }
        }
        __syncthreads();                                 // the next tile's patch overwrites this one
    }
}

// bf_op_first_conv for k x k, 3 -> 32 (k = 3, 5, 7), split-f16 arithmetic
extern "C" int bf_op_first_conv_h3k(const void* in, int in_is_u8, float* out, const float* w, int B, int Hs, int Ws, int H, int W, int k,
                                    int normalize, float v_min, float v_max, int act, float alpha, void* stream)
{
    if (!in || !out || !w || B <= 0 || Hs <= 0 || Ws <= 0 || H < Hs || W < Ws) return BF_EINVAL;
    if (normalize && !(v_max > v_min)) return BF_EINVAL;
    if ((uintptr_t)out % 16) return BF_EINVAL;
    if (act < 0 || act > 3) return BF_EINVAL;
    if (act == 2 && !(alpha >= 0.f && alpha <= 1.f)) return BF_EINVAL;
    if (k != 3 && k != 5 && k != 7) return BF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (k != 7) {
        if (B > 65535 || (H + UF_TH - 1) / UF_TH > 65535) return BF_EUNSUPPORTED;
        const dim3 grid3((W + UF_TW - 1) / UF_TW, (H + UF_TH - 1) / UF_TH, B);
        if (k == 3) hipLaunchKernelGGL(uf_first_conv_tile_kernel<3>, grid3, dim3(256), 0, s, in, in_is_u8, out, w, Hs, Ws, H, W, normalize, v_min, v_max, act, alpha);
        else hipLaunchKernelGGL(uf_first_conv_tile_kernel<5>, grid3, dim3(256), 0, s, in, in_is_u8, out, w, Hs, Ws, H, W, normalize, v_min, v_max, act, alpha);
        return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
    }
    const int64_t nt = (int64_t)B * ((H + UF_TH - 1) / UF_TH) * ((W + UF_TW - 1) / UF_TW);
    if (nt >= ((int64_t)1 << 31)) return BF_EUNSUPPORTED;
    const int ntiles = (int)nt;
    const dim3 grid((unsigned)std::min<int64_t>(nt, UF_GRID(k)));
    hipLaunchKernelGGL((uf_first_conv_kernel<7>), grid, dim3(256), 0, s, in, in_is_u8, out, w, Hs, Ws, H, W, normalize, v_min, v_max, act, alpha, ntiles);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}

// the shape the unet_laplacian builder emits (5x5, 3 -> 32)
extern "C" int bf_op_first_conv_h3(const void* in, int in_is_u8, float* out, const float* w, int B, int Hs, int Ws, int H, int W,
                                   int normalize, float v_min, float v_max, int act, float alpha, void* stream)
{
    return bf_op_first_conv_h3k(in, in_is_u8, out, w, B, Hs, Ws, H, W, 5, normalize, v_min, v_max, act, alpha, stream);
}
