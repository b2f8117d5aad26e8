// 3x3, 16 -> 16 channel, stride-1 SAME convolutions on the gfx950 matrix cores.
//
// Reference semantics: keras Conv2D(3x3, use_bias=False, padding="same") inside conv2d_wrapper
// (bfcnn/utilities.py:196) as the resnet blocks call it (bfcnn/backbone_blocks.py:174-196), the
// BatchNormalization / activation that follow it (utilities.py:207-215) and the residual Add
// (backbone_blocks.py:242).
//
// GEMM view per tap: D[cout][pixel] += W[cout][cin] * X[cin][pixel], exact fp32 on
// v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain).  The weights are the A operand and live in 36
// VGPRs per lane for the whole kernel; the activations are the B operand, one ds_read_b128 per
// tap and 16-pixel group feeds four MFMAs (k-slot q of MFMA kk carries cin = 4q+kk).  The D
// fragment leaves every lane with 4 consecutive output channels of one pixel, so stores are
// 16 B per lane, 1 KiB contiguous per wave-instruction in NHWC.
#include "bf_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// timing-only ablations of the fused kernel (tools/ablate.sh builds them into lib/variants/; results
// are WRONG when any is set): 1 = no barriers, 2 = no next-tile fetch/stage, 4 = no global stores
#ifndef BF_ABLATE
#define BF_ABLATE 0
#endif
// 8 = in-kernel s_memtime stamps per phase (diagnostic build; sums per wave go to args.dbg)
#if BF_ABLATE & 8
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        unsigned long long now_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        stamp_sum[k] += now_ - stamp_prev;                                                               \
        stamp_prev = now_;                                                                               \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
#if BF_ABLATE & 1
#define FUSED_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define FUSED_SYNC() __syncthreads()
#endif

// ------------------------------------------------------------------------------------------
// weight packing: HWIO [3,3,16,16] -> wpack[(tap*4+kk)*64 + lane] = W[tap][4*(lane>>4)+kk][lane&15]
// transpose_flip = 1 packs the data-gradient kernel W'[tap][ci][co] = W[8-tap][co][ci].
// ------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wpack, int transpose_flip)
{
    for (int idx = threadIdx.x; idx < BF_WPACK_FLOATS; idx += blockDim.x) {
        const int i = idx >> 6, l = idx & 63;
        const int tap = i >> 2, kk = i & 3;
        const int cin = 4 * (l >> 4) + kk, cout = l & 15;
        wpack[idx] = transpose_flip ? w[((8 - tap) * 16 + cout) * 16 + cin]
                                    : w[(tap * 16 + cin) * 16 + cout];
    }
}

hipError_t bf_launch_pack_conv(const float* w_hwio, float* wpack, int transpose_flip, hipStream_t s)
{
    hipLaunchKernelGGL(pack_conv_kernel, dim3(1), dim3(256), 0, s, w_hwio, wpack, transpose_flip);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// generic single convolution with selectable epilogue
// ------------------------------------------------------------------------------------------
constexpr int CT_H = 16, CT_W = 32;            // output tile
constexpr int CT_IH = CT_H + 2, CT_IW = CT_W + 2;

template <int EPI>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float tile[CT_IH * CT_IW * 16];   // 39,168 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (a.W + CT_W - 1) / CT_W, tiles_y = (a.H + CT_H - 1) / CT_H;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * CT_H, x0 = tx * CT_W;
    const size_t img = (size_t)b * a.H * a.W * 16;
    const float* inb = a.in + img;

    for (int n = tid; n < CT_IH * CT_IW * 4; n += 256) {
        const int row = n / (CT_IW * 4);
        const int rem = n - row * (CT_IW * 4);
        const int gy = y0 - 1 + row, gx = x0 - 1 + (rem >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v = *reinterpret_cast<const float4*>(inb + ((size_t)gy * a.W + gx) * 16 + (rem & 3) * 4);
        *reinterpret_cast<float4*>(tile + n * 4) = v;
    }
    float w[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) w[i] = a.wpack[i * 64 + lane];
    __syncthreads();

    const int p = lane & 15, q = lane >> 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (EPI & EPI_AFFINE) {
        sc = *reinterpret_cast<const f32x4*>(a.scale + q * 4);
        sh = *reinterpret_cast<const f32x4*>(a.shift + q * 4);
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 acc[4];
        int base[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int gi = half * 4 + j;
            const int r = 4 * wave + (gi >> 1), xo = (gi & 1) * 16;
            base[j] = (r * CT_IW + xo + p) * 16 + q * 4;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = ((tap / 3) * CT_IW + (tap % 3)) * 16;
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const f32x4*>(tile + base[j] + off);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = MFMA(w[tap * 4 + kk], bv[j][kk], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gi = half * 4 + j;
            const int gy = y0 + 4 * wave + (gi >> 1), gx = x0 + (gi & 1) * 16 + p;
            if (gy < a.H && gx < a.W) {
                const size_t idx = img + ((size_t)gy * a.W + gx) * 16 + q * 4;
                f32x4 v = bf_acc_ready(acc[j]);
                if (EPI & EPI_STATS) { s1 += v; s2 += v * v; }
                if (EPI & EPI_AFFINE) v = v * sc + sh;
                if (EPI & EPI_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (EPI & EPI_MASK) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(a.mask + idx);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                    v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
                if (EPI & EPI_RES) v += *reinterpret_cast<const f32x4*>(a.res + idx);
                *reinterpret_cast<f32x4*>(a.out + idx) = v;
            }
        }
    }

    if (EPI & EPI_STATS) {
        // reduce over the 16 pixel lanes that share a channel quad, then over the 4 waves
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2[c] += __shfl_xor(s2[c], m);
            }
        }
        __syncthreads();                       // tile no longer needed
        float* red = tile;                     // [4 waves][32]
        if (p == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[wave * 32 + q * 4 + c] = s1[c];
                red[wave * 32 + 16 + q * 4 + c] = s2[c];
            }
        }
        __syncthreads();
        if (tid < 32)
            a.stats[(size_t)blockIdx.x * 32 + tid] = (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
    }
}

int bf_conv3x3_c16_grid(int B, int H, int W)
{
    return B * ((H + CT_H - 1) / CT_H) * ((W + CT_W - 1) / CT_W);
}

hipError_t bf_launch_conv3x3_c16(const ConvArgs& a, int epi, hipStream_t s)
{
    const dim3 grid(bf_conv3x3_c16_grid(a.B, a.H, a.W)), block(256);
#define BF_CASE(E) case E: hipLaunchKernelGGL(conv3x3_c16_kernel<E>, grid, block, 0, s, a); break;
    switch (epi) {
        BF_CASE(0)
        BF_CASE(EPI_RELU)
        BF_CASE(EPI_STATS)
        BF_CASE(EPI_AFFINE | EPI_RES)
        BF_CASE(EPI_AFFINE)
        BF_CASE(EPI_AFFINE | EPI_RELU)
        BF_CASE(EPI_RES)
        BF_CASE(EPI_MASK)
        default: return hipErrorInvalidValue;
    }
#undef BF_CASE
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// fused residual block (inference): out = x + scale * conv2(act(conv1(x))) + shift
// One HBM read and one HBM write of the 16-channel activation per BLOCK (72 FLOP/B instead of
// 36); the intermediate activation only ever exists in LDS.  Tile 14x32 outputs: input tile
// 18x36 px (41,472 B) + intermediate 16x34 px (34,816 B) = 76,288 B LDS -> two 4-wave
// workgroups per CU.  The 16x34 = 544 intermediate pixels are exactly 34 MFMA groups, the
// 14x32 = 448 outputs exactly 28 (groups that lie wholly outside the image are skipped).
// Workgroups are persistent: weights (72 VGPRs) are fetched once; consecutive tiles of one
// XCD-label (blockIdx % 8) are neighbours in the image so halos are L2 hits.
// Software pipeline: the NEXT tile's 41 KB are fetched into registers (11 x 16 B per lane)
// before conv1 of the current tile starts and only written to LDS after conv2 has finished, so
// the HBM/L2 latency and the chip-wide load burst hide behind ~18k cycles of MFMA work per tile
// (rocprof r01_v1: without it the MFMA pipe was 64 % busy, waves 29 % in s_waitcnt/s_barrier).
// ------------------------------------------------------------------------------------------
// Tile geometry is a template parameter (TH x TW outputs, NW waves per workgroup) so that shapes can
// be A/B-ed in one binary (bf_set_option "fused_tile").  TW is a multiple of 16, so an output row
// is TW/16 MFMA groups.  The intermediate region is (TH+2) x (TW+2): per row TW/16 "row groups"
// (columns 0..TW-1) and the two remaining columns of every 8 rows form one "strip group"
// (lane p -> row 8s + p/2, column TW + p%2).  With this decomposition EVERY LDS and global address
// is (wave-uniform scalar) + (one of three per-lane constants): a pass prologue/epilogue costs a
// handful of VALU instructions instead of ~30 (the MFMA + LDS micro-benchmark
// tools/exp/mfma_loop.hip prices 30 epilogue VALU per group at 9 % of the kernel -- VALU work of a
// pass is not hidden behind the partner wave's MFMAs).
template <int TH_, int TW_, int NW_>
struct FusedCfg {
    static constexpr int TH = TH_, TW = TW_, NW = NW_, NT = NW_ * 64;
    static constexpr int MH = TH + 2, MW = TW + 2;        // intermediate region
    static constexpr int IH = TH + 4, IW = TW + 4;        // input region
    static constexpr int GPR = TW / 16;                   // groups per row
    static constexpr int RG = MH * GPR;                   // conv1 row groups
    static constexpr int SG = (MH + 7) / 8;               // conv1 strip groups
    static constexpr int MG = RG + SG;                    // conv1 groups
    static constexpr int OG = TH * GPR;                   // conv2 groups
    static constexpr int C1K = (MG + NW - 1) / NW;        // conv1 groups per wave (max)
    static constexpr int C2K = (OG + NW - 1) / NW;        // conv2 groups per wave (max)
    static constexpr int IN4 = IH * IW * 4;               // float4 per input tile
    static constexpr int PF = (IN4 + NT - 1) / NT;        // float4 per lane in the prefetch
    static constexpr int TIN_FLOATS = IH * IW * 16;
    static constexpr int LDS_BYTES = (TIN_FLOATS + MH * MW * 16) * 4;
    static_assert(TW % 16 == 0, "output rows must be whole MFMA groups");
};

template <int NG>
__device__ __forceinline__ void conv_groups(const float* __restrict__ src, const int (&base)[NG],
                                            const int row_pitch, const float (&w)[36], f32x4 (&acc)[NG])
{
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int off = ((tap / 3) * row_pitch + (tap % 3)) * 16;
        f32x4 bv[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) bv[j] = *reinterpret_cast<const f32x4*>(src + base[j] + off);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < NG; ++j) acc[j] = MFMA(w[tap * 4 + kk], bv[j][kk], acc[j]);
    }
}

struct FusedTile {
    int y0, x0;
    size_t img;
};

template <class Cfg>
__device__ __forceinline__ FusedTile fused_tile(const FusedBlockArgs& a, int t)
{
    FusedTile r;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    r.y0 = ty * Cfg::TH;
    r.x0 = tx * Cfg::TW;
    r.img = (size_t)b * a.H * a.W * 16;
    return r;
}

// a tile whose whole input region lies inside the image needs no bounds check anywhere
template <class Cfg>
__device__ __forceinline__ bool tile_interior(const FusedBlockArgs& a, const FusedTile& t)
{
    return t.y0 >= 2 && t.y0 + Cfg::TH + 2 <= a.H && t.x0 >= 2 && t.x0 + Cfg::TW + 2 <= a.W;
}

// per-lane constants (tile-invariant)
struct FusedLane {
    int p, q;
    int row_c;       // p*16 + q*4                       : row-group pixel p, channel quad q
    int strip_in;    // ((p/2)*IW + p%2)*16 + q*4         : strip-group pixel in the input tile
    int strip_mid;   // ((p/2)*MW + p%2)*16 + q*4         : strip-group pixel in the intermediate tile
};

// issue the tile's global loads into registers (2 px halo); no wait.  Every load is UNCONDITIONAL:
// out-of-image elements read offset 0 of the image and are zeroed in fused_stage through the returned
// bit mask.  (A per-element `if (inside) load` makes hipcc branch around each load and wait vmcnt(0)
// per element: the 11 loads of a border tile were serialised, ~500 cycles each -- s_memtime stamps.)
// (row, col) of element n = tid + i*NT advance incrementally; addresses are (uniform image base) +
// (unsigned 32-bit per-lane byte offset) so the loads take the saddr form.
template <class Cfg, bool INTERIOR>
__device__ __forceinline__ unsigned fused_fetch(const FusedBlockArgs& a, const FusedTile& t, int tid, float4 (&pf)[Cfg::PF])
{
    constexpr int RW = Cfg::IW * 4;                                  // float4 per tile row
    const char* img = reinterpret_cast<const char*>(a.in + t.img);
    int row = tid / RW, rem = tid - row * RW;
    unsigned zero_mask = 0;
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const bool in_tile = (i + 1) * Cfg::NT <= Cfg::IN4 || tid + i * Cfg::NT < Cfg::IN4;
        const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + (rem >> 2);
        const bool inside = INTERIOR || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
        unsigned off = ((unsigned)(gy * a.W + gx) * 16u + (unsigned)(rem & 3) * 4u) * 4u;
        if (!(in_tile && inside)) { off = 0; zero_mask |= 1u << i; }
        pf[i] = *reinterpret_cast<const float4*>(img + off);
        rem += Cfg::NT % RW;
        row += Cfg::NT / RW;
        if (rem >= RW) { rem -= RW; ++row; }
    }
    return zero_mask;
}

template <class Cfg>
__device__ __forceinline__ void fused_stage(float* __restrict__ tin, int tid, const float4 (&pf)[Cfg::PF], const unsigned zero_mask)
{
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const int n = tid + i * Cfg::NT;
        float4 v = pf[i];
        if (zero_mask & (1u << i)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((i + 1) * Cfg::NT <= Cfg::IN4 || n < Cfg::IN4) *reinterpret_cast<float4*>(tin + n * 4) = v;
    }
}

// conv1 (+activation) of NG intermediate groups g[j] (wave-uniform): input tile -> intermediate tile.
// INTERIOR = false additionally zeroes pixels outside the image: conv2 must see ZERO padding there,
// not conv1 evaluated outside the image.
template <class Cfg, int NG, bool INTERIOR>
__device__ __forceinline__ void conv1_pass(const FusedBlockArgs& a, const float* __restrict__ tin, float* __restrict__ tmid,
                                           const float (&w1)[36], const int (&g)[NG], const FusedLane& L, const FusedTile& t)
{
    int base[NG], dst[NG], my[NG], mx[NG];
    f32x4 acc[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        if (g[j] < Cfg::RG) {                                        // wave-uniform branch
            const int r = g[j] / Cfg::GPR, xg = (g[j] - r * Cfg::GPR) * 16;
            base[j] = (r * Cfg::IW + xg) * 16 + L.row_c;
            dst[j] = (r * Cfg::MW + xg) * 16 + L.row_c;
            my[j] = r; mx[j] = xg + L.p;
        } else {
            const int r0 = (g[j] - Cfg::RG) * 8;
            if (Cfg::MH % 8 == 0 || r0 + 8 <= Cfg::MH) {
                base[j] = (r0 * Cfg::IW + Cfg::TW) * 16 + L.strip_in;
                dst[j] = (r0 * Cfg::MW + Cfg::TW) * 16 + L.strip_mid;
                my[j] = r0 + (L.p >> 1);
            } else {                                                 // partial last strip: clamp the row
                my[j] = min(r0 + (L.p >> 1), Cfg::MH - 1);
                base[j] = (my[j] * Cfg::IW + Cfg::TW + (L.p & 1)) * 16 + L.q * 4;
                dst[j] = (my[j] * Cfg::MW + Cfg::TW + (L.p & 1)) * 16 + L.q * 4;
            }
            mx[j] = Cfg::TW + (L.p & 1);
        }
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    conv_groups<NG>(tin, base, Cfg::IW, w1, acc);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        f32x4 v = bf_acc_ready(acc[j]);
        if (a.act1_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!INTERIOR) {
            const int gy = t.y0 - 1 + my[j], gx = t.x0 - 1 + mx[j];
            if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(tmid + dst[j]) = v;
    }
}

// conv2 + folded BN + residual of NG output groups: intermediate tile -> global.
// RES_GLOBAL: the residual is re-read from global memory (an L2 hit, the workgroup fetched those
// lines for its input tile one phase earlier) so the LDS input tile is free for the next tile's DMA.
template <class Cfg, int NG, bool INTERIOR, bool RES_GLOBAL = false>
__device__ __forceinline__ void conv2_pass(const FusedBlockArgs& a, const float* __restrict__ tin,
                                           const float* __restrict__ tmid, float* __restrict__ out_tile,
                                           const float (&w2)[36], const f32x4 sc, const f32x4 sh, const int (&g)[NG],
                                           const FusedLane& L, const FusedTile& t)
{
    int base[NG];
    f32x4 acc[NG], res[NG];
    const float* in_tile = a.in + (out_tile - a.out);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int oy = g[j] / Cfg::GPR, xg = (g[j] - oy * Cfg::GPR) * 16;
        base[j] = (oy * Cfg::MW + xg) * 16 + L.row_c;
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (RES_GLOBAL) {
            res[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (INTERIOR || (t.y0 + oy < a.H && t.x0 + xg + L.p < a.W))
                res[j] = *reinterpret_cast<const f32x4*>(in_tile + ((size_t)oy * a.W + xg) * 16 + (unsigned)L.row_c);
        }
    }
    conv_groups<NG>(tmid, base, Cfg::MW, w2, acc);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int oy = g[j] / Cfg::GPR, xg = (g[j] - oy * Cfg::GPR) * 16;
        if (INTERIOR || (t.y0 + oy < a.H && t.x0 + xg + L.p < a.W)) {
            if (!RES_GLOBAL) res[j] = *reinterpret_cast<const f32x4*>(tin + ((oy + 2) * Cfg::IW + xg + 2) * 16 + L.row_c);
            const f32x4 v = bf_acc_ready(acc[j]) * sc + sh + res[j];
            if (BF_ABLATE & 4) { if (v.x == 12345.678f) out_tile[0] = v.y; }       // keeps the MFMAs live
            else *reinterpret_cast<f32x4*>(out_tile + ((size_t)oy * a.W + xg) * 16 + (unsigned)L.row_c) = v;
        }
    }
}

// An intermediate group is needed iff one of its pixels can be read by an in-image output:
// image row <= H and column <= W (rows/columns beyond that are never read).
template <class Cfg>
__device__ __forceinline__ bool mid_group_needed(const FusedBlockArgs& a, const FusedTile& t, int g)
{
    if (g < Cfg::RG) {
        const int r = g / Cfg::GPR, xg = (g - r * Cfg::GPR) * 16;
        return t.y0 - 1 + r <= a.H && t.x0 - 1 + xg <= a.W;
    }
    return t.y0 - 1 + (g - Cfg::RG) * 8 <= a.H && t.x0 - 1 + Cfg::TW <= a.W;
}

template <class Cfg>
__device__ __forceinline__ bool out_group_needed(const FusedBlockArgs& a, const FusedTile& t, int g)
{
    const int oy = g / Cfg::GPR, xg = (g - oy * Cfg::GPR) * 16;
    return t.y0 + oy < a.H && t.x0 + xg < a.W;
}

// this wave's groups slot K0.. of an interior tile, in passes of PASS (compile-time group slots)
template <class Cfg, int K0>
__device__ __forceinline__ void conv1_interior(const FusedBlockArgs& a, const float* __restrict__ tin, float* __restrict__ tmid,
                                               const float (&w1)[36], const FusedLane& L, const FusedTile& t, const int wave,
                                               const int n1)
{
    if constexpr (K0 < Cfg::C1K) {
        const int rem = n1 - K0;
        const int g0 = wave + Cfg::NW * K0, g1 = g0 + Cfg::NW, g2 = g1 + Cfg::NW;
        if (rem >= 3)      { const int gs[3] = {g0, g1, g2}; conv1_pass<Cfg, 3, true>(a, tin, tmid, w1, gs, L, t); }
        else if (rem == 2) { const int gs[2] = {g0, g1};     conv1_pass<Cfg, 2, true>(a, tin, tmid, w1, gs, L, t); }
        else if (rem == 1) { const int gs[1] = {g0};         conv1_pass<Cfg, 1, true>(a, tin, tmid, w1, gs, L, t); }
        conv1_interior<Cfg, K0 + 3>(a, tin, tmid, w1, L, t, wave, n1);
    }
}

template <class Cfg, int K0, bool RES_GLOBAL = false>
__device__ __forceinline__ void conv2_interior(const FusedBlockArgs& a, const float* __restrict__ tin,
                                               const float* __restrict__ tmid, float* __restrict__ out_tile,
                                               const float (&w2)[36], const f32x4 sc, const f32x4 sh, const FusedLane& L,
                                               const FusedTile& t, const int wave, const int n2)
{
    if constexpr (K0 < Cfg::C2K) {
        const int rem = n2 - K0;
        const int g0 = wave + Cfg::NW * K0, g1 = g0 + Cfg::NW, g2 = g1 + Cfg::NW, g3 = g2 + Cfg::NW;
        if (rem >= 4)      { const int gs[4] = {g0, g1, g2, g3}; conv2_pass<Cfg, 4, true, RES_GLOBAL>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, t); }
        else if (rem == 3) { const int gs[3] = {g0, g1, g2};     conv2_pass<Cfg, 3, true, RES_GLOBAL>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, t); }
        else if (rem == 2) { const int gs[2] = {g0, g1};         conv2_pass<Cfg, 2, true, RES_GLOBAL>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, t); }
        else if (rem == 1) { const int gs[1] = {g0};             conv2_pass<Cfg, 1, true, RES_GLOBAL>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, t); }
        conv2_interior<Cfg, K0 + 4, RES_GLOBAL>(a, tin, tmid, out_tile, w2, sc, sh, L, t, wave, n2);
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_kernel(FusedBlockArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tin = lds;                                  // [IH][IW][16]
    float* tmid = lds + Cfg::TIN_FLOATS;               // [MH][MW][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FusedLane L0;
    L0.p = lane & 15;
    L0.q = lane >> 4;
    L0.row_c = L0.p * 16 + L0.q * 4;
    L0.strip_in = ((L0.p >> 1) * Cfg::IW + (L0.p & 1)) * 16 + L0.q * 4;
    L0.strip_mid = ((L0.p >> 1) * Cfg::MW + (L0.p & 1)) * 16 + L0.q * 4;

    float w1[36], w2[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) { w1[i] = a.w1pack[i * 64 + lane]; w2[i] = a.w2pack[i * 64 + lane]; }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + L0.q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + L0.q * 4);

    // XCD-aware persistent schedule: label = blockIdx % 8 owns a contiguous chunk of the tile
    // sequence; its workgroups walk the chunk together (speed only, never correctness).
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;            // gridDim.x is a multiple of nxcd
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);

    int t = t_begin + slot;
    if (t >= t_end) return;                            // uniform per workgroup
    const int n1 = (Cfg::MG - wave + Cfg::NW - 1) / Cfg::NW;      // this wave's conv1 / conv2 group counts
    const int n2 = (Cfg::OG - wave + Cfg::NW - 1) / Cfg::NW;

    float4 pf[Cfg::PF];
    FusedTile cur = fused_tile<Cfg>(a, t);
    unsigned zero_mask = fused_fetch<Cfg, false>(a, cur, tid, pf);
    fused_stage<Cfg>(tin, tid, pf, zero_mask);
    FUSED_SYNC();
#if BF_ABLATE & 8
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif

    for (; t < t_end; t += per_label) {
        const int tn = t + per_label;
        const bool has_next = tn < t_end;
        FusedTile nxt = cur;
        if (has_next) {                                // prefetch: in flight during conv1 + conv2
            nxt = fused_tile<Cfg>(a, tn);
            STAMP(7);                                  // tile index arithmetic
            if (!(BF_ABLATE & 2)) {
                if (tile_interior<Cfg>(a, nxt)) zero_mask = fused_fetch<Cfg, true>(a, nxt, tid, pf);
                else zero_mask = fused_fetch<Cfg, false>(a, nxt, tid, pf);
            }
        }
        STAMP(0);                                      // fetch issue
        float* out_tile = a.out + cur.img + ((size_t)cur.y0 * a.W + cur.x0) * 16;
        // make the per-lane constants opaque inside the loop body: otherwise LICM hoists every
        // (scalar + lane constant) address of every unrolled pass out of the tile loop and the
        // kernel spills (hipcc 7.2: 256 VGPRs + 160..476 B scratch per lane)
        FusedLane L = L0;
        asm volatile("" : "+v"(L.row_c), "+v"(L.strip_in), "+v"(L.strip_mid));

        if (tile_interior<Cfg>(a, cur)) {
            conv1_interior<Cfg, 0>(a, tin, tmid, w1, L, cur, wave, n1);
            STAMP(1);                                  // conv1
            FUSED_SYNC();
            STAMP(2);                                  // barrier after conv1
            conv2_interior<Cfg, 0>(a, tin, tmid, out_tile, w2, sc, sh, L, cur, wave, n2);
            STAMP(3);                                  // conv2
        } else {
            // border tile: groups wholly outside the image are skipped, the rest is bounds-checked
            for (int g = wave; g < Cfg::MG;) {
                int g0 = -1, g1 = -1, g2 = -1;
                for (; g < Cfg::MG && g2 < 0; g += Cfg::NW) {
                    if (!mid_group_needed<Cfg>(a, cur, g)) continue;
                    if (g0 < 0) g0 = g; else if (g1 < 0) g1 = g; else g2 = g;
                }
                if (g2 >= 0)      { const int gs[3] = {g0, g1, g2}; conv1_pass<Cfg, 3, false>(a, tin, tmid, w1, gs, L, cur); }
                else if (g1 >= 0) { const int gs[2] = {g0, g1};     conv1_pass<Cfg, 2, false>(a, tin, tmid, w1, gs, L, cur); }
                else if (g0 >= 0) { const int gs[1] = {g0};         conv1_pass<Cfg, 1, false>(a, tin, tmid, w1, gs, L, cur); }
            }
            FUSED_SYNC();
            for (int g = wave; g < Cfg::OG;) {
                int g0 = -1, g1 = -1, g2 = -1, g3 = -1;
                for (; g < Cfg::OG && g3 < 0; g += Cfg::NW) {
                    if (!out_group_needed<Cfg>(a, cur, g)) continue;
                    if (g0 < 0) g0 = g; else if (g1 < 0) g1 = g; else if (g2 < 0) g2 = g; else g3 = g;
                }
                if (g3 >= 0)      { const int gs[4] = {g0, g1, g2, g3}; conv2_pass<Cfg, 4, false>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g2 >= 0) { const int gs[3] = {g0, g1, g2};     conv2_pass<Cfg, 3, false>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g1 >= 0) { const int gs[2] = {g0, g1};         conv2_pass<Cfg, 2, false>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g0 >= 0) { const int gs[1] = {g0};             conv2_pass<Cfg, 1, false>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
            }
        }
        FUSED_SYNC();                                  // every wave is done with tin / tmid
        STAMP(4);                                      // barrier after conv2 (+ whole border tiles)
        if (has_next && !(BF_ABLATE & 2)) {
            fused_stage<Cfg>(tin, tid, pf, zero_mask); // waits for the prefetch here, one tile late
            STAMP(5);                                  // vmcnt wait + ds_write
            FUSED_SYNC();
            STAMP(6);                                  // barrier after stage
        }
        cur = nxt;
    }
#if BF_ABLATE & 8
    if (a.dbg && lane == 0) {
        for (int k = 0; k < 8; ++k) a.dbg[((size_t)blockIdx.x * Cfg::NW + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

// ---- LDS-DMA variant -----------------------------------------------------------------------------
// The next tile goes global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
// VGPR staging, no ds_write pass): it is issued right after the conv1 -> conv2 barrier, when nobody
// reads the input tile any more (the residual comes from global/L2), and lands while conv2 runs.
// Two barriers per tile; 44 fewer VGPRs.  Out-of-image elements are sourced from a zero-filled
// global line.  LDS = intermediate tile, then the input tile padded to whole wave-instructions.
template <class Cfg>
struct FusedDmaCfg {
    static constexpr int TIN_PAD_FLOATS = Cfg::PF * Cfg::NT * 4;
    static constexpr int TMID_FLOATS = Cfg::MH * Cfg::MW * 16;
    static constexpr int LDS_BYTES = (TMID_FLOATS + TIN_PAD_FLOATS) * 4;
};

template <class Cfg, bool INTERIOR>
__device__ __forceinline__ void fused_dma(const FusedBlockArgs& a, const FusedTile& t, float* __restrict__ tin, int tid, int wave)
{
    constexpr int RW = Cfg::IW * 4;
    const char* img = reinterpret_cast<const char*>(a.in + t.img);
    int row = tid / RW, rem = tid - row * RW;
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const bool in_tile = (i + 1) * Cfg::NT <= Cfg::IN4 || tid + i * Cfg::NT < Cfg::IN4;
        const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + (rem >> 2);
        const bool inside = INTERIOR || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
        const unsigned off = ((unsigned)(gy * a.W + gx) * 16u + (unsigned)(rem & 3) * 4u) * 4u;
        const char* src = (in_tile && inside) ? img + off : reinterpret_cast<const char*>(a.zeros);
        // LDS destination = wave-uniform base + lane*16 (hardware); element n = tid + i*NT -> byte n*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tin + (i * Cfg::NT + wave * 64) * 4),
                                         16, 0, 0);
        rem += Cfg::NT % RW;
        row += Cfg::NT / RW;
        if (rem >= RW) { rem -= RW; ++row; }
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_dma_kernel(FusedBlockArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tmid = lds;                                       // [MH][MW][16]
    float* tin = lds + FusedDmaCfg<Cfg>::TMID_FLOATS;        // [IH][IW][16] + pad
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FusedLane L0;
    L0.p = lane & 15;
    L0.q = lane >> 4;
    L0.row_c = L0.p * 16 + L0.q * 4;
    L0.strip_in = ((L0.p >> 1) * Cfg::IW + (L0.p & 1)) * 16 + L0.q * 4;
    L0.strip_mid = ((L0.p >> 1) * Cfg::MW + (L0.p & 1)) * 16 + L0.q * 4;

    float w1[36], w2[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) { w1[i] = a.w1pack[i * 64 + lane]; w2[i] = a.w2pack[i * 64 + lane]; }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + L0.q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + L0.q * 4);

    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);
    int t = t_begin + slot;
    if (t >= t_end) return;
    const int n1 = (Cfg::MG - wave + Cfg::NW - 1) / Cfg::NW;
    const int n2 = (Cfg::OG - wave + Cfg::NW - 1) / Cfg::NW;

    FusedTile cur = fused_tile<Cfg>(a, t);
    fused_dma<Cfg, false>(a, cur, tin, tid, wave);
    __syncthreads();                                         // drains the DMA (vmcnt(0)) and publishes tin

    for (; t < t_end; t += per_label) {
        const int tn = t + per_label;
        const bool has_next = tn < t_end;
        FusedTile nxt = cur;
        if (has_next) nxt = fused_tile<Cfg>(a, tn);
        float* out_tile = a.out + cur.img + ((size_t)cur.y0 * a.W + cur.x0) * 16;
        FusedLane L = L0;
        asm volatile("" : "+v"(L.row_c), "+v"(L.strip_in), "+v"(L.strip_mid));
        const bool interior = tile_interior<Cfg>(a, cur);

        if (interior) {
            conv1_interior<Cfg, 0>(a, tin, tmid, w1, L, cur, wave, n1);
        } else {
            for (int g = wave; g < Cfg::MG;) {
                int g0 = -1, g1 = -1, g2 = -1;
                for (; g < Cfg::MG && g2 < 0; g += Cfg::NW) {
                    if (!mid_group_needed<Cfg>(a, cur, g)) continue;
                    if (g0 < 0) g0 = g; else if (g1 < 0) g1 = g; else g2 = g;
                }
                if (g2 >= 0)      { const int gs[3] = {g0, g1, g2}; conv1_pass<Cfg, 3, false>(a, tin, tmid, w1, gs, L, cur); }
                else if (g1 >= 0) { const int gs[2] = {g0, g1};     conv1_pass<Cfg, 2, false>(a, tin, tmid, w1, gs, L, cur); }
                else if (g0 >= 0) { const int gs[1] = {g0};         conv1_pass<Cfg, 1, false>(a, tin, tmid, w1, gs, L, cur); }
            }
        }
        __syncthreads();                                     // tmid complete; tin is dead (residual from global)
        if (has_next) {
            if (tile_interior<Cfg>(a, nxt)) fused_dma<Cfg, true>(a, nxt, tin, tid, wave);
            else fused_dma<Cfg, false>(a, nxt, tin, tid, wave);
        }
        if (interior) {
            conv2_interior<Cfg, 0, true>(a, tin, tmid, out_tile, w2, sc, sh, L, cur, wave, n2);
        } else {
            for (int g = wave; g < Cfg::OG;) {
                int g0 = -1, g1 = -1, g2 = -1, g3 = -1;
                for (; g < Cfg::OG && g3 < 0; g += Cfg::NW) {
                    if (!out_group_needed<Cfg>(a, cur, g)) continue;
                    if (g0 < 0) g0 = g; else if (g1 < 0) g1 = g; else if (g2 < 0) g2 = g; else g3 = g;
                }
                if (g3 >= 0)      { const int gs[4] = {g0, g1, g2, g3}; conv2_pass<Cfg, 4, false, true>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g2 >= 0) { const int gs[3] = {g0, g1, g2};     conv2_pass<Cfg, 3, false, true>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g1 >= 0) { const int gs[2] = {g0, g1};         conv2_pass<Cfg, 2, false, true>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
                else if (g0 >= 0) { const int gs[1] = {g0};             conv2_pass<Cfg, 1, false, true>(a, tin, tmid, out_tile, w2, sc, sh, gs, L, cur); }
            }
        }
        __syncthreads();                                     // drains the DMA; tin = next tile, tmid free
        cur = nxt;
    }
}

template <class Cfg>
static hipError_t launch_fused_dma(FusedBlockArgs a, int wgs_per_cu, hipStream_t s)
{
    if (!a.zeros) return hipErrorInvalidValue;
    a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
    a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(fused_block_dma_kernel<Cfg>), FusedDmaCfg<Cfg>::LDS_BYTES);      // once per device
        if (ea != hipSuccess) return ea;
    }
    const int resident = 256 * wgs_per_cu;
    int grid = a.ntiles < resident ? a.ntiles : resident;
    if (grid >= 8) grid -= grid % 8;
    hipLaunchKernelGGL(fused_block_dma_kernel<Cfg>, dim3(grid), dim3(Cfg::NT), FusedDmaCfg<Cfg>::LDS_BYTES, s, a);
    return hipGetLastError();
}

// ---- v4: LDS-DMA + immediate-offset addressing ------------------------------------------------------
// Measured on MI355X (profiles/, tools/exp): a non-MFMA VALU instruction costs ~7 SIMD cycles of matrix
// throughput, and the register-prefetch kernel executes 0.74 of them per MFMA (tile fetch address math,
// LDS staging, pass prologues/epilogues) -- that, not barriers or tiling, held it at 61 %.  This variant
// removes them structurally:
//   * wave w owns groups g = w + NW*k.  With RG % NW == 0 and NW % GPR == 0 the group's row is
//     (w/GPR) + (NW/GPR)*k and its column (w%GPR)*16, so every LDS address is ONE per-wave VGPR constant
//     plus a compile-time immediate (ds_read/ds_write offset field); global addresses are a scalar base
//     plus one per-wave VGPR constant.  A pass needs no address VALU at all.
//   * the next tile goes global -> LDS by DMA (global_load_lds_dwordx4) with per-lane constant offsets:
//     no VGPR staging, no ds_write pass, no fetch address math; it is issued after the conv1 -> conv2
//     barrier (the residual is re-read from global/L2) and lands while conv2 runs.  Two barriers per tile.
template <class Cfg>
struct V4 {
    static_assert(Cfg::RG % Cfg::NW == 0 && Cfg::NW % Cfg::GPR == 0, "v4 addressing needs RG % NW == 0 and NW % GPR == 0");
    static constexpr int RSTEP = Cfg::NW / Cfg::GPR;              // rows between a wave's consecutive groups
    static constexpr int K1R = Cfg::RG / Cfg::NW;                  // conv1 row groups per wave
    static constexpr int K1S = (Cfg::SG + Cfg::NW - 1) / Cfg::NW;  // conv1 strip-group slots per wave
    static constexpr int K2 = Cfg::OG / Cfg::NW;                   // conv2 groups per wave (OG % NW == 0 follows)
    static_assert(Cfg::OG % Cfg::NW == 0, "conv2 groups must divide evenly");
    static constexpr int TIN_PAD_FLOATS = Cfg::PF * Cfg::NT * 4;
    static constexpr int TMID_FLOATS = Cfg::MH * Cfg::MW * 16;
    static constexpr int LDS_BYTES = (TMID_FLOATS + TIN_PAD_FLOATS) * 4;
};

struct V4Lane {
    int in_c;        // float offset in the input tile of this wave's row-group pixel (k = 0)
    int mid_c;       // float offset in the intermediate tile (k = 0): conv1 writes, conv2 reads
    int strip_in;    // float offset in the input tile of this wave's strip-group pixel (slot 0)
    int strip_mid;   // float offset in the intermediate tile of this wave's strip-group pixel
    unsigned glob_c; // byte offset in the image of this wave's output pixel (k = 0) relative to the tile origin
    int px;          // column of the lane's pixel inside the tile for row groups: (w%GPR)*16 + p
    int sp;          // p (strip groups: row 8s + p/2, column TW + p%2)
};

// K consecutive groups of this wave starting at slot K0: conv1 (+activation) -> intermediate tile
template <class Cfg, int NG, int K0, bool INTERIOR>
__device__ __forceinline__ void v4_conv1_rows(const FusedBlockArgs& a, const float* __restrict__ tin, float* __restrict__ tmid,
                                              const float (&w1)[36], const V4Lane& L, const FusedTile& t, const int wrow)
{
    constexpr int RS = V4<Cfg>::RSTEP;
    int base[NG];
    f32x4 acc[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        base[j] = L.in_c + (K0 + j) * RS * Cfg::IW * 16;            // VGPR + immediate
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    conv_groups<NG>(tin, base, Cfg::IW, w1, acc);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        f32x4 v = bf_acc_ready(acc[j]);
        if (a.act1_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!INTERIOR) {
            // conv2 must see ZERO padding outside the image, not conv1 evaluated there
            const int gy = t.y0 - 1 + wrow + (K0 + j) * RS;         // scalar
            const int gx = t.x0 - 1 + L.px;
            if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(tmid + L.mid_c + (K0 + j) * RS * Cfg::MW * 16) = v;
    }
}

template <class Cfg, bool INTERIOR>
__device__ __forceinline__ void v4_conv1_strip(const FusedBlockArgs& a, const float* __restrict__ tin, float* __restrict__ tmid,
                                               const float (&w1)[36], const V4Lane& L, const FusedTile& t, const int s)
{
    // strip group s: rows 8s .. 8s+7 (clamped to MH-1 in a partial last group), columns TW, TW+1
    int base[1];
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    int my = 8 * s + (L.sp >> 1);
    int dst;
    if (Cfg::MH % 8 == 0 || 8 * s + 8 <= Cfg::MH) {
        base[0] = L.strip_in;
        dst = L.strip_mid;
    } else {
        my = min(my, Cfg::MH - 1);
        base[0] = (my * Cfg::IW + Cfg::TW + (L.sp & 1)) * 16 + (L.in_c & 12);
        dst = (my * Cfg::MW + Cfg::TW + (L.sp & 1)) * 16 + (L.in_c & 12);
    }
    conv_groups<1>(tin, base, Cfg::IW, w1, acc);
    f32x4 v = bf_acc_ready(acc[0]);
    if (a.act1_relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (!INTERIOR) {
        const int gy = t.y0 - 1 + my, gx = t.x0 - 1 + Cfg::TW + (L.sp & 1);
        if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<f32x4*>(tmid + dst) = v;
}

// conv2 + folded BN + residual (from global) of NG groups starting at slot K0
template <class Cfg, int NG, int K0, bool INTERIOR>
__device__ __forceinline__ void v4_conv2(const FusedBlockArgs& a, const float* __restrict__ tmid, const char* __restrict__ in_tile,
                                         char* __restrict__ out_tile, const float (&w2)[36], const f32x4 sc, const f32x4 sh,
                                         const V4Lane& L, const FusedTile& t, const int wrow)
{
    constexpr int RS = V4<Cfg>::RSTEP;
    int base[NG];
    f32x4 acc[NG], res[NG];
    const size_t rowstep = (size_t)a.W * 64;                         // bytes per image row
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        base[j] = L.mid_c + (K0 + j) * RS * Cfg::MW * 16;
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        res[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int gy = t.y0 + wrow + (K0 + j) * RS;                  // scalar
        if (INTERIOR || (gy < a.H && t.x0 + L.px < a.W))
            res[j] = *reinterpret_cast<const f32x4*>(in_tile + (K0 + j) * RS * rowstep + L.glob_c);
    }
    conv_groups<NG>(tmid, base, Cfg::MW, w2, acc);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int gy = t.y0 + wrow + (K0 + j) * RS;
        if (INTERIOR || (gy < a.H && t.x0 + L.px < a.W)) {
            const f32x4 v = bf_acc_ready(acc[j]) * sc + sh + res[j];
            *reinterpret_cast<f32x4*>(out_tile + (K0 + j) * RS * rowstep + L.glob_c) = v;
        }
    }
}

template <class Cfg, int K0, bool INTERIOR>
__device__ __forceinline__ void v4_conv1_all(const FusedBlockArgs& a, const float* __restrict__ tin, float* __restrict__ tmid,
                                             const float (&w1)[36], const V4Lane& L, const FusedTile& t, const int wrow)
{
    constexpr int K = V4<Cfg>::K1R;
    if constexpr (K0 < K) {
        constexpr int NG = K - K0 >= 4 ? (K - K0 == 4 || K - K0 >= 7 ? 4 : 3) : K - K0;   // 8 -> 4+4, 9 -> 3+3+3, 6 -> 3+3, 5 -> 3+2
        const bool needed = INTERIOR || (t.y0 - 1 + wrow + K0 * V4<Cfg>::RSTEP <= a.H);  // rows beyond H+1 are never read
        if (needed) v4_conv1_rows<Cfg, NG, K0, INTERIOR>(a, tin, tmid, w1, L, t, wrow);
        v4_conv1_all<Cfg, K0 + NG, INTERIOR>(a, tin, tmid, w1, L, t, wrow);
    }
}

template <class Cfg, int K0, bool INTERIOR>
__device__ __forceinline__ void v4_conv2_all(const FusedBlockArgs& a, const float* __restrict__ tmid, const char* __restrict__ in_tile,
                                             char* __restrict__ out_tile, const float (&w2)[36], const f32x4 sc, const f32x4 sh,
                                             const V4Lane& L, const FusedTile& t, const int wrow)
{
    constexpr int K = V4<Cfg>::K2;
    if constexpr (K0 < K) {
        constexpr int NG = K - K0 >= 4 ? (K - K0 == 4 || K - K0 >= 7 ? 4 : 3) : K - K0;   // 7 -> 4+3
        const bool needed = INTERIOR || (t.y0 + wrow + K0 * V4<Cfg>::RSTEP < a.H);
        if (needed) v4_conv2<Cfg, NG, K0, INTERIOR>(a, tmid, in_tile, out_tile, w2, sc, sh, L, t, wrow);
        v4_conv2_all<Cfg, K0 + NG, INTERIOR>(a, tmid, in_tile, out_tile, w2, sc, sh, L, t, wrow);
    }
}

// DMA of a tile: interior = per-lane constant offsets from the tile origin, nothing else
template <class Cfg, bool INTERIOR>
__device__ __forceinline__ void v4_dma(const FusedBlockArgs& a, const FusedTile& t, float* __restrict__ tin, int tid, int wave,
                                       const unsigned (&pfoff)[Cfg::PF])
{
    const char* origin = reinterpret_cast<const char*>(a.in + t.img) + ((ptrdiff_t)(t.y0 - 2) * a.W + (t.x0 - 2)) * 64;
    constexpr int RW = Cfg::IW * 4;
    int row = tid / RW, rem = tid - row * RW;
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const char* src = origin + pfoff[i];
        if (!INTERIOR || (i + 1) * Cfg::NT > Cfg::IN4) {
            const bool in_tile = (i + 1) * Cfg::NT <= Cfg::IN4 || tid + i * Cfg::NT < Cfg::IN4;
            bool inside = true;
            if (!INTERIOR) {
                const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + (rem >> 2);
                inside = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                rem += Cfg::NT % RW;
                row += Cfg::NT / RW;
                if (rem >= RW) { rem -= RW; ++row; }
            }
            if (!(in_tile && inside)) src = reinterpret_cast<const char*>(a.zeros);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tin + (i * Cfg::NT + wave * 64) * 4),
                                         16, 0, 0);
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_v4_kernel(FusedBlockArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tmid = lds;                                       // [MH][MW][16]
    float* tin = lds + V4<Cfg>::TMID_FLOATS;                 // [IH][IW][16] + pad to whole wave-instructions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 15, q = lane >> 4;
    const int wrow = wave / Cfg::GPR, wcol = (wave % Cfg::GPR) * 16;

    V4Lane L0;
    L0.in_c = (wrow * Cfg::IW + wcol + p) * 16 + q * 4;
    L0.mid_c = (wrow * Cfg::MW + wcol + p) * 16 + q * 4;
    L0.strip_in = ((8 * wave + (p >> 1)) * Cfg::IW + Cfg::TW + (p & 1)) * 16 + q * 4;     // strip slot 0: s = wave
    L0.strip_mid = ((8 * wave + (p >> 1)) * Cfg::MW + Cfg::TW + (p & 1)) * 16 + q * 4;
    L0.glob_c = ((unsigned)(wrow * a.W + wcol + p) * 16u + (unsigned)q * 4u) * 4u;
    L0.px = wcol + p;
    L0.sp = p;
    unsigned pfoff[Cfg::PF];
    {
        constexpr int RW = Cfg::IW * 4;
        int row = tid / RW, rem = tid - row * RW;
#pragma unroll
        for (int i = 0; i < Cfg::PF; ++i) {
            pfoff[i] = ((unsigned)(row * a.W + (rem >> 2)) * 16u + (unsigned)(rem & 3) * 4u) * 4u;
            rem += Cfg::NT % RW;
            row += Cfg::NT / RW;
            if (rem >= RW) { rem -= RW; ++row; }
        }
    }

    float w1[36], w2[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) { w1[i] = a.w1pack[i * 64 + lane]; w2[i] = a.w2pack[i * 64 + lane]; }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + q * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + q * 4);

    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);
    int t = t_begin + slot;
    if (t >= t_end) return;

    FusedTile cur = fused_tile<Cfg>(a, t);
    v4_dma<Cfg, false>(a, cur, tin, tid, wave, pfoff);
    __syncthreads();                                         // drains the DMA (vmcnt(0)) and publishes tin

    for (; t < t_end; t += per_label) {
        const int tn = t + per_label;
        const bool has_next = tn < t_end;
        FusedTile nxt = cur;
        if (has_next) nxt = fused_tile<Cfg>(a, tn);
        V4Lane L = L0;
        // opaque inside the loop body: keeps LICM from hoisting every (constant + immediate) address
        asm volatile("" : "+v"(L.in_c), "+v"(L.mid_c), "+v"(L.glob_c));
        const bool interior = tile_interior<Cfg>(a, cur);
        const char* in_tile = reinterpret_cast<const char*>(a.in + cur.img) + ((size_t)cur.y0 * a.W + cur.x0) * 64;
        char* out_tile = reinterpret_cast<char*>(a.out + cur.img) + ((size_t)cur.y0 * a.W + cur.x0) * 64;

        // ---- conv1: input tile -> intermediate tile -------------------------------------------
        if (interior) v4_conv1_all<Cfg, 0, true>(a, tin, tmid, w1, L, cur, wrow);
        else v4_conv1_all<Cfg, 0, false>(a, tin, tmid, w1, L, cur, wrow);
#pragma unroll
        for (int ks = 0; ks < V4<Cfg>::K1S; ++ks) {
            const int s = wave + Cfg::NW * ks;               // strip group index
            if (s < Cfg::SG) {
                V4Lane Ls = L;
                Ls.strip_in = L.strip_in + ks * Cfg::NW * 8 * Cfg::IW * 16;
                Ls.strip_mid = L.strip_mid + ks * Cfg::NW * 8 * Cfg::MW * 16;
                if (interior) v4_conv1_strip<Cfg, true>(a, tin, tmid, w1, Ls, cur, s);
                else if (cur.y0 - 1 + 8 * s <= a.H && cur.x0 - 1 + Cfg::TW <= a.W) v4_conv1_strip<Cfg, false>(a, tin, tmid, w1, Ls, cur, s);
            }
        }
        __syncthreads();                                     // tmid complete; tin is dead (residual comes from global)

        if (has_next) {
            if (tile_interior<Cfg>(a, nxt)) v4_dma<Cfg, true>(a, nxt, tin, tid, wave, pfoff);
            else v4_dma<Cfg, false>(a, nxt, tin, tid, wave, pfoff);
        }
        // ---- conv2 + folded BN + residual -> global ---------------------------------------------
        if (interior) v4_conv2_all<Cfg, 0, true>(a, tmid, in_tile, out_tile, w2, sc, sh, L, cur, wrow);
        else v4_conv2_all<Cfg, 0, false>(a, tmid, in_tile, out_tile, w2, sc, sh, L, cur, wrow);
        __syncthreads();                                     // drains the DMA; tin = next tile, tmid free
        cur = nxt;
    }
}

template <class Cfg>
static hipError_t launch_fused_v4(FusedBlockArgs a, int wgs_per_cu, hipStream_t s)
{
    if (!a.zeros) return hipErrorInvalidValue;
    a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
    a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(fused_block_v4_kernel<Cfg>), V4<Cfg>::LDS_BYTES);      // once per device
        if (ea != hipSuccess) return ea;
    }
    const int resident = 256 * wgs_per_cu;
    int grid = a.ntiles < resident ? a.ntiles : resident;
    if (grid >= 8) grid -= grid % 8;
    hipLaunchKernelGGL(fused_block_v4_kernel<Cfg>, dim3(grid), dim3(Cfg::NT), V4<Cfg>::LDS_BYTES, s, a);
    return hipGetLastError();
}

// 4 (default): v4 LDS-DMA + immediate-offset addressing, 14x32 x4 waves ; 0: register-prefetch 14x32 x4 waves
// (2 workgroups/CU) ; 1: 32x32 x8 waves ; 2: 16x64 x8 waves ; 3: LDS-DMA 14x32 x4 waves.  A negative value
// restores the default.
constexpr int kDefaultFusedTile = 4;
static int g_fused_tile = kDefaultFusedTile;
void bf_set_fused_tile(int v) { g_fused_tile = v < 0 ? kDefaultFusedTile : v; }

template <class Cfg>
static hipError_t launch_fused(FusedBlockArgs a, int wgs_per_cu, hipStream_t s)
{
    a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
    a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(fused_block_kernel<Cfg>), Cfg::LDS_BYTES);      // once per device
        if (ea != hipSuccess) return ea;
    }
    const int resident = 256 * wgs_per_cu;
    int grid = a.ntiles < resident ? a.ntiles : resident;
    if (grid >= 8) grid -= grid % 8;
    hipLaunchKernelGGL(fused_block_kernel<Cfg>, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
}

// name of the kernel bf_launch_fused_block launches (bench.py looks its counter traffic up by this name)
const char* bf_fused_block_kernel_name()
{
    switch (g_fused_tile) {
        case 1: case 2: case 0: return "fused_block_kernel";
        case 3: return "fused_block_dma_kernel";
        default: return "fused_block_v4_kernel";
    }
}

hipError_t bf_launch_fused_block(const FusedBlockArgs& a, hipStream_t s)
{
    switch (g_fused_tile) {
        case 1: return launch_fused<FusedCfg<32, 32, 8>>(a, 1, s);
        case 2: return launch_fused<FusedCfg<16, 64, 8>>(a, 1, s);
        case 3: return launch_fused_dma<FusedCfg<14, 32, 4>>(a, 2, s);
        case 0: return launch_fused<FusedCfg<14, 32, 4>>(a, 2, s);
        default: return launch_fused_v4<FusedCfg<14, 32, 4>>(a, 2, s);
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient of the 3x3 16->16 convolution:
//   dW[tap][ci][co] = sum over pixels of X[pixel + tap][ci] * dY[pixel][co]
// GEMM view: M = ci, N = co, K = pixels (4 per MFMA; k-slot q of MFMA kk is pixel 4kk+q of a
// 16-pixel group, which keeps the ds_read_b32 of both operands bank-conflict free).  One dY
// fragment is shared by the nine taps (nine accumulators = 36 VGPRs).  Persistent workgroups
// keep their partial dW in registers across tiles; partials are reduced in a fixed order
// (no float atomics -> bitwise reproducible).
// ------------------------------------------------------------------------------------------
constexpr int WG_TH = 16, WG_TW = 32;
constexpr int WG_IH = WG_TH + 2, WG_IW = WG_TW + 2;

__global__ __launch_bounds__(256, 2) void wgrad3x3_c16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int B, int H, int W,
                                                              int tiles_x, int tiles_y, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tx_ = lds;                               // [18][34][16]  x with halo
    float* td = lds + WG_IH * WG_IW * 16;           // [16][32][16]  dy
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int tt = t;
        const int txi = tt % tiles_x; tt /= tiles_x;
        const int tyi = tt % tiles_y;
        const int b = tt / tiles_y;
        const int y0 = tyi * WG_TH, x0 = txi * WG_TW;
        const size_t img = (size_t)b * H * W * 16;
        for (int n = tid; n < WG_IH * WG_IW * 4; n += 256) {
            const int row = n / (WG_IW * 4);
            const int rem = n - row * (WG_IW * 4);
            const int gy = y0 - 1 + row, gx = x0 - 1 + (rem >> 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(x + img + ((size_t)gy * W + gx) * 16 + (rem & 3) * 4);
            *reinterpret_cast<float4*>(tx_ + n * 4) = v;
        }
        for (int n = tid; n < WG_TH * WG_TW * 4; n += 256) {
            const int row = n / (WG_TW * 4);
            const int rem = n - row * (WG_TW * 4);
            const int gy = y0 + row, gx = x0 + (rem >> 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < H && gx < W)
                v = *reinterpret_cast<const float4*>(dy + img + ((size_t)gy * W + gx) * 16 + (rem & 3) * 4);
            *reinterpret_cast<float4*>(td + n * 4) = v;
        }
        __syncthreads();
        // wave handles rows 4w..4w+3 of the tile: 8 groups of 16 pixels
        for (int gi = 0; gi < 8; ++gi) {
            const int r = 4 * wave + (gi >> 1), xo = (gi & 1) * 16;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int px = xo + 4 * kk + q;
                const float bv = td[(r * WG_TW + px) * 16 + p];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float av = tx_[((r + tap / 3) * WG_IW + px + tap % 3) * 16 + p];
                    acc[tap] = MFMA(av, bv, acc[tap]);
                }
            }
        }
        __syncthreads();
    }
    // cross-wave reduction through LDS: [4][9][256]
    float* red = lds;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            red[(wave * 9 + tap) * 256 + (4 * q + j) * 16 + p] = acc[tap][j];
    __syncthreads();
    for (int i = tid; i < 2304; i += 256)
        partial[(size_t)blockIdx.x * 2304 + i] = (red[i] + red[2304 + i]) + (red[2 * 2304 + i] + red[3 * 2304 + i]);
}

int bf_wgrad_grid(int B, int H, int W)
{
    const int ntiles = B * ((H + WG_TH - 1) / WG_TH) * ((W + WG_TW - 1) / WG_TW);
    return ntiles < 512 ? ntiles : 512;
}

hipError_t bf_launch_wgrad3x3_c16(const float* x, const float* dy, float* partial, float* dw,
                                  int B, int H, int W, hipStream_t s)
{
    const int tiles_x = (W + WG_TW - 1) / WG_TW, tiles_y = (H + WG_TH - 1) / WG_TH;
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = bf_wgrad_grid(B, H, W);
    constexpr int lds_bytes = (WG_IH * WG_IW + WG_TH * WG_TW) * 16 * 4;   // 71,936
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(wgrad3x3_c16_kernel), lds_bytes);      // once per device
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(wgrad3x3_c16_kernel, dim3(grid), dim3(256), lds_bytes, s, x, dy, partial, B, H, W,
                       tiles_x, tiles_y, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return bf_launch_reduce_partials(partial, grid, 2304, dw, 1.0f, s);
}
