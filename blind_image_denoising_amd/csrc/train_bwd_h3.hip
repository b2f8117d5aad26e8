// Backward of one 3x3 16->16 convolution of a residual block in ONE kernel (split-f16 arithmetic, fp32 NHWC tensors):
//
//     dw[tap][ci][co] = sum_px x[px + tap][ci] * g[px][co]                    (weight gradient, wgrad3x3_h3_kernel's work)
//     dx[px][ci]      = sum_tap sum_co g[px - tap][co] * w[tap][ci][co]       (data gradient, conv3x3_h3_kernel's work)
//
// and, when the convolution is followed by a BatchNorm, that BatchNorm's backward on the way in:
// g = k1[c] * dy + k2[c] * conv_out + k3[c] (bn_bwd_apply_kernel's work) formed while the tile is staged.
//
// Why one kernel: the training step is HBM-bound (every kernel of it runs at ~5 TB/s).  As three kernels the
// gradient g is written once and read twice, dy and conv_out are read once more, x twice: 8 tensor passes for the second
// convolution of a block, 6 for the first.  Here every operand is read once and dx written once: 4 and 5 passes.
// (bfcnn/train_loop.py:273-294 is what this computes a part of: tape.gradient through backbone_blocks.py:174-246.)
//
// Tile = 16 x 32 pixels + 1-pixel halo for x and g, split into hi / lo f16 planes [4][18][34][8 x f16] while staged (the
// layout of conv3x3_h3_kernel), one image for x, one for g.  The weight gradient reads both images with the transposing
// ds_read_b64_tr_b16 (pixel index along K; every lane supplies the address of its own 8-byte chunk, so the planar layout
// serves as well as a pixel-major one), the data gradient streams rows of the g image exactly as conv3x3_h3_kernel does.
// Persistent workgroups (<= 512) walk the tiles; weight-gradient accumulators and BatchNorm sums stay in registers across
// tiles and leave as per-workgroup partials (fixed summation order: bitwise reproducible).
#include "h3_rows.h"

typedef __fp16 tb_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

#ifndef BWD_H3_TH
#define BWD_H3_TH 16               // tile rows: 16 = two workgroups per CU (78.8 KB of images), 8 = three (44 KB; A/B)
#endif
struct BwdH3Geom {
    static constexpr int TH = BWD_H3_TH, TW = 32, IH = TH + 2, IW = TW + 2, R = TH / 2;
    static constexpr int WG_PER_CU = TH == 16 ? 2 : 3;
    static_assert(TH == 16 || TH == 8, "tile rows");
    // plane stride = 128 mod 256: the transposed reads of a half-wave touch planes 0 and 1 of 8 neighbouring pixels
    static constexpr int PLANE = (IH * IW * 16 + 127) / 256 * 256 + 128;               // 9856 >= 9792 (5504 >= 5440)
    static constexpr int IMG = 4 * PLANE;
    static constexpr int LDS_BYTES = 2 * IMG;                                          // 78,848: two workgroups per CU
    static_assert(PLANE >= IH * IW * 16 && PLANE % 256 == 128, "plane stride");
    static_assert(4 * 9 * 256 * 4 <= LDS_BYTES, "weight-gradient reduction reuses the images");
};

__device__ __forceinline__ h8 tb_tr_operand(const char* img, const int addr)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const tb_fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tb_fp16x4*)(img + addr));
    const tb_fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tb_fp16x4*)(img + addr + 16 * 16));
    const u2 ua = __builtin_bit_cast(u2, a), ub = __builtin_bit_cast(u2, b);
    return __builtin_bit_cast(h8, (u4){ua[0], ua[1], ub[0], ub[1]});
}

extern __shared__ __attribute__((aligned(16))) char tb_lds[];       // the kernels' LDS arena (dynamic)

template <int EPI>
struct BwdH3Epi {
    struct Pre {};
    enum { EXTRA_MFMA = 0 };
    // A fresh object per tile, everything but the BatchNorm sums const: per-tile fields that were `mutable` members of one
    // long-lived object ended up in scratch memory (a scratch load per output row), and so did sums reached through a pointer.
    const BwdH3Args& a; const float inv_s;
    const int mask_off;             // arena offset of this lane's 4 channels of output row 0 of its strip in the x image (hi plane)
    const size_t base; const int gy0, gx;
    mutable f32x4 s1, s2;
    __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
    __device__ __forceinline__ f32x4 finish(const int, const f32x4 v, const Pre&) const { return v; }
    __device__ __forceinline__ void operator()(const int o, const f32x4 av) const
    {
        if (gy0 + o < a.H && gx < a.W) {
            const size_t idx = base + (size_t)o * a.W * 16;
            f32x4 v = av * inv_s;
            if (EPI & EPI_MASK) {
                // the activated input IS the mask, and its hi half sits in LDS: x > 0 <=> hi > 0 (the staging keeps x >= 2^-24)
                // (a second global read of the tile missed L2 more often than not: 117 KB per tile, 64 tiles in flight per XCD)
                const h4 mh = *reinterpret_cast<const h4*>(tb_lds + mask_off + o * (BwdH3Geom::IW * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (float)mh[k] > 0.f ? v[k] : 0.f;
            }
            if (EPI & EPI_RES) v += *reinterpret_cast<const f32x4*>(a.res + idx);
            if (EPI & EPI_BNBWD) { s1 += v; s2 += v * *reinterpret_cast<const f32x4*>(a.bnc + idx); }
            *reinterpret_cast<f32x4*>(a.out + idx) = v;
        }
    }
};

template <bool BNAPPLY, int EPI>
__global__ __launch_bounds__(256, BwdH3Geom::WG_PER_CU) void bwd3x3_h3_kernel(BwdH3Args a)
{
    using G = BwdH3Geom;
    char* xs = tb_lds;
    char* gs = tb_lds + G::IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;

    const float inv_s = a.wpack[BF_H3R_WPACK_FLOATS];

    // BatchNorm-backward coefficients of this thread's channel quad (tid & 3 is the quad of every element it stages)
    f32x4 k1 = {1.f, 1.f, 1.f, 1.f}, k2 = {0.f, 0.f, 0.f, 0.f}, k3 = {0.f, 0.f, 0.f, 0.f};
    if (BNAPPLY) {
        k1 = *reinterpret_cast<const f32x4*>(a.coef + (tid & 3) * 4);
        k2 = *reinterpret_cast<const f32x4*>(a.coef + 16 + (tid & 3) * 4);
        k3 = *reinterpret_cast<const f32x4*>(a.coef + 32 + (tid & 3) * 4);
    }

    // transposed-read address of this lane inside a 32-pixel row chunk: pixel 4g + q', channels 4j .. 4j+3 (j = lane & 3)
    const int tr_off = ((lane & 3) >> 1) * G::PLANE + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8;
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int strip = wave & 1, half = wave >> 1;
    const int px_l = strip * 16 + n;                              // column inside the tile
    const int o0 = half * G::R;                                   // first output row of this wave
    const int b1 = (q & 1) * G::PLANE + (o0 * G::IW + px_l) * 16;
    const int mask_lds = (q >> 1) * G::PLANE + ((o0 + 1) * G::IW + px_l + 1) * 16 + (q & 1) * 8;
    f32x4 bs1 = {0.f, 0.f, 0.f, 0.f}, bs2 = {0.f, 0.f, 0.f, 0.f};          // BatchNorm sums across the tiles

    constexpr int NX = (G::IH * G::IW * 4 + 255) / 256;
    // Without the BatchNorm operand there are registers to spare: the x operand of the NEXT tile is requested behind the staging
    // of the current one and stays in flight through its matrix work (+ 1/3 more bytes in flight per workgroup).
    constexpr bool PREX = !BNAPPLY;
    f32x4 px_next[PREX ? NX : 1];
    auto load_x = [&](const int t0n, f32x4 (&dst)[PREX ? NX : 1]) {
        const int t = a.reverse ? a.ntiles - 1 - t0n : t0n;
        const int txi = t % a.tiles_x, rest = t / a.tiles_x;
        const int y0 = (rest % a.tiles_y) * G::TH, x0 = txi * G::TW;
        const size_t img = (size_t)(rest / a.tiles_y) * a.H * a.W * 16;
#pragma unroll
        for (int i = 0; i < (PREX ? NX : 1); ++i) {
            const int e = tid + i * 256;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::IW, col = px - row * G::IW;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            dst[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH * G::IW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                dst[i] = *reinterpret_cast<const f32x4*>(a.x + img + ((size_t)gy * a.W + gx) * 16 + quad * 4);
        }
    };
    if (PREX && (int)blockIdx.x < a.ntiles) load_x(blockIdx.x, px_next);
    // Tiles are dealt round-robin (tile = workgroup + k * grid): at any moment the grid covers one contiguous window of the
    // tensors (every HBM channel busy; a contiguous RUN of tiles per workgroup put all workgroups on the same channels and
    // was 10 % slower), x-neighbours sit on neighbouring workgroups and, with eight tiles per image row, y-neighbours on the
    // same XCD.  a.reverse walks the window from the end of the tensors to the start: the caller alternates it from launch to
    // launch, so that a kernel starts on the part of its input the previous kernel wrote last (still in the Infinity Cache).
    for (int t0 = blockIdx.x; t0 < a.ntiles; t0 += gridDim.x) {
        const int t = a.reverse ? a.ntiles - 1 - t0 : t0;
        int tt = t;
        const int txi = tt % a.tiles_x; tt /= a.tiles_x;
        const int tyi = tt % a.tiles_y;
        const int b = tt / a.tiles_y;
        const int y0 = tyi * G::TH, x0 = txi * G::TW;
        const size_t img = (size_t)b * a.H * a.W * 16;

        // ---- stage x and g (1-pixel halo, zero outside the image): all loads of a thread first, then split + store ----
        {
            f32x4 rx[NX], rg[NX], rc[BNAPPLY ? NX : 1];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = tid + i * 256;
                const int px = e >> 2, quad = e & 3;
                const int row = px / G::IW, col = px - row * G::IW;
                const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                rg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (BNAPPLY) rc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (PREX) rx[i] = px_next[i];
                if (e < G::IH * G::IW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    const size_t idx = img + ((size_t)gy * a.W + gx) * 16 + quad * 4;
                    if (!PREX) rx[i] = *reinterpret_cast<const f32x4*>(a.x + idx);
                    rg[i] = *reinterpret_cast<const f32x4*>(a.g + idx);
                    if (BNAPPLY) rc[i] = *reinterpret_cast<const f32x4*>(a.c + idx);
                }
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int e = tid + i * 256;
                if (e < G::IH * G::IW * 4) {
                    const int px = e >> 2, quad = e & 3;
                    const int off = (quad >> 1) * G::PLANE + px * 16 + (quad & 1) * 8;
                    h4 hi, lo;
                    f32x4 xv = rx[i];
                    if (EPI & EPI_MASK) {
                        // the epilogue takes the ReLU mask from this image (x > 0 <=> hi > 0): a positive activation below the
                        // smallest f16 (2^-24) must not vanish in the split -- it is raised to 2^-24 (one element in 8 million of
                        // a normal sample; the absolute error stays inside the split's own 2^-25 .. 2^-24 floor)
#pragma unroll
                        for (int k = 0; k < 4; ++k) xv[k] = xv[k] > 0.f ? fmaxf(xv[k], 0x1p-24f) : xv[k];
                    }
                    h3_split(xv, hi, lo);
                    *reinterpret_cast<h4*>(xs + off) = hi;
                    *reinterpret_cast<h4*>(xs + off + 2 * G::PLANE) = lo;
                    f32x4 gv = rg[i];
                    if (BNAPPLY) {
                        // dc = k1 dy + k2 c + k3 inside the image, 0 outside (SAME padding of the data gradient)
                        const int row = px / G::IW, col = px - row * G::IW;
                        const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                        const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                        for (int k = 0; k < 4; ++k) gv[k] = in ? fmaf(k1[k], rg[i][k], fmaf(k2[k], rc[i][k], k3[k])) : 0.f;
                    }
                    h3_split(gv, hi, lo);
                    *reinterpret_cast<h4*>(gs + off) = hi;
                    *reinterpret_cast<h4*>(gs + off + 2 * G::PLANE) = lo;
                }
            }
        }
        __syncthreads();
        if (PREX && t0 + (int)gridDim.x < a.ntiles) load_x(t0 + gridDim.x, px_next);

        // data-gradient weights: 12 A-operand images (pack_h3_train_kernel, transposed + flipped pack).  Fetched per tile (L2
        // hits, in flight behind the weight-gradient MFMAs) rather than once per kernel: 52 registers that would otherwise be
        // live through the staging phase, whose 30 outstanding 16-byte loads per thread then spill (41 VGPRs measured).
        h8 w[13];
        {
            int opaque = 0;
            asm volatile("" : "+s"(opaque));                    // keeps hipcc from hoisting the loads out of the tile loop
            const h8* wp = reinterpret_cast<const h8*>(a.wpack + opaque) + lane;
#pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = wp[i * 64];
            w[12] = w[0];
        }

        // ---- weight gradient: wave handles rows 4w .. 4w+3 of the tile (TH / 4 rows), one K chunk of 32 pixels per row ----
#pragma unroll
        for (int rr = 0; rr < G::TH / 4; ++rr) {
            const int r = (G::TH / 4) * wave + rr;
            const int gaddr = ((r + 1) * G::IW + 1) * 16 + tr_off;
            const h8 bh = tb_tr_operand(gs, gaddr);
            const h8 bl = tb_tr_operand(gs + 2 * G::PLANE, gaddr);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ax = ((r + tap / 3) * G::IW + tap % 3) * 16 + tr_off;
                const h8 ah = tb_tr_operand(xs, ax);
                const h8 al = tb_tr_operand(xs + 2 * G::PLANE, ax);
                acc[tap] = MFMA_H(ah, bh, acc[tap]);
                acc[tap] = MFMA_H(al, bh, acc[tap]);
                acc[tap] = MFMA_H(ah, bl, acc[tap]);
            }
        }

        // ---- data gradient: 8 rows of one 16-column strip per wave, streamed from the g image ----
        {
            const int gx = x0 + px_l;
            const BwdH3Epi<EPI> epi{a, inv_s, mask_lds, img + ((size_t)(y0 + o0) * a.W + gx) * 16 + q * 4, y0 + o0, gx, bs1, bs2};
            h3r_rows<G::R, G::IW * 16, 2 * G::PLANE>(gs, b1 + (q >> 1) * 16, b1 + 32 + (q >> 1) * 2 * G::PLANE, w, epi, H3NoHook{});
            bs1 = epi.s1; bs2 = epi.s2;
        }
        __syncthreads();                       // images free for the next tile
    }

    // ---- per-workgroup partials: weight gradient [9][16][16] (D[ci = 4q + j][co = p] per lane), BatchNorm sums [32] ----
    float* red = reinterpret_cast<float*>(tb_lds);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = bf_acc_ready(acc[tap]);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(wave * 9 + tap) * 256 + (4 * q + j) * 16 + n] = v[j];
    }
    __syncthreads();
    for (int i = tid; i < 2304; i += 256)
        a.wpartial[(size_t)blockIdx.x * 2304 + i] = (red[i] + red[2304 + i]) + (red[2 * 2304 + i] + red[3 * 2304 + i]);
    if (EPI & EPI_BNBWD) {
        f32x4 s1 = bs1, s2 = bs2;
        // over the 16 pixel lanes that share a channel quad, then over the 4 waves (fixed order)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2[c] += __shfl_xor(s2[c], m);
            }
        }
        __syncthreads();
        if (n == 0) {
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

// ------------------------------------------------------------------------------------------------------------------
// The same work with the staging of tile t+1 overlapped with the arithmetic of tile t: 512 threads, ONE workgroup per CU,
// two sets of LDS images (157 KB).  A thread issues the global loads of the next tile (15 x 16 bytes with the BatchNorm
// operand, 10 without) BEFORE the matrix work of the current one and converts / stores them into the other image set
// after it; one barrier per tile.  The two-workgroups-per-CU form above only overlaps when the two workgroups happen to be
// in different phases and runs at 4.3-5.0 TB/s; plain streaming kernels with the same read : write mix reach ~6 on this
// chip (tools/exp/copy_bw.py).  Waves: weight gradient rows 2w, 2w+1; data gradient strip w & 1, rows 4 (w >> 1) .. +3.
// ------------------------------------------------------------------------------------------------------------------
template <bool BNAPPLY, int EPI>
__global__ __launch_bounds__(512, 1) void bwd3x3_h3d_kernel(BwdH3Args a)
{
    using G = BwdH3Geom;
    constexpr int NT = 512, NW = 8, RQ = G::TH / (NW / 2);              // 4 rows of a strip per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const float inv_s = a.wpack[BF_H3R_WPACK_FLOATS];
    f32x4 k1 = {1.f, 1.f, 1.f, 1.f}, k2 = {0.f, 0.f, 0.f, 0.f}, k3 = {0.f, 0.f, 0.f, 0.f};
    if (BNAPPLY) {
        k1 = *reinterpret_cast<const f32x4*>(a.coef + (tid & 3) * 4);
        k2 = *reinterpret_cast<const f32x4*>(a.coef + 16 + (tid & 3) * 4);
        k3 = *reinterpret_cast<const f32x4*>(a.coef + 32 + (tid & 3) * 4);
    }
    const int tr_off = ((lane & 3) >> 1) * G::PLANE + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8;
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int strip = wave & 1, quarter = wave >> 1;
    const int px_l = strip * 16 + n;
    const int o0 = quarter * RQ;
    const int b1 = (q & 1) * G::PLANE + (o0 * G::IW + px_l) * 16;
    const int mask_off = (q >> 1) * G::PLANE + ((o0 + 1) * G::IW + px_l + 1) * 16 + (q & 1) * 8;
    f32x4 bs1 = {0.f, 0.f, 0.f, 0.f}, bs2 = {0.f, 0.f, 0.f, 0.f};

    constexpr int NX = (G::IH * G::IW * 4 + NT - 1) / NT;                 // 5
    f32x4 rx[NX], rg[NX], rc[BNAPPLY ? NX : 1];
    struct Tile { int y0, x0; size_t img; };
    auto tile_of = [&](const int t0) {
        const int t = a.reverse ? a.ntiles - 1 - t0 : t0;
        const int txi = t % a.tiles_x, rest = t / a.tiles_x;
        return Tile{(rest % a.tiles_y) * G::TH, txi * G::TW, (size_t)(rest / a.tiles_y) * a.H * a.W * 16};
    };
    auto fetch = [&](const Tile& tl) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * NT;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::IW, col = px - row * G::IW;
            const int gy = tl.y0 - 1 + row, gx = tl.x0 - 1 + col;
            rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (BNAPPLY) rc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH * G::IW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t idx = tl.img + ((size_t)gy * a.W + gx) * 16 + quad * 4;
                rx[i] = *reinterpret_cast<const f32x4*>(a.x + idx);
                rg[i] = *reinterpret_cast<const f32x4*>(a.g + idx);
                if (BNAPPLY) rc[i] = *reinterpret_cast<const f32x4*>(a.c + idx);
            }
        }
    };
    auto stash = [&](const Tile& tl, char* xs, char* gs) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * NT;
            if (e < G::IH * G::IW * 4) {
                const int px = e >> 2, quad = e & 3;
                const int off = (quad >> 1) * G::PLANE + px * 16 + (quad & 1) * 8;
                h4 hi, lo;
                f32x4 xv = rx[i];
                if (EPI & EPI_MASK) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xv[k] = xv[k] > 0.f ? fmaxf(xv[k], 0x1p-24f) : xv[k];      // see the kernel above
                }
                h3_split(xv, hi, lo);
                *reinterpret_cast<h4*>(xs + off) = hi;
                *reinterpret_cast<h4*>(xs + off + 2 * G::PLANE) = lo;
                f32x4 gv = rg[i];
                if (BNAPPLY) {
                    const int row = px / G::IW, col = px - row * G::IW;
                    const int gy = tl.y0 - 1 + row, gx = tl.x0 - 1 + col;
                    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) gv[k] = in ? fmaf(k1[k], rg[i][k], fmaf(k2[k], rc[i][k], k3[k])) : 0.f;
                }
                h3_split(gv, hi, lo);
                *reinterpret_cast<h4*>(gs + off) = hi;
                *reinterpret_cast<h4*>(gs + off + 2 * G::PLANE) = lo;
            }
        }
    };

    int t0 = blockIdx.x;
    Tile cur = tile_of(t0 < a.ntiles ? t0 : 0);
    if (t0 < a.ntiles) {
        fetch(cur);
        stash(cur, tb_lds, tb_lds + G::IMG);
    }
    __syncthreads();
    for (int k = 0; t0 < a.ntiles; t0 += gridDim.x, ++k) {
        char* xs = tb_lds + (k & 1) * 2 * G::IMG;
        char* gs = xs + G::IMG;
        const bool more = t0 + (int)gridDim.x < a.ntiles;
        const Tile nxt = tile_of(more ? t0 + (int)gridDim.x : t0);
        if (more) fetch(nxt);                                   // in flight behind the matrix work below

        h8 w[13];
        {
            int opaque = 0;
            asm volatile("" : "+s"(opaque));
            const h8* wp = reinterpret_cast<const h8*>(a.wpack + opaque) + lane;
#pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = wp[i * 64];
            w[12] = w[0];
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr;
            const int gaddr = ((r + 1) * G::IW + 1) * 16 + tr_off;
            const h8 bh = tb_tr_operand(gs, gaddr);
            const h8 bl = tb_tr_operand(gs + 2 * G::PLANE, gaddr);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ax = ((r + tap / 3) * G::IW + tap % 3) * 16 + tr_off;
                const h8 ah = tb_tr_operand(xs, ax);
                const h8 al = tb_tr_operand(xs + 2 * G::PLANE, ax);
                acc[tap] = MFMA_H(ah, bh, acc[tap]);
                acc[tap] = MFMA_H(al, bh, acc[tap]);
                acc[tap] = MFMA_H(ah, bl, acc[tap]);
            }
        }
        {
            const int gx = cur.x0 + px_l;
            const BwdH3Epi<EPI> epi{a, inv_s, (k & 1) * 2 * G::IMG + mask_off, cur.img + ((size_t)(cur.y0 + o0) * a.W + gx) * 16 + q * 4, cur.y0 + o0, gx,
                                    bs1, bs2};
            h3r_rows<RQ, G::IW * 16, 2 * G::PLANE>(gs, b1 + (q >> 1) * 16, b1 + 32 + (q >> 1) * 2 * G::PLANE, w, epi, H3NoHook{});
            bs1 = epi.s1; bs2 = epi.s2;
        }
        if (more) stash(nxt, tb_lds + ((k + 1) & 1) * 2 * G::IMG, tb_lds + ((k + 1) & 1) * 2 * G::IMG + G::IMG);
        cur = nxt;
        __syncthreads();                       // the other image set is complete; this one is free
    }

    float* red = reinterpret_cast<float*>(tb_lds);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = bf_acc_ready(acc[tap]);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(wave * 9 + tap) * 256 + (4 * q + j) * 16 + n] = v[j];
    }
    __syncthreads();
    for (int i = tid; i < 2304; i += NT) {
        float sacc = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) sacc += red[wv * 2304 + i];
        a.wpartial[(size_t)blockIdx.x * 2304 + i] = sacc;
    }
    if (EPI & EPI_BNBWD) {
        f32x4 s1 = bs1, s2 = bs2;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2[c] += __shfl_xor(s2[c], m);
            }
        }
        __syncthreads();
        if (n == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[wave * 32 + q * 4 + c] = s1[c];
                red[wave * 32 + 16 + q * 4 + c] = s2[c];
            }
        }
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) sacc += red[wv * 32 + tid];
            a.stats[(size_t)blockIdx.x * 32 + tid] = sacc;
        }
    }
}

int bf_bwd3x3_h3_grid_ex(int B, int H, int W, int dbuf)
{
    using G = BwdH3Geom;
    const int64_t ntiles = (int64_t)B * ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
    const int cap = dbuf ? 256 : 256 * G::WG_PER_CU;
    return (int)(ntiles < cap ? ntiles : cap);
}

int bf_bwd3x3_h3_grid(int B, int H, int W)
{
    using G = BwdH3Geom;
    const int64_t ntiles = (int64_t)B * ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
    return (int)(ntiles < 256 * G::WG_PER_CU ? ntiles : 256 * G::WG_PER_CU);
}

// a.wpartial: [grid][2304] floats, a.stats: [grid][32]; dw <- sum of the partials (fixed order; skipped when dw is nullptr).  epi: 0, EPI_MASK, EPI_RES
// or EPI_RES | EPI_BNBWD; a.coef != nullptr selects the BatchNorm-backward staging (a.c = the BatchNorm's input).
// a.out must not alias a.x, a.g or a.c (all three are read with a halo); it may alias a.res.
hipError_t bf_launch_bwd3x3_h3(const BwdH3Args& a0, int epi, float* dw, hipStream_t s)
{
    using G = BwdH3Geom;
    BwdH3Args a = a0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    if (a.out == a.x || a.out == a.g || (a.coef && a.out == a.c)) return hipErrorInvalidValue;
    int grid = bf_bwd3x3_h3_grid(a.B, a.H, a.W);
    const bool bn = a.coef != nullptr;
    if (a.dbuf) {
        // one 512-thread workgroup per CU; the partial buffers are sized for bf_bwd3x3_h3_grid rows (>= 256)
        grid = a.ntiles < 256 ? a.ntiles : 256;
#define BF_CASE_D(BN, E)                                                                                                  \
    if (bn == BN && epi == (E)) {                                                                                        \
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(bwd3x3_h3d_kernel<BN, E>), 2 * G::LDS_BYTES);  \
        if (ea != hipSuccess) return ea;                                                                                 \
        hipLaunchKernelGGL((bwd3x3_h3d_kernel<BN, E>), dim3(grid), dim3(512), 2 * G::LDS_BYTES, s, a);                    \
    } else
        BF_CASE_D(true, EPI_MASK)
        BF_CASE_D(true, 0)
        BF_CASE_D(false, EPI_MASK)
        BF_CASE_D(false, 0)
        BF_CASE_D(false, EPI_RES)
        BF_CASE_D(false, EPI_RES | EPI_BNBWD)
        return hipErrorInvalidValue;
#undef BF_CASE_D
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        // rows grid .. bf_bwd3x3_h3_grid - 1 of the partial buffers are not written: sum the rows that are
        if (a.grid_out) *a.grid_out = grid;
        return dw ? bf_launch_reduce_partials(a.wpartial, grid, 2304, dw, 1.0f, s) : hipSuccess;
    }
    if (a.grid_out) *a.grid_out = grid;
#define BF_CASE(BN, E)                                                                                                    \
    if (bn == BN && epi == (E)) {                                                                                        \
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(bwd3x3_h3_kernel<BN, E>), G::LDS_BYTES);       \
        if (ea != hipSuccess) return ea;                                                                                 \
        hipLaunchKernelGGL((bwd3x3_h3_kernel<BN, E>), dim3(grid), dim3(256), G::LDS_BYTES, s, a);                         \
    } else
    BF_CASE(true, EPI_MASK)
    BF_CASE(true, 0)
    BF_CASE(false, EPI_MASK)
    BF_CASE(false, 0)
    BF_CASE(false, EPI_RES)
    BF_CASE(false, EPI_RES | EPI_BNBWD)
    return hipErrorInvalidValue;
#undef BF_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return dw ? bf_launch_reduce_partials(a.wpartial, grid, 2304, dw, 1.0f, s) : hipSuccess;      // nullptr: the caller sums a.wpartial later
}
