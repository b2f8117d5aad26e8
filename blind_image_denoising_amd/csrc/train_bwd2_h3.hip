// Backward of BOTH convolutions of a [3,3] residual block in ONE kernel (split-f16 arithmetic, fp32 NHWC tensors):
//     dc  = k1 dy + k2 c + k3                                   BatchNorm backward of the block's second convolution (on load)
//     dw2 = T^T dc ,  dT = dgrad2(dc) * (T > 0)                 second convolution (T = relu(conv1 A), its input)
//     dw1 = A^T dT ,  dA' = dgrad1(dT) + dy [+ sums for the BatchNorm of the block in front]      first convolution + skip
// (tape.gradient through bfcnn/backbone_blocks.py:174-246, bfcnn/train_loop.py:273-294.)
//
// As two launches of bwd3x3_h3_kernel (train_bwd_h3.hip) dT is written once and read once, dy is read twice: 9 tensor passes per block;
// here dT only ever exists in LDS: T, dy, c, A [, the next BatchNorm's input] read once, dA' written once: 6 [7] passes.
//
// A workgroup (512 threads, one per CU) owns a 16 x 32 tile.  Phase 1 works one pixel further out than the tile: dc is staged on the tile
// + 2 pixels (20 x 36, in a 20 x 50 image whose last 14 columns are zero), the data gradient of the second convolution is formed on the
// tile + 1 pixel (18 x 34: three 16-column strips x two 9-row runs on six waves, the third strip computing 14 columns nobody keeps), masked by
// T > 0, zeroed outside the image (it is the first convolution's SAME padding) and written as a split-f16 LDS image; the weight gradient of
// the second convolution sums over the tile's own 16 x 32 pixels only.  Phase 2 is bwd3x3_h3d_kernel's work with that LDS image as its
// gradient operand and A (requested during phase 1, staged over T's image) as its input.  The T / dy / c tiles of the NEXT tile are
// requested behind phase 2 and stay in registers across the tile boundary.
#include "h3_rows.h"

// 1: A is requested before phase 1 and stays in registers through it (hidden latency, ~50 spilled registers); 0: requested after it
#ifndef BWD2_PREFETCH_A
#define BWD2_PREFETCH_A 1
#endif
typedef __fp16 tb2_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct Bwd2Geom {
    static constexpr int TH = 16, TW = 32, NT = 512, NW = 8;
    static constexpr int IH1 = TH + 2, IW1 = TW + 2;                                   // T / A / dT images: tile + 1
    static constexpr int PLANE1 = ((IH1 * IW1 * 16 + 255) / 256) * 256 + 128 - 256;    // 9856 (128 mod 256: transposed reads)
    static constexpr int IMG1 = 4 * PLANE1;
    static constexpr int IH2 = TH + 4, IW2 = 3 * 16 + 2, RW2 = TW + 4;                 // dc image: tile + 2 rows, three strips + halo columns
    static constexpr int PLANE2 = IH2 * IW2 * 16;                                      // 16000 = 128 mod 256
    static constexpr int IMG2 = 4 * PLANE2;
    static constexpr int LDS_BYTES = IMG2 + 2 * IMG1;                                  // 142,848
    static constexpr int R2 = 9, RQ = TH / (NW / 2);                                   // rows per run: phase 1 (six waves), phase 2
    static_assert(PLANE1 >= IH1 * IW1 * 16 && PLANE1 % 256 == 128 && PLANE2 % 256 == 128, "plane strides");
    static_assert(LDS_BYTES <= 160 * 1024 && NW * 5 * 256 * 4 <= LDS_BYTES, "LDS");
};

extern __shared__ __attribute__((aligned(16))) char tb2_lds[];

__device__ __forceinline__ h8 tb2_tr_operand(const char* img, const int addr)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const tb2_fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tb2_fp16x4*)(img + addr));
    const tb2_fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tb2_fp16x4*)(img + addr + 16 * 16));
    const u2 ua = __builtin_bit_cast(u2, a), ub = __builtin_bit_cast(u2, b);
    return __builtin_bit_cast(h8, (u4){ua[0], ua[1], ub[0], ub[1]});
}

// phase 1 epilogue: dT of the tile + 1 region -> masked, zero outside the image, split into the dT image (fresh object per tile, all const)
struct Bwd2MidEpi {
    struct Pre {};
    enum { EXTRA_MFMA = 0 };
    const float inv_s;
    const int t_off;                // arena offset of this lane's 4 channels of row 0 of its run in the T image (hi plane)
    const int d_off;                // same position in the dT image
    const int gy0, gx, H, W;        // image coordinates of row 0 of the run / of the lane's column
    const bool col_kept;            // the lane's column is inside the 34 columns of the region
    __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
    __device__ __forceinline__ f32x4 finish(const int, const f32x4 v, const Pre&) const { return v; }
    __device__ __forceinline__ void operator()(const int o, const f32x4 av) const
    {
        if (!col_kept) return;
        f32x4 v = av * inv_s;
        const h4 mh = *reinterpret_cast<const h4*>(tb2_lds + t_off + o * (Bwd2Geom::IW1 * 16));
        const bool in = gy0 + o >= 0 && gy0 + o < H && gx >= 0 && gx < W;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (in && (float)mh[k] > 0.f) ? v[k] : 0.f;
        h4 hi, lo;
        h3_split(v, hi, lo);
        char* p = tb2_lds + d_off + o * (Bwd2Geom::IW1 * 16);
        *reinterpret_cast<h4*>(p) = hi;
        *reinterpret_cast<h4*>(p + 2 * Bwd2Geom::PLANE1) = lo;
    }
};

// phase 2 epilogue: dA' = dgrad1 + skip [+ BatchNorm sums of the block in front]
template <bool BNBWD>
struct Bwd2OutEpi {
    struct Pre {};
    enum { EXTRA_MFMA = 0 };
    const Bwd2H3Args& a; const float inv_s;
    const size_t base; const int gy0, gx;
    mutable f32x4 s1, s2;
    __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
    __device__ __forceinline__ f32x4 finish(const int, const f32x4 v, const Pre&) const { return v; }
    __device__ __forceinline__ void operator()(const int o, const f32x4 av) const
    {
        if (gy0 + o < a.H && gx < a.W) {
            const size_t idx = base + (size_t)o * a.W * 16;
            f32x4 v = av * inv_s + *reinterpret_cast<const f32x4*>(a.dy + idx);
            if (BNBWD) { s1 += v; s2 += v * *reinterpret_cast<const f32x4*>(a.bnc + idx); }
            *reinterpret_cast<f32x4*>(a.out + idx) = v;
        }
    }
};

template <bool BNBWD>
__global__ __launch_bounds__(Bwd2Geom::NT, 1) void bwd2_h3_kernel(Bwd2H3Args a)
{
    using G = Bwd2Geom;
    char* dcs = tb2_lds;                               // dc image  [4 planes][20][50][8 x f16]
    char* xs = tb2_lds + G::IMG2;                      // T, then A [4 planes][18][34][8]
    char* ds = xs + G::IMG1;                           // dT        [4 planes][18][34][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const float inv_s2 = a.wpack2[BF_H3R_WPACK_FLOATS], inv_s1 = a.wpack1[BF_H3R_WPACK_FLOATS];
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(a.coef + (tid & 3) * 4);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(a.coef + 16 + (tid & 3) * 4);
    const f32x4 k3 = *reinterpret_cast<const f32x4*>(a.coef + 32 + (tid & 3) * 4);

    // the dc image's columns 36 .. 49 are never staged: zero once (the third strip of phase 1 reads them)
    for (int i = tid * 16; i < G::IMG2; i += G::NT * 16) *reinterpret_cast<f32x4*>(dcs + i) = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tr1 = ((lane & 3) >> 1) * G::PLANE1 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8;
    const int tr2 = ((lane & 3) >> 1) * G::PLANE2 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8;
    // weight-gradient accumulators: wave w sums taps 5 (w >> 2) .. (five, then four) over rows 4 (w & 3) .. + 3 of the tile: 5 + 5
    // accumulators instead of 9 + 9 (the two convolutions' sets are both live through the whole kernel)
    constexpr int NTAP = 5;
    const int tap0 = NTAP * (wave >> 2), wrow0 = 4 * (wave & 3);
    f32x4 acc2[NTAP], acc1[NTAP];
#pragma unroll
    for (int i = 0; i < NTAP; ++i) { acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    f32x4 bs1 = {0.f, 0.f, 0.f, 0.f}, bs2 = {0.f, 0.f, 0.f, 0.f};

    // phase 1 runs: waves 0..5 = strip (wave % 3) x half (wave / 3) of the 18 x 34 region
    const int s2 = wave % 3, h2 = wave / 3;
    const int c2 = s2 * 16 + n, o02 = h2 * G::R2;
    const int b12 = (q & 1) * G::PLANE2 + (o02 * G::IW2 + c2) * 16;
    // phase 2 runs: strip wave & 1, rows 4 (wave >> 1) .. + 3 of the tile
    const int strip = wave & 1, quarter = wave >> 1;
    const int px_l = strip * 16 + n, o0 = quarter * G::RQ;
    const int b11 = (q & 1) * G::PLANE1 + (o0 * G::IW1 + px_l) * 16;

    constexpr int NX1 = (G::IH1 * G::IW1 * 4 + G::NT - 1) / G::NT;       // 5: T / A elements per thread
    constexpr int NX2 = (G::IH2 * G::RW2 * 4 + G::NT - 1) / G::NT;       // 6: dy / c elements per thread
    struct Tile { int y0, x0; size_t img; };
    auto tile_of = [&](const int t0) {
        const int t = a.reverse ? a.ntiles - 1 - t0 : t0;
        const int txi = t % a.tiles_x, rest = t / a.tiles_x;
        return Tile{(rest % a.tiles_y) * G::TH, txi * G::TW, (size_t)(rest / a.tiles_y) * a.H * a.W * 16};
    };
    f32x4 rt[NX1], rdy[NX2], rc[NX2], ra[NX1];
    // T (or A) on the tile + 1, dy and c on the tile + 2; zero outside the image
    // (the element index math of the staging helpers hangs on an OPAQUE copy of tid made at every call: hipcc otherwise hoists the ~40
    // per-thread offsets of the eleven elements out of the tile loop and keeps them in registers through both matrix phases)
    auto opaque_tid = [&]() { int v = tid; asm volatile("" : "+v"(v)); return v; };
    auto fetch1 = [&](const Tile& tl, const float* __restrict__ src, f32x4 (&dst)[NX1]) {
        const int tz = opaque_tid();
#pragma unroll
        for (int i = 0; i < NX1; ++i) {
            const int e = tz + i * G::NT;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::IW1, col = px - row * G::IW1;
            const int gy = tl.y0 - 1 + row, gx = tl.x0 - 1 + col;
            dst[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH1 * G::IW1 * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                dst[i] = *reinterpret_cast<const f32x4*>(src + tl.img + ((size_t)gy * a.W + gx) * 16 + quad * 4);
        }
    };
    auto fetch2 = [&](const Tile& tl) {
        const int tz = opaque_tid();
#pragma unroll
        for (int i = 0; i < NX2; ++i) {
            const int e = tz + i * G::NT;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::RW2, col = px - row * G::RW2;
            const int gy = tl.y0 - 2 + row, gx = tl.x0 - 2 + col;
            rdy[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            rc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH2 * G::RW2 * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t idx = tl.img + ((size_t)gy * a.W + gx) * 16 + quad * 4;
                rdy[i] = *reinterpret_cast<const f32x4*>(a.dy + idx);
                rc[i] = *reinterpret_cast<const f32x4*>(a.c + idx);
            }
        }
    };
    auto stash1 = [&](const f32x4 (&src)[NX1], const bool keep_mask) {
        const int tz = opaque_tid();
#pragma unroll
        for (int i = 0; i < NX1; ++i) {
            const int e = tz + i * G::NT;
            if (e < G::IH1 * G::IW1 * 4) {
                const int px = e >> 2, quad = e & 3;
                const int off = (quad >> 1) * G::PLANE1 + px * 16 + (quad & 1) * 8;
                f32x4 xv = src[i];
                if (keep_mask) {
                    // the mask of phase 1 is read from this image (T > 0 <=> hi > 0): see bwd3x3_h3_kernel
#pragma unroll
                    for (int k = 0; k < 4; ++k) xv[k] = xv[k] > 0.f ? fmaxf(xv[k], 0x1p-24f) : xv[k];
                }
                h4 hi, lo;
                h3_split(xv, hi, lo);
                *reinterpret_cast<h4*>(xs + off) = hi;
                *reinterpret_cast<h4*>(xs + off + 2 * G::PLANE1) = lo;
            }
        }
    };
    auto stash2 = [&](const Tile& tl) {
        const int tz = opaque_tid();
#pragma unroll
        for (int i = 0; i < NX2; ++i) {
            const int e = tz + i * G::NT;
            if (e < G::IH2 * G::RW2 * 4) {
                const int px = e >> 2, quad = e & 3;
                const int row = px / G::RW2, col = px - row * G::RW2;
                const int gy = tl.y0 - 2 + row, gx = tl.x0 - 2 + col;
                const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                f32x4 gv;
                // dc = k1 dy + k2 c + k3 inside the image, 0 outside (SAME padding of the data gradient)
#pragma unroll
                for (int k = 0; k < 4; ++k) gv[k] = in ? fmaf(k1[k], rdy[i][k], fmaf(k2[k], rc[i][k], k3[k])) : 0.f;
                h4 hi, lo;
                h3_split(gv, hi, lo);
                const int off = (quad >> 1) * G::PLANE2 + (row * G::IW2 + col) * 16 + (quad & 1) * 8;
                *reinterpret_cast<h4*>(dcs + off) = hi;
                *reinterpret_cast<h4*>(dcs + off + 2 * G::PLANE2) = lo;
            }
        }
    };
    auto load_w = [&](const float* pack, h8 (&w)[13]) {
        int opaque = 0;
        asm volatile("" : "+s"(opaque));                    // keeps hipcc from hoisting the loads out of the tile loop
        const h8* wp = reinterpret_cast<const h8*>(pack + opaque) + lane;
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = wp[i * 64];
        w[12] = w[0];
    };

    int t0 = blockIdx.x;
    Tile cur = tile_of(t0 < a.ntiles ? t0 : 0);
    if (t0 < a.ntiles) {
        fetch1(cur, a.t, rt);
        fetch2(cur);
    }
    __syncthreads();                                        // (the zero fill of the dc image)
    for (; t0 < a.ntiles; t0 += gridDim.x) {
        // ---- stage T and dc of this tile (requested behind the previous tile's phase 2), request A ----
        stash1(rt, true);
        stash2(cur);
#if BWD2_PREFETCH_A
        fetch1(cur, a.a, ra);
#endif
        __syncthreads();

        // ---- phase 1: weight gradient of the second convolution over the tile, its data gradient over the tile + 1 ----
        {
            h8 w[13];
            load_w(a.wpack2, w);
#pragma unroll 1
            for (int rr = 0; rr < 4; ++rr) {
                const int r = wrow0 + rr;
                const int gaddr = ((r + 2) * G::IW2 + 2) * 16 + tr2;
                const h8 bh = tb2_tr_operand(dcs, gaddr);
                const h8 bl = tb2_tr_operand(dcs + 2 * G::PLANE2, gaddr);
#pragma unroll
                for (int k = 0; k < NTAP; ++k) {
                    const int tap = min(tap0 + k, 8);                        // (the second group's fifth slot repeats tap 8: discarded)
                    const int ax = ((r + tap / 3) * G::IW1 + tap % 3) * 16 + tr1;
                    const h8 ah = tb2_tr_operand(xs, ax);
                    const h8 al = tb2_tr_operand(xs + 2 * G::PLANE1, ax);
                    acc2[k] = MFMA_H(ah, bh, acc2[k]);
                    acc2[k] = MFMA_H(al, bh, acc2[k]);
                    acc2[k] = MFMA_H(ah, bl, acc2[k]);
                }
            }
            if (wave < 6) {
                const int pos = (q >> 1) * G::PLANE1 + (o02 * G::IW1 + c2) * 16 + (q & 1) * 8;
                const Bwd2MidEpi epi{inv_s2, G::IMG2 + pos, G::IMG2 + G::IMG1 + pos, cur.y0 - 1 + o02, cur.x0 - 1 + c2, a.H, a.W,
                                     c2 < G::IW1};
                h3r_rows<G::R2, G::IW2 * 16, 2 * G::PLANE2>(dcs, b12 + (q >> 1) * 16, b12 + 32 + (q >> 1) * 2 * G::PLANE2, w, epi, H3NoHook{});
            }
        }
        __syncthreads();                                    // dT complete; T no longer read

        // ---- A over T's image; the next tile's T / dy / c requested ----
#if !BWD2_PREFETCH_A
        fetch1(cur, a.a, ra);
#endif
        stash1(ra, false);
        const bool more = t0 + (int)gridDim.x < a.ntiles;
        const Tile nxt = tile_of(more ? t0 + (int)gridDim.x : t0);
        if (more) {
            fetch1(nxt, a.t, rt);
            fetch2(nxt);
        }
        __syncthreads();

        // ---- phase 2: weight gradient of the first convolution, its data gradient + skip ----
        {
            h8 w[13];
            load_w(a.wpack1, w);
#pragma unroll 1
            for (int rr = 0; rr < 4; ++rr) {
                const int r = wrow0 + rr;
                const int gaddr = ((r + 1) * G::IW1 + 1) * 16 + tr1;
                const h8 bh = tb2_tr_operand(ds, gaddr);
                const h8 bl = tb2_tr_operand(ds + 2 * G::PLANE1, gaddr);
#pragma unroll
                for (int k = 0; k < NTAP; ++k) {
                    const int tap = min(tap0 + k, 8);
                    const int ax = ((r + tap / 3) * G::IW1 + tap % 3) * 16 + tr1;
                    const h8 ah = tb2_tr_operand(xs, ax);
                    const h8 al = tb2_tr_operand(xs + 2 * G::PLANE1, ax);
                    acc1[k] = MFMA_H(ah, bh, acc1[k]);
                    acc1[k] = MFMA_H(al, bh, acc1[k]);
                    acc1[k] = MFMA_H(ah, bl, acc1[k]);
                }
            }
            const int gx = cur.x0 + px_l;
            const Bwd2OutEpi<BNBWD> epi{a, inv_s1, cur.img + ((size_t)(cur.y0 + o0) * a.W + gx) * 16 + q * 4, cur.y0 + o0, gx, bs1, bs2};
            h3r_rows<G::RQ, G::IW1 * 16, 2 * G::PLANE1>(ds, b11 + (q >> 1) * 16, b11 + 32 + (q >> 1) * 2 * G::PLANE1, w, epi, H3NoHook{});
            bs1 = epi.s1; bs2 = epi.s2;
        }
        cur = nxt;
        __syncthreads();                                    // the images are free for the next tile
    }

    // ---- per-workgroup partials: both weight gradients [9][16][16] (D[ci = 4q + j][co = n] per lane), BatchNorm sums [32] ----
    float* red = reinterpret_cast<float*>(tb2_lds);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int k = 0; k < NTAP; ++k) {
            const f32x4 v = bf_acc_ready(pass == 0 ? acc2[k] : acc1[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * NTAP + k) * 256 + (4 * q + j) * 16 + n] = v[j];
        }
        __syncthreads();
        float* dst = pass == 0 ? a.wpartial2 : a.wpartial1;
        for (int i = tid; i < 2304; i += G::NT) {
            // tap i / 256 lives in slot tap - 5 g of the four waves 4 g .. 4 g + 3 of tap group g (fixed order)
            const int tap = i >> 8, g = tap >= NTAP, k = tap - NTAP * g, idx = i & 255;
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) sacc += red[((4 * g + wv) * NTAP + k) * 256 + idx];
            dst[(size_t)blockIdx.x * 2304 + i] = sacc;
        }
        __syncthreads();
    }
    if (BNBWD) {
        f32x4 s1 = bs1, s2v = bs2;
        // over the 16 pixel lanes that share a channel quad, then over the waves (fixed order)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2v[c] += __shfl_xor(s2v[c], m);
            }
        }
        if (n == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[wave * 32 + q * 4 + c] = s1[c];
                red[wave * 32 + 16 + q * 4 + c] = s2v[c];
            }
        }
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
#pragma unroll
            for (int wv = 0; wv < G::NW; ++wv) sacc += red[wv * 32 + tid];
            a.stats[(size_t)blockIdx.x * 32 + tid] = sacc;
        }
    }
}

int bf_bwd2_h3_grid(int B, int H, int W)
{
    using G = Bwd2Geom;
    const int64_t ntiles = (int64_t)B * ((H + G::TH - 1) / G::TH) * ((W + G::TW - 1) / G::TW);
    return (int)(ntiles < 256 ? ntiles : 256);
}

// a.wpartial2 / a.wpartial1: [grid][2304] floats each, a.stats: [grid][32] (a.bnc != nullptr).  a.out must not alias any input.
hipError_t bf_launch_bwd2_h3(const Bwd2H3Args& a0, hipStream_t s)
{
    using G = Bwd2Geom;
    Bwd2H3Args a = a0;
    if (!a.t || !a.a || !a.dy || !a.c || !a.coef || !a.wpack2 || !a.wpack1 || !a.out || !a.wpartial2 || !a.wpartial1) return hipErrorInvalidValue;
    if (a.out == a.t || a.out == a.a || a.out == a.dy || a.out == a.c || (a.bnc && (a.out == a.bnc || !a.stats))) return hipErrorInvalidValue;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    const int grid = bf_bwd2_h3_grid(a.B, a.H, a.W);
    if (a.grid_out) *a.grid_out = grid;
    if (a.bnc) {
        const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(bwd2_h3_kernel<true>), G::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bwd2_h3_kernel<true>, dim3(grid), dim3(G::NT), G::LDS_BYTES, s, a);
    } else {
        const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(bwd2_h3_kernel<false>), G::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bwd2_h3_kernel<false>, dim3(grid), dim3(G::NT), G::LDS_BYTES, s, a);
    }
    return hipGetLastError();
}
