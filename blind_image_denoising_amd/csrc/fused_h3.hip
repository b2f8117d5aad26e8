// Fused residual block on the f16 matrix cores with split-f16 ("f16x3") operands.
//
// Same reference semantics as fused_block_kernel in conv3x3_c16.hip (bfcnn/backbone_blocks.py:174-242 with
// the inference BatchNormalization folded to scale/shift): out = x + scale * conv2(act(conv1 x)) + shift.
//
// Arithmetic.  An fp32 value v is carried as two f16 numbers hi = f16(v), lo = f16(v - hi) (22 mantissa
// bits together).  A 3x3 16->16 convolution is evaluated as
//     conv(x_hi, w_hi) + conv(x_lo, w_hi) + conv(x_hi, w_lo)          (the lo*lo term is < 2^-22 relative)
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation: 14 MFMAs of 16 cycles per 16-pixel group instead of
// 36 MFMAs of 32 cycles on the f32 matrix path (5.1x fewer matrix cycles).  The weights of one kernel are
// pre-scaled by a power of two (max |w| * s in [2^13, 2^14)) so that w_lo stays a normal f16 number; 1/s is
// folded into the epilogues (exact).  tools/exp/emulate_f16x3.py sizes the error of this arithmetic against
// the fp64 oracle: 2.2e-7 normalised MAE through 1x18 (exact-fp32 arithmetic: 1.9e-7; the bar is 1e-4).
// Precondition: |activation| < 65504 (f16 range); the exact-fp32 kernels stay selectable.
//
// K packing.  One MFMA contracts K = 32 = 2 taps x 16 input channels: lanes q = 0,1 (k-slots 0..15) carry
// tap A, lanes q = 2,3 tap B, each lane 8 consecutive channels of one pixel = ONE ds_read_b128.  Tap pairs:
// (0,0)|(1,0), (0,1)|(1,1), (0,2)|(1,2), (2,0)|(2,1); the ninth tap (2,2) uses [w_hi | w_lo] x [x_hi | x_hi]
// and [w_hi | 0] x [x_lo | x_lo]: 4*3 + 2 = 14 MFMAs per group.
//
// Layout ("split-planar", HBM and LDS alike): per image 4 planes [rows][cols][8 x f16] = hi(c0..7),
// hi(c8..15), lo(c0..7), lo(c8..15); 64 B per pixel in total, the same as fp32 NHWC.  A b128 operand read
// is bank-conflict free (16 consecutive pixels x 16 B per lane group, both channel halves one plane apart
// = a multiple of 256 B), the D fragment (4 consecutive output channels of one pixel per lane) goes back
// as one 8-byte hi and one 8-byte lo write, and tiles move HBM -> LDS by DMA without a conversion pass.
//
// Schedule.  One 8-wave workgroup per CU (2 waves per SIMD), persistent, XCD-contiguous tile chunks.  The
// input tile is DOUBLE buffered: tile t+1 is requested at the top of tile t and has the whole tile to land;
// the residual comes from the LDS input tile (its centre), so the loop holds no vector-memory operation
// except the DMA and the 8 result stores per wave, and the one explicit vmcnt(8) per tile is exact.
// Barriers are bare s_barrier with lgkmcnt(0) only (hipcc puts vmcnt(0) in front of every __syncthreads it
// can see, which would drain the DMA).
#include "bf_common.h"
#include "h3_core.h"

template <int TH_, int TW_, int NW_>
struct H3Cfg {
    static constexpr int TH = TH_, TW = TW_, NW = NW_, NT = NW_ * 64;
    static constexpr int MH = TH + 2, MW = TW + 2;             // intermediate region
    static constexpr int IH = TH + 4, IW = TW + 4;             // input region
    static constexpr int GPR = TW / 16;                        // 16-pixel groups per row
    static constexpr int RSTEP = NW / GPR;                     // rows between a wave's consecutive groups
    static constexpr int K1 = (MH + RSTEP - 1) / RSTEP;        // conv1 row-group slots per wave (last one partial)
    static constexpr int K1_FULL = MH / RSTEP;                 // slots every wave has
    static constexpr int K2 = TH / RSTEP;                      // conv2 groups per wave
    static constexpr int SG = (MH + 7) / 8;                    // strip groups (columns TW, TW+1 of the intermediate region)
    static constexpr int STRIP_W0 = (MH % RSTEP) * GPR;        // first wave with K1_FULL row groups only: strips go there
    static constexpr int IN_PLANE = IH * IW * 16;              // bytes per input-tile plane
    static constexpr int MID_PLANE = (MH * MW * 16 + 255) / 256 * 256;
    static constexpr int IN_ELEMS = 4 * IH * IW;               // 16-byte elements per input tile
    static constexpr int PF = (IN_ELEMS + NT - 1) / NT;        // DMA wave-instructions per wave (max)
    static constexpr int TIN_BYTES = PF * NT * 16;             // 4 planes + pad to PF whole workgroup-instructions
    static constexpr int LDS_BYTES = 4 * MID_PLANE + 2 * TIN_BYTES;
    static_assert(TW % 16 == 0 && NW % GPR == 0, "rows must be whole MFMA groups, waves whole rows");
    static constexpr int WG_PER_CU = (160 * 1024) / LDS_BYTES >= 2 && NW <= 4 ? 2 : 1;
    static_assert(TH % RSTEP == 0, "conv2 groups must divide evenly over the waves");
    static_assert(IN_PLANE % 256 == 0, "input planes must keep the two channel halves 256-B congruent");
    static_assert(IN_ELEMS % 64 == 0, "DMA moves whole wave-instructions");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct H3Tile {
    int y0, x0;
    size_t img;      // byte offset of the image in the split-planar tensor
};

template <class Cfg>
__device__ __forceinline__ H3Tile h3_tile(const FusedH3Args& a, int t)
{
    H3Tile r;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    r.y0 = ty * Cfg::TH;
    r.x0 = tx * Cfg::TW;
    r.img = (size_t)b * a.H * a.W * 64;
    return r;
}

template <class Cfg>
__device__ __forceinline__ bool h3_interior(const FusedH3Args& a, const H3Tile& t)
{
    return t.y0 >= 2 && t.y0 + Cfg::TH + 2 <= a.H && t.x0 >= 2 && t.x0 + Cfg::TW + 2 <= a.W;
}

// NG groups x 14 MFMAs.  va / vb / vc: per-group LDS byte address of the lane's 16-byte record of tap (0,0)
// with the pair deltas of the upper lane half already added (va: +1 row, vb: +1 pixel, vc: none).
template <int NG, int PITCH, int LO>
__device__ __forceinline__ void h3_mma(const char* __restrict__ src, const int (&va)[NG], const int (&vb)[NG],
                                       const int (&vc)[NG], const h8 (&w)[10], f32x4 (&acc)[NG])
{
#pragma unroll
    for (int P = 0; P < 4; ++P) {
        h8 bh[NG], bl[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int ad = P < 3 ? va[j] + P * 16 : vb[j] + 2 * PITCH;
            bh[j] = *reinterpret_cast<const h8*>(src + ad);
            bl[j] = *reinterpret_cast<const h8*>(src + ad + LO);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = MFMA_H(w[P], bh[j], acc[j]);          // w_hi x x_hi
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = MFMA_H(w[4 + P], bh[j], acc[j]);      // w_lo x x_hi
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = MFMA_H(w[P], bl[j], acc[j]);          // w_hi x x_lo
    }
    {
        h8 bh[NG], bl[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int ad = vc[j] + 2 * PITCH + 32;
            bh[j] = *reinterpret_cast<const h8*>(src + ad);
            bl[j] = *reinterpret_cast<const h8*>(src + ad + LO);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = MFMA_H(w[8], bh[j], acc[j]);          // [w_hi | w_lo] x [x_hi | x_hi]
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[j] = MFMA_H(w[9], bl[j], acc[j]);          // [w_hi | 0] x [x_lo | x_lo]
    }
}

// per-lane constants (tile-invariant)
struct H3Lane {
    int r1;          // conv1 operand read: lane's record of (row wrow, column wcol + n) in the input tile (channel half q&1)
    int s1;          // same for this wave's strip group
    int w1;          // conv1 result write: (row wrow, column wcol + n) of the intermediate tile, plane q>>1, half q&1
    int ws;          // same for the strip group
    int r2;          // conv2 operand read in the intermediate tile
    int rr;          // residual read: lane's 8-byte record of output (row wrow, column wcol + n) in the INPUT tile
    unsigned g;      // byte offset of the lane's 8-byte record of output (row wrow, column wcol + n) from the tile origin
    int d_row_in, d_row_mid, d_px;   // upper-lane-half deltas: +1 row (input pitch / intermediate pitch), +1 pixel
    int px;          // wcol + n
    int srow, scol;  // strip group: intermediate row / column of this lane's pixel
};

// conv1 (+ activation) of NG groups -> intermediate tile (hi and lo planes)
template <class Cfg, int NG, bool INTERIOR>
__device__ __forceinline__ void h3_conv1(const FusedH3Args& a, const char* __restrict__ tin, char* __restrict__ tmid,
                                         const h8 (&w)[10], const float inv_s, const int (&rd)[NG], const int (&wr)[NG],
                                         const int (&my)[NG], const int (&mx)[NG], const H3Lane& L, const H3Tile& t)
{
    int va[NG], vb[NG];
    f32x4 acc[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        va[j] = rd[j] + L.d_row_in;
        vb[j] = rd[j] + L.d_px;
        acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    h3_mma<NG, Cfg::IW * 16, 2 * Cfg::IN_PLANE>(tin, va, vb, rd, w, acc);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        f32x4 v = acc[j] * inv_s;
        if (a.act1_relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!INTERIOR) {
            // conv2 must see ZERO padding outside the image, not conv1 evaluated there
            const int gy = t.y0 - 1 + my[j], gx = t.x0 - 1 + mx[j];
            if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        h4 hi, lo;
        h3_split(v, hi, lo);
        *reinterpret_cast<h4*>(tmid + wr[j]) = hi;
        *reinterpret_cast<h4*>(tmid + wr[j] + 2 * Cfg::MID_PLANE) = lo;
    }
}

// DMA of one input tile (2-pixel halo) into an LDS buffer.  Element e = tid + i*NT of the tile (plane-major,
// then row, then column) comes from tile origin + pfoff[i]; out-of-image elements come from a zero line.
// EVERY wave issues exactly PF wave-instructions per call, whatever the tile (elements past the tile's end
// and the whole tile when !live are zero-line reads into the buffer's pad), so that the number of operations
// younger than a tile's DMA is a compile-time constant for the vmcnt at the end of the tile.
template <class Cfg, bool INTERIOR>
__device__ __forceinline__ void h3_dma(const FusedH3Args& a, const H3Tile& t, char* __restrict__ tin, const int tid,
                                       const int wave, const unsigned (&pfoff)[Cfg::PF], const bool live)
{
    const char* origin = reinterpret_cast<const char*>(a.in) + t.img + ((ptrdiff_t)(t.y0 - 2) * a.W + (t.x0 - 2)) * 16;
    const char* zeros = reinterpret_cast<const char*>(a.zeros);
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const char* src = origin + pfoff[i];
        bool use = live;
        if ((i + 1) * Cfg::NT > Cfg::IN_ELEMS) use = use && (i * Cfg::NT + wave * 64) < Cfg::IN_ELEMS;     // wave-uniform
        if (!INTERIOR) {
            const int e = tid + i * Cfg::NT;
            const int r = e % (Cfg::IH * Cfg::IW);
            const int row = r / Cfg::IW, col = r - row * Cfg::IW;
            const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + col;
            use = use && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        if (!use) src = zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(tin + (i * Cfg::NT + wave * 64) * 16),
                                         16, 0, 0);
    }
}

// one of the PF wave-instructions of h3_dma (I compile-time): lets the row-streaming kernel spread the DMA issue
// over conv1's rows.  Measured with s_memtime stamps (tools/stamp_h3.py): issuing the 6 instructions back to back
// at the top of the tile costs a wave 1000-2100 cycles -- the 48 KB of the 8 waves queue up in the vector-memory
// issue path (~32 cycles per 1-KiB wave-instruction per CU) and every wave stalls in-order behind its own.
template <class Cfg, bool INTERIOR, int I>
__device__ __forceinline__ void h3_dma_one(const FusedH3Args& a, const H3Tile& t, const char* origin, char* __restrict__ tin,
                                           const int tid, const int wave, const unsigned (&pfoff)[Cfg::PF], const bool live)
{
    const char* src = origin + pfoff[I];
    bool use = live;
    if ((I + 1) * Cfg::NT > Cfg::IN_ELEMS) use = use && (I * Cfg::NT + wave * 64) < Cfg::IN_ELEMS;     // wave-uniform
    if (!INTERIOR) {
        const int e = tid + I * Cfg::NT;
        const int r = e % (Cfg::IH * Cfg::IW);
        const int row = r / Cfg::IW, col = r - row * Cfg::IW;
        const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + col;
        use = use && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    if (!use) src = reinterpret_cast<const char*>(a.zeros);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(tin + (I * Cfg::NT + wave * 64) * 16), 16, 0, 0);
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_h3_kernel(FusedH3Args a)
{
    static_assert(Cfg::K1 == Cfg::K1_FULL + 1 && Cfg::K1_FULL == Cfg::K2, "wave plan assumes K2 full conv1 slots + one partial slot");
    static_assert(Cfg::STRIP_W0 + Cfg::SG <= Cfg::NW, "strip groups must fit on the waves with fewer row groups");
    extern __shared__ __attribute__((aligned(16))) char h3_lds[];
    char* tmid = h3_lds;                                        // [4][MH][MW][8 f16] (+ pad per plane)
    char* tin0 = h3_lds + 4 * Cfg::MID_PLANE;                   // [4][IH][IW][8 f16], two buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wrow = wave / Cfg::GPR, wcol = (wave % Cfg::GPR) * 16;
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;       // bytes per global plane

    H3Lane L0;
    L0.px = wcol + n;
    const int sg = wave - Cfg::STRIP_W0;                        // this wave's strip group (valid if 0 <= sg < SG)
    L0.srow = min(8 * max(sg, 0) + (n >> 1), Cfg::MH - 1);      // partial last group: clamp (duplicate work, same values)
    L0.scol = Cfg::TW + (n & 1);
    L0.r1 = (q & 1) * Cfg::IN_PLANE + (wrow * Cfg::IW + L0.px) * 16;
    L0.s1 = (q & 1) * Cfg::IN_PLANE + (L0.srow * Cfg::IW + L0.scol) * 16;
    L0.w1 = (q >> 1) * Cfg::MID_PLANE + (wrow * Cfg::MW + L0.px) * 16 + (q & 1) * 8;
    L0.ws = (q >> 1) * Cfg::MID_PLANE + (L0.srow * Cfg::MW + L0.scol) * 16 + (q & 1) * 8;
    L0.r2 = (q & 1) * Cfg::MID_PLANE + (wrow * Cfg::MW + L0.px) * 16;
    L0.rr = (q >> 1) * Cfg::IN_PLANE + ((wrow + 2) * Cfg::IW + L0.px + 2) * 16 + (q & 1) * 8;
    L0.g = (unsigned)(q >> 1) * plane_g + (unsigned)(wrow * a.W + L0.px) * 16u + (unsigned)(q & 1) * 8u;
    L0.d_row_in = (q >> 1) * Cfg::IW * 16;
    L0.d_row_mid = (q >> 1) * Cfg::MW * 16;
    L0.d_px = (q >> 1) * 16;

    unsigned pfoff[Cfg::PF];
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const int e = tid + i * Cfg::NT;
        const int pl = e / (Cfg::IH * Cfg::IW), r = e - pl * (Cfg::IH * Cfg::IW);
        const int row = r / Cfg::IW, col = r - row * Cfg::IW;
        pfoff[i] = (unsigned)pl * plane_g + (unsigned)(row * a.W + col) * 16u;
    }

    h8 w1[10], w2[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        w1[i] = reinterpret_cast<const h8*>(a.w1)[i * 64 + lane];
        w2[i] = reinterpret_cast<const h8*>(a.w2)[i * 64 + lane];
    }
    const float inv_s1 = a.aux[0];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.aux + 16 + q * 4);     // folded BN scale / s2
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.aux + 32 + q * 4);

    // XCD-aware persistent schedule (as fused_block_v4_kernel): label = blockIdx % 8 owns a contiguous chunk
    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);
    int t = t_begin + slot;
    if (t >= t_end) return;

    // prologue: the first tile
    {
        const H3Tile c0 = h3_tile<Cfg>(a, t);
        h3_dma<Cfg, false>(a, c0, tin0, tid, wave, pfoff, true);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // also retires the weight / scale loads above
        h3_barrier();
    }

    int buf = 0;
    for (; t < t_end; t += per_label, buf ^= 1) {
        const H3Tile cur = h3_tile<Cfg>(a, t);
        const int t1 = t + per_label;
        const bool has1 = t1 < t_end;
        char* tin = tin0 + buf * Cfg::TIN_BYTES;
        const bool interior = h3_interior<Cfg>(a, cur);
        H3Lane L = L0;
        // opaque inside the loop body: keeps LICM from hoisting every (constant + immediate) address
        asm volatile("" : "+v"(L.r1), "+v"(L.s1), "+v"(L.w1), "+v"(L.ws), "+v"(L.r2), "+v"(L.rr), "+v"(L.g));

        // ---- next tile: global -> the other LDS buffer, in flight for this whole tile --------------------
        // (that buffer was last read by the previous tile's conv2 residual, one barrier ago)
        {
            const H3Tile nx = h3_tile<Cfg>(a, has1 ? t1 : t);  // !has1: PF zero-line reads into the dead buffer
            char* tnx = tin0 + (buf ^ 1) * Cfg::TIN_BYTES;
            if (h3_interior<Cfg>(a, nx)) h3_dma<Cfg, true>(a, nx, tnx, tid, wave, pfoff, has1);
            else h3_dma<Cfg, false>(a, nx, tnx, tid, wave, pfoff, has1);
        }

        // ---- conv1: input tile -> intermediate tile --------------------------------------------------
        // slots 0 .. K1_FULL-1 on every wave; slot K1_FULL on the waves with wrow < MH % RSTEP; strip group sg on
        // waves STRIP_W0 .. STRIP_W0+SG-1.  Passes: {0,1}, then {2 .. K1_FULL-1 (+ the extra slot or the strip)}.
        constexpr int RS_IN = Cfg::RSTEP * Cfg::IW * 16, RS_MID = Cfg::RSTEP * Cfg::MW * 16;
        static_assert(Cfg::K1_FULL == 4, "pass plan below is written for 4 full slots");
#define H3_CONV1(NGv, RD, WR, MY, MX)                                                                              \
        do {                                                                                                       \
            if (interior) h3_conv1<Cfg, NGv, true>(a, tin, tmid, w1, inv_s1, RD, WR, MY, MX, L, cur);               \
            else h3_conv1<Cfg, NGv, false>(a, tin, tmid, w1, inv_s1, RD, WR, MY, MX, L, cur);                       \
        } while (0)
        {
            const int rd[2] = {L.r1, L.r1 + RS_IN}, wr[2] = {L.w1, L.w1 + RS_MID};
            const int my[2] = {wrow, wrow + Cfg::RSTEP}, mx[2] = {L.px, L.px};
            H3_CONV1(2, rd, wr, my, mx);
        }
        if (wrow < Cfg::MH % Cfg::RSTEP) {                       // wave-uniform: 3 row groups
            const int rd[3] = {L.r1 + 2 * RS_IN, L.r1 + 3 * RS_IN, L.r1 + 4 * RS_IN};
            const int wr[3] = {L.w1 + 2 * RS_MID, L.w1 + 3 * RS_MID, L.w1 + 4 * RS_MID};
            const int my[3] = {wrow + 2 * Cfg::RSTEP, wrow + 3 * Cfg::RSTEP, wrow + 4 * Cfg::RSTEP}, mx[3] = {L.px, L.px, L.px};
            H3_CONV1(3, rd, wr, my, mx);
        } else if (sg < Cfg::SG) {                               // 2 row groups + a strip group
            const int rd[3] = {L.r1 + 2 * RS_IN, L.r1 + 3 * RS_IN, L.s1};
            const int wr[3] = {L.w1 + 2 * RS_MID, L.w1 + 3 * RS_MID, L.ws};
            const int my[3] = {wrow + 2 * Cfg::RSTEP, wrow + 3 * Cfg::RSTEP, L.srow}, mx[3] = {L.px, L.px, L.scol};
            H3_CONV1(3, rd, wr, my, mx);
        } else {
            const int rd[2] = {L.r1 + 2 * RS_IN, L.r1 + 3 * RS_IN}, wr[2] = {L.w1 + 2 * RS_MID, L.w1 + 3 * RS_MID};
            const int my[2] = {wrow + 2 * Cfg::RSTEP, wrow + 3 * Cfg::RSTEP}, mx[2] = {L.px, L.px};
            H3_CONV1(2, rd, wr, my, mx);
        }
#undef H3_CONV1
        h3_barrier();                                            // tmid complete

        // ---- conv2 + folded BN + residual (from the LDS input tile) -> global ---------------------------
        static_assert(Cfg::K2 == 4, "conv2 pass below is written for 4 groups per wave");
        {
            char* out_tile = reinterpret_cast<char*>(a.out) + cur.img + ((size_t)cur.y0 * a.W + cur.x0) * 16;
            const size_t rowstep = (size_t)a.W * 16 * Cfg::RSTEP;    // bytes between a wave's consecutive output rows
            const size_t lo_g = 2 * (size_t)plane_g;
            int va[4], vb[4], vc[4];
            f32x4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vc[j] = L.r2 + j * RS_MID;
                va[j] = vc[j] + L.d_row_mid;
                vb[j] = vc[j] + L.d_px;
                acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            h3_mma<4, Cfg::MW * 16, 2 * Cfg::MID_PLANE>(tmid, va, vb, vc, w2, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h4 rh = *reinterpret_cast<const h4*>(tin + L.rr + j * RS_IN);
                const h4 rl = *reinterpret_cast<const h4*>(tin + L.rr + j * RS_IN + 2 * Cfg::IN_PLANE);
                const f32x4 res = __builtin_convertvector(rh, f32x4) + __builtin_convertvector(rl, f32x4);
                const f32x4 v = acc[j] * sc + sh + res;
                h4 hi, lo;
                h3_split(v, hi, lo);
                // out-of-image lanes store to a dump line so that every wave issues the same number of
                // vector-memory operations per tile (the vmcnt below is exact)
                char* p = out_tile + j * rowstep + L.g;
                char* pl = p + lo_g;
                if (!interior && !(cur.y0 + wrow + j * Cfg::RSTEP < a.H && cur.x0 + L.px < a.W)) {
                    p = reinterpret_cast<char*>(a.dump) + lane * 8;
                    pl = p;
                }
                *reinterpret_cast<h4*>(p) = hi;
                *reinterpret_cast<h4*>(pl) = lo;
            }
        }
        // The next tile's DMA (issued at the top) must have landed before the barrier publishes its buffer.  Younger
        // than it on this wave: exactly the 8 stores above.  No other vector-memory operation exists in the loop --
        // in particular no ordinary global LOAD: hipcc waits vmcnt(0) for any load that is pending together with
        // LDS-DMA (measured: tools/exp/dma_waitcnt.hip), which would drain the DMA in the conv2 epilogue.
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(8));
        h3_barrier();                                            // tmid free; next tile's input visible to all waves
    }
}

// ==========================================================================================================
// Row-streaming variant (fused_block_h3r_kernel): same arithmetic, layout, DMA pipeline and barriers as above, but
// a wave owns a RUN of R consecutive rows of one 16-column strip and walks down the rows.  Taps are paired
// HORIZONTALLY -- (dy,0)|(dy,1) in one K=32 MFMA, tap (dy,2) as [w_hi | w_lo] x [x_hi | x_hi] and [w_hi | 0] x
// [x_lo | x_lo] -- so a B fragment depends on the INPUT ROW only: the four fragments of input row i (pair hi, pair
// lo, single hi, single lo = 4 ds_read_b128) serve the three output rows i, i-1, i-2.  Per output row that is
// 4*(R+2)/R reads instead of 10 (6 for R = 4) against 15 MFMAs instead of 14; the next row's fragments are in
// flight while the current row's 15 MFMAs run (16 VGPRs of prefetch), and only three accumulators are live.
// rocprof of the group-per-pass kernel above (profiles/r01_h3_groups_pmc.txt): matrix pipe 36 % busy, LDS 34 %,
// waves parked in s_waitcnt 40 % of their cycles -- latency, not throughput; this variant attacks exactly that.
// ==========================================================================================================
#include "h3_rows.h"

template <class Cfg>
struct H3RPlan {
    static constexpr int NRUN = Cfg::RSTEP;                            // row runs per strip = waves per strip
    static constexpr int R1_SMALL = Cfg::MH / NRUN, R1_BIG = R1_SMALL + 1, N_BIG = Cfg::MH % NRUN;   // conv1: 18 = 5+5+4+4
    static constexpr int R2 = Cfg::TH / NRUN;                          // conv2: 16 = 4x4
    static constexpr int N_SHORT = (NRUN - N_BIG) * Cfg::GPR;          // waves with the short run: they take the strip groups,
    static constexpr int SG_PER_WAVE = (Cfg::SG + N_SHORT - 1) / N_SHORT;  // strip group sg + k * N_SHORT on short wave sg
    static constexpr int NROWS_MIN = R1_SMALL + 2;                      // conv1 input rows of the shortest run (DMA issue slots)
    static_assert(Cfg::TH % NRUN == 0, "conv2 rows must divide over the runs");
    static_assert(N_BIG > 0 && N_BIG < NRUN, "plan assumes both run lengths occur");
};

template <class Cfg, bool INTERIOR>
__device__ __forceinline__ void h3r_conv1_store(const FusedH3Args& a, char* __restrict__ tmid, const int wr, f32x4 v,
                                                const float inv_s, const float relu_floor, const int gy, const int gx)
{
    if (!(H3_ABLATE & 64)) {
        // activation without a branch and without the canonicalising v_max x,x hipcc puts in front of fmaxf:
        // max(v, floor) with floor = 0 (relu) or -inf (linear) as median(v, floor, +inf)
        v = v * inv_s;
        v.x = __builtin_amdgcn_fmed3f(v.x, relu_floor, __builtin_inff()); v.y = __builtin_amdgcn_fmed3f(v.y, relu_floor, __builtin_inff());
        v.z = __builtin_amdgcn_fmed3f(v.z, relu_floor, __builtin_inff()); v.w = __builtin_amdgcn_fmed3f(v.w, relu_floor, __builtin_inff());
    }
    if (!INTERIOR) {
        // conv2 must see ZERO padding outside the image, not conv1 evaluated there
        if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    *reinterpret_cast<h8*>(tmid + wr) = h3_split_record(v);      // wr: plane (q>>1) + 2*(q&1), pixel * 16
}

struct H3RLane {
    int p1, s1;      // conv1 pair / single fragment address of the run's first input row (input tile)
    int w1;          // conv1 result write of the run's first row (intermediate tile)
    int p2, s2;      // conv2 pair / single fragment address (intermediate tile)
    int rr;          // residual read of the run's first output row (input tile centre)
    unsigned g;      // global byte offset of the lane's 8-byte record of the run's first output row from the tile origin
    int px;
};

// issues the next tile's DMA wave-instructions I, I + NROWS, ... after conv1's input row I
template <class Cfg, bool NX_INTERIOR, int NROWS>
struct H3DmaHook {
    const FusedH3Args& a;
    const H3Tile& nx;
    const char* origin;
    char* tnx;
    int tid, wave;
    const unsigned (&pfoff)[Cfg::PF];
    bool live;
    template <int I> __device__ __forceinline__ void row() const
    {
        if constexpr (I < NROWS && I < Cfg::PF) {
            if (!(H3_ABLATE & 1)) h3_dma_one<Cfg, NX_INTERIOR, I>(a, nx, origin, tnx, tid, wave, pfoff, live);
            if constexpr (I + NROWS < Cfg::PF) {
                static_assert(I + 2 * NROWS >= Cfg::PF, "at most two DMA instructions per row");
                if (!(H3_ABLATE & 1)) h3_dma_one<Cfg, NX_INTERIOR, I + NROWS>(a, nx, origin, tnx, tid, wave, pfoff, live);
            }
        }
    }
};

template <class Cfg, int R, bool INTERIOR, class Hook>
__device__ __forceinline__ void h3r_conv1_run(const FusedH3Args& a, const char* __restrict__ tin, char* __restrict__ tmid,
                                              const h8 (&w)[13], const float inv_s, const float relu_floor, const H3RLane& L,
                                              const H3Tile& t, const int o0, const Hook& hook)
{
    // (plain values, no references to the lane-constant struct: a reference member kept the whole struct in scratch memory)
    struct Epi {
        struct Pre {};
        enum { EXTRA_MFMA = 0 };
        const FusedH3Args& a; char* __restrict__ tmid; float inv_s, relu_floor; int w1, gy0, gx;
        __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
        __device__ __forceinline__ f32x4 finish(const int, const f32x4 acc, const Pre&) const { return acc; }
        __device__ __forceinline__ void operator()(const int o, const f32x4 v) const
        {
            h3r_conv1_store<Cfg, INTERIOR>(a, tmid, w1 + o * Cfg::MW * 16, v, inv_s, relu_floor, gy0 + o, gx);
        }
    } const epi{a, tmid, inv_s, relu_floor, L.w1, t.y0 - 1 + o0, t.x0 - 1 + L.px};
    h3r_rows<R, Cfg::IW * 16, 2 * Cfg::IN_PLANE>(tin, L.p1, L.s1, w, epi, hook);
}

// HEAD = 1: the last block of a network with a linear denoiser head: y . wh (16 x 3, premultiplied), tanh, denormalise,
// [round, uint8] happen in the conv2 epilogue; the block output is never written and the head kernel (one more pass over
// the 64 B / pixel activation) disappears.
template <class Cfg, int HEAD = 0>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_h3r_kernel(FusedH3Args a)
{
    using Plan = H3RPlan<Cfg>;
    extern __shared__ __attribute__((aligned(16))) char h3_lds[];
    char* tmid = h3_lds;                                        // [4][MH][MW][8 f16] (+ pad per plane)
    char* tin0 = h3_lds + 4 * Cfg::MID_PLANE;                   // [4][IH][IW][8 f16] (+ pad), two buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int run = wave / Cfg::GPR, wcol = (wave % Cfg::GPR) * 16;
#if H3_ABLATE & 256
    if (wave >= Cfg::NW / 2) __builtin_amdgcn_s_setprio(1);    // experiment: favour the younger wave of each SIMD
#endif
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;       // bytes per global plane
    const bool big = run < Plan::N_BIG;
    const int o1 = big ? run * Plan::R1_BIG : Plan::N_BIG * Plan::R1_BIG + (run - Plan::N_BIG) * Plan::R1_SMALL;   // conv1 first row
    const int o2 = run * Plan::R2;                                                                               // conv2 first row
    const int sg = wave - Plan::N_BIG * Cfg::GPR;               // first strip group of this wave (short-run waves: sg >= 0)

    H3RLane L0;
    L0.px = wcol + n;
    {
        const int b1 = (q & 1) * Cfg::IN_PLANE + (o1 * Cfg::IW + L0.px) * 16;
        L0.p1 = b1 + (q >> 1) * 16;
        L0.s1 = b1 + 32 + (q >> 1) * 2 * Cfg::IN_PLANE;
        L0.w1 = ((q >> 1) + 2 * (q & 1)) * Cfg::MID_PLANE + (o1 * Cfg::MW + L0.px) * 16;
        const int b2 = (q & 1) * Cfg::MID_PLANE + (o2 * Cfg::MW + L0.px) * 16;
        L0.p2 = b2 + (q >> 1) * 16;
        L0.s2 = b2 + 32 + (q >> 1) * 2 * Cfg::MID_PLANE;
        // residual operand [x_hi | x_lo] of the centre pixel: lanes q < 2 read the hi planes, q >= 2 the lo planes
        L0.rr = ((q & 1) + 2 * (q >> 1)) * Cfg::IN_PLANE + ((o2 + 2) * Cfg::IW + L0.px + 2) * 16;
        L0.g = (unsigned)((q >> 1) + 2 * (q & 1)) * plane_g + (unsigned)(o2 * a.W + L0.px) * 16u;
    }

    unsigned pfoff[Cfg::PF];
#pragma unroll
    for (int i = 0; i < Cfg::PF; ++i) {
        const int e = tid + i * Cfg::NT;
        const int pl = e / (Cfg::IH * Cfg::IW), r = e - pl * (Cfg::IH * Cfg::IW);
        const int row = r / Cfg::IW, col = r - row * Cfg::IW;
        pfoff[i] = (unsigned)pl * plane_g + (unsigned)(row * a.W + col) * 16u;
    }

    h8 w1[13], w2[13];                                          // [12]: s2 * identity (conv2 only: adds the residual)
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        w1[i] = reinterpret_cast<const h8*>(a.w1)[i * 64 + lane];
        w2[i] = reinterpret_cast<const h8*>(a.w2)[i * 64 + lane];
    }
    const float inv_s1 = a.aux[0];
    const float inv_s2 = a.aux[48];                             // BN scale is folded into the row-layout w2
    const float relu_floor = a.act1_relu ? 0.f : -__builtin_inff();
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.aux + 32 + q * 4);
    f32x4 whr[4];                                               // HEAD: rows 4q .. 4q+3 of the 16 x 4 head matrix
#pragma unroll
    for (int j = 0; j < 4; ++j) whr[j] = HEAD ? *reinterpret_cast<const f32x4*>(a.head_wh + (4 * q + j) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);
    int t = t_begin + slot;
    if (t >= t_end) return;

    {
        const H3Tile c0 = h3_tile<Cfg>(a, t);
        h3_dma<Cfg, false>(a, c0, tin0, tid, wave, pfoff, true);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // also retires the weight / scale loads above
        h3_barrier();
    }

#if H3_ABLATE & 32
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    int buf = 0;
    H3Tile carry = h3_tile<Cfg>(a, t);
    // tile coordinates advance incrementally: per_label = (sb * tiles_y + sy) * tiles_x + sx, two carries per step
    // (the three integer divisions of h3_tile cost ~350 scalar-dependent cycles per tile on every wave)
    int c_tx = carry.x0 / Cfg::TW, c_ty = carry.y0 / Cfg::TH;
    const int sx = per_label % a.tiles_x, sy = (per_label / a.tiles_x) % a.tiles_y;
    const size_t img_bytes = (size_t)a.H * a.W * 64;
    const size_t sb_bytes = (size_t)(per_label / a.tiles_x / a.tiles_y) * img_bytes;
    for (; t < t_end; t += per_label, buf ^= 1) {
        const H3Tile cur = carry;                                // (a fresh const per iteration: the epilogue structs hold references)
        const int t1 = t + per_label;
        const bool has1 = t1 < t_end;
        char* tin = tin0 + buf * Cfg::TIN_BYTES;
        const bool interior = h3_interior<Cfg>(a, cur);
        H3RLane L = L0;
        asm volatile("" : "+v"(L.p1), "+v"(L.w1), "+v"(L.p2), "+v"(L.rr), "+v"(L.g));
        H3_STAMP(6);                                             // tile index math

        // next tile: global -> the other LDS buffer (last read by the previous tile's conv2 residual, one barrier
        // ago); its PF DMA instructions are issued one per input row inside conv1 and land during conv2
        H3Tile nx = cur;
        if (has1) {
            c_tx += sx;
            const int cx = c_tx >= a.tiles_x;
            c_tx -= cx ? a.tiles_x : 0;
            c_ty += sy + cx;
            const int cy = c_ty >= a.tiles_y;
            c_ty -= cy ? a.tiles_y : 0;
            nx.x0 = c_tx * Cfg::TW;
            nx.y0 = c_ty * Cfg::TH;
            nx.img = cur.img + sb_bytes + (cy ? img_bytes : 0);
        }
        const char* nx_origin = reinterpret_cast<const char*>(a.in) + nx.img + ((ptrdiff_t)(nx.y0 - 2) * a.W + (nx.x0 - 2)) * 16;
        char* tnx = tin0 + (buf ^ 1) * Cfg::TIN_BYTES;
        const bool nx_interior = h3_interior<Cfg>(a, nx);
        H3_STAMP(0);                                             // next-tile index math
        // ---- conv1: input tile -> intermediate tile ------------------------------------------------------
#define H3R_CONV1(RV)                                                                                                       \
        do {                                                                                                               \
            if (nx_interior) {                                                                                             \
                const H3DmaHook<Cfg, true, Plan::NROWS_MIN> hook{a, nx, nx_origin, tnx, tid, wave, pfoff, has1};           \
                if (interior) h3r_conv1_run<Cfg, RV, true>(a, tin, tmid, w1, inv_s1, relu_floor, L, cur, o1, hook);                    \
                else h3r_conv1_run<Cfg, RV, false>(a, tin, tmid, w1, inv_s1, relu_floor, L, cur, o1, hook);                            \
            } else {                                                                                                       \
                const H3DmaHook<Cfg, false, Plan::NROWS_MIN> hook{a, nx, nx_origin, tnx, tid, wave, pfoff, has1};          \
                if (interior) h3r_conv1_run<Cfg, RV, true>(a, tin, tmid, w1, inv_s1, relu_floor, L, cur, o1, hook);                    \
                else h3r_conv1_run<Cfg, RV, false>(a, tin, tmid, w1, inv_s1, relu_floor, L, cur, o1, hook);                            \
            }                                                                                                              \
        } while (0)
        if (big) {
            H3R_CONV1(Plan::R1_BIG);
        } else {
            // strip group FIRST: at the end of conv1 its read -> MFMA -> write chain ran alone (the partner wave on the
            // SIMD was already parked at the barrier) and cost ~1100 cycles, all of it barrier time for the other waves
            // (8 rows x 2 columns per group; its lane addresses are rebuilt here, ~10 VALU, instead of living in VGPRs)
#pragma unroll
            for (int k = 0; k < Plan::SG_PER_WAVE; ++k) {
                const int g = sg + k * Plan::N_SHORT;
                if (g < Cfg::SG) {
                    const int srow = min(8 * g + (n >> 1), Cfg::MH - 1);      // partial last group: clamp (same values twice)
                    const int scol = Cfg::TW + (n & 1);
                    const int bg = (q & 1) * Cfg::IN_PLANE + (srow * Cfg::IW + scol) * 16;
                    const int gw = ((q >> 1) + 2 * (q & 1)) * Cfg::MID_PLANE + (srow * Cfg::MW + scol) * 16;
                    const f32x4 v = bf_acc_ready(h3r_group<Cfg::IW * 16, 2 * Cfg::IN_PLANE>(tin, bg + (q >> 1) * 16, bg + 32 + (q >> 1) * 2 * Cfg::IN_PLANE, w1));
                    if (interior) h3r_conv1_store<Cfg, true>(a, tmid, gw, v, inv_s1, relu_floor, 0, 0);
                    else h3r_conv1_store<Cfg, false>(a, tmid, gw, v, inv_s1, relu_floor, cur.y0 - 1 + srow, cur.x0 - 1 + scol);
                }
            }
            H3R_CONV1(Plan::R1_SMALL);
        }
#undef H3R_CONV1
        H3_STAMP(1);                                             // conv1 (+ DMA issue)
        h3_barrier();                                            // tmid complete
        H3_STAMP(2);                                             // barrier A

        // ---- conv2 + folded BN + residual (from the LDS input tile) -> global ---------------------------
        {
            char* out_row0 = reinterpret_cast<char*>(a.out) + cur.img + ((size_t)cur.y0 * a.W + cur.x0) * 16;
            const size_t rowbytes = (size_t)a.W * 16;
            const size_t lo_g = 2 * (size_t)plane_g;
            struct Epi2 {
                typedef h8 Pre;
                enum { EXTRA_MFMA = 1 };
                const FusedH3Args& a; const char* __restrict__ tin; h8 wres; char* out_row0; size_t rowbytes, lo_g;
                float inv_s2; f32x4 sh; bool interior; int rr, y_base, x_px, lane; unsigned g;
                f32x4 wh0, wh1, wh2, wh3; char* head_row0; int q;
                // residual operand [x_hi | x_lo] of the centre pixel of output row o: requested one row step early
                __device__ __forceinline__ Pre pre(const int o) const { return *reinterpret_cast<const h8*>(tin + rr + o * Cfg::IW * 16); }
                // residual on the matrix pipe: acc += (s2 * I) x [x_hi | x_lo] -- exact (power-of-two times f16 in
                // fp32) and one MFMA + one ds_read_b128 instead of two ds_read_b64 and 12 conversions / additions
                __device__ __forceinline__ f32x4 finish(const int, const f32x4 acc, const Pre& xr) const { return MFMA_H(wres, xr, acc); }
                __device__ __forceinline__ void operator()(const int o, const f32x4 acc) const
                {
                    const f32x4 v = (H3_ABLATE & 64) ? acc : acc * inv_s2 + sh;
                    if constexpr (HEAD) {
                        // the lane's 4 channels times their rows of the head matrix, summed over the 4 lanes of the pixel
                        f32x4 hsum = wh0 * v.x + wh1 * v.y + wh2 * v.z + wh3 * v.w;
                        const bool finite = fabsf(v.x) <= 3.0e38f && fabsf(v.y) <= 3.0e38f && fabsf(v.z) <= 3.0e38f && fabsf(v.w) <= 3.0e38f;
                        if (!finite && a.status) atomicOr(a.status, BF_STATUS_F16_RANGE);      // never on valid activations
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            hsum[k] += __shfl_xor(hsum[k], 16, 64);
                            hsum[k] += __shfl_xor(hsum[k], 32, 64);
                        }
                        // after the reduction the 4 lanes of a pixel hold the same three sums: lane group q writes output
                        // channel q (q = 3: dump line), ONE store per row and wave, every lane with its own tanh
                        const float hk = q == 0 ? hsum[0] : (q == 1 ? hsum[1] : hsum[2]);
                        const bool live = q < 3 && y_base + o < a.Ho && x_px < a.Wo;
                        // tanh(2h) = 1 - 2 / (exp(4h) + 1): a handful of instructions inside the matrix loop (libm tanhf: ~30)
                        float r = (1.0f - 2.0f / (__expf(4.0f * hk) + 1.0f)) * 0.51f;
                        if (a.denormalize) r = (fminf(fmaxf(r, -0.5f), 0.5f) + 0.5f) * (a.v_max - a.v_min) + a.v_min;
                        if (a.head_u8) {
                            unsigned char* p = live ? reinterpret_cast<unsigned char*>(head_row0) + (size_t)o * a.Wo * 3 + q
                                                    : reinterpret_cast<unsigned char*>(a.dump) + lane * 16;
                            *p = (unsigned char)fminf(fmaxf(rintf(r), 0.f), 255.f);              // rintf = round-half-even
                        } else {
                            float* p = live ? reinterpret_cast<float*>(head_row0) + (size_t)o * a.Wo * 3 + q
                                            : reinterpret_cast<float*>(a.dump) + lane * 4;
                            *p = r;
                        }
                        return;
                    }
                    const h8 rec = h3_split_record(v);
                    // out-of-image lanes store to a dump line: every wave issues exactly R2 stores per tile
                    char* p = out_row0 + o * rowbytes + g;
                    if (!(y_base + o < a.H && x_px < a.W))   /* branch-free on purpose: see h3_split_record's neighbour comment */ p = reinterpret_cast<char*>(a.dump) + lane * 16;
                    if (H3_ABLATE & 2) { if (v.x == 12345.678f) *reinterpret_cast<h8*>(p) = rec; }   // keeps the work live
                    else *reinterpret_cast<h8*>(p) = rec;
                }
            };
            char* head_row0 = nullptr;
            if constexpr (HEAD) {
                const size_t bimg = cur.img / img_bytes;             // image index (once per tile)
                head_row0 = reinterpret_cast<char*>(a.head_out) +
                            ((bimg * a.Ho + (size_t)(cur.y0 + o2)) * a.Wo + (size_t)(cur.x0 + L.px)) * 3 * (a.head_u8 ? 1 : 4);
            }
            const Epi2 epi2{a, tin, w2[12], out_row0, rowbytes, lo_g, inv_s2, sh, interior, L.rr, cur.y0 + o2, cur.x0 + L.px, lane, L.g,
                            whr[0], whr[1], whr[2], whr[3], head_row0, q};
            h3r_rows<Plan::R2, Cfg::MW * 16, 2 * Cfg::MID_PLANE>(tmid, L.p2, L.s2, w2, epi2, H3NoHook{});
        }
        // next tile's DMA landed <=> at most the R2 stores above are outstanding (see fused_block_h3_kernel)
        H3_STAMP(3);                                             // conv2 + stores
        __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3_ABLATE & 2) ? 0 : Plan::R2));      // HEAD: also one store per row
        H3_STAMP(4);                                             // wait for the next tile's DMA
        h3_barrier();
        H3_STAMP(5);                                             // barrier B
        carry = nx;                                              // tile coordinates are computed once per tile
    }
#if H3_ABLATE & 32
    if (a.dbg && lane == 0) {
        for (int k = 0; k < 8; ++k) a.dbg[((size_t)blockIdx.x * Cfg::NW + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

// ==========================================================================================================
// Wave-specialised variant (fused_block_h3s_kernel): same arithmetic, layout and row streaming, but the workgroup's
// waves split into conv1 waves (0..3) and conv2 waves (4..7) -- wave w and w+4 share a SIMD -- and the two
// convolutions of CONSECUTIVE tiles run concurrently: in iteration k the conv1 waves turn tile k's input into its
// intermediate tile while the conv2 waves finish tile k-1.  Both the input tile and the intermediate tile are double
// buffered (14x32 tiles: 2 x 45 KB + 2 x 34 KB of LDS).  Versus the kernel above:
//   * ONE barrier per tile instead of two, and the fill / drain steps of the two row pipelines are out of phase on
//     every SIMD (stamps of the kernel above: 13 % of a wave's cycles at barriers; run fill/drain = R+2 steps per R rows);
//   * runs of 8 and 7 rows instead of 5 and 4 (fewer fill/drain steps per row);
//   * a wave holds ONE kernel's weights (52 VGPRs instead of 104), which pays for the 7 residual fragments a conv2
//     wave fetches from the input tile one iteration early (so that the input buffer is free for the DMA of tile k+1
//     right after the barrier).
// The conv1 waves own the DMA (next tile, spread over their row steps, vmcnt(0) before the barrier: nothing else is
// in their vector-memory queue); the conv2 waves own the stores and never wait for them.
// ==========================================================================================================
template <int TH_, int TW_>
struct H3SCfg {
    static constexpr int TH = TH_, TW = TW_, NW = 8, NT = 512, NA = 4, NB = 4;
    static constexpr int MH = TH + 2, MW = TW + 2, IH = TH + 4, IW = TW + 4;
    static constexpr int GPR = TW / 16;                                 // strips
    static constexpr int IN_PLANE = (IH * IW * 16 + 255) / 256 * 256;   // bytes per input-tile plane (padded)
    static constexpr int IN_PLANE_ELEMS = IN_PLANE / 16;
    static constexpr int MID_PLANE = (MH * MW * 16 + 255) / 256 * 256;
    static constexpr int IN_ELEMS = 4 * IN_PLANE_ELEMS;                 // 16-byte elements per input tile incl. plane pads
    static constexpr int DMA_LANES = NA * 64;                           // the conv1 waves move the tile
    static constexpr int PF = (IN_ELEMS + DMA_LANES - 1) / DMA_LANES;
    static constexpr int TIN_BYTES = PF * DMA_LANES * 16;
    static constexpr int TMID_BYTES = 4 * MID_PLANE;
    static constexpr int LDS_BYTES = 2 * TIN_BYTES + 2 * TMID_BYTES;
    static constexpr int R1 = MH / (NA / GPR), R2 = TH / (NB / GPR);    // rows per conv1 / conv2 wave
    static constexpr int SG = (MH + 7) / 8;                             // strip groups
    static constexpr int WG_PER_CU = 1;
    static_assert(GPR == 2 && NA % GPR == 0 && MH % (NA / GPR) == 0 && TH % (NB / GPR) == 0, "wave plan");
    static_assert(SG <= NA, "one strip group per conv1 wave at most");
    static_assert(IN_ELEMS % 64 == 0, "DMA moves whole wave-instructions");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <class Cfg>
__device__ __forceinline__ bool h3s_interior(const FusedH3Args& a, const H3Tile& t)
{
    return t.y0 >= 2 && t.y0 + Cfg::TH + 2 <= a.H && t.x0 >= 2 && t.x0 + Cfg::TW + 2 <= a.W;
}

// DMA wave-instruction I (of PF) of one conv1 wave: element e = (wave*64 + lane) + I*DMA_LANES of the padded tile
template <class Cfg, bool INTERIOR, int I>
__device__ __forceinline__ void h3s_dma_one(const FusedH3Args& a, const H3Tile& t, const char* origin, char* __restrict__ tin,
                                            const int dlane, const int wave, const unsigned (&pfoff)[Cfg::PF], const bool live)
{
    const char* src = origin + pfoff[I];
    const int e = dlane + I * Cfg::DMA_LANES;
    const int r = e % Cfg::IN_PLANE_ELEMS;
    bool use = live && r < Cfg::IH * Cfg::IW;                       // plane pad -> zero line
    if ((I + 1) * Cfg::DMA_LANES > Cfg::IN_ELEMS) use = use && e < Cfg::IN_ELEMS;
    if (!INTERIOR) {
        const int row = r / Cfg::IW, col = r - row * Cfg::IW;
        const int gy = t.y0 - 2 + row, gx = t.x0 - 2 + col;
        use = use && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    }
    if (!use) src = reinterpret_cast<const char*>(a.zeros);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(tin + (I * Cfg::DMA_LANES + wave * 64) * 16), 16, 0, 0);
}

// DMA instructions I, I + NROWS, ... after conv1's input row I
template <class Cfg, bool NX_INTERIOR, int NROWS>
struct H3SDmaHook {
    const FusedH3Args& a;
    const H3Tile& nx;
    const char* origin;
    char* tnx;
    int dlane, wave;
    const unsigned (&pfoff)[Cfg::PF];
    bool live;
    template <int I> __device__ __forceinline__ void row() const
    {
        if constexpr (I < NROWS && I < Cfg::PF) {
            h3s_dma_one<Cfg, NX_INTERIOR, I>(a, nx, origin, tnx, dlane, wave, pfoff, live);
            if constexpr (I + NROWS < Cfg::PF) {
                static_assert(I + 2 * NROWS >= Cfg::PF, "at most two DMA instructions per row");
                h3s_dma_one<Cfg, NX_INTERIOR, I + NROWS>(a, nx, origin, tnx, dlane, wave, pfoff, live);
            }
        }
    }
};

// next tile of this workgroup's sequence (incremental: no divisions in the loop)
struct H3SWalk {
    int tx, ty, sx, sy, tiles_x, tiles_y;
    size_t img, sb_bytes, img_bytes;
    template <class Cfg> __device__ __forceinline__ H3Tile tile() const
    {
        H3Tile r;
        r.x0 = tx * Cfg::TW; r.y0 = ty * Cfg::TH; r.img = img;
        return r;
    }
    __device__ __forceinline__ void advance()
    {
        tx += sx;
        const int cx = tx >= tiles_x;
        tx -= cx ? tiles_x : 0;
        ty += sy + cx;
        const int cy = ty >= tiles_y;
        ty -= cy ? tiles_y : 0;
        img += sb_bytes + (cy ? img_bytes : 0);
    }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT, 2) void fused_block_h3s_kernel(FusedH3Args a)
{
    extern __shared__ __attribute__((aligned(16))) char h3_lds[];
    char* tmid0 = h3_lds;                                       // 2 x [4][MH][MW][8 f16]
    char* tin0 = h3_lds + 2 * Cfg::TMID_BYTES;                  // 2 x [4][IH][IW][8 f16] (+ pad)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const bool role_a = wave < Cfg::NA;                         // conv1 waves
    const int rw = role_a ? wave : wave - Cfg::NA;              // index within the role
    const int wcol = (rw % Cfg::GPR) * 16, half = rw / Cfg::GPR;
    const int px = wcol + n;
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;       // bytes per global plane

    const int nxcd = gridDim.x >= 8 ? 8 : 1;
    const int label = blockIdx.x % nxcd, slot = blockIdx.x / nxcd;
    const int per_label = gridDim.x / nxcd;
    const int chunk = (a.ntiles + nxcd - 1) / nxcd;
    const int t_begin = label * chunk;
    const int t_end = min(a.ntiles, t_begin + chunk);
    const int t0 = t_begin + slot;
    if (t0 >= t_end) return;
    const int ntl = (t_end - t0 + per_label - 1) / per_label;   // tiles of this workgroup

    H3SWalk walk;
    {
        const H3Tile c0 = h3_tile<Cfg>(a, t0);
        walk.tx = c0.x0 / Cfg::TW; walk.ty = c0.y0 / Cfg::TH; walk.img = c0.img;
        walk.tiles_x = a.tiles_x; walk.tiles_y = a.tiles_y;
        walk.sx = per_label % a.tiles_x; walk.sy = (per_label / a.tiles_x) % a.tiles_y;
        walk.img_bytes = (size_t)a.H * a.W * 64;
        walk.sb_bytes = (size_t)(per_label / a.tiles_x / a.tiles_y) * walk.img_bytes;
    }

    if (role_a) {
        // ------------------------------------------------------------------ conv1 waves --------------------------
        const int o1 = half * Cfg::R1;
        const int dlane = wave * 64 + lane;
        unsigned pfoff[Cfg::PF];
#pragma unroll
        for (int i = 0; i < Cfg::PF; ++i) {
            const int e = dlane + i * Cfg::DMA_LANES;
            const int pl = e / Cfg::IN_PLANE_ELEMS, r = e - pl * Cfg::IN_PLANE_ELEMS;
            const int row = r / Cfg::IW, col = r - row * Cfg::IW;
            pfoff[i] = (unsigned)pl * plane_g + (unsigned)(row * a.W + col) * 16u;
        }
        h8 w1[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) w1[i] = reinterpret_cast<const h8*>(a.w1)[i * 64 + lane];
        const float inv_s1 = a.aux[0];
        const float relu_floor = a.act1_relu ? 0.f : -__builtin_inff();
        const int b1 = (q & 1) * Cfg::IN_PLANE + (o1 * Cfg::IW + px) * 16;
        int c_p1 = b1 + (q >> 1) * 16, c_w1 = ((q >> 1) + 2 * (q & 1)) * Cfg::MID_PLANE + (o1 * Cfg::MW + px) * 16;

        H3Tile cur = walk.tile<Cfg>();
        {   // prologue: the first tile, all PF instructions at once
            const char* origin = reinterpret_cast<const char*>(a.in) + cur.img + ((ptrdiff_t)(cur.y0 - 2) * a.W + (cur.x0 - 2)) * 16;
            const H3SDmaHook<Cfg, false, Cfg::PF> all{a, cur, origin, tin0, dlane, wave, pfoff, true};
            all.template row<0>(); all.template row<1>(); all.template row<2>(); all.template row<3>(); all.template row<4>();
            all.template row<5>(); all.template row<6>(); all.template row<7>(); all.template row<8>(); all.template row<9>();
            all.template row<10>(); all.template row<11>();
            static_assert(Cfg::PF <= 12, "prologue issues rows 0..11");
            __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // also retires the weight / scale loads above
            h3_barrier();
        }
        for (int k = 0; k <= ntl; ++k) {
            if (k < ntl) {
                const int buf = k & 1;
                const char* tin = tin0 + buf * Cfg::TIN_BYTES;
                char* tmid = tmid0 + buf * Cfg::TMID_BYTES;
                const bool interior = h3s_interior<Cfg>(a, cur);
                const bool has1 = k + 1 < ntl;
                H3Tile nx = cur;
                if (has1) { walk.advance(); nx = walk.tile<Cfg>(); }
                const char* nx_origin = reinterpret_cast<const char*>(a.in) + nx.img + ((ptrdiff_t)(nx.y0 - 2) * a.W + (nx.x0 - 2)) * 16;
                char* tnx = tin0 + (buf ^ 1) * Cfg::TIN_BYTES;
                const bool nx_interior = h3s_interior<Cfg>(a, nx);
                int p1 = c_p1, wr1 = c_w1;
                asm volatile("" : "+v"(p1), "+v"(wr1));           // keeps LICM from hoisting every (constant + immediate) address
                // strip group first (8 rows x 2 columns), on the first SG conv1 waves
                if (wave < Cfg::SG) {
                    const int srow = min(8 * wave + (n >> 1), Cfg::MH - 1);
                    const int scol = Cfg::TW + (n & 1);
                    const int bg = (q & 1) * Cfg::IN_PLANE + (srow * Cfg::IW + scol) * 16;
                    const int gw = ((q >> 1) + 2 * (q & 1)) * Cfg::MID_PLANE + (srow * Cfg::MW + scol) * 16;
                    const f32x4 v = bf_acc_ready(h3r_group<Cfg::IW * 16, 2 * Cfg::IN_PLANE>(tin, bg + (q >> 1) * 16, bg + 32 + (q >> 1) * 2 * Cfg::IN_PLANE, w1));
                    if (interior) h3r_conv1_store<Cfg, true>(a, tmid, gw, v, inv_s1, relu_floor, 0, 0);
                    else h3r_conv1_store<Cfg, false>(a, tmid, gw, v, inv_s1, relu_floor, cur.y0 - 1 + srow, cur.x0 - 1 + scol);
                }
#define H3S_CONV1(INT, NXI)                                                                                                  \
                do {                                                                                                         \
                    const H3SDmaHook<Cfg, NXI, Cfg::R1 + 2> hook{a, nx, nx_origin, tnx, dlane, wave, pfoff, has1};            \
                    struct Epi {                                                                                              \
                        struct Pre {};                                                                                        \
                        enum { EXTRA_MFMA = 0 };                                                                              \
                        const FusedH3Args& a; char* __restrict__ tmid; float inv_s, relu_floor; int w1, gy0, gx;              \
                        __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }                                 \
                        __device__ __forceinline__ f32x4 finish(const int, const f32x4 acc, const Pre&) const { return acc; } \
                        __device__ __forceinline__ void operator()(const int o, const f32x4 v) const                          \
                        {                                                                                                     \
                            h3r_conv1_store<Cfg, INT>(a, tmid, w1 + o * Cfg::MW * 16, v, inv_s, relu_floor, gy0 + o, gx);     \
                        }                                                                                                     \
                    } const epi{a, tmid, inv_s1, relu_floor, wr1, cur.y0 - 1 + o1, cur.x0 - 1 + px};                          \
                    h3r_rows<Cfg::R1, Cfg::IW * 16, 2 * Cfg::IN_PLANE>(tin, p1, p1 - (q >> 1) * 16 + 32 + (q >> 1) * 2 * Cfg::IN_PLANE, w1, epi, hook); \
                } while (0)
                if (interior) { if (nx_interior) H3S_CONV1(true, true); else H3S_CONV1(true, false); }
                else { if (nx_interior) H3S_CONV1(false, true); else H3S_CONV1(false, false); }
#undef H3S_CONV1
                cur = nx;
                __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));           // the next tile has landed (only DMA in this wave's queue)
            }
            h3_barrier();
        }
    } else {
        // ------------------------------------------------------------------ conv2 waves --------------------------
        const int o2 = half * Cfg::R2;
        h8 w2[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) w2[i] = reinterpret_cast<const h8*>(a.w2)[i * 64 + lane];
        const float inv_s2 = a.aux[48];                             // BN scale is folded into the row-layout w2
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.aux + 32 + q * 4);
        const int b2 = (q & 1) * Cfg::MID_PLANE + (o2 * Cfg::MW + px) * 16;
        int c_p2 = b2 + (q >> 1) * 16;
        // residual operand [x_hi | x_lo] of the centre pixel: lanes q < 2 read the hi planes, q >= 2 the lo planes
        int c_rr = ((q & 1) + 2 * (q >> 1)) * Cfg::IN_PLANE + ((o2 + 2) * Cfg::IW + px + 2) * 16;
        unsigned c_g = (unsigned)((q >> 1) + 2 * (q & 1)) * plane_g + (unsigned)(o2 * a.W + px) * 16u;

        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));                   // weight / scale loads
        h3_barrier();                                            // prologue barrier: tile 0 is in tin0
        h8 res[Cfg::R2];
        H3Tile prev = walk.tile<Cfg>();                          // tile k-1 of iteration k
        for (int k = 0; k <= ntl; ++k) {
            if (k >= 1) {
                const int buf = (k - 1) & 1;
                const char* tmid = tmid0 + buf * Cfg::TMID_BYTES;
                const bool interior = h3s_interior<Cfg>(a, prev);
                char* out_row0 = reinterpret_cast<char*>(a.out) + prev.img + ((size_t)prev.y0 * a.W + prev.x0) * 16;
                int p2 = c_p2;
                unsigned g = c_g;
                asm volatile("" : "+v"(p2), "+v"(g));
                struct Epi2 {
                    struct Pre {};
                    enum { EXTRA_MFMA = 1 };
                    const FusedH3Args& a; h8 wres; char* out_row0; size_t rowbytes, lo_g;
                    float inv_s2; f32x4 sh; bool interior; int y_base, x_px, lane; unsigned g;
                    h8 res[Cfg::R2];
                    __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
                    // residual on the matrix pipe: acc += (s2 * I) x [x_hi | x_lo] (exact)
                    __device__ __forceinline__ f32x4 finish(const int o, const f32x4 acc, const Pre&) const { return MFMA_H(wres, res[o], acc); }
                    __device__ __forceinline__ void operator()(const int o, const f32x4 acc) const
                    {
                        const f32x4 v = acc * inv_s2 + sh;
                        char* p = out_row0 + o * rowbytes + g;
                        if (!(y_base + o < a.H && x_px < a.W))   /* branch-free on purpose: see h3_split_record's neighbour comment */ p = reinterpret_cast<char*>(a.dump) + lane * 16;
                        *reinterpret_cast<h8*>(p) = h3_split_record(v);
                    }
                };
                Epi2 epi2{a, w2[12], out_row0, (size_t)a.W * 16, 2 * (size_t)plane_g, inv_s2, sh, interior,
                          prev.y0 + o2, prev.x0 + px, lane, g, {}};
#pragma unroll
                for (int o = 0; o < Cfg::R2; ++o) epi2.res[o] = res[o];
                h3r_rows<Cfg::R2, Cfg::MW * 16, 2 * Cfg::MID_PLANE>(tmid, p2, p2 - (q >> 1) * 16 + 32 + (q >> 1) * 2 * Cfg::MID_PLANE, w2, epi2, H3NoHook{});
                prev = walk.tile<Cfg>();
            }
            if (k < ntl) {
                // residual fragments of tile k, one iteration early: the input buffer is then free for the DMA of
                // tile k+2 right after the barrier below
                if (k >= 1) { walk.advance(); prev = walk.tile<Cfg>(); }
                const char* tin = tin0 + (k & 1) * Cfg::TIN_BYTES;
                int rr = c_rr;
                asm volatile("" : "+v"(rr));
#pragma unroll
                for (int o = 0; o < Cfg::R2; ++o) res[o] = *reinterpret_cast<const h8*>(tin + rr + o * Cfg::IW * 16);
            }
            h3_barrier();
        }
    }
}

using H3Spec = H3SCfg<14, 32>;

// ==========================================================================================================
// Single 3x3 16->16 convolution on fp32 NHWC tensors with the split-f16 arithmetic and the row-streaming inner loop of
// the fused kernels above: the training convolutions (forward conv1 / conv2, data gradients) -- same epilogue stages
// as conv3x3_c16_kernel ([ReLU] [mask] [+residual] [BN statistics]), same 16x32 tiles and grid, so it is a drop-in.
// The fp32 tile is split into hi / lo f16 planes while it is staged into LDS (8 vector instructions per 4 values);
// a wave streams 8 rows of one 16-column strip; the fp32 result goes straight from the accumulator to HBM.
// 15 MFMAs of 16 cycles per 16 pixels instead of 36 MFMAs of 32 cycles.
// ==========================================================================================================
struct ConvH3Geom {
    static constexpr int TH = 16, TW = 32, IH = TH + 2, IW = TW + 2, R = 8;
    static constexpr int PLANE = (IH * IW * 16 + 255) / 256 * 256;
    static constexpr int LDS_BYTES = 4 * PLANE;
};

template <int EPI, bool PRE = false>
__global__ __launch_bounds__(256, 2) void conv3x3_h3_kernel(ConvArgs a)
{
    using G = ConvH3Geom;
    __shared__ __attribute__((aligned(16))) char tile[G::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int tiles_x = (a.W + G::TW - 1) / G::TW, tiles_y = (a.H + G::TH - 1) / G::TH;
    int t = a.reverse ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int y0 = ty * G::TH, x0 = tx * G::TW;
    const size_t img = (size_t)b * a.H * a.W * 16;

    // weights: 12 A-operand images + 1/s (pack_h3_train_kernel)
    h8 w[13];
#pragma unroll
    for (int i = 0; i < 12; ++i) w[i] = reinterpret_cast<const h8*>(a.wpack)[i * 64 + lane];
    w[12] = w[0];
    const float inv_s = a.wpack[BF_H3R_WPACK_FLOATS];

    // stage: fp32 NHWC (1-pixel halo, zero outside the image) -> hi / lo planes [4][IH][IW][8 x f16].  All the loads of a
    // thread are issued before the first one is consumed (a rolled load -> split -> store loop pays one memory round trip
    // per element: hipcc does not pipeline it)
    {
        constexpr int NX = (G::IH * G::IW * 4 + 255) / 256;
        f32x4 rx[NX], rc[PRE ? NX : 1];
        f32x4 psc = {0.f, 0.f, 0.f, 0.f}, psh = {0.f, 0.f, 0.f, 0.f};
        if (PRE) {                              // tid & 3 is the channel quad of every element this thread stages
            psc = *reinterpret_cast<const f32x4*>(a.pre_scale + (tid & 3) * 4);
            psh = *reinterpret_cast<const f32x4*>(a.pre_shift + (tid & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * 256;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::IW, col = px - row * G::IW;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (PRE) rc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH * G::IW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                const size_t idx = img + ((size_t)gy * a.W + gx) * 16 + quad * 4;
                rx[i] = *reinterpret_cast<const f32x4*>(a.in + idx);
                if (PRE) rc[i] = *reinterpret_cast<const f32x4*>(a.pre_c + idx);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * 256;
            if (e < G::IH * G::IW * 4) {
                const int px = e >> 2, quad = e & 3;
                f32x4 v = rx[i];
                if (PRE) {
                    // y = x + (scale * c + shift) as affine_add_kernel rounds it; 0 outside the image (SAME padding); the tile's
                    // own pixels (not the halo, which the neighbours own) go back to HBM: the block input the backward pass needs
                    const int row = px / G::IW, col = px - row * G::IW;
                    const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                    const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = in ? rx[i][k] + fmaf(psc[k], rc[i][k], psh[k]) : 0.f;
                    if (in && row >= 1 && row <= G::TH && col >= 1 && col <= G::TW)
                        *reinterpret_cast<f32x4*>(a.pre_out + img + ((size_t)gy * a.W + gx) * 16 + quad * 4) = v;
                }
                h4 hi, lo;
                h3_split(v, hi, lo);
                char* p = tile + (quad >> 1) * G::PLANE + px * 16 + (quad & 1) * 8;
                *reinterpret_cast<h4*>(p) = hi;
                *reinterpret_cast<h4*>(p + 2 * G::PLANE) = lo;
            }
        }
    }
    __syncthreads();

    const int strip = wave & 1, half = wave >> 1;
    const int px_l = strip * 16 + n;                              // column inside the tile
    const int o0 = half * G::R;                                   // first output row of this wave
    const int b1 = (q & 1) * G::PLANE + (o0 * G::IW + px_l) * 16;
    const int gx = x0 + px_l;
    struct Epi {
        struct Pre {};
        enum { EXTRA_MFMA = 0 };
        const ConvArgs& a; size_t base; int gy0, gx, q; float inv_s; f32x4 sc, sh;
        f32x4* s1; f32x4* s2;
        __device__ __forceinline__ Pre pre(const int) const { return Pre{}; }
        __device__ __forceinline__ f32x4 finish(const int, const f32x4 acc, const Pre&) const { return acc; }
        __device__ __forceinline__ void operator()(const int o, const f32x4 acc) const
        {
            if (gy0 + o < a.H && gx < a.W) {
                const size_t idx = base + (size_t)o * a.W * 16;
                f32x4 v = acc * inv_s;
                if (EPI & EPI_STATS) { *s1 += v; *s2 += v * v; }
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
                if (EPI & EPI_BNBWD) { *s1 += v; *s2 += v * *reinterpret_cast<const f32x4*>(a.bnc + idx); }
                *reinterpret_cast<f32x4*>(a.out + idx) = v;
            }
        }
    };
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (EPI & EPI_AFFINE) {
        sc = *reinterpret_cast<const f32x4*>(a.scale + q * 4);
        sh = *reinterpret_cast<const f32x4*>(a.shift + q * 4);
    }
    const Epi epi{a, img + ((size_t)(y0 + o0) * a.W + gx) * 16 + q * 4, y0 + o0, gx, q, inv_s, sc, sh, &s1, &s2};
    h3r_rows<G::R, G::IW * 16, 2 * G::PLANE>(tile, b1 + (q >> 1) * 16, b1 + 32 + (q >> 1) * 2 * G::PLANE, w, epi, H3NoHook{});

    if (EPI & (EPI_STATS | EPI_BNBWD)) {
        // reduce over the 16 pixel lanes that share a channel quad, then over the 4 waves (fixed order)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2[c] += __shfl_xor(s2[c], m);
            }
        }
        __syncthreads();                       // tile no longer needed
        float* red = reinterpret_cast<float*>(tile);      // [4 waves][32]
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

hipError_t bf_launch_conv3x3_h3(const ConvArgs& a, int epi, hipStream_t s)
{
    const dim3 grid(bf_conv3x3_c16_grid(a.B, a.H, a.W)), block(256);
    if (a.pre_c) {
        // affine + add on load: in front of a block's first convolution only ([activation] epilogue)
        if (!a.pre_scale || !a.pre_shift || !a.pre_out || a.pre_out == a.in || a.pre_out == a.pre_c) return hipErrorInvalidValue;
        if (epi == EPI_RELU) hipLaunchKernelGGL((conv3x3_h3_kernel<EPI_RELU, true>), grid, block, 0, s, a);
        else if (epi == 0) hipLaunchKernelGGL((conv3x3_h3_kernel<0, true>), grid, block, 0, s, a);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#define BF_CASE(E) case E: hipLaunchKernelGGL(conv3x3_h3_kernel<E>, grid, block, 0, s, a); break;
    switch (epi) {
        BF_CASE(0)
        BF_CASE(EPI_RELU)
        BF_CASE(EPI_STATS)
        BF_CASE(EPI_RES)
        BF_CASE(EPI_RES | EPI_BNBWD)
        BF_CASE(EPI_MASK)
        default: return hipErrorInvalidValue;
    }
#undef BF_CASE
    return hipGetLastError();
}

// training packs: one workgroup per (layer, which) with which = 0 w1 forward, 1 w2 forward, 2 w1 data gradient,
// 3 w2 data gradient (W'[tap][ci][co] = W[8-tap][co][ci]); row layout of the fused kernels without BN folding or
// identity; dst = [12 x 64 x 16 B][1/s broadcast x 64 floats]
__global__ __launch_bounds__(256) void pack_h3_train_kernel(const float* __restrict__ params, int64_t p_blocks, int64_t p_stride,
                                                            float* __restrict__ dst, int64_t d_stride, int nconv, int unit)
{
    // blockIdx.x = layer * 2 * nconv + which ; which < nconv: forward pack of convolution `which`, else the data-gradient pack
    // of convolution which - nconv.  Convolution j of a block sits at j * 2304 (+ (j - 1) * 16 behind the gammas: unit = 2320)
    __shared__ float red[256];
    __shared__ float s_scale;
    const int per = 2 * nconv;
    const int layer = blockIdx.x / per, which = blockIdx.x % per;
    const int cj = which % nconv;
    const float* w = params + p_blocks + layer * p_stride + (cj == 0 ? 0 : 2304 + (int64_t)(cj - 1) * unit);
    const int tf = which / nconv;
    float m = 0.f;
    for (int i = threadIdx.x; i < 2304; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float sr = 1.f;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            ex = max(-100, min(100, ex));
            sr = ldexpf(1.f, 14 - ex);
        }
        s_scale = sr;
    }
    __syncthreads();
    const float sr = s_scale;
    float* out = dst + ((int64_t)layer * per + which) * d_stride;
    _Float16* orow = reinterpret_cast<_Float16*>(out);
    for (int idx = threadIdx.x; idx < 12 * 64 * 8; idx += 256) {
        const int i = idx >> 9, l = (idx >> 3) & 63, j = idx & 7;
        const int cout = l & 15, kslot = 8 * (l >> 4) + j, half = kslot >> 4, cin = kslot & 15;
        const int dy = i >> 2, kind = i & 3;
        int tap, part;
        if (kind == 0) { tap = dy * 3 + half; part = 0; }
        else if (kind == 1) { tap = dy * 3 + half; part = 1; }
        else if (kind == 2) { tap = dy * 3 + 2; part = 0; }
        else { tap = dy * 3 + 2; part = half ? 2 : 1; }
        const float wv = tf ? w[((8 - tap) * 16 + cout) * 16 + cin] : w[(tap * 16 + cin) * 16 + cout];
        const float ws = wv * sr;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        orow[idx] = part == 0 ? hi : (part == 1 ? lo : (_Float16)0.f);
    }
    for (int idx = 12 * 64 * 8 + threadIdx.x; idx < 13 * 64 * 8; idx += 256) orow[idx] = (_Float16)0.f;     // unused 13th image
    if (threadIdx.x < 64) out[BF_H3R_WPACK_FLOATS + threadIdx.x] = 1.0f / sr;
}

hipError_t bf_launch_pack_h3_train(const float* params, int64_t p_blocks, int64_t p_stride, float* dst, int layers, int nconv,
                                   int unit, hipStream_t s)
{
    if (layers <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_h3_train_kernel, dim3(layers * 2 * nconv), dim3(256), 0, s, params, p_blocks, p_stride, dst,
                       (int64_t)BF_H3_TRAIN_PACK_FLOATS, nconv, unit);
    return hipGetLastError();
}

// ==========================================================================================================
// Weight gradient of the 3x3 16->16 convolution with the split-f16 arithmetic:
//   dW[tap][ci][co] = sum over pixels of X[pixel + tap][ci] * dY[pixel][co]
//                   ~ X_hi.dY_hi + X_lo.dY_hi + X_hi.dY_lo                       (fp32 accumulation)
// GEMM view per tap: M = ci, N = co, K = pixels, 32 pixels (one tile row) per v_mfma_f32_16x16x32_f16.  Both operands
// need the PIXEL index along K while the tiles are pixel-major in memory: the LDS images stay [pixel][16 channels]
// (32 B per pixel, written with 8-byte stores while the fp32 tile is split) and ds_read_b64_tr_b16 delivers them
// transposed -- lane 16g+i receives channel i of pixels 4g..4g+3 -- two reads per operand and K chunk (k-slots 0..3 of
// lane group g = pixels 4g..4g+3, k-slots 4..7 = pixels 16+4g..16+4g+3: the two 32-lane halves of a read touch
// disjoint 256-B windows, no bank conflicts).  27 MFMAs of 16 cycles per 32 pixels instead of 72 of 32 cycles, and 40
// LDS reads instead of 80.  Nine accumulators stay in registers across the tiles of a persistent workgroup; partials
// are reduced in a fixed order (no float atomics -> bitwise reproducible), exactly as wgrad3x3_c16_kernel does.
// ==========================================================================================================
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct WgradH3Geom {
    static constexpr int TH = 16, TW = 32, IH = TH + 2, IW = TW + 2;
    static constexpr int X_IMG = IH * IW * 32, D_IMG = TH * TW * 32;       // bytes per f16 image
    static constexpr int LDS_BYTES = 2 * X_IMG + 2 * D_IMG;                // 71,936
};

__device__ __forceinline__ h8 h3_tr_operand(const char* img, const int addr)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(img + addr));
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(img + addr + 16 * 32));
    const u2 ua = __builtin_bit_cast(u2, a), ub = __builtin_bit_cast(u2, b);
    return __builtin_bit_cast(h8, (u4){ua[0], ua[1], ub[0], ub[1]});
}

__global__ __launch_bounds__(256, 2) void wgrad3x3_h3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ partial, int B, int H, int W, int tiles_x,
                                                             int tiles_y, int ntiles)
{
    using G = WgradH3Geom;
    extern __shared__ __attribute__((aligned(16))) char wg_lds[];
    char* xh = wg_lds;                      // [IH][IW][16] f16 hi
    char* xl = wg_lds + G::X_IMG;           // lo
    char* dh = wg_lds + 2 * G::X_IMG;       // [TH][TW][16] f16 hi
    char* dl = dh + G::D_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // transposed-read address of this lane inside a 32-pixel row chunk: pixel 4g + q', channels 4p'..4p'+3
    const int tr_off = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    f32x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The fp32 elements of a tile are fetched into registers one tile ahead (NX + ND 16-byte loads per thread) and split /
    // written to LDS after the matrix work of the previous tile: the global-memory latency of tile t+1 hides behind the
    // MFMAs of tile t inside the workgroup, instead of relying on the second workgroup of the CU alone.
    constexpr int NX = (G::IH * G::IW * 4 + 255) / 256, ND = G::TH * G::TW * 4 / 256;
    f32x4 rx[NX], rd[ND];
    auto fetch = [&](const int t) {
        int tt = t;
        const int txi = tt % tiles_x; tt /= tiles_x;
        const int tyi = tt % tiles_y;
        const int b = tt / tiles_y;
        const int y0 = tyi * G::TH, x0 = txi * G::TW;
        const size_t img = (size_t)b * H * W * 16;
#pragma unroll
        for (int i = 0; i < NX; ++i) {                     // x with a 1-pixel halo (zero outside the image)
            const int e = tid + i * 256;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::IW, col = px - row * G::IW;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e < G::IH * G::IW * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                rx[i] = *reinterpret_cast<const f32x4*>(x + img + ((size_t)gy * W + gx) * 16 + quad * 4);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {                     // dy (zero outside the image)
            const int e = tid + i * 256;
            const int px = e >> 2, quad = e & 3;
            const int row = px / G::TW, col = px - row * G::TW;
            const int gy = y0 + row, gx = x0 + col;
            rd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W) rd[i] = *reinterpret_cast<const f32x4*>(dy + img + ((size_t)gy * W + gx) * 16 + quad * 4);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // split + store the prefetched tile
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + i * 256;
            if (e < G::IH * G::IW * 4) {
                h4 hi, lo;
                h3_split(rx[i], hi, lo);
                *reinterpret_cast<h4*>(xh + (e >> 2) * 32 + (e & 3) * 8) = hi;
                *reinterpret_cast<h4*>(xl + (e >> 2) * 32 + (e & 3) * 8) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int e = tid + i * 256;
            h4 hi, lo;
            h3_split(rd[i], hi, lo);
            *reinterpret_cast<h4*>(dh + (e >> 2) * 32 + (e & 3) * 8) = hi;
            *reinterpret_cast<h4*>(dl + (e >> 2) * 32 + (e & 3) * 8) = lo;
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        // wave handles rows 4w .. 4w+3 of the tile: four K chunks of 32 pixels
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            const h8 bh = h3_tr_operand(dh, r * G::TW * 32 + tr_off);
            const h8 bl = h3_tr_operand(dl, r * G::TW * 32 + tr_off);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ax = ((r + tap / 3) * G::IW + tap % 3) * 32 + tr_off;
                const h8 ah = h3_tr_operand(xh, ax);
                const h8 al = h3_tr_operand(xl, ax);
                acc[tap] = MFMA_H(ah, bh, acc[tap]);
                acc[tap] = MFMA_H(al, bh, acc[tap]);
                acc[tap] = MFMA_H(ah, bl, acc[tap]);
            }
        }
        __syncthreads();
    }
    // cross-wave reduction through LDS: [4][9][256], D[ci = 4q + j][co = p] per lane
    float* red = reinterpret_cast<float*>(wg_lds);
    const int p = lane & 15, q = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = bf_acc_ready(acc[tap]);
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(wave * 9 + tap) * 256 + (4 * q + j) * 16 + p] = v[j];
    }
    __syncthreads();
    for (int i = tid; i < 2304; i += 256)
        partial[(size_t)blockIdx.x * 2304 + i] = (red[i] + red[2304 + i]) + (red[2 * 2304 + i] + red[3 * 2304 + i]);
}

hipError_t bf_launch_wgrad3x3_h3(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W, hipStream_t s)
{
    using G = WgradH3Geom;
    const int tiles_x = (W + G::TW - 1) / G::TW, tiles_y = (H + G::TH - 1) / G::TH;
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = bf_wgrad_grid(B, H, W);
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(wgrad3x3_h3_kernel), G::LDS_BYTES);      // once per device
        if (ea != hipSuccess) return ea;
    }
    hipLaunchKernelGGL(wgrad3x3_h3_kernel, dim3(grid), dim3(256), G::LDS_BYTES, s, x, dy, partial, B, H, W, tiles_x, tiles_y, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return bf_launch_reduce_partials(partial, grid, 2304, dw, 1.0f, s);
}

using H3Default = H3Cfg<16, 32, 8>;
using H3Small = H3Cfg<16, 16, 4>;

// Default kernel (variant < 0 everywhere): the full-row streaming kernel (4, fused_h3v.hip) for images up to 256 columns when
// the batch holds enough rows to amortise a band's ~10-step fill (B * H >= 3072: from ~12 rows per workgroup on; measured at
// 256 x 256: batch 8 0.46 ms (tiles) vs 0.49, batch 16 0.78 vs 0.75, batch 128 5.5 vs 4.5), the row-streaming tile kernel (1, or
// 2 for very small inputs) otherwise -- a single 256 x 256 image takes 162 us through the 18 blocks on tiles, 367 us on one-row bands.
static int g_h3_variant = -1;                         // override of the default for debug entries without a handle (tests, A/B)
void bf_set_h3_variant(int v) { g_h3_variant = v < 0 ? -1 : v; }
// Below one 16 x 32 tile per CU (or between one and two) the 16 x 16 tiles of variant 2 -- two 4-wave workgroups per CU -- put more of
// the chip to work: resnet 1x18 on one 256 x 256 image 149 us for 177, one 128 x 128 image 138 for 166, three 256 x 256 images 241 for
// 267; from 512 tiles on the larger tile is 3-5 % faster (tools/exp/small_batch_variants.py).
static int h3_default_variant(const FusedH3Args& a)
{
    // (short AND narrow images -- under 24 rows, up to 128 columns -- stay on tiles: 2 000 x 4 x 64 1 186 us on tiles, 1 559 streaming)
    if (!a.head_wh && bf_fused_block_h3v_supports(a.H, a.W) && (int64_t)a.B * a.H >= 3072 && (a.H >= 24 || a.W > 128)) return 4;
    const int64_t tiles32 = (int64_t)a.B * ((a.H + 15) / 16) * ((a.W + 31) / 32);
    return (!a.head_wh && !a.compact && tiles32 < 512 && tiles32 != 256) ? 2 : 1;
}

template <class Cfg, int VARIANT>
static hipError_t launch_h3(void (*kernel)(FusedH3Args), const FusedH3Args& a, hipStream_t s)
{
    {
        const hipError_t ea = bf_set_max_lds(reinterpret_cast<const void*>(kernel), Cfg::LDS_BYTES);      // once per device
        if (ea != hipSuccess) return ea;
    }
    const int resident = 256 * Cfg::WG_PER_CU;                  // persistent: every workgroup resident at once
    int grid = a.ntiles < resident ? a.ntiles : resident;
    if (grid >= 8) grid -= grid % 8;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, s, a);
    return hipGetLastError();
}

bool bf_fused_block_h3_is_streaming(const FusedH3Args& a)
{
    const int variant = (a.variant >= 0 ? a.variant : (g_h3_variant >= 0 ? g_h3_variant : h3_default_variant(a))) & 255;
    return variant == 4 && !a.head_wh && bf_fused_block_h3v_supports(a.H, a.W);
}

// Two blocks per launch (fused_h3w.hip) beyond the shapes of the one-block streaming kernel: its 128-column strips take any image width,
// and at two blocks per launch a band's fill is amortised earlier -- with the library's default selection, from 4 096 rows of strips per
// forward on (B * H * ceil(W / 128)) consecutive blocks run two per launch; an odd block count runs its single block on whatever
// bf_launch_fused_block_h3 picks.  Measured (tools/exp/regime_sweep.py, resnet 1x18, us per forward, pairs / 16 x 32 tiles):
// 8 x 256^2 403 / 429, 6 x 256^2 357 / 355, 2 x 512^2 404 / 424, 1 x 512^2 295 / 282, 32 x 128^2 397 / 420, 16 x 128^2 291 / 281,
// 32 x 512^2 4 300 / 5 570, 1 x 1080 x 1920 2 207 / 2 747.
// Images of fewer than 24 rows are the exception (a band's 12-step fill per handful of rows): 512 x 8 x 256 runs 830 us on the one-block
// streaming kernel, 899 in pairs; 300 x 20 x 100 867 on tiles, 969 in pairs -- they keep the one-block selection of h3_default_variant.
// A variant forced through set_option / bf_debug_set_h3_variant pairs exactly where it selects the streaming kernel (tests, A/B).
bool bf_fused_block_h3_use_pairs(const FusedH3Args& a)
{
    if (a.head_wh || a.compact) return false;
    if (a.variant >= 0 || g_h3_variant >= 0) return bf_fused_block_h3_is_streaming(a);
    const int64_t nstrips = (a.W + 127) / 128;
    return a.W >= 1 && a.H >= 24 && (int64_t)a.B * a.H * nstrips >= 4096;
}

// name of the kernel bf_launch_fused_block_h3 launches for these arguments
const char* bf_fused_block_h3_kernel_name(const FusedH3Args& a)
{
    if (bf_fused_block_h3_is_streaming(a)) return "fused_block_h3v_kernel";
    const int variant = (a.variant >= 0 ? a.variant : (g_h3_variant >= 0 ? g_h3_variant : h3_default_variant(a))) & 255;
    return variant == 0 ? "fused_block_h3_kernel" : (variant == 3 ? "fused_block_h3s_kernel" : "fused_block_h3r_kernel");
}

hipError_t bf_launch_fused_block_h3(const FusedH3Args& args, hipStream_t s)
{
    FusedH3Args a = args;
    if (!a.zeros || !a.dump) return hipErrorInvalidValue;
    if ((int64_t)a.H * a.W * 64 >= ((int64_t)1 << 32)) return hipErrorInvalidValue;      // 32-bit in-image offsets
    int variant = a.variant >= 0 ? a.variant : (g_h3_variant >= 0 ? g_h3_variant : h3_default_variant(a));
    if (variant & 256) a.reverse_tiles = 1;                      // tests: the bottom-up walk of the full-row streaming kernel
    variant &= 255;
    // full-row streaming kernel: images up to 256 columns, no head epilogue (the tile kernel below takes the rest)
    if (variant == 4 && !a.head_wh && bf_fused_block_h3v_supports(a.H, a.W)) return bf_launch_fused_block_h3v(a, s);
    if (a.compact) return hipErrorInvalidValue;                 // only the streaming kernel reads the compact layout
    if (variant == 2) {                                    // two 4-wave workgroups per CU on 16x16 tiles
        using Cfg = H3Small;
        a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
        a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
        a.ntiles = a.B * a.tiles_x * a.tiles_y;
        a.w1 = a.w1r; a.w2 = a.w2r;
        return launch_h3<Cfg, 2>(fused_block_h3r_kernel<Cfg>, a, s);
    }
    if (variant == 3) {                                    // wave-specialised: conv1 waves / conv2 waves, 14x32 tiles
        using Cfg = H3Spec;
        a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
        a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
        a.ntiles = a.B * a.tiles_x * a.tiles_y;
        a.w1 = a.w1r; a.w2 = a.w2r;
        return launch_h3<Cfg, 3>(fused_block_h3s_kernel<Cfg>, a, s);
    }
    using Cfg = H3Default;
    a.tiles_x = (a.W + Cfg::TW - 1) / Cfg::TW;
    a.tiles_y = (a.H + Cfg::TH - 1) / Cfg::TH;
    a.ntiles = a.B * a.tiles_x * a.tiles_y;
    if (variant == 0) return launch_h3<Cfg, 0>(fused_block_h3_kernel<Cfg>, a, s);
    a.w1 = a.w1r; a.w2 = a.w2r;                                 // horizontally paired weights
    if (a.head_wh) return launch_h3<Cfg, 5>(fused_block_h3r_kernel<Cfg, 1>, a, s);
    return launch_h3<Cfg, 1>(fused_block_h3r_kernel<Cfg>, a, s);
}

// ------------------------------------------------------------------------------------------
// weight packing for the kernel above.  One workgroup per (layer, conv): power-of-two scale from max |w|,
// then the ten A-operand register images [i][lane][8 x f16]:
//   i = 0..3  w_hi of tap pair i ; i = 4..7  w_lo of tap pair i-4 ; i = 8  [w_hi | w_lo] of tap (2,2) ;
//   i = 9  [w_hi | 0] of tap (2,2).     lane l: output channel l & 15, k-slots 8*(l >> 4) .. +7
//   (k-slot < 16: first tap of the pair, >= 16: second tap; input channel = k-slot & 15).
// aux[0..15] = 1/s1, aux[16..31] = folded BN scale / s2, aux[32..47] = folded BN shift.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_h3_kernel(const float* __restrict__ params, const float* __restrict__ state,
                                                      int64_t p_blocks, int64_t p_stride, float* __restrict__ dst,
                                                      int64_t d_stride, int use_bn, float eps,
                                                      const float* __restrict__ ext_scale, const float* __restrict__ ext_shift)
{
    __shared__ float red[256];
    __shared__ float s_scale;
    const int layer = blockIdx.x >> 1, which = blockIdx.x & 1;
    const float* w = params + p_blocks + layer * p_stride + which * 2304;      // HWIO [3][3][16][16]
    float m = 0.f;
    for (int i = threadIdx.x; i < 2304; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float s = 1.f;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);                   // mx = f * 2^ex, f in [0.5, 1)
            ex = max(-100, min(100, ex));
            s = ldexpf(1.f, 14 - ex);                // mx * s in [2^13, 2^14)
        }
        s_scale = s;
    }
    __syncthreads();
    const float s = s_scale;
    _Float16* o = reinterpret_cast<_Float16*>(dst + layer * d_stride + which * BF_H3_WPACK_FLOATS);
    const int tapA[4] = {0, 1, 2, 6}, tapB[4] = {3, 4, 5, 7};
    for (int idx = threadIdx.x; idx < 10 * 64 * 8; idx += 256) {
        const int i = idx >> 9, l = (idx >> 3) & 63, j = idx & 7;
        const int cout = l & 15, kslot = 8 * (l >> 4) + j, half = kslot >> 4, cin = kslot & 15;
        int tap, part;                                // part: 0 = hi, 1 = lo, 2 = zero
        if (i < 4) { tap = half ? tapB[i] : tapA[i]; part = 0; }
        else if (i < 8) { tap = half ? tapB[i - 4] : tapA[i - 4]; part = 1; }
        else if (i == 8) { tap = 8; part = half; }
        else { tap = 8; part = half ? 2 : 0; }
        const float ws = w[(tap * 16 + cin) * 16 + cout] * s;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        o[idx] = part == 0 ? hi : (part == 1 ? lo : (_Float16)0.f);
    }
    // row-streaming layout: [dy*4 + {pair hi, pair lo, single [hi | hi], single [lo | 0]}][lane][8], [12] = sr * identity.
    // conv2 (which == 1): the folded BN scale is multiplied INTO the weights (per output channel) and the kernel adds the
    // residual as (sr * I) x [x_hi | x_lo] on the matrix pipe, so sr must itself be an f16 number: sr <= 2^15.
    __shared__ float s_fold[16];
    __shared__ float s_scale_r;
    if (threadIdx.x < 16) {
        float f = 1.f;
        if (which == 1) {
            if (ext_scale) f = ext_scale[threadIdx.x];
            else if (use_bn) f = params[p_blocks + layer * p_stride + 4608 + threadIdx.x] / sqrtf(state[layer * 32 + 16 + threadIdx.x] + eps);
        }
        s_fold[threadIdx.x] = f;
    }
    __syncthreads();
    m = 0.f;
    for (int i = threadIdx.x; i < 2304; i += 256) m = fmaxf(m, fabsf(w[i] * s_fold[i & 15]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float sr = 1.f;
        const float mx = red[0];
        if (mx > 0.f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            ex = max(-100, min(100, ex));
            sr = ldexpf(1.f, 14 - ex);
        }
        if (which == 1) sr = fminf(sr, 32768.f);
        s_scale_r = sr;
    }
    __syncthreads();
    const float sr = s_scale_r;
    _Float16* orow = reinterpret_cast<_Float16*>(dst + layer * d_stride + 2 * BF_H3_WPACK_FLOATS + 64 + which * BF_H3R_WPACK_FLOATS);
    for (int idx = threadIdx.x; idx < 13 * 64 * 8; idx += 256) {
        const int i = idx >> 9, l = (idx >> 3) & 63, j = idx & 7;
        const int cout = l & 15, kslot = 8 * (l >> 4) + j, half = kslot >> 4, cin = kslot & 15;
        if (i == 12) {
            orow[idx] = (which == 1 && cin == cout) ? (_Float16)sr : (_Float16)0.f;
            continue;
        }
        const int dy = i >> 2, kind = i & 3;
        int tap, part;
        if (kind == 0) { tap = dy * 3 + half; part = 0; }
        else if (kind == 1) { tap = dy * 3 + half; part = 1; }
        else if (kind == 2) { tap = dy * 3 + 2; part = 0; }                 // [w_hi | w_hi] x [x_hi | x_lo]
        else { tap = dy * 3 + 2; part = half ? 2 : 1; }                       // [w_lo | 0]    x [x_hi | x_lo]
        const float ws = w[(tap * 16 + cin) * 16 + cout] * s_fold[cout] * sr;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        orow[idx] = part == 0 ? hi : (part == 1 ? lo : (_Float16)0.f);
    }
    float* aux = dst + layer * d_stride + 2 * BF_H3_WPACK_FLOATS;
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        if (which == 0) {
            aux[c] = 1.0f / sr;                       // == 1/s: conv1 folds nothing
        } else {
            float sc = 1.f, sh = 0.f;
            if (ext_scale) {                          // debug entry: caller's scale / shift
                sc = ext_scale[c];
                sh = ext_shift[c];
            } else if (use_bn) {     // keras BatchNormalization(training=False): gamma*(x-mean)*rsqrt(var+eps)
                const float g = params[p_blocks + layer * p_stride + 4608 + c];
                const float mean = state[layer * 32 + c], var = state[layer * 32 + 16 + c];
                sc = g / sqrtf(var + eps);
                sh = -sc * mean;
            }
            aux[16 + c] = sc * (1.0f / s);            // group-per-pass kernel: scale in the epilogue
            aux[32 + c] = sh;
            aux[48 + c] = 1.0f / sr;                  // row-streaming kernel: scale folded into the weights
        }
    }
}

hipError_t bf_launch_pack_h3(const float* params, const float* state, int64_t p_blocks, int64_t p_stride, float* dst,
                             int64_t d_stride, int layers, int use_bn, float eps, const float* ext_scale,
                             const float* ext_shift, hipStream_t s)
{
    if (layers <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_h3_kernel, dim3(layers * 2), dim3(256), 0, s, params, state, p_blocks, p_stride, dst, d_stride,
                       use_bn, eps, ext_scale, ext_shift);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// fp32 NHWC [B,H,W,16]  <->  split-planar [B][4][H][W][8 x f16]   (tests, debug entry points)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h3_from_f32_kernel(const float* __restrict__ x, char* __restrict__ y, int64_t npix, int64_t hw)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix * 2; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i >> 1;
        const int half = (int)(i & 1);
        const int64_t b = pix / hw, p = pix - b * hw;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + pix * 16 + half * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + pix * 16 + half * 8 + 4);
        h4 h0, l0, h1, l1;
        h3_split(v0, h0, l0);
        h3_split(v1, h1, l1);
        char* base = y + b * hw * 64 + p * 16;
        h8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        h8 lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        *reinterpret_cast<h8*>(base + (int64_t)half * hw * 16) = hi;
        *reinterpret_cast<h8*>(base + (int64_t)(2 + half) * hw * 16) = lo;
    }
}

__global__ __launch_bounds__(256) void h3_to_f32_kernel(const char* __restrict__ y, float* __restrict__ x, int64_t npix, int64_t hw)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix * 2; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i >> 1;
        const int half = (int)(i & 1);
        const int64_t b = pix / hw, p = pix - b * hw;
        const char* base = y + b * hw * 64 + p * 16;
        const h8 hi = *reinterpret_cast<const h8*>(base + (int64_t)half * hw * 16);
        const h8 lo = *reinterpret_cast<const h8*>(base + (int64_t)(2 + half) * hw * 16);
        float* o = x + pix * 16 + half * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (float)hi[c] + (float)lo[c];
    }
}

hipError_t bf_launch_h3_from_f32(const float* x, void* y, int B, int H, int W, hipStream_t s)
{
    const int64_t npix = (int64_t)B * H * W;
    const int64_t g = (npix * 2 + 255) / 256;
    hipLaunchKernelGGL(h3_from_f32_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, s, x, (char*)y, npix, (int64_t)H * W);
    return hipGetLastError();
}

hipError_t bf_launch_h3_to_f32(const void* y, float* x, int B, int H, int W, hipStream_t s)
{
    const int64_t npix = (int64_t)B * H * W;
    const int64_t g = (npix * 2 + 255) / 256;
    hipLaunchKernelGGL(h3_to_f32_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(256), 0, s, (const char*)y, x, npix, (int64_t)H * W);
    return hipGetLastError();
}
