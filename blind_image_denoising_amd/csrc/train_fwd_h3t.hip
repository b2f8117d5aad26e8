// Training-mode forward of one residual block in ONE kernel (fwd_block_h3t_kernel), split-f16 arithmetic, fp32 NHWC tensors,
// full-row streaming schedule of fused_block_h3v_kernel (fused_h3v.hip).
//
// Reference: one iteration of the block loop of bfcnn/backbone_blocks.py:167-246 under training=True (train_loop.py:249-251, 277):
//     A_i = A_{i-1} + bn(C_{i-1})                  (the Add + BatchNormalization of the block in FRONT, formed on load)
//     T_i = relu(conv_0 A_i)                        (LDS only -- or also written, for a backward pass that reads it)
//     C_i = conv_1 T_i                              (raw output in front of this block's BatchNormalization)
//     sum C_i, sum C_i^2 per channel                (the batch statistics of that BatchNormalization)
// As two kernels (conv3x3_h3_kernel<.., PRE> then conv3x3_h3_kernel<EPI_STATS>) that is 6 tensor passes -- A_{i-1}, C_{i-1} read,
// A_i, T_i written, T_i read, C_i written -- at the ~5 TB/s the chip sustains on mixed streams; here it is 4 (5 with T_i written).
//
// A workgroup owns ALL columns of an image (W <= 256) and walks down a band of rows, one image row per step; twelve waves:
//   * waves 0..3 ("A") conv_0 + ReLU: input ring row s -> three vertical-tap contributions, intermediate row s-2 completes
//     and goes to the intermediate ring (split f16, scaled by 1/s, zero outside the image: it is conv_1's padding);
//   * waves 4..7 ("B") conv_1: intermediate row s-3 -> output rows s-3, s-4, s-5; row s-5 completes: scaled, added to the
//     wave's per-channel sums, written as fp32 to the staging ring;
//   * waves 8, 9 (loaders) read row s+2 of A_{i-1} [and C_{i-1}] into REGISTERS (fp32 cannot go through LDS-DMA: it has to be
//     split), and turn the row they requested a step earlier into hi / lo f16 planes of input ring row s+1
//     (y = x + (scale c + shift) first, rounded as affine_add_kernel rounds it); they never store to global memory, so their
//     vector-memory queue holds loads only and completes in order (fused_h3v.hip on why loaders and storers are different waves);
//   * waves 10, 11 (storers) write C_i row s-6 from the staging ring, A_i row s-1 re-assembled from the input ring (hi + lo:
//     the 22 bits the convolution itself saw, so a backward pass that recomputes T_i from the stored A_i gets the SAME T_i bit
//     for bit) and, on request, T_i row s-3 from the intermediate ring.
// ONE barrier per step, nrows + 6 steps per band.
#include "bf_common.h"
#include "h3_core.h"
#include "h3v_core.h"

struct H3TGeom {
    static constexpr int WMAX = 256;                   // columns a workgroup covers (whole image rows)
    static constexpr int G = 4;                        // 16-column groups per matrix wave
    static constexpr int NW = 12, NT = 768;
    static constexpr int PITCH = (WMAX + 2) * 16;      // bytes per plane-row of the input / intermediate rings; ring column = image column + 1
    static constexpr int NRI = 3, NRM = 2, NRO = 2;    // ring depths (rows): input rows s-1 (A_i store), s (conv_0), s+1 (being written)
    static constexpr int UNROLL = 6;
    // plane stride of the input ring = 128 mod 256: the 8-byte records the loaders write / the storers read touch planes 0 and 1 of
    // 8 neighbouring pixels per half-wave
    static constexpr int IN_PLANE = (NRI * PITCH + 255) / 256 * 256 + 128;
    static constexpr int MID_PLANE = (NRM * PITCH + 255) / 256 * 256;
    // staging ring: fp32, per slot four channel-quad planes of [WMAX] 16-byte records, plane stride 64 mod 256 (B writes 16
    // consecutive pixels of one quad, the storers read the four quads of 4 consecutive pixels: both conflict-free)
    static constexpr int OUT_QUAD = WMAX * 16 + 64, OUT_SLOT = 4 * OUT_QUAD;
    static constexpr int IN_BYTES = 4 * IN_PLANE, MID_BYTES = 4 * MID_PLANE, OUT_BYTES = NRO * OUT_SLOT;
    static constexpr int LDS_BYTES = IN_BYTES + MID_BYTES + OUT_BYTES;
    static constexpr int NJ = WMAX * 4 / 128;          // 16-byte elements of a row per lane of a two-wave team: 8
    static_assert(UNROLL % NRM == 0 && UNROLL % NRO == 0 && UNROLL % NRI == 0 && UNROLL % 3 == 0 && UNROLL % 2 == 0, "static slots");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(3 * IN_PLANE + NRI * PITCH < 65536 && 3 * MID_PLANE + NRM * PITCH < 65536, "fragment offsets fit the 16-bit ds offset");
};

struct H3TTile {
    int y0, nrows;
    size_t img;                  // byte offset of the image in an fp32 NHWC tensor of 16 channels (64 bytes per pixel)
    int ybase, ystep;            // image row of band-relative row k: ybase + ystep * k (a reversed band walks bottom-up)
    __device__ __forceinline__ int y(const int k) const { return ybase + ystep * k; }
};

#ifndef H3T_XCD_ORDER
#define H3T_XCD_ORDER 1
#endif
__device__ __forceinline__ H3TTile h3t_tile(const FwdBlockH3Args& a, const int t)
{
    H3TTile r;
    // workgroup ids go round the 8 XCDs: tile t of the launch order becomes tile (t mod 8) * ntiles / 8 + t / 8, so that an XCD walks a
    // contiguous eighth of the bands and the halo rows two vertically adjacent bands both read are found in ITS L2 (H3T_XCD_ORDER 0:
    // neighbours on different XCDs, every halo row fetched twice from the Infinity Cache / HBM: 626 MB per launch for 537 MB algorithmic)
    const int tp = (H3T_XCD_ORDER && (a.ntiles & 7) == 0) ? (t & 7) * (a.ntiles >> 3) + (t >> 3) : t;
    const int tt = a.reverse ? a.ntiles - 1 - tp : tp;
    const int b = tt / a.tiles_y, ty = tt - b * a.tiles_y;
    r.y0 = ty * a.rows_per_tile;
    r.nrows = min(a.rows_per_tile, a.H - r.y0);
    r.img = (size_t)b * a.H * a.W * 64;
    r.ybase = a.reverse ? r.y0 + r.nrows - 1 : r.y0;
    r.ystep = a.reverse ? -1 : 1;
    return r;
}

// weight image i = dy * 4 + kind as the code's tap row dy: mirrored for a band that walks bottom-up
__device__ __forceinline__ int h3t_wimage(const FwdBlockH3Args& a, const int i) { return a.reverse ? (2 - i / 4) * 4 + i % 4 : i; }

// the 15 MFMAs of one 16-pixel group and step: vertical taps 2 / 1 / 0 of the ring row in `cur` go to the accumulators of the
// rows that are two / one / zero steps old; micro-ops of the PREVIOUS group's epilogue ride behind MFMA 2, 3, ...
template <int J, class Epi>
__device__ __forceinline__ void h3t_mfmas(const h8 (&w)[13], f32x4& acc2, f32x4& acc1, f32x4& c0, const H3VFrag& cur, Epi* epi)
{
    if constexpr (J < 15) {
        constexpr int k = J / 3, which = J % 3;
        if constexpr (which == 0) acc2 = h3v_mfma(cur, w, 2, k, acc2);
        else if constexpr (which == 1) acc1 = h3v_mfma(cur, w, 1, k, acc1);
        else c0 = h3v_mfma(cur, w, 0, k, c0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (J >= 2) {
            if (epi) epi->template pair<J - 2>();
        }
        h3t_mfmas<J + 1>(w, acc2, acc1, c0, cur, epi);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// role A: conv_0 + activation -> intermediate ring.  acc[g][3]: intermediate rows s, s-1, s-2 (modulo 3).
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3TRoleA {
    using Gm = H3TGeom;
    const FwdBlockH3Args& a;
    const char* tin;
    char* tmid;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs, wr;
    float inv_s, relu_floor;
    float lane_scale[Gm::G];     // !FULLW: inv_s where the lane's column is inside the image, else 0

    __device__ __forceinline__ H3VFrag load(const int islot_bytes, const int g) const
    {
        H3VFrag f;
        const char* p = tin + rp + islot_bytes;
        f.ph = *reinterpret_cast<const h8*>(p + g * 256);
        f.pl = *reinterpret_cast<const h8*>(p + g * 256 + 2 * Gm::IN_PLANE);
        f.s = *reinterpret_cast<const h8*>(tin + rs + islot_bytes + g * 256);
        return f;
    }
    __device__ __forceinline__ H3VEpi<true> epilogue(const int g, const int mslot, const f32x4 v, const bool rowok) const
    {
        H3VEpi<true> e;
        e.v = v;
        const float sc = FULLW ? inv_s : lane_scale[g];
        e.sc = rowok ? sc : 0.f;                         // rows outside the image are conv_1's zero padding
        e.floor_ = relu_floor;
        e.p = tmid + wr + mslot * Gm::PITCH + g * 256;
        e.lo_off = 2 * Gm::MID_PLANE;
        return e;
    }
    // step s (s % UNROLL == PH): input ring row s in slot `islot`
    template <int PH>
    __device__ __forceinline__ void step(const H3TTile& t, const int s, const int islot)
    {
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // accumulators of intermediate rows s, s-1, s-2
        constexpr int mslot = PH % Gm::NRM;                                    // (s - 2) mod 2
        const int m = s - 2, ym = t.y(m - 1);
        const bool rowok = (m >= 0) & (ym >= 0) & (ym < a.H);
        const int ib = islot * Gm::PITCH;
        H3VFrag cur = load(ib, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(ib, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if (g > 0) {
                H3VEpi<true> e = epilogue(g - 1, mslot, acc[g - 1][a2], rowok);
                h3t_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, &e);
            } else {
                h3t_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, (H3VEpi<true>*)nullptr);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        H3VEpi<true> e = epilogue(Gm::G - 1, mslot, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
        e.all();
    }
};

// epilogue of one 16-pixel group of conv_1 as micro-ops: scale, add to the wave's sums, one 16-byte fp32 record to the staging ring
struct H3TEpiOut {
    static constexpr int NOPS = 13;
    f32x4 v, s1, s2;
    float sc;
    char* p;
    template <int I> __device__ __forceinline__ void op()
    {
        if constexpr (I < 4) v[I] *= sc;
        else if constexpr (I < 8) s1[I - 4] += v[I - 4];
        else if constexpr (I < 12) s2[I - 8] = fmaf(v[I - 8], v[I - 8], s2[I - 8]);
        else *reinterpret_cast<f32x4*>(p) = v;
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int SLOT> __device__ __forceinline__ void pair()
    {
        if constexpr (2 * SLOT < NOPS) op<2 * SLOT>();
        if constexpr (2 * SLOT + 1 < NOPS) op<2 * SLOT + 1>();
    }
    template <int I = 0> __device__ __forceinline__ void all()
    {
        if constexpr (I < NOPS) {
            op<I>();
            all<I + 1>();
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// role B: conv_1 -> staging ring (fp32) + the per-channel sums of the block's BatchNormalization
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3TRoleB {
    using Gm = H3TGeom;
    const FwdBlockH3Args& a;
    const char* tmid;
    char* tout;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs, wo;
    float inv_s;
    float lane_scale[Gm::G];
    f32x4 s1, s2;                // sum / sum of squares of this lane's four channels over the pixels it has finished

    __device__ __forceinline__ H3VFrag load(const int slot, const int g) const
    {
        H3VFrag f;
        const int o = slot * Gm::PITCH + g * 256;
        f.ph = *reinterpret_cast<const h8*>(tmid + rp + o);
        f.pl = *reinterpret_cast<const h8*>(tmid + rp + o + 2 * Gm::MID_PLANE);
        f.s = *reinterpret_cast<const h8*>(tmid + rs + o);
        return f;
    }
    __device__ __forceinline__ H3TEpiOut epilogue(const int g, const int oslot, const f32x4 v, const bool rowok) const
    {
        H3TEpiOut e;
        e.v = v;
        e.s1 = s1;
        e.s2 = s2;
        const float sc = FULLW ? inv_s : lane_scale[g];
        e.sc = rowok ? sc : 0.f;                         // rows / columns outside the band add nothing to the sums (and are not stored)
        e.p = tout + wo + oslot * Gm::OUT_SLOT + g * 256;
        return e;
    }
    // step s (s % UNROLL == PH): intermediate ring row s-3, completes output row s-5
    template <int PH>
    __device__ __forceinline__ void step(const H3TTile& t, const int s)
    {
        constexpr int mslot = (PH + 1) % Gm::NRM;                              // (s - 3) mod 2
        constexpr int oslot = (PH + 1) % Gm::NRO;                              // (s - 5) mod 2
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // output rows s-3, s-4, s-5
        const int o = s - 5;
        const bool rowok = (o >= 0) & (o < t.nrows);
        H3VFrag cur = load(mslot, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(mslot, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if (g > 0) {
                H3TEpiOut e = epilogue(g - 1, oslot, acc[g - 1][a2], rowok);
                h3t_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, &e);
                s1 = e.s1;
                s2 = e.s2;
            } else {
                h3t_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, (H3TEpiOut*)nullptr);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        H3TEpiOut e = epilogue(Gm::G - 1, oslot, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
        e.all();
        s1 = e.s1;
        s2 = e.s2;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// loaders (waves 8, 9): a row is 1 024 sixteen-byte elements (256 pixels x 4 channel quads); lane l of loader wave lw owns the
// elements e = 128 j + 64 lw + l, j = 0..7: consecutive lanes read consecutive 16 bytes (1 KiB per wave-instruction), every
// element of a lane has the same channel quad l & 3.
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW, bool PRE>
struct H3TLoader {
    using Gm = H3TGeom;
    const FwdBlockH3Args& a;
    char* tin;
    int e0;                      // 64 lw + lane
    f32x4 psc, psh;
    // ONE row of registers: element j of the row requested a step ago is converted and its registers are re-loaded with element j
    // of the next row at once (two row buffers -- 128 registers with the BatchNorm operand -- spilled)
    f32x4 x[Gm::NJ], c[PRE ? Gm::NJ : 1];
    bool ok;                     // the row in the registers lies inside the image (wave-uniform)

    __device__ __forceinline__ bool row_ok(const H3TTile& t, const int r) const
    {
        const int y = t.y(r - 2);
        return (y >= 0) & (y < a.H) & (r < t.nrows + 4);        // rows outside the image / past the rows conv_0 needs: zeros, no loads
    }
    // UNCONDITIONAL loads (a branch around a load makes hipcc wait for vmcnt(0) at the join, which drains the prefetch): a row
    // outside the image / a column past its width reads a clamped address and is zeroed when it is converted
    __device__ __forceinline__ void load(const H3TTile& t, const int r, const int j)
    {
        const int y = min(max(t.y(r - 2), 0), a.H - 1);
        const int e = 128 * j + e0;
        const int px = FULLW ? (e >> 2) : min(e >> 2, a.W - 1);
        const size_t off = t.img + ((size_t)y * a.W + px) * 64 + (e & 3) * 16;
        x[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.x) + off);
        if (PRE) c[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.pre_c) + off);
    }
    // ring row r (image row y(r - 2)) into the registers
    __device__ __forceinline__ void request(const H3TTile& t, const int r)
    {
        ok = row_ok(t, r);
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) load(t, r, j);
    }
    // the row in the registers -> hi / lo planes of ring slot `slot`; with NEXT: ring row rn takes its place element by element
    template <bool NEXT>
    __device__ __forceinline__ void commit(const int slot, const H3TTile& t, const int rn)
    {
        const int quad = e0 & 3;
        char* base = tin + (quad >> 1) * Gm::IN_PLANE + slot * Gm::PITCH + 16 + (quad & 1) * 8;
        const bool ok_next = NEXT ? row_ok(t, rn) : false;
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) {
            const int px = (128 * j + e0) >> 2;
            f32x4 v;
            // y = x + (scale * c + shift) as affine_add_kernel rounds it; 0 outside the image (SAME padding of conv_0)
            const bool in = ok && (FULLW || px < a.W);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = in ? (PRE ? x[j][k] + fmaf(psc[k], c[j][k], psh[k]) : x[j][k]) : 0.f;
            if (NEXT) {
                __builtin_amdgcn_sched_barrier(0);               // the next row's loads leave as soon as the registers are free
                load(t, rn, j);
                __builtin_amdgcn_sched_barrier(0);
            }
            h4 hi, lo;
            h3_split(v, hi, lo);
            *reinterpret_cast<h4*>(base + px * 16) = hi;
            *reinterpret_cast<h4*>(base + px * 16 + 2 * Gm::IN_PLANE) = lo;
        }
        if (NEXT) ok = ok_next;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// storers (waves 10, 11): same element ownership as the loaders
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3TStorer {
    using Gm = H3TGeom;
    const FwdBlockH3Args& a;
    const char* tin;
    const char* tmid;
    const char* tout;
    int e0;

    // fp32 row of the staging ring -> C_i
    __device__ __forceinline__ void store_c(const H3TTile& t, const int o, const int oslot) const
    {
        if (!((o >= 0) & (o < t.nrows))) return;                                      // wave-uniform
        char* dst = reinterpret_cast<char*>(a.c_out) + t.img + (size_t)t.y(o) * a.W * 64;
        const int quad = e0 & 3;
        f32x4 v[Gm::NJ];
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) {
            const int px = (128 * j + e0) >> 2;
            v[j] = *reinterpret_cast<const f32x4*>(tout + oslot * Gm::OUT_SLOT + quad * Gm::OUT_QUAD + px * 16);
        }
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) {
            const int e = 128 * j + e0;
            if (FULLW || (e >> 2) < a.W) *reinterpret_cast<f32x4*>(dst + (size_t)e * 16) = v[j];
        }
    }
    // hi + lo of a split-f16 ring row -> fp32 row k of the band in `out` (A_i from the input ring, T_i from the intermediate ring)
    __device__ __forceinline__ void store_split(const H3TTile& t, const int k, const char* ring, const int plane, const int slot_off,
                                                float* out) const
    {
        if (!((k >= 0) & (k < t.nrows))) return;                                      // wave-uniform
        char* dst = reinterpret_cast<char*>(out) + t.img + (size_t)t.y(k) * a.W * 64;
        const int quad = e0 & 3;
        const char* base = ring + (quad >> 1) * plane + slot_off + 16 + (quad & 1) * 8;
        h4 hi[Gm::NJ], lo[Gm::NJ];
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) {
            const int px = (128 * j + e0) >> 2;
            hi[j] = *reinterpret_cast<const h4*>(base + px * 16);
            lo[j] = *reinterpret_cast<const h4*>(base + px * 16 + 2 * plane);
        }
#pragma unroll
        for (int j = 0; j < Gm::NJ; ++j) {
            const int e = 128 * j + e0;
            const f32x4 v = __builtin_convertvector(hi[j], f32x4) + __builtin_convertvector(lo[j], f32x4);
            if (FULLW || (e >> 2) < a.W) *reinterpret_cast<f32x4*>(dst + (size_t)e * 16) = v;
        }
    }
};

template <bool FULLW, bool PRE, bool WRITE_T>
__global__ __launch_bounds__(H3TGeom::NT, 3) void fwd_block_h3t_kernel(FwdBlockH3Args a)
{
    using Gm = H3TGeom;
    extern __shared__ __attribute__((aligned(16))) char h3t_lds[];
    char* tin = h3t_lds;                                       // [4 planes][NRI rows][WMAX + 2 columns][8 f16]
    char* tmid = h3t_lds + Gm::IN_BYTES;                       // [4 planes][NRM rows][WMAX + 2 columns][8 f16]
    char* tout = tmid + Gm::MID_BYTES;                         // [NRO rows][4 quads][WMAX columns][4 fp32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int role = wave >> 2, rw = wave & 3;

    // The BatchNorm of the block in front, finalised here (FwdBlockH3Args::fin_partial): the arithmetic of bn_finalize_kernel (fp64, fixed
    // order: 24 stripes of rows, then the stripes) in every workgroup -- they all arrive at the same scale / shift --, the results and the
    // moving statistics written by workgroup 0.  The LDS it uses is cleared right below.
    f32x4 fin_sc = {0.f, 0.f, 0.f, 0.f}, fin_sh = {0.f, 0.f, 0.f, 0.f};
    if (PRE && a.fin_partial) {
        double* red = reinterpret_cast<double*>(h3t_lds);       // [24][32]
        float* fin = reinterpret_cast<float*>(h3t_lds + 24 * 32 * 8);
        {
            const int ch = tid & 31, stripe = tid >> 5;
            double sum = 0.0;
            for (int r = stripe; r < a.fin_nblk; r += Gm::NT / 32) sum += (double)a.fin_partial[(size_t)r * 32 + ch];
            red[stripe * 32 + ch] = sum;
        }
        __syncthreads();
        if (tid < 16) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < Gm::NT / 32; ++k) { s1 += red[k * 32 + tid]; s2 += red[k * 32 + 16 + tid]; }
            const double count = a.fin_count;
            const double mean = s1 / count;
            double var = s2 / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + (double)a.fin_eps);
            const double g = a.fin_gamma ? (double)a.fin_gamma[tid] : 1.0;
            fin[tid] = (float)(g * inv);
            fin[16 + tid] = (float)(-g * inv * mean);
            if (blockIdx.x == 0) {
                a.fin_scale[tid] = fin[tid];
                a.fin_scale[16 + tid] = fin[16 + tid];
                a.fin_meaninv[tid] = (float)mean;
                a.fin_meaninv[16 + tid] = (float)inv;
                const double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0)), mo = (double)a.fin_momentum;
                a.fin_mm[tid] = (float)((double)a.fin_mm[tid] * mo + mean * (1.0 - mo));
                a.fin_mv[tid] = (float)((double)a.fin_mv[tid] * mo + unbiased * (1.0 - mo));
            }
        }
        __syncthreads();
        fin_sc = *reinterpret_cast<const f32x4*>(fin + (lane & 3) * 4);
        fin_sh = *reinterpret_cast<const f32x4*>(fin + 16 + (lane & 3) * 4);
        __syncthreads();
    }
    // ring columns 0 and W+1.. of the input and intermediate rings are the zero padding: cleared once, never written
    for (int i = tid * 16; i < Gm::LDS_BYTES; i += Gm::NT * 16) *reinterpret_cast<f32x4*>(h3t_lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int c0 = 64 * rw + n;                                 // matrix waves: lane's image column in group 0
    if (role == 0) {
        __builtin_amdgcn_s_setprio(1);
        H3TRoleA<FULLW> A{a, tin, tmid};
#pragma unroll
        for (int i = 0; i < 12; ++i) A.w[i] = reinterpret_cast<const h8*>(a.wpack0)[h3t_wimage(a, i) * 64 + lane];
        A.w[12] = A.w[0];
        A.inv_s = a.wpack0[BF_H3R_WPACK_FLOATS];
        A.relu_floor = a.act_relu ? 0.f : -__builtin_inff();
        A.rp = (q & 1) * Gm::IN_PLANE + (c0 + (q >> 1)) * 16;
        A.rs = ((q & 1) + 2 * (q >> 1)) * Gm::IN_PLANE + (c0 + 2) * 16;
        A.wr = (q >> 1) * Gm::MID_PLANE + (c0 + 1) * 16 + (q & 1) * 8;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) A.lane_scale[g] = (c0 + 16 * g < a.W) ? A.inv_s : 0.f;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) A.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // weight / scale loads

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3TTile t = h3t_tile(a, ti);
            h3_barrier();                                        // prologue: input row 0 is in slot 0
            const int nsteps = t.nrows + 6;
            int islot = 0;                                       // s mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3T_STEP_A(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if (s < t.nrows + 4) A.template step<PH>(t, s, islot);    /* input rows 0 .. nrows+3 */   \
                    islot = h3v_wrap(islot + 1, Gm::NRI);                                                     \
                    h3_barrier();                                                                             \
                } while (0)
                H3T_STEP_A(0); H3T_STEP_A(1); H3T_STEP_A(2); H3T_STEP_A(3); H3T_STEP_A(4); H3T_STEP_A(5);
#undef H3T_STEP_A
            }
            h3_barrier();
        }
        h3_barrier();                                            // (the sums of the B waves)
        h3_barrier();
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(1);
        H3TRoleB<FULLW> Bv{a, tmid, tout};
#pragma unroll
        for (int i = 0; i < 12; ++i) Bv.w[i] = reinterpret_cast<const h8*>(a.wpack1)[h3t_wimage(a, i) * 64 + lane];
        Bv.w[12] = Bv.w[0];
        Bv.inv_s = a.wpack1[BF_H3R_WPACK_FLOATS];
        Bv.rp = (q & 1) * Gm::MID_PLANE + (c0 + (q >> 1)) * 16;
        Bv.rs = ((q & 1) + 2 * (q >> 1)) * Gm::MID_PLANE + (c0 + 2) * 16;
        Bv.wo = q * Gm::OUT_QUAD + c0 * 16;
        Bv.s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        Bv.s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) Bv.lane_scale[g] = (c0 + 16 * g < a.W) ? Bv.inv_s : 0.f;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) Bv.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3TTile t = h3t_tile(a, ti);
            h3_barrier();
            const int nsteps = t.nrows + 6;
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3T_STEP_B(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if ((s >= 3) & (s < t.nrows + 5)) Bv.template step<PH>(t, s);   /* intermediate rows 0 .. nrows+1 */ \
                    h3_barrier();                                                                             \
                } while (0)
                H3T_STEP_B(0); H3T_STEP_B(1); H3T_STEP_B(2); H3T_STEP_B(3); H3T_STEP_B(4); H3T_STEP_B(5);
#undef H3T_STEP_B
            }
            h3_barrier();
        }
        // per-workgroup partial of the batch statistics: over the 16 pixel lanes that share a channel quad, then over the four
        // B waves (fixed order: bitwise reproducible)
        f32x4 s1 = Bv.s1, s2 = Bv.s2;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s1[c] += __shfl_xor(s1[c], m);
                s2[c] += __shfl_xor(s2[c], m);
            }
        }
        float* red = reinterpret_cast<float*>(tout);             // [4 waves][32]; the staging ring is free after the last band
        if (n == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[rw * 32 + q * 4 + c] = s1[c];
                red[rw * 32 + 16 + q * 4 + c] = s2[c];
            }
        }
        h3_barrier();
        if (rw == 0 && lane < 32)
            a.stats[(size_t)blockIdx.x * 32 + lane] = (red[lane] + red[32 + lane]) + (red[64 + lane] + red[96 + lane]);
        h3_barrier();
    } else if (rw < 2) {
        H3TLoader<FULLW, PRE> L{a, tin, 64 * rw + lane};
        L.psc = (f32x4){0.f, 0.f, 0.f, 0.f};
        L.psh = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (PRE && a.fin_partial) {
            L.psc = fin_sc;
            L.psh = fin_sh;
        } else if (PRE) {
            L.psc = *reinterpret_cast<const f32x4*>(a.pre_scale + (lane & 3) * 4);
            L.psh = *reinterpret_cast<const f32x4*>(a.pre_shift + (lane & 3) * 4);
        }
        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3TTile t = h3t_tile(a, ti);
            // prologue: row 0 in slot 0, row 1 in the registers (converted in step 0)
            L.request(t, 0);
            L.template commit<true>(0, t, 1);
            h3_barrier();
            const int nsteps = t.nrows + 6;
            int wslot = 1;                                       // (s + 1) mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#pragma unroll
                for (int ph = 0; ph < Gm::UNROLL; ++ph) {
                    // step s: row s+1 (requested a step ago) -> ring slot (s+1) mod NRI, row s+2 into its registers
                    L.template commit<true>(wslot, t, s0 + ph + 2);
                    wslot = h3v_wrap(wslot + 1, Gm::NRI);
                    h3_barrier();
                }
            }
            h3_barrier();
        }
        h3_barrier();
        h3_barrier();
    } else {
        const H3TStorer<FULLW> S{a, tin, tmid, tout, 64 * (rw - 2) + lane};
        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3TTile t = h3t_tile(a, ti);
            h3_barrier();
            const int nsteps = t.nrows + 6;
            int aslot = Gm::NRI - 1;                             // (s - 1) mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3T_STEP_S(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    S.store_c(t, s - 6, PH % Gm::NRO);           /* staged by B in step s-1: (s - 6) mod 2 */ \
                    if (PRE) S.store_split(t, s - 3, tin, Gm::IN_PLANE, aslot * Gm::PITCH, a.a_out);     /* ring row s-1 = band row s-3 */ \
                    if (WRITE_T) S.store_split(t, s - 4, tmid, Gm::MID_PLANE, ((PH + 1) % Gm::NRM) * Gm::PITCH, a.t_out);  /* intermediate row s-3 = band row s-4 */ \
                    aslot = h3v_wrap(aslot + 1, Gm::NRI);                                                     \
                    h3_barrier();                                                                             \
                } while (0)
                H3T_STEP_S(0); H3T_STEP_S(1); H3T_STEP_S(2); H3T_STEP_S(3); H3T_STEP_S(4); H3T_STEP_S(5);
#undef H3T_STEP_S
            }
            h3_barrier();
        }
        h3_barrier();
        h3_barrier();
    }
}

bool bf_fwd_block_h3t_supports(int H, int W) { return W >= 1 && W <= H3TGeom::WMAX && H >= 1; }

// bands: as the inference kernel (fused_h3v.hip): rows per band such that the slowest CU finishes earliest
static int h3t_rows_per_tile(const int B, const int H, const int cus)
{
    int best = H;
    long best_cost = -1;
    for (int ty = 1; ty <= (H + 7) / 8; ++ty) {
        const int rows = (H + ty - 1) / ty;
        if ((H + rows - 1) / rows != ty) continue;
        const long tiles = (long)B * ty;
        const long cost = ((tiles + cus - 1) / cus) * (rows + 10);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = rows;
        }
    }
    return best;
}

// workgroups (= rows of a.stats) a launch uses
int bf_fwd_block_h3t_grid(int B, int H, int W)
{
    (void)W;
    const int rows = h3t_rows_per_tile(B, H, 256);
    const long tiles = (long)B * ((H + rows - 1) / rows);
    return (int)(tiles < 256 ? tiles : 256);
}

hipError_t bf_launch_fwd_block_h3t(const FwdBlockH3Args& args, hipStream_t s)
{
    using Gm = H3TGeom;
    FwdBlockH3Args a = args;
    if (!bf_fwd_block_h3t_supports(a.H, a.W) || !a.x || !a.c_out || !a.wpack0 || !a.wpack1 || !a.stats) return hipErrorInvalidValue;
    if (a.pre_c && ((!a.fin_partial && (!a.pre_scale || !a.pre_shift)) || !a.a_out || a.a_out == a.x || a.a_out == a.pre_c)) return hipErrorInvalidValue;
    if (a.fin_partial && (!a.pre_c || a.fin_partial == a.stats || a.fin_nblk <= 0 || !(a.fin_count > 0.0) || !a.fin_mm || !a.fin_mv || !a.fin_scale || !a.fin_meaninv))
        return hipErrorInvalidValue;
    if (a.c_out == a.x || a.c_out == a.pre_c || a.t_out == a.x || (a.t_out && a.t_out == a.pre_c)) return hipErrorInvalidValue;
    const int cus = 256;
    a.rows_per_tile = h3t_rows_per_tile(a.B, a.H, cus);
    a.tiles_y = (a.H + a.rows_per_tile - 1) / a.rows_per_tile;
    a.ntiles = a.B * a.tiles_y;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    const bool fullw = a.W == Gm::WMAX, pre = a.pre_c != nullptr, wt = a.t_out != nullptr;
    void (*kernel)(FwdBlockH3Args) = nullptr;
#define BF_PICK(F, P, T) if (fullw == F && pre == P && wt == T) kernel = fwd_block_h3t_kernel<F, P, T>;
    BF_PICK(true, true, true) BF_PICK(true, true, false) BF_PICK(true, false, true) BF_PICK(true, false, false)
    BF_PICK(false, true, true) BF_PICK(false, true, false) BF_PICK(false, false, true) BF_PICK(false, false, false)
#undef BF_PICK
    const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(kernel), Gm::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Gm::NT), Gm::LDS_BYTES, s, a);
    return hipGetLastError();
}
