// Row-streaming inner loop of the split-f16 3x3 16->16 convolutions (fused_block_h3r_kernel, the training convolutions of
// fused_h3.hip and the fused backward kernel of train_bwd_h3.hip): a wave owns a run of R consecutive rows of one 16-column
// strip of an LDS image laid out as hi / lo planes [row][col][8 x f16].  See the comment block in fused_h3.hip.
#pragma once
#include "h3_core.h"

template <int PITCH, int LO>
struct H3RowFrag {
    // ph / pl: taps (dy,0)|(dy,1) from the hi / lo planes.  s: tap (dy,2) as [x_hi | x_lo] -- lanes q < 2 read the hi plane,
    // q >= 2 the lo plane of the SAME pixel (the caller's vs carries the + LO of the upper lane half): one read where
    // [x_hi | x_hi] and [x_lo | x_lo] took two, and an LDS read costs a wave ~16 issue cycles it cannot spend on MFMAs
    h8 ph, pl, s;
    __device__ __forceinline__ void load(const char* __restrict__ src, const int vp, const int vs, const int row)
    {
        ph = *reinterpret_cast<const h8*>(src + vp + row * PITCH);
        pl = *reinterpret_cast<const h8*>(src + vp + row * PITCH + LO);
        s = *reinterpret_cast<const h8*>(src + vs + row * PITCH);
    }
};

// the five MFMAs of one (input row, dy) pair; w: [dy*4 + {pair hi, pair lo, single [hi|hi], single [lo|0]}]
template <int PITCH, int LO>
__device__ __forceinline__ f32x4 h3r_tap_row(const H3RowFrag<PITCH, LO>& x, const h8 (&w)[13], const int dy, f32x4 acc)
{
    // ablations 4 / 8: conv2 reads the intermediate tile (PITCH = MW*16 is not a multiple of 64 for TW = 32), conv1 the input tile
    if (((H3_ABLATE & 4) && PITCH % 64 != 0) || ((H3_ABLATE & 8) && PITCH % 64 == 0)) {
        acc[0] += (float)x.ph[0] + (float)x.pl[1] + (float)x.s[2];      // keeps the LDS reads live
        return acc;
    }
    acc = MFMA_H(w[dy * 4 + 0], x.ph, acc);
    acc = MFMA_H(w[dy * 4 + 1], x.ph, acc);
    acc = MFMA_H(w[dy * 4 + 0], x.pl, acc);
    acc = MFMA_H(w[dy * 4 + 2], x.s, acc);
    acc = MFMA_H(w[dy * 4 + 3], x.s, acc);
    return acc;
}

// MFMA m (0..4) of the five of one (input row, dy) pair
template <int PITCH, int LO>
__device__ __forceinline__ f32x4 h3r_tap_mfma(const H3RowFrag<PITCH, LO>& x, const h8 (&w)[13], const int dy, const int m, f32x4 acc)
{
    if (((H3_ABLATE & 4) && PITCH % 64 != 0) || ((H3_ABLATE & 8) && PITCH % 64 == 0)) {
        if (m == 0) acc[0] += (float)x.ph[0] + (float)x.pl[1] + (float)x.s[2];
        return acc;
    }
    switch (m) {
        case 0: return MFMA_H(w[dy * 4 + 0], x.ph, acc);
        case 1: return MFMA_H(w[dy * 4 + 1], x.ph, acc);
        case 2: return MFMA_H(w[dy * 4 + 0], x.pl, acc);
        case 3: return MFMA_H(w[dy * 4 + 2], x.s, acc);
        default: return MFMA_H(w[dy * 4 + 3], x.s, acc);
    }
}

// R output rows of one strip: vp / vs = lane's LDS byte address of input row 0 for the pair / single fragments
struct H3NoHook {
    template <int I> __device__ __forceinline__ void row() const {}
};

// One input row I (= 0 .. R+1) of a run.  Issue order, pinned with a scheduling barrier because hipcc otherwise sinks
// the prefetch below the MFMAs and every row step then eats a full LDS round trip (tools/ablate.sh 76: with NO MFMA
// and NO epilogue arithmetic the kernel still took 216 us of 320 -- it was LDS-latency bound, 16 row steps per tile):
//   1. ds_reads of row I+1 (+ whatever the epilogue wants early: epi.pre(O) for output row O = I-2)
//   2. the epilogue of output row I-3 (finished in the previous step) interleaved with the 15 MFMAs of output rows
//      I-2 (which completes: + epi.finish), I-1 and I
//   3. hook.row<I>()
template <int R, int PITCH, int LO, int I, class Epi, class Hook>
__device__ __forceinline__ void h3r_rows_step(const char* __restrict__ src, const int vp, const int vs, const h8 (&w)[13],
                                              const f32x4 acc_done, f32x4 acc_m2, f32x4 acc_m1, const H3RowFrag<PITCH, LO> cur,
                                              const H3RowFrag<PITCH, LO> nxt, const typename Epi::Pre pre_m2, const Epi& epi,
                                              const Hook& hook)
{
    // acc_done / acc_m2 / acc_m1: accumulators of output rows I-3 (complete) / I-2 / I-1 (rolling values, not an array:
    // hipcc left a 5-row accumulator array in scratch memory once the epilogue call moved, and scratch traffic drains
    // the tile DMA)
    if constexpr (I < R + 2) {
        // prefetch distance 2: the fragments of row I+2 are requested while rows I and I+1 are already in registers / in
        // flight.  With distance 1 the run's first and last steps (5 and 10 MFMAs) were LDS-latency bound: 7 steps of
        // ~480 cycles for 75 MFMAs (stamps), 480 = what 2 x 15 MFMAs of the two waves of a SIMD need in steady state.
        H3RowFrag<PITCH, LO> nx2;
        if (I + 2 < R + 2) nx2.load(src, vp, vs, I + 2);
        typename Epi::Pre pre_m1 = {};
        if constexpr (I >= 1 && I - 1 < R) pre_m1 = epi.pre(I - 1);   // consumed in the NEXT step: a full row of MFMAs away
        __builtin_amdgcn_sched_barrier(0);
#if !(H3_ABLATE & 2048)
        // epilogue of the row that finished ONE STEP AGO: its arithmetic, LDS write / global store are independent of this
        // step's MFMAs, so they can issue in the MFMAs' shadows (8 of every 16 cycles of the vector issue port are free)
        // instead of between two MFMA bursts; no hazard padding needed either (the accumulator is a whole step old)
        if constexpr (I >= 3) epi(I - 3, acc_done);
#endif
        f32x4 acc_0 = {0.f, 0.f, 0.f, 0.f};
        // the three accumulators round-robin: consecutive MFMAs are independent (a chain on one accumulator only
        // issues back to back when hipcc happens to keep vDst == SrcC)
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            if constexpr (I >= 2) acc_m2 = h3r_tap_mfma<PITCH, LO>(cur, w, 2, m, acc_m2);
            if constexpr (I >= 1 && I - 1 < R) acc_m1 = h3r_tap_mfma<PITCH, LO>(cur, w, 1, m, acc_m1);
            if constexpr (I < R) acc_0 = h3r_tap_mfma<PITCH, LO>(cur, w, 0, m, acc_0);
        }
        if constexpr (I >= 2) acc_m2 = epi.finish(I - 2, acc_m2, pre_m2);     // conv2: + residual MFMA
#if H3_ABLATE & 2048
        if constexpr (I >= 2) epi(I - 2, bf_acc_ready(acc_m2));
#endif
        hook.template row<I>();
#if !(H3_ABLATE & 128) && !(H3_ABLATE & 2048)
        // one vector instruction in the shadow of every MFMA
        if constexpr (I >= 3) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
#endif
        h3r_rows_step<R, PITCH, LO, I + 1>(src, vp, vs, w, acc_m2, acc_m1, acc_0, nxt, nx2, pre_m1, epi, hook);
    } else {
#if !(H3_ABLATE & 2048)
        epi(R - 1, bf_acc_ready(acc_done));                     // last row: nothing left to hide it under
#endif
    }
}

template <int R, int PITCH, int LO, class Epi, class Hook>
__device__ __forceinline__ void h3r_rows(const char* __restrict__ src, const int vp, const int vs, const h8 (&w)[13],
                                         const Epi& epi, const Hook& hook)
{
    H3RowFrag<PITCH, LO> cur, nxt;
    cur.load(src, vp, vs, 0);
    nxt.load(src, vp, vs, 1);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    h3r_rows_step<R, PITCH, LO, 0>(src, vp, vs, w, z, z, z, cur, nxt, typename Epi::Pre{}, epi, hook);
}

// one 16-pixel group of arbitrary shape (the 2-column strip groups): 12 reads, 15 MFMAs, no row reuse
template <int PITCH, int LO>
__device__ __forceinline__ f32x4 h3r_group(const char* __restrict__ src, const int vp, const int vs, const h8 (&w)[13])
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        H3RowFrag<PITCH, LO> x;
        x.load(src, vp, vs, dy);
        acc = h3r_tap_row<PITCH, LO>(x, w, dy, acc);
    }
    return acc;
}
