// Device helpers shared by the split-f16 ("f16x3") kernels: fused_h3.hip (tile kernels, training convolutions) and
// fused_h3v.hip (row-streaming full-width kernel).  Arithmetic and layout: see the header of fused_h3.hip.
#pragma once
#include "bf_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// timing-only ablations of fused_block_h3r_kernel (tools/ablate.sh; results are WRONG when any is set):
// 1 = no next-tile DMA, 2 = no global stores, 4 = no conv2 MFMA work, 8 = no conv1 MFMA work, 16 = no barriers,
// 64 = no epilogue arithmetic (raw accumulator bits are stored)
// 32 = s_memtime stamps per phase (diagnostic build; per-wave sums go to args.dbg, tools/stamp_h3.py)
#ifndef H3_ABLATE
#define H3_ABLATE 0
#endif
#if H3_ABLATE & 32
#define H3_STAMP(k)                                                                                      \
    do {                                                                                                 \
        unsigned long long now_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        stamp_sum[k] += now_ - stamp_prev;                                                               \
        stamp_prev = now_;                                                                               \
    } while (0)
#else
#define H3_STAMP(k) do { } while (0)
#endif

// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]).  The waits go through
// the builtin, not inline asm, so that hipcc's own waitcnt bookkeeping sees them: with an asm wait in the prologue
// it believed the weight / scale loads issued before the tile loop were still pending at the loop header and put a
// vmcnt(0) in front of the first MFMA of conv1 AND conv2 of every tile -- which drained the tile DMA right after
// it had been issued (385 us per launch instead of the numbers in DESIGN.md).
constexpr int h3_vmcnt(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }
constexpr int H3_LGKMCNT0 = 0xC07F;

// workgroup barrier that publishes LDS writes but leaves vector-memory operations (the tile DMA) in flight
// (s_barrier stays inline asm: hipcc puts "s_waitcnt vmcnt(0) lgkmcnt(0)" in front of every barrier it can see)
__device__ __forceinline__ void h3_barrier()
{
    __builtin_amdgcn_s_waitcnt(H3_LGKMCNT0);
#if H3_ABLATE & 16
    asm volatile("" ::: "memory");
#else
    asm volatile("s_barrier" ::: "memory");
#endif
}

// v - float(one half of the packed f16 pair hh) in ONE instruction (hipcc never selects v_fma_mix_f32 for this)
__device__ __forceinline__ float h3_sub_half(const float v, const unsigned hh, const bool high)
{
    float r;
    if (high) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hh), "v"(v));
    return r;
}

// hi = f16(v) (round-to-nearest-even), lo = f16(v - hi): 2 x v_cvt_pk + 4 x v_fma_mix + 2 x v_cvt_pk
__device__ __forceinline__ void h3_split(const f32x4 v, h4& hi, h4& lo)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#if H3_ABLATE & 64
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(h4, (f32x2){v[0], v[1]});           // timing only: no conversion VALU at all
    lo = __builtin_bit_cast(h4, (f32x2){v[2], v[3]});
    return;
#endif
    hi = __builtin_convertvector(v, h4);
    const unsigned a = __builtin_bit_cast(unsigned, (h2){hi[0], hi[1]}), b = __builtin_bit_cast(unsigned, (h2){hi[2], hi[3]});
    const f32x4 d = {h3_sub_half(v[0], a, false), h3_sub_half(v[1], a, true), h3_sub_half(v[2], b, false), h3_sub_half(v[3], b, true)};
    lo = __builtin_convertvector(d, h4);
}

// NOTE on `interior` shortcuts in the conv2 epilogues: `if (!interior && out_of_image) p = dump` made hipcc branch over the
// select on the uniform `interior`, and on the taken path its hazard recognizer left ONE wait state between the last
// MFMA of a row and the v_pk_fma that reads the accumulator: stale .zw halves on interior tiles of the wave-specialised
// kernel (caught by the parity tests).  The out-of-image select is therefore unconditional (per-lane condition, no branch).
//
// hi / lo of the lane's four channels, then a row exchange (v_permlane16_swap_b32: result 0 = [a.row0, b.row0, a.row2,
// b.row2], result 1 = [a.row1, b.row1, a.row3, b.row3], rows = 16-lane groups = q) so that every lane ends up with ONE
// 16-byte record of EIGHT channels: q = 0 -> hi(c0..7), q = 1 -> lo(c0..7), q = 2 -> hi(c8..15), q = 3 -> lo(c8..15) of its
// pixel, i.e. plane (q >> 1) + 2 * (q & 1).  One 16-byte store (ds_write_b128 / global dwordx4) per group instead of
// two 8-byte ones: half the LDS-write and vector-memory instructions of the epilogues.  Needs EXEC all ones.
__device__ __forceinline__ h8 h3_split_record(const f32x4 v)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    h4 hi, lo;
    h3_split(v, hi, lo);
    u2 H = __builtin_bit_cast(u2, hi), L = __builtin_bit_cast(u2, lo);
    // inline asm with explicit wait states on both sides: with the builtin, hipcc 7.2 issued the swap in the slot right
    // after the v_cvt_pk that produces its operand (and the store right after the swap) in the tightest code paths and
    // the wave-specialised kernel then stored stale halves on interior tiles (parity tests) -- a data hazard of this
    // new gfx950 instruction the compiler does not pad
    unsigned h0 = H[0], l0 = L[0], h1 = H[1], l1 = L[1];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(h0), "+v"(l0), "+v"(h1), "+v"(l1));
    return __builtin_bit_cast(h8, (u4){h0, h1, l0, l1});
}

