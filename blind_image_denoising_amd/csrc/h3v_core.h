// Pieces shared by the full-row streaming kernels: fused_block_h3v_kernel (fused_h3v.hip, one residual block per launch) and
// fused_block2_h3w_kernel (fused_h3w.hip, two residual blocks per launch on 128-column strips).
#pragma once
#include "bf_common.h"
#include "h3_core.h"

// timing-only ablations (tools/ablate_unit.sh fused_h3v H3V_ABLATE ...; results are WRONG when any is set):
// 1 = no DMA, 2 = no global stores, 4 = no conv2 MFMAs, 8 = no conv1 MFMAs, 16 = no per-step barrier,
// 32 = s_memtime stamps (per-wave sums to args.dbg, tools/stamp_h3v.py), 64 = no epilogue arithmetic
#ifndef H3V_ABLATE
#define H3V_ABLATE 0
#endif
#ifndef H3V_PRIO_B
#define H3V_PRIO_B 1
#endif
// experiment: priority flip in the middle of a step (0 = static priorities).  1 / 2: conv1 waves high in the first / second half
// of their step, low in the other; 3 / 4: the same for the conv2 waves
#ifndef H3V_PRIO_MODE
#define H3V_PRIO_MODE 0
#endif

// B fragments of one 16-pixel group of one ring row: ph / pl = taps (dy,0)|(dy,1) from the hi / lo planes, s = tap (dy,2)
// as [x_hi | x_lo] (see H3RowFrag in fused_h3.hip)
struct H3VFrag {
    h8 ph, pl, s;
};

// the five MFMAs of one (row, dy) pair, w: [dy*4 + {pair hi, pair lo, single [hi|hi], single [lo|0]}]
__device__ __forceinline__ f32x4 h3v_mfma(const H3VFrag& x, const h8 (&w)[13], const int dy, const int m, f32x4 acc)
{
    switch (m) {
        case 0: return MFMA_H(w[dy * 4 + 0], x.ph, acc);
        case 1: return MFMA_H(w[dy * 4 + 1], x.ph, acc);
        case 2: return MFMA_H(w[dy * 4 + 0], x.pl, acc);
        case 3: return MFMA_H(w[dy * 4 + 2], x.s, acc);
        default: return MFMA_H(w[dy * 4 + 3], x.s, acc);
    }
}

__device__ __forceinline__ void h3v_barrier()
{
#if H3V_ABLATE & 16
    __builtin_amdgcn_s_waitcnt(H3_LGKMCNT0);
    asm volatile("" ::: "memory");
#else
    h3_barrier();
#endif
}

#if H3V_ABLATE & 32
#define H3V_STAMP(k)                                                                                     \
    do {                                                                                                 \
        unsigned long long now_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        stamp_sum[k] += now_ - stamp_prev;                                                               \
        stamp_prev = now_;                                                                               \
    } while (0)
#else
#define H3V_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int h3v_wrap(const int v, const int n) { return v >= n ? v - n : v; }

// Epilogue of one 16-pixel group as single-instruction micro-ops, so that they can be placed one by one in the shadows
// of the NEXT group's MFMAs (an MFMA holds the vector-issue port for half of its 16 cycles; an in-order wave gets its other
// vector instructions for free only if they sit right there, and hipcc neither interleaves inline asm nor selects the
// mix instructions from C):
//   [RELU: 4 x v_max]  hi = f16(v * sc): 4 x v_fma_mixlo/hi_f16   lo = f16(v * sc - hi): 4 x v_fma_mixlo/hi_f16 with the f16 hi as
//   third source   2 x ds_write_b64 (hi plane, lo plane).  v * sc is exact (sc a power of two, or 0 for a row / column outside
//   the image), so hi is the correctly rounded f16 of the value and the scale costs no instruction.
// Every micro-op reads accumulator registers no earlier than two MFMAs after the MFMA that finished them (the caller's
// placement), which covers the MFMA -> VALU hazard hipcc does not pad for inline asm.
typedef unsigned h3v_u2 __attribute__((ext_vector_type(2)));
template <bool RELU>
struct H3VEpi {
    static constexpr int NOPS = (RELU ? 4 : 0) + 10;
    f32x4 v;
    float sc, floor_;
    unsigned h0, h1, l0, l1;
    char* p;
    int lo_off;
    template <int I> __device__ __forceinline__ void op()
    {
        constexpr int K = RELU ? I - 4 : I;
        if constexpr (RELU && I < 4) {
            // max(v, floor), floor = 0 (relu) or -inf (linear): ONE instruction (fmaxf / fmed3f come with a canonicalising
            // v_max v, v in front)
            if (!(H3V_ABLATE & 64)) {
                float r;
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v[I]), "v"(floor_));
                v[I] = r;
            }
        } else if constexpr (K == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v.x), "v"(sc));
        else if constexpr (K == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v.z), "v"(sc));
        else if constexpr (K == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v.y), "v"(sc));
        else if constexpr (K == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v.w), "v"(sc));
        // lo = f16(v * sc - hi) in ONE instruction: the third source is the f16 half of the packed hi pair (op_sel_hi[2] = 1:
        // f16 source, op_sel[2]: which half); v * sc - hi is exact in fp32, so this rounds once, like a cvt of the difference
        else if constexpr (K == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(sc), "v"(h0));
        else if constexpr (K == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(sc), "v"(h1));
        else if constexpr (K == 6) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(sc), "v"(h0));
        else if constexpr (K == 7) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(sc), "v"(h1));
        else if constexpr (K == 8) *reinterpret_cast<h3v_u2*>(p) = (H3V_ABLATE & 64) ? (h3v_u2){__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)} : (h3v_u2){h0, h1};
        else if constexpr (K == 9) *reinterpret_cast<h3v_u2*>(p + lo_off) = (H3V_ABLATE & 64) ? (h3v_u2){__builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)} : (h3v_u2){l0, l1};
        __builtin_amdgcn_sched_barrier(0);
    }
    // micro-ops 2*slot, 2*slot+1 (no-ops past the end)
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
