// Fused residual block, split-f16 arithmetic, FULL-ROW STREAMING schedule (fused_block_h3v_kernel).
//
// Same reference semantics, arithmetic, weight images and split-planar activation layout as fused_block_h3r_kernel
// (fused_h3.hip; bfcnn/backbone_blocks.py:174-242 with the inference BatchNormalization folded):
//     out = x + scale * conv2(act(conv1 x)) + shift.
// What changes is the decomposition.  The tile kernels cut an image into 16x32 tiles: two workgroup barriers per tile,
// row pipelines of 4-5 rows that fill and drain (R+2 steps per R rows), 19 % of conv1 recomputed on tile halos, and a
// whole tile (+ a double buffer) resident in LDS.  Here a workgroup owns ALL columns of an image (W <= 256) and walks
// DOWN a band of rows, one image row per step:
//
//   * waves 0..3 ("A") run conv1, waves 4..7 ("B") conv2; wave k of a role owns columns [64k, 64k+64) = four 16-pixel
//     MFMA groups.  Wave w and w+4 share a SIMD, so every SIMD hosts one conv1 and one conv2 stream, and a wave holds ONE
//     kernel's weights (52 VGPRs);
//   * LDS holds rings of rows, not tiles: 6 input rows (filled by LDS-DMA 4 rows ahead) and 3 intermediate rows;
//   * step s: A turns input row s into its three vertical-tap contributions (mid rows s, s-1, s-2; the last one completes
//     and goes to the mid ring); B does the same with mid row s-3 for output rows s-3, s-4, s-5 (s-5 completes and is
//     stored).  The residual enters an output row's accumulator first, as (s2 * I) x [x_hi | x_lo] on the matrix pipe;
//   * ONE barrier per step; no fill / drain inside a band (5 steps per band of 64-256 rows), no halo columns at all
//     (columns -1 and W are the zero padding itself: LDS columns nobody writes), 2 halo rows per band.
//
// Per step and SIMD: 60 + 64 MFMAs of 16 cycles; per step and CU 16 KiB in (16 DMA wave-instructions) and 16 KiB out.
#include "bf_common.h"
#include "h3_core.h"

// timing-only ablations (tools/ablate.sh, ABLATE_MACRO=H3V_ABLATE; results are WRONG when any is set):
// 1 = no DMA, 2 = no global stores, 4 = no conv2 MFMAs, 8 = no conv1 MFMAs, 16 = no per-step barrier,
// 32 = s_memtime stamps (per-wave sums to args.dbg), 64 = no epilogue arithmetic
#ifndef H3V_ABLATE
#define H3V_ABLATE 0
#endif

struct H3VGeom {
    static constexpr int WMAX = 256;                   // columns a workgroup covers (whole image rows)
    static constexpr int G = 4;                        // 16-column groups per wave
    static constexpr int NR = 4, NW = 8, NT = 512;     // waves per role, per workgroup
    static constexpr int PITCH = (WMAX + 2) * 16;      // bytes per plane-row; ring column = image column + 1
    static constexpr int NRI = 6, NRM = 3, PD = 4;     // ring depths (rows), DMA distance (rows ahead of conv1)
    static constexpr int UNROLL = 6;                   // steps per loop iteration: ring slots and accumulator rotation static
    static constexpr int IN_PLANE = (NRI * PITCH + 255) / 256 * 256;
    static constexpr int MID_PLANE = (NRM * PITCH + 255) / 256 * 256;
    static constexpr int IN_BYTES = 4 * IN_PLANE, MID_BYTES = 4 * MID_PLANE;
    static constexpr int LDS_BYTES = IN_BYTES + MID_BYTES;
    static constexpr int DMA_PER_WAVE = WMAX / 64;     // role-A wave k moves plane k of a row: 4 x 1 KiB
    static_assert(UNROLL % NRI == 0 && UNROLL % NRM == 0 && UNROLL % 3 == 0, "static slots");
    static_assert(NRI >= PD + 2, "a row's slot is reused only after its last reader (the residual, one step after conv1)");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(2 * IN_PLANE < 65536 && 3 * MID_PLANE < 65536, "ds offsets are 16-bit");
};

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

struct H3VTile {
    int y0, nrows;
    size_t img;
};

__device__ __forceinline__ H3VTile h3v_tile(const FusedH3Args& a, const int t)
{
    H3VTile r;
    const int b = t / a.tiles_y, ty = t - b * a.tiles_y;
    r.y0 = ty * a.rows_per_tile;
    r.nrows = min(a.rows_per_tile, a.H - r.y0);
    r.img = (size_t)b * a.H * a.W * 64;
    return r;
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

// ---------------------------------------------------------------------------------------------------------------------
// role A: conv1.  State that lives across steps: acc[g][3] (mid rows s, s-1, s-2 modulo 3), the prefetched fragment of
// the next step's first group.
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3VRoleA {
    using Gm = H3VGeom;
    const FusedH3Args& a;
    const char* tin;
    char* tmid;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    H3VFrag nxt;                 // fragment of (row s+1, group 0), requested during step s
    int rp, rs;                  // lane's LDS byte address of ring row 0, group 0: pair fragment (hi planes), single fragment
    int wr;                      // lane's LDS byte address of its record in mid ring row 0, group 0
    float inv_s, relu_floor;
    float lane_scale[Gm::G];     // !FULLW: inv_s where the lane's column is inside the image, else 0
    unsigned dma_off[Gm::DMA_PER_WAVE];   // lane's byte offset inside a plane-row for DMA instruction j
    bool dma_ok[Gm::DMA_PER_WAVE];        // !FULLW: the lane's column is inside the image
    int plane;                   // the plane this wave moves (= wave index within the role)
    unsigned plane_g;

    __device__ __forceinline__ H3VFrag load(const int slot, const int g) const
    {
        H3VFrag f;
        const int o = slot * Gm::PITCH + g * 256;
        f.ph = *reinterpret_cast<const h8*>(tin + rp + o);
        f.pl = *reinterpret_cast<const h8*>(tin + rp + o + 2 * Gm::IN_PLANE);
        f.s = *reinterpret_cast<const h8*>(tin + rs + o);
        return f;
    }

    // source of ring row r (image row y0 - 2 + r) for this wave's plane: wave-uniform base pointer and offset mask; rows
    // outside the image or past the band's last halo row come from the zero line (mask 0: every lane reads its first 16 B)
    struct RowSrc {
        const char* base;
        unsigned mask;
    };
    __device__ __forceinline__ RowSrc row_src(const H3VTile& t, const int r) const
    {
        const int y = t.y0 - 2 + r;
        const bool ok = (y >= 0) & (y < a.H) & (r < t.nrows + 4);
        RowSrc rs;
        rs.base = ok ? reinterpret_cast<const char*>(a.in) + t.img + (size_t)plane * plane_g + (size_t)y * a.W * 16
                     : reinterpret_cast<const char*>(a.zeros);
        rs.mask = ok ? ~0u : 0u;
        return rs;
    }
    // DMA instruction j of a ring row: columns 64j .. 64j+63 of plane `plane` -> ring slot `slot`
    __device__ __forceinline__ void dma(const RowSrc& rs, const int slot, const int j) const
    {
        if (H3V_ABLATE & 1) return;
        const char* src = rs.base + (dma_off[j] & rs.mask);
        if (!FULLW) {
            if (!dma_ok[j]) src = reinterpret_cast<const char*>(a.zeros);
        }
        char* dst = const_cast<char*>(tin) + plane * Gm::IN_PLANE + slot * Gm::PITCH + (1 + 64 * j) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }

    // mid row m (image row ym) of group g: scale, activation, split, one 16-byte record per lane -> mid ring
    __device__ __forceinline__ void epilogue(const int g, const int mslot, f32x4 v, const bool rowok) const
    {
        if (!(H3V_ABLATE & 64)) {
            float sc = FULLW ? inv_s : lane_scale[g];
            sc = rowok ? sc : 0.f;                       // rows outside the image are conv2's zero padding
            v = v * sc;
            v.x = __builtin_amdgcn_fmed3f(v.x, relu_floor, __builtin_inff()); v.y = __builtin_amdgcn_fmed3f(v.y, relu_floor, __builtin_inff());
            v.z = __builtin_amdgcn_fmed3f(v.z, relu_floor, __builtin_inff()); v.w = __builtin_amdgcn_fmed3f(v.w, relu_floor, __builtin_inff());
        }
        *reinterpret_cast<h8*>(tmid + wr + mslot * Gm::PITCH + g * 256) = h3_split_record(v);
    }

    // step s (s % UNROLL == PH): input ring row s
    template <int PH>
    __device__ __forceinline__ void step(const H3VTile& t, const int s)
    {
        constexpr int islot = PH % Gm::NRI, nslot = (PH + 1) % Gm::NRI, dslot = (PH + Gm::PD) % Gm::NRI;
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // accumulators of mid rows s, s-1, s-2
        constexpr int mslot = (PH + 1) % Gm::NRM;                              // (s - 2) mod 3
        const int m = s - 2, ym = t.y0 - 1 + m;
        const bool rowok = (m >= 0) & (ym >= 0) & (ym < a.H);
        const RowSrc src = row_src(t, s + Gm::PD);
        H3VFrag cur = nxt;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(islot, g + 1);
            else nx = load(nslot, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g > 0) epilogue(g - 1, mslot, acc[g - 1][a2], rowok);
            if (!(H3V_ABLATE & 8)) {
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    acc[g][a2] = h3v_mfma(cur, w, 2, k, acc[g][a2]);
                    acc[g][a1] = h3v_mfma(cur, w, 1, k, acc[g][a1]);
                    c0 = h3v_mfma(cur, w, 0, k, c0);
                }
                acc[g][a0] = c0;
            } else {
                acc[g][a2][0] += (float)cur.ph[0] + (float)cur.pl[1] + (float)cur.s[2];
            }
            dma(src, dslot, g);
            if (g > 0) {
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            cur = nx;
        }
        nxt = cur;
        epilogue(Gm::G - 1, mslot, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// role B: conv2 + folded BN + residual -> global
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3VRoleB {
    using Gm = H3VGeom;
    const FusedH3Args& a;
    const char* tin;
    const char* tmid;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's LDS byte address in mid ring row 0, group 0: pair / single fragments
    int rr;                      // lane's LDS byte address of the residual operand [x_hi | x_lo] in input ring row 0, group 0
    unsigned g_off;              // lane's byte offset of its record from the start of an output row (plane + column), group 0
    bool col_ok[Gm::G];          // !FULLW
    float inv_s2;
    f32x4 sh;
    int lane;

    __device__ __forceinline__ H3VFrag load(const int slot, const int g) const
    {
        H3VFrag f;
        const int o = slot * Gm::PITCH + g * 256;
        f.ph = *reinterpret_cast<const h8*>(tmid + rp + o);
        f.pl = *reinterpret_cast<const h8*>(tmid + rp + o + 2 * Gm::MID_PLANE);
        f.s = *reinterpret_cast<const h8*>(tmid + rs + o);
        return f;
    }
    __device__ __forceinline__ h8 load_res(const int slot, const int g) const
    {
        return *reinterpret_cast<const h8*>(tin + rr + slot * Gm::PITCH + g * 256);
    }

    __device__ __forceinline__ void epilogue(const int g, const f32x4 accv, char* out_row, const bool rowok) const
    {
        const f32x4 v = (H3V_ABLATE & 64) ? accv : accv * inv_s2 + sh;
        const h8 rec = h3_split_record(v);
        if (H3V_ABLATE & 2) {
            if (v.x == 12345.678f) *reinterpret_cast<h8*>(out_row + g_off + g * 256) = rec;
            return;
        }
        // rows outside the band (pipeline fill / drain) and columns outside the image go to the dump line: a select, not a
        // branch -- a branch would cut the scheduling region and the epilogue could no longer sit in the MFMAs' shadows
        char* p = out_row + g_off + g * 256;
        bool ok = rowok;
        if (!FULLW) ok = ok && col_ok[g];
        if (!ok) p = reinterpret_cast<char*>(a.dump) + lane * 16;
        *reinterpret_cast<h8*>(p) = rec;
    }

    // step s (s % UNROLL == PH): mid ring row s-3, residual from input ring row s-1, completes output row s-5
    template <int PH>
    __device__ __forceinline__ void step(const H3VTile& t, const int s)
    {
        constexpr int mslot = PH % Gm::NRM;                                    // (s - 3) mod 3
        constexpr int xslot = (PH + Gm::NRI - 1) % Gm::NRI;                     // (s - 1) mod 6
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // output rows s-3, s-4, s-5
        const int o = s - 5;
        const bool rowok = (o >= 0) & (o < t.nrows);
        char* out_row = reinterpret_cast<char*>(a.out) + t.img + (size_t)(t.y0 + (rowok ? o : 0)) * a.W * 16;
        H3VFrag cur = load(mslot, 0);
        h8 xr = load_res(xslot, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            h8 xn;
            if (g + 1 < Gm::G) {
                nx = load(mslot, g + 1);
                xn = load_res(xslot, g + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g > 0) epilogue(g - 1, acc[g - 1][a2], out_row, rowok);
            if (!(H3V_ABLATE & 4)) {
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
                c0 = MFMA_H(w[12], xr, c0);                     // residual: (s2 * I) x [x_hi | x_lo], exact
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    acc[g][a2] = h3v_mfma(cur, w, 2, k, acc[g][a2]);
                    acc[g][a1] = h3v_mfma(cur, w, 1, k, acc[g][a1]);
                    c0 = h3v_mfma(cur, w, 0, k, c0);
                }
                acc[g][a0] = c0;
            } else {
                acc[g][a2][0] += (float)cur.ph[0] + (float)cur.pl[1] + (float)cur.s[2] + (float)xr[3];
            }
            if (g > 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            if (g + 1 < Gm::G) {
                cur = nx;
                xr = xn;
            }
        }
        epilogue(Gm::G - 1, bf_acc_ready(acc[Gm::G - 1][a2]), out_row, rowok);
    }
};

template <bool FULLW>
__global__ __launch_bounds__(H3VGeom::NT, 2) void fused_block_h3v_kernel(FusedH3Args a)
{
    using Gm = H3VGeom;
    extern __shared__ __attribute__((aligned(16))) char h3v_lds[];
    char* tin = h3v_lds;                                       // [4 planes][NRI rows][WMAX + 2 columns][8 f16]
    char* tmid = h3v_lds + Gm::IN_BYTES;                       // [4 planes][NRM rows][WMAX + 2 columns][8 f16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const bool role_a = wave < Gm::NR;
    const int rw = wave & (Gm::NR - 1);
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;      // bytes per global plane

    // columns 0 and W+1.. of both rings are the zero padding: cleared once, never written (DMA and conv1 write ring
    // columns 1 .. WMAX only, and zeros where the image is narrower)
    for (int i = tid * 16; i < Gm::LDS_BYTES; i += Gm::NT * 16) *reinterpret_cast<f32x4*>(h3v_lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int c0 = 64 * rw + n;                                 // lane's image column in group 0
#if H3V_ABLATE & 32
    unsigned long long stamp_sum[4] = {0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    if (role_a) {
        H3VRoleA<FULLW> A{a, tin, tmid};
#pragma unroll
        for (int i = 0; i < 13; ++i) A.w[i] = reinterpret_cast<const h8*>(a.w1r)[i * 64 + lane];
        A.inv_s = a.aux[0];
        A.relu_floor = a.act1_relu ? 0.f : -__builtin_inff();
        // taps dx = 0,1 of image column c are ring columns c, c+1 (ring column = image column + 1, centre tap dx = 1)
        A.rp = (q & 1) * Gm::IN_PLANE + (c0 + (q >> 1)) * 16;
        A.rs = ((q & 1) + 2 * (q >> 1)) * Gm::IN_PLANE + (c0 + 2) * 16;
        A.wr = ((q >> 1) + 2 * (q & 1)) * Gm::MID_PLANE + (c0 + 1) * 16;
        A.plane = rw;
        A.plane_g = plane_g;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) A.lane_scale[g] = (c0 + 16 * g < a.W) ? A.inv_s : 0.f;
#pragma unroll
        for (int j = 0; j < Gm::DMA_PER_WAVE; ++j) {
            A.dma_off[j] = (unsigned)(64 * j + lane) * 16u;
            A.dma_ok[j] = 64 * j + lane < a.W;
        }
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) A.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // weight / scale loads retired before any counted wait

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3VTile t = h3v_tile(a, ti);
            // prologue: rows 0 .. PD-1 requested, rows 0 and 1 landed
#pragma unroll
            for (int r = 0; r < Gm::PD; ++r) {
                const auto src = A.row_src(t, r);
#pragma unroll
                for (int j = 0; j < Gm::DMA_PER_WAVE; ++j) A.dma(src, r % Gm::NRI, j);
            }
            __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3V_ABLATE & 1) ? 0 : 2 * Gm::DMA_PER_WAVE));
            h3_barrier();
            A.nxt = A.load(0, 0);
            const int nsteps = t.nrows + 5;
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3V_STEP_A(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if (s < nsteps - 1) A.template step<PH>(t, s);     /* input rows 0 .. nrows+3 */         \
                    H3V_STAMP(0);                                                                             \
                    /* row s+2 (requested two steps ago) has landed <=> at most the 2 younger rows are outstanding */ \
                    __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3V_ABLATE & 1) ? 0 : 2 * Gm::DMA_PER_WAVE));            \
                    H3V_STAMP(1);                                                                             \
                    h3_barrier();                                                                             \
                    H3V_STAMP(2);                                                                             \
                } while (0)
                H3V_STEP_A(0); H3V_STEP_A(1); H3V_STEP_A(2); H3V_STEP_A(3); H3V_STEP_A(4); H3V_STEP_A(5);
#undef H3V_STEP_A
            }
            // the rows requested past the band's end (zero-line reads into dead slots) must not land in the next band's rows
            __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
            h3_barrier();
        }
    } else {
        H3VRoleB<FULLW> Bv{a, tin, tmid};
#pragma unroll
        for (int i = 0; i < 13; ++i) Bv.w[i] = reinterpret_cast<const h8*>(a.w2r)[i * 64 + lane];
        Bv.inv_s2 = a.aux[48];
        Bv.sh = *reinterpret_cast<const f32x4*>(a.aux + 32 + q * 4);
        Bv.rp = (q & 1) * Gm::MID_PLANE + (c0 + (q >> 1)) * 16;
        Bv.rs = ((q & 1) + 2 * (q >> 1)) * Gm::MID_PLANE + (c0 + 2) * 16;
        Bv.rr = ((q & 1) + 2 * (q >> 1)) * Gm::IN_PLANE + (c0 + 1) * 16;
        Bv.g_off = (unsigned)((q >> 1) + 2 * (q & 1)) * plane_g + (unsigned)c0 * 16u;
        Bv.lane = lane;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) Bv.col_ok[g] = c0 + 16 * g < a.W;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) Bv.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3VTile t = h3v_tile(a, ti);
            h3_barrier();                                        // prologue barrier
            const int nsteps = t.nrows + 5;
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3V_STEP_B(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if (s >= 3 && s < nsteps) Bv.template step<PH>(t, s);    /* mid rows 0 .. nrows+1 */      \
                    H3V_STAMP(0);                                                                             \
                    h3_barrier();                                                                             \
                    H3V_STAMP(2);                                                                             \
                } while (0)
                H3V_STEP_B(0); H3V_STEP_B(1); H3V_STEP_B(2); H3V_STEP_B(3); H3V_STEP_B(4); H3V_STEP_B(5);
#undef H3V_STEP_B
            }
            h3_barrier();
        }
    }
#if H3V_ABLATE & 32
    if (a.dbg && lane == 0) {
        for (int k = 0; k < 4; ++k) a.dbg[((size_t)blockIdx.x * Gm::NW + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

// bands: every image is cut into ceil(H / rows) bands of `rows` rows; one band = one unit of work of a workgroup.
// rows is chosen so that the slowest CU (ceil(bands / CUs) bands of rows + 8 steps each) finishes earliest.
static int h3v_rows_per_tile(const int B, const int H, const int cus)
{
    int best = H;
    long best_cost = -1;
    for (int ty = 1; ty <= (H + 7) / 8; ++ty) {
        const int rows = (H + ty - 1) / ty;
        if ((H + rows - 1) / rows != ty) continue;
        const long tiles = (long)B * ty;
        const long cost = ((tiles + cus - 1) / cus) * (rows + 8);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = rows;
        }
    }
    return best;
}

bool bf_fused_block_h3v_supports(int H, int W) { return W >= 1 && W <= H3VGeom::WMAX && H >= 1; }

hipError_t bf_launch_fused_block_h3v(const FusedH3Args& args, hipStream_t s)
{
    using Gm = H3VGeom;
    FusedH3Args a = args;
    if (!a.zeros || !a.dump || !bf_fused_block_h3v_supports(a.H, a.W)) return hipErrorInvalidValue;
    if ((int64_t)a.H * a.W * 64 >= ((int64_t)1 << 32)) return hipErrorInvalidValue;      // 32-bit in-image offsets
    const int cus = 256;
    a.rows_per_tile = h3v_rows_per_tile(a.B, a.H, cus);
    a.tiles_x = 1;
    a.tiles_y = (a.H + a.rows_per_tile - 1) / a.rows_per_tile;
    a.ntiles = a.B * a.tiles_y;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    const bool fullw = a.W == Gm::WMAX;
    void (*kernel)(FusedH3Args) = fullw ? fused_block_h3v_kernel<true> : fused_block_h3v_kernel<false>;
    const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(kernel), Gm::LDS_BYTES);      // once per device
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Gm::NT), Gm::LDS_BYTES, s, a);
    return hipGetLastError();
}
