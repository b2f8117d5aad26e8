// Fused residual block, split-f16 arithmetic, FULL-ROW STREAMING schedule (fused_block_h3v_kernel).
//
// Same reference semantics, arithmetic, weight images and split-planar activation layout as fused_block_h3r_kernel
// (fused_h3.hip; bfcnn/backbone_blocks.py:174-242 with the inference BatchNormalization folded):
//     out = x + scale * conv2(act(conv1 x)) + shift.
// What changes is the decomposition.  The tile kernels cut an image into 16x32 tiles: two workgroup barriers per tile,
// row pipelines of 4-5 rows that fill and drain (R+2 steps per R rows), 19 % of conv1 recomputed on tile halos, and a
// whole tile (+ a double buffer) resident in LDS.  Here a workgroup owns ALL columns of an image (W <= 256) and walks
// DOWN a band of rows, one image row per step, with three kinds of waves (12 per workgroup, one of each kind per SIMD):
//
//   * waves 0..3 ("A") run conv1, waves 4..7 ("B") conv2; wave k of a role owns columns [64k, 64k+64) = four 16-pixel
//     MFMA groups and holds ONE kernel's weights (52 VGPRs).  Neither role ever issues a vector-memory instruction;
//   * waves 8..11 ("C") move the data: waves 8-9 request input row s+3 into the input ring by LDS-DMA (two planes each),
//     waves 10-11 store the finished output row from its LDS staging slot to global memory -- loaders and storers are
//     DIFFERENT waves because `s_waitcnt vmcnt(N)` counts a wave's loads and stores together while stores retire early: a
//     wave that did both saw its count reached with DMA pieces still in flight.  A vector-memory instruction costs the issuing wave 100-200 cycles when the
//     CU streams 32 KiB per step; on the matrix waves those stalls did not overlap with anything (ablations of the first
//     version, where A issued the DMA and B the stores: 248 us per launch, 150 us without memory instructions, 95 us
//     without matrix / vector work -- the sum, not the maximum);
//   * LDS holds rings of rows, not tiles: 5 input rows, 2 intermediate rows, 2 output rows;
//   * step s: A turns input row s into its three vertical-tap contributions (mid rows s, s-1, s-2; the last one completes
//     and goes to the mid ring); B does the same with mid row s-3 for output rows s-3, s-4, s-5 (s-5 completes and goes to
//     the staging ring); C stores output row s-6.  The residual enters an output row's accumulator first, as
//     (s2 * I) x [x_hi | x_lo] on the matrix pipe;
//   * ONE barrier per step; no fill / drain inside a band (6 steps per band of 64-256 rows), no halo columns at all
//     (columns -1 and W are the zero padding itself: LDS columns nobody writes), 2 halo rows per band.
//
// Per step and SIMD: 60 + 64 MFMAs of 16 cycles; per step and CU 16 KiB in (16 DMA wave-instructions) and 16 KiB out.
#include "bf_common.h"
#include "h3_core.h"

#include "h3v_core.h"

struct H3VGeom {
    static constexpr int WMAX = 256;                   // columns a workgroup covers (whole image rows)
    static constexpr int G = 4;                        // 16-column groups per matrix wave
    static constexpr int NR = 4, NW = 12, NT = 768;    // waves per role, per workgroup
    static constexpr int PITCH = (WMAX + 2) * 16;      // bytes per plane-row of the input / mid rings; ring column = image column + 1
    static constexpr int NRI = 5, NRM = 2, NRO = 2;    // ring depths (rows)
    static constexpr int PD = 3;                       // DMA distance: row s+3 is requested in step s, awaited at the end of step s+2
    static constexpr int UNROLL = 6;                   // steps per loop iteration: mid / out slots and the accumulator rotation static
    static constexpr int IN_PLANE = (NRI * PITCH + 255) / 256 * 256;
    static constexpr int MID_PLANE = (NRM * PITCH + 255) / 256 * 256;
    static constexpr int OUT_PLANE = WMAX * 16, OUT_SLOT = 4 * OUT_PLANE;
    static constexpr int IN_BYTES = 4 * IN_PLANE, MID_BYTES = 4 * MID_PLANE, OUT_BYTES = NRO * OUT_SLOT;
    static constexpr int LDS_BYTES = IN_BYTES + MID_BYTES + OUT_BYTES;
    static constexpr int PIECES = WMAX / 64;           // 1-KiB wave-instructions per plane-row; C wave k moves plane k
    static constexpr int NSTAMP = 4;
    static_assert(UNROLL % NRM == 0 && UNROLL % NRO == 0 && UNROLL % 3 == 0, "static slots");
    // input ring: rows s-1 (residual), s (conv1), s+1 (landed), s+2, s+3 (in flight)
    static_assert(NRI == PD + 2, "input ring");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(2 * IN_PLANE + 1024 < 65536 && 2 * MID_PLANE + NRM * PITCH + 1024 < 65536, "fragment offsets fit the 16-bit ds offset");
};

struct H3VTile {
    int y0, nrows;
    size_t img;
    int ybase, ystep;            // image row of band-relative row k: ybase + ystep * k (a reversed band walks bottom-up)
    __device__ __forceinline__ int y(const int k) const { return ybase + ystep * k; }
};

// A launch walks its bands top-down or (reverse_tiles) bottom-up and last band first: consecutive blocks alternate, so a
// block starts on the rows the previous one wrote last -- the ones still in the 256 MB Infinity Cache.  Walking up only
// mirrors the vertical taps (the weight images are loaded with dy -> 2 - dy) and the row addresses.
__device__ __forceinline__ H3VTile h3v_tile(const FusedH3Args& a, const int t)
{
    H3VTile r;
    const int tt = a.reverse_tiles ? a.ntiles - 1 - t : t;
    const int b = tt / a.tiles_y, ty = tt - b * a.tiles_y;
    r.y0 = ty * a.rows_per_tile;
    r.nrows = min(a.rows_per_tile, a.H - r.y0);
    r.img = (size_t)b * a.H * a.W * 64;
    r.ybase = a.reverse_tiles ? r.y0 + r.nrows - 1 : r.y0;
    r.ystep = a.reverse_tiles ? -1 : 1;
    return r;
}

// weight image i = dy * 4 + kind (12 = s2 * identity) as the code's tap row dy: mirrored for a band that walks bottom-up
__device__ __forceinline__ int h3v_wimage(const FusedH3Args& a, const int i)
{
    return (a.reverse_tiles && i < 12) ? (2 - i / 4) * 4 + i % 4 : i;
}

// ---------------------------------------------------------------------------------------------------------------------
// role A: conv1.  State that lives across steps: acc[g][3] (mid rows s, s-1, s-2 modulo 3).
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3VRoleA {
    using Gm = H3VGeom;
    const FusedH3Args& a;
    const char* tin;
    char* tmid;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's LDS byte address of ring slot 0, group 0: pair fragment (hi planes), single fragment
    int wr;                      // lane's LDS byte address of its 8-byte hi record in mid ring slot 0, group 0 (lo: + 2 planes)
    float inv_s, relu_floor;
    float lane_scale[Gm::G];     // !FULLW: inv_s where the lane's column is inside the image, else 0

    __device__ __forceinline__ H3VFrag load(const int islot_bytes, const int g) const
    {
        H3VFrag f;
        const char* p = tin + rp + islot_bytes;           // (one v_add per step; everything else folds into the ds offsets)
        f.ph = *reinterpret_cast<const h8*>(p + g * 256);
        f.pl = *reinterpret_cast<const h8*>(p + g * 256 + 2 * Gm::IN_PLANE);
        f.s = *reinterpret_cast<const h8*>(tin + rs + islot_bytes + g * 256);
        return f;
    }

    // epilogue state of mid row m, group g (micro-ops: H3VEpi)
    __device__ __forceinline__ H3VEpi<true> epilogue(const int g, const int mslot, const f32x4 v, const bool rowok) const
    {
        H3VEpi<true> e;
        e.v = v;
        const float sc = FULLW ? inv_s : lane_scale[g];
        e.sc = rowok ? sc : 0.f;                         // rows outside the image are conv2's zero padding
        e.floor_ = relu_floor;
        e.p = tmid + wr + mslot * Gm::PITCH + g * 256;
        e.lo_off = 2 * Gm::MID_PLANE;
        return e;
    }

    // the 15 MFMAs of group g; micro-ops of the previous group's epilogue after MFMA 2, 3, ...
    template <int J, class Epi>
    __device__ __forceinline__ void mfmas(const int g, const int a0, const int a1, const int a2, const H3VFrag& cur, f32x4& c0, Epi* epi)
    {
        if constexpr (J < 15) {
            constexpr int k = J / 3, which = J % 3;
            if (!(H3V_ABLATE & 8)) {
                if constexpr (which == 0) acc[g][a2] = h3v_mfma(cur, w, 2, k, acc[g][a2]);
                else if constexpr (which == 1) acc[g][a1] = h3v_mfma(cur, w, 1, k, acc[g][a1]);
                else c0 = h3v_mfma(cur, w, 0, k, c0);
            } else if (J == 0) {
                acc[g][a2][0] += (float)cur.ph[0] + (float)cur.pl[1] + (float)cur.s[2];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J >= 2) {
                if (epi) epi->template pair<J - 2>();
            }
            mfmas<J + 1>(g, a0, a1, a2, cur, c0, epi);
        }
    }

    // step s (s % UNROLL == PH): input ring row s in slot `islot`
    template <int PH>
    __device__ __forceinline__ void step(const H3VTile& t, const int s, const int islot)
    {
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // accumulators of mid rows s, s-1, s-2
        constexpr int mslot = PH % Gm::NRM;                                    // (s - 2) mod 2
        const int m = s - 2, ym = t.y(m - 1);
        const bool rowok = (m >= 0) & (ym >= 0) & (ym < a.H);
        const int ib = islot * Gm::PITCH;
        H3VFrag cur = load(ib, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(ib, g + 1);
            if (H3V_PRIO_MODE == 1 && g == 0) __builtin_amdgcn_s_setprio(2);
            if (H3V_PRIO_MODE == 1 && g == 2) __builtin_amdgcn_s_setprio(0);
            if (H3V_PRIO_MODE == 2 && g == 0) __builtin_amdgcn_s_setprio(0);
            if (H3V_PRIO_MODE == 2 && g == 2) __builtin_amdgcn_s_setprio(2);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if (g > 0) {
                H3VEpi<true> e = epilogue(g - 1, mslot, acc[g - 1][a2], rowok);
                mfmas<0>(g, a0, a1, a2, cur, c0, &e);
            } else {
                mfmas<0>(g, a0, a1, a2, cur, c0, (H3VEpi<true>*)nullptr);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        H3VEpi<true> e = epilogue(Gm::G - 1, mslot, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
        e.all();
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// role B: conv2 + folded BN + residual -> output staging ring
// ---------------------------------------------------------------------------------------------------------------------
struct H3VRoleB {
    using Gm = H3VGeom;
    const FusedH3Args& a;
    const char* tin;
    const char* tmid;
    char* tout;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's LDS byte address in mid ring slot 0, group 0: pair / single fragments
    int rr;                      // lane's LDS byte address of the residual operand [x_hi | x_lo] in input ring slot 0, group 0
    int wo;                      // lane's LDS byte address of its 8-byte hi record in staging slot 0, group 0 (lo: + 2 planes)
    float inv_s2;
    f32x4 shs;                   // folded BN shift of the lane's four channels times s2 (the accumulators' scale)

    __device__ __forceinline__ H3VFrag load(const int slot, const int g) const
    {
        H3VFrag f;
        const int o = slot * Gm::PITCH + g * 256;
        f.ph = *reinterpret_cast<const h8*>(tmid + rp + o);
        f.pl = *reinterpret_cast<const h8*>(tmid + rp + o + 2 * Gm::MID_PLANE);
        f.s = *reinterpret_cast<const h8*>(tmid + rs + o);
        return f;
    }
    __device__ __forceinline__ h8 load_res(const int xslot_bytes, const int g) const
    {
        return *reinterpret_cast<const h8*>(tin + rr + xslot_bytes + g * 256);
    }

    // the accumulator already holds s2 * (scale * conv2 + x + shift) (the shift is the C operand of the row's first MFMA):
    // epilogue state of output row o, group g (micro-ops: H3VEpi)
    __device__ __forceinline__ H3VEpi<false> epilogue(const int g, const int oslot, const f32x4 accv) const
    {
        H3VEpi<false> e;
        e.v = accv;
        e.sc = inv_s2;
        e.floor_ = 0.f;
        e.p = tout + wo + oslot * Gm::OUT_SLOT + g * 256;
        e.lo_off = 2 * Gm::OUT_PLANE;
        return e;
    }

    // the 1 + 15 MFMAs of group g; micro-ops of the previous group's epilogue after MFMA 2, 3, ...
    template <int J, class Epi>
    __device__ __forceinline__ void mfmas(const int g, const int a0, const int a1, const int a2, const H3VFrag& cur, const h8 xr,
                                          f32x4& c0, Epi* epi)
    {
        if constexpr (J < 16) {
            if (!(H3V_ABLATE & 4)) {
                if constexpr (J == 0) {
                    // residual: (s2 * I) x [x_hi | x_lo], exact, on top of the folded BN shift (times s2) as the C operand
                    c0 = MFMA_H(w[12], xr, shs);
                } else {
                    constexpr int k = (J - 1) / 3, which = (J - 1) % 3;
                    if constexpr (which == 0) acc[g][a2] = h3v_mfma(cur, w, 2, k, acc[g][a2]);
                    else if constexpr (which == 1) acc[g][a1] = h3v_mfma(cur, w, 1, k, acc[g][a1]);
                    else c0 = h3v_mfma(cur, w, 0, k, c0);
                }
            } else if (J == 0) {
                acc[g][a2][0] += (float)cur.ph[0] + (float)cur.pl[1] + (float)cur.s[2] + (float)xr[3];
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (J >= 2) {
                if (epi) epi->template pair<J - 2>();
            }
            mfmas<J + 1>(g, a0, a1, a2, cur, xr, c0, epi);
        }
    }

    // step s (s % UNROLL == PH): mid ring row s-3, residual from input ring row s-1 (slot `xslot`), completes output row s-5
    template <int PH>
    __device__ __forceinline__ void step(const int xslot)
    {
        constexpr int mslot = (PH + 1) % Gm::NRM;                              // (s - 3) mod 2
        constexpr int oslot = (PH + 1) % Gm::NRO;                              // (s - 5) mod 2
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // output rows s-3, s-4, s-5
        const int xb = xslot * Gm::PITCH;
        H3VFrag cur = load(mslot, 0);
        h8 xr = load_res(xb, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            h8 xn;
            if (g + 1 < Gm::G) {
                nx = load(mslot, g + 1);
                xn = load_res(xb, g + 1);
            }
            if (H3V_PRIO_MODE == 3 && g == 0) __builtin_amdgcn_s_setprio(2);
            if (H3V_PRIO_MODE == 3 && g == 2) __builtin_amdgcn_s_setprio(0);
            if (H3V_PRIO_MODE == 4 && g == 0) __builtin_amdgcn_s_setprio(0);
            if (H3V_PRIO_MODE == 4 && g == 2) __builtin_amdgcn_s_setprio(2);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0;
            if (g > 0) {
                H3VEpi<false> e = epilogue(g - 1, oslot, acc[g - 1][a2]);
                mfmas<0>(g, a0, a1, a2, cur, xr, c0, &e);
            } else {
                mfmas<0>(g, a0, a1, a2, cur, xr, c0, (H3VEpi<false>*)nullptr);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) {
                cur = nx;
                xr = xn;
            }
        }
        H3VEpi<false> e = epilogue(Gm::G - 1, oslot, bf_acc_ready(acc[Gm::G - 1][a2]));
        e.all();
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// role C: the only waves that touch global memory.  Waves 8, 9 request the input rows (LDS-DMA, two planes = eight 1-KiB
// pieces per step each); waves 10, 11 store the output row B finished one step earlier (two planes each).  Loads and
// stores are kept on different waves because the exact vmcnt of the loaders needs a queue that completes in issue order:
// with stores in the same queue the count reached its target while the oldest DMA pieces were still in flight (stores
// retire early) and conv1 read rows that had not landed (parity test, 240 workgroups).  EVERY step issues exactly eight
// pieces per loader (rows outside the band: reads of the zero line).
// ---------------------------------------------------------------------------------------------------------------------
template <bool FULLW>
struct H3VRoleC {
    using Gm = H3VGeom;
    const FusedH3Args& a;
    char* tin;
    const char* tout;
    int plane0, lane;                  // this wave moves planes plane0 and plane0 + 1
    unsigned plane_g;
    unsigned col_off[Gm::PIECES];      // lane's byte offset inside a plane-row for piece j: (64 j + lane) * 16
    bool col_ok[Gm::PIECES];           // !FULLW: the lane's column is inside the image

    // DMA of ring row r (image row y0 - 2 + r) into ring slot `slot`; rows outside the image / past the band: zero line
    __device__ __forceinline__ void dma_row(const H3VTile& t, const int r, const int slot) const
    {
        if (H3V_ABLATE & 1) return;
        const int y = t.y(r - 2);
        const bool ok = (y >= 0) & (y < a.H) & (r < t.nrows + 4);                   // wave-uniform
        const unsigned mask = ok ? ~0u : 0u;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int plane = plane0 + pl;
            const char* base = ok ? reinterpret_cast<const char*>(a.in) + t.img + (size_t)plane * plane_g + (size_t)y * a.W * 16
                                  : reinterpret_cast<const char*>(a.zeros);
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j) {
                const char* src = base + (col_off[j] & mask);
                if (!FULLW) {
                    if (!col_ok[j]) src = reinterpret_cast<const char*>(a.zeros);
                }
                char* dst = tin + plane * Gm::IN_PLANE + slot * Gm::PITCH + (1 + 64 * j) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    }

    // ---- compact layout (a.compact; bf_common.h): the lo planes are fp8 in memory, 8 bytes per pixel.  LDS-DMA cannot convert, so
    // the loader of the lo planes (wave 9) takes them through registers: lane l owns the pixels 64 j + l of both planes (one
    // 8-byte load each; consecutive lanes = consecutive 16-byte ring records, the bank-conflict-free pattern of the DMA),
    // requests row s+3 in step s, and in step s+1 -- a whole step later, the data has arrived -- decodes it to f16 and writes it
    // to the ring: row s+2 is complete at the end of step s, one step before conv1 reads it (the DMA rows are complete one step
    // later).  The storer of the lo planes (wave 11) encodes while it stores.
    bf_u2 lo_raw[2][Gm::PIECES];       // [plane - 2][j]: 8 fp8 of one pixel of ring row "pending"

    __device__ __forceinline__ void lo_request(const H3VTile& t, const int r)
    {
        const int y = t.y(r - 2);
        const bool ok = (y >= 0) & (y < a.H) & (r < t.nrows + 4);                   // wave-uniform
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const char* base = reinterpret_cast<const char*>(a.in) + t.img + 2 * (size_t)plane_g + (size_t)pl * (plane_g / 2) +
                               (size_t)(ok ? y : 0) * a.W * 8;
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j) {
                bf_u2 v = {0u, 0u};
                if (ok && (FULLW || col_ok[j])) v = *reinterpret_cast<const bf_u2*>(base + (col_off[j] >> 1));
                lo_raw[pl][j] = v;
            }
        }
    }

    __device__ __forceinline__ void lo_commit(const int slot) const
    {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j)
                *reinterpret_cast<h8*>(tin + (2 + pl) * Gm::IN_PLANE + slot * Gm::PITCH + (1 + 64 * j + lane) * 16) = bf_h3c_decode8(lo_raw[pl][j]);
    }

    __device__ __forceinline__ void lo_store_row(const H3VTile& t, const int o, const int oslot) const
    {
        if (H3V_ABLATE & 2) return;
        const bool ok = (o >= 0) & (o < t.nrows);                                    // wave-uniform
        if (!ok) return;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            char* base = reinterpret_cast<char*>(a.out) + t.img + 2 * (size_t)plane_g + (size_t)pl * (plane_g / 2) + (size_t)t.y(o) * a.W * 8;
            h8 rec[Gm::PIECES];
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j)
                rec[j] = *reinterpret_cast<const h8*>(tout + oslot * Gm::OUT_SLOT + (2 + pl) * Gm::OUT_PLANE + (64 * j + lane) * 16);
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j) {
                if (FULLW || col_ok[j]) *reinterpret_cast<bf_u2*>(base + (col_off[j] >> 1)) = bf_h3c_encode8(rec[j]);
            }
        }
    }

    // output row o (image row y0 + o) from staging slot `oslot` to global memory
    __device__ __forceinline__ void store_row(const H3VTile& t, const int o, const int oslot) const
    {
        if (H3V_ABLATE & 2) return;
        const bool ok = (o >= 0) & (o < t.nrows);                                    // wave-uniform
        if (!ok) return;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int plane = plane0 + pl;
            char* base = reinterpret_cast<char*>(a.out) + t.img + (size_t)plane * plane_g + (size_t)t.y(o) * a.W * 16;
            h8 rec[Gm::PIECES];
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j)
                rec[j] = *reinterpret_cast<const h8*>(tout + oslot * Gm::OUT_SLOT + plane * Gm::OUT_PLANE + (64 * j + lane) * 16);
#pragma unroll
            for (int j = 0; j < Gm::PIECES; ++j) {
                if (FULLW || col_ok[j]) *reinterpret_cast<h8*>(base + col_off[j]) = rec[j];
            }
        }
    }
};

template <bool FULLW>
__global__ __launch_bounds__(H3VGeom::NT, 3) void fused_block_h3v_kernel(FusedH3Args a)
{
    using Gm = H3VGeom;
    extern __shared__ __attribute__((aligned(16))) char h3v_lds[];
    char* tin = h3v_lds;                                       // [4 planes][NRI rows][WMAX + 2 columns][8 f16]
    char* tmid = h3v_lds + Gm::IN_BYTES;                       // [4 planes][NRM rows][WMAX + 2 columns][8 f16]
    char* tout = tmid + Gm::MID_BYTES;                         // [NRO rows][4 planes][WMAX columns][8 f16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
#if H3V_ABLATE & 128
    const int role = (wave >> 2) == 2 ? 2 : 1 - (wave >> 2);       // experiment: conv2 on the older waves
#else
    const int role = wave >> 2;
#endif
    const int rw = wave & 3;
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;      // bytes per global plane

    // columns 0 and W+1.. of the input and mid rings are the zero padding: cleared once, never written (DMA and conv1
    // write ring columns 1 .. WMAX only, and zeros where the image is narrower)
    for (int i = tid * 16; i < Gm::LDS_BYTES; i += Gm::NT * 16) *reinterpret_cast<f32x4*>(h3v_lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int c0 = 64 * rw + n;                                 // matrix waves: lane's image column in group 0
#if H3V_ABLATE & 32
    unsigned long long stamp_sum[Gm::NSTAMP] = {0, 0, 0, 0}, stamp_prev, real0;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0), "=s"(stamp_prev)::"memory");
#endif
    if (role == 0) {
        __builtin_amdgcn_s_setprio(1);
        H3VRoleA<FULLW> A{a, tin, tmid};
#pragma unroll
        for (int i = 0; i < 13; ++i) A.w[i] = reinterpret_cast<const h8*>(a.w1r)[h3v_wimage(a, i) * 64 + lane];
        A.inv_s = a.aux[0];
        A.relu_floor = a.act1_relu ? 0.f : -__builtin_inff();
        // taps dx = 0,1 of image column c are ring columns c, c+1 (ring column = image column + 1, centre tap dx = 1)
        A.rp = (q & 1) * Gm::IN_PLANE + (c0 + (q >> 1)) * 16;
        A.rs = ((q & 1) + 2 * (q >> 1)) * Gm::IN_PLANE + (c0 + 2) * 16;
        A.wr = (q >> 1) * Gm::MID_PLANE + (c0 + 1) * 16 + (q & 1) * 8;       // channels 4q .. 4q+3: plane q>>1, half-record q&1
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) A.lane_scale[g] = (c0 + 16 * g < a.W) ? A.inv_s : 0.f;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) A.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // weight / scale loads

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3VTile t = h3v_tile(a, ti);
            h3v_barrier();                                       // prologue: input row 0 has landed
            const int nsteps = t.nrows + 6;
            int islot = 0;                                       // s mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3V_STEP_A(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if (s < t.nrows + 4) A.template step<PH>(t, s, islot);    /* input rows 0 .. nrows+3 */   \
                    islot = h3v_wrap(islot + 1, Gm::NRI);                                                     \
                    H3V_STAMP(0);                                                                             \
                    h3v_barrier();                                                                            \
                    H3V_STAMP(2);                                                                             \
                } while (0)
                H3V_STEP_A(0); H3V_STEP_A(1); H3V_STEP_A(2); H3V_STEP_A(3); H3V_STEP_A(4); H3V_STEP_A(5);
#undef H3V_STEP_A
            }
            h3v_barrier();
        }
    } else if (role == 1) {
        __builtin_amdgcn_s_setprio(H3V_PRIO_B);
        H3VRoleB Bv{a, tin, tmid, tout};
#pragma unroll
        for (int i = 0; i < 13; ++i) Bv.w[i] = reinterpret_cast<const h8*>(a.w2r)[h3v_wimage(a, i) * 64 + lane];
        Bv.inv_s2 = a.aux[48];
        Bv.shs = *reinterpret_cast<const f32x4*>(a.aux + 32 + q * 4) * (1.0f / Bv.inv_s2);     // inv_s2 is a power of two: exact
        Bv.rp = (q & 1) * Gm::MID_PLANE + (c0 + (q >> 1)) * 16;
        Bv.rs = ((q & 1) + 2 * (q >> 1)) * Gm::MID_PLANE + (c0 + 2) * 16;
        Bv.rr = ((q & 1) + 2 * (q >> 1)) * Gm::IN_PLANE + (c0 + 1) * 16;
        Bv.wo = (q >> 1) * Gm::OUT_PLANE + c0 * 16 + (q & 1) * 8;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) Bv.acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));

        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3VTile t = h3v_tile(a, ti);
            h3v_barrier();                                       // prologue barrier
            const int nsteps = t.nrows + 6;
            int xslot = Gm::NRI - 1;                             // (s - 1) mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3V_STEP_B(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if ((s >= 3) & (s < t.nrows + 5)) Bv.template step<PH>(xslot);   /* mid rows 0 .. nrows+1 */ \
                    xslot = h3v_wrap(xslot + 1, Gm::NRI);                                                     \
                    H3V_STAMP(0);                                                                             \
                    h3v_barrier();                                                                            \
                    H3V_STAMP(2);                                                                             \
                } while (0)
                H3V_STEP_B(0); H3V_STEP_B(1); H3V_STEP_B(2); H3V_STEP_B(3); H3V_STEP_B(4); H3V_STEP_B(5);
#undef H3V_STEP_B
            }
            h3v_barrier();
        }
    } else {
        H3VRoleC<FULLW> Cv{a, tin, tout};
        const bool loader = rw < 2;
        Cv.plane0 = 2 * (rw & 1);
        Cv.lane = lane;
        Cv.plane_g = plane_g;
#pragma unroll
        for (int j = 0; j < Gm::PIECES; ++j) {
            Cv.col_off[j] = (unsigned)(64 * j + lane) * 16u;
            Cv.col_ok[j] = 64 * j + lane < a.W;
        }
        constexpr int INFLIGHT = 2 * 2 * Gm::PIECES;           // pieces of the two rows younger than the one awaited
        const bool lo_regs = a.compact && Cv.plane0 == 2;      // compact layout: this wave's planes are the fp8 lo planes
        for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {
            const H3VTile t = h3v_tile(a, ti);
            if (loader && lo_regs) {
                // prologue: rows 0 and 1 decoded into their slots, row 2 requested
                Cv.lo_request(t, 0); Cv.lo_commit(0);
                Cv.lo_request(t, 1); Cv.lo_commit(1);
                Cv.lo_request(t, 2);
            } else if (loader) {
                // prologue: rows 0 .. PD-1 requested, row 0 landed
#pragma unroll
                for (int r = 0; r < Gm::PD; ++r) Cv.dma_row(t, r, r);
                __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3V_ABLATE & 1) ? 0 : INFLIGHT));
            }
            h3v_barrier();
            const int nsteps = t.nrows + 6;
            int dslot = Gm::PD;                                  // (s + PD) mod NRI
            for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {
#define H3V_STEP_C(PH)                                                                                        \
                do {                                                                                          \
                    const int s = s0 + PH;                                                                    \
                    if (loader && lo_regs) {                                                                  \
                        Cv.lo_commit(h3v_wrap(dslot + Gm::NRI - 1, Gm::NRI));    /* row s+2, requested a step ago */ \
                        Cv.lo_request(t, s + Gm::PD);                                                         \
                        H3V_STAMP(0);                                                                         \
                    } else if (loader) {                                                                      \
                        Cv.dma_row(t, s + Gm::PD, dslot);                                                     \
                        H3V_STAMP(0);                                                                         \
                        /* row s+1 (requested two steps ago) has landed <=> at most the two younger rows are outstanding */ \
                        __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3V_ABLATE & 1) ? 0 : INFLIGHT));                 \
                        H3V_STAMP(1);                                                                         \
                    } else if (lo_regs) {                                                                     \
                        Cv.lo_store_row(t, s - 6, PH % Gm::NRO);                                              \
                        H3V_STAMP(0);                                                                         \
                    } else {                                                                                  \
                        Cv.store_row(t, s - 6, PH % Gm::NRO);    /* staged by B in step s-1: (s - 6) mod 2 */ \
                        H3V_STAMP(0);                                                                         \
                    }                                                                                         \
                    dslot = h3v_wrap(dslot + 1, Gm::NRI);                                                     \
                    h3v_barrier();                                                                            \
                    H3V_STAMP(2);                                                                             \
                } while (0)
                H3V_STEP_C(0); H3V_STEP_C(1); H3V_STEP_C(2); H3V_STEP_C(3); H3V_STEP_C(4); H3V_STEP_C(5);
#undef H3V_STEP_C
            }
            // the rows requested past the band's end (zero-line reads into dead slots) must not land in the next band's rows
            __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
            h3v_barrier();
        }
    }
#if H3V_ABLATE & 32
    {
        unsigned long long real1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");
        stamp_sum[3] = real1 - real0;                               // 100 MHz ticks: clock = cycles / ticks * 100 MHz
    }
    if (a.dbg && lane == 0) {
        for (int k = 0; k < Gm::NSTAMP; ++k) a.dbg[((size_t)blockIdx.x * Gm::NW + wave) * 8 + k] = stamp_sum[k];
    }
#endif
}

// bands: every image is cut into ceil(H / rows) bands of `rows` rows; one band = one unit of work of a workgroup.
// rows is chosen so that the slowest CU (ceil(bands / CUs) bands of rows + 10 steps each) finishes earliest.
static int h3v_rows_per_tile(const int B, const int H, const int cus)
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
