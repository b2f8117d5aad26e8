// TWO fused residual blocks per launch, split-f16 arithmetic, row streaming on 128-column STRIPS (fused_block2_h3w_kernel).
//
// Reference semantics (bfcnn/backbone_blocks.py:167-246, the loop over the blocks of a resnet level, with the inference
// BatchNormalization folded), two iterations of it per launch:
//     x1 = x0 + scale_a * conv2a(act(conv1a x0)) + shift_a ;   x2 = x1 + scale_b * conv2b(act(conv1b x1)) + shift_b
// Same arithmetic, weight images (pack_h3_kernel's row-streaming images) and split-planar activation layout as
// fused_block_h3v_kernel (fused_h3v.hip); x1 and both intermediate activations only ever exist in LDS, so a pair of blocks
// moves 128 B per pixel through HBM where two launches of the one-block kernel move 256.
//
// The one-block kernel keeps whole 256-column rows in its rings (16.5 KB per ring row) and has no room for a second block.
// Here a workgroup owns a STRIP of 128 output columns and walks down a band of rows, one image row per step, through a
// chain of four convolutions.  Every convolution works on the same GRID of 144 columns = nine 16-pixel MFMA groups that
// contains the strip plus its halo (x0 needs 4 columns beyond the strip on an interior side, the first intermediate 3, x1 2,
// the second intermediate 1); at an image edge the grid starts / ends ON the edge, so a 256-column image is two strips with
// grids [0, 144) and [112, 256): 9 groups per row and convolution where 8 are output (12.5 % more matrix work, 6 % more
// input bytes).  Columns of the grid that no output depends on hold finite garbage.
//
//   * 12 waves = 4 roles x 3 waves, three matrix waves per SIMD; wave k of a role owns grid columns [48k, 48k+48) = three MFMA
//     groups and holds ONE kernel's weights (52 VGPRs).  A1 = conv1a, B1 = conv2a (+ residual x0, folded BN) -> x1 ring,
//     A2 = conv1b, B2 = conv2b (+ residual x1) -> staging rows;
//   * the memory instructions ride on matrix waves: the A2 waves request x0 row s+3 by LDS-DMA (four 1-KiB pieces each),
//     the A1 waves store the staged output row (loads and stores on DIFFERENT waves: `s_waitcnt vmcnt(N)` counts both and
//     stores retire early, see fused_h3v.hip);
//   * LDS rings (ring column = grid column + 1; ring columns 0 and 145 are never written: the zero padding at an image
//     edge): x0 5 rows, first intermediate 2, x1 3, second intermediate 2, staging 2 -> 131 KB;
//   * step s, band-relative rows (x0 row k is consumed at step k+4): A1 turns x0 row s-4 into its three vertical-tap
//     contributions and completes intermediate row s-5; B1 consumes intermediate row s-6 and completes x1 row s-7 (its
//     accumulator starts as the residual (s2 I) x [x_hi | x_lo] on top of the folded shift); A2 consumes x1 row s-8 ->
//     second intermediate row s-9; B2 consumes it in step s+1 and completes output row s-11 in step s; stored in s+1.
//     ONE barrier per step, nrows + 12 steps per band;
//   * rows and columns outside the image are forced to zero in every ring (they are the NEXT convolution's zero padding,
//     not values computed from padded input).
#include "h3v_core.h"

// wave priorities (s_setprio) per role; H3W_PRIO_SET picks a preset for A/B builds: 1 = younger roles higher (0,1,2,3),
// 2 = older roles higher (3,2,1,0), 3 = second convolutions high (0,2,0,2), 4 = first convolutions high (2,0,2,0)
#ifndef H3W_PRIO_SET
#define H3W_PRIO_SET 0
#endif
#if H3W_PRIO_SET == 1
#define H3W_PRIO_A1 0
#define H3W_PRIO_B1 1
#define H3W_PRIO_A2 2
#define H3W_PRIO_B2 3
#elif H3W_PRIO_SET == 2
#define H3W_PRIO_A1 3
#define H3W_PRIO_B1 2
#define H3W_PRIO_A2 1
#define H3W_PRIO_B2 0
#elif H3W_PRIO_SET == 3
#define H3W_PRIO_A1 0
#define H3W_PRIO_B1 2
#define H3W_PRIO_A2 0
#define H3W_PRIO_B2 2
#elif H3W_PRIO_SET == 4
#define H3W_PRIO_A1 2
#define H3W_PRIO_B1 0
#define H3W_PRIO_A2 2
#define H3W_PRIO_B2 0
#endif
#ifndef H3W_PRIO_A1
#define H3W_PRIO_A1 1
#endif
#ifndef H3W_PRIO_B1
#define H3W_PRIO_B1 1
#endif
#ifndef H3W_PRIO_A2
#define H3W_PRIO_A2 1
#endif
#ifndef H3W_PRIO_B2
#define H3W_PRIO_B2 1
#endif

struct H3WGeom {
    static constexpr int NG = 9, GW = 16 * NG;         // grid: nine 16-column groups
    static constexpr int G = 3;                        // groups per matrix wave
    static constexpr int NR = 3, NW = 12, NT = 768;    // waves per role, per workgroup, threads
    static constexpr int SW = 128;                     // output columns of a strip
    static constexpr int PITCH = (GW + 2) * 16;        // bytes per plane-row of a ring; ring column = grid column + 1
    static constexpr int NRX0 = 5, NRM = 2, NRX1 = 3, NRO = 2;      // ring depths (rows)
    static constexpr int PD = 3;                       // DMA distance: x0 row s+3 is requested in step s, awaited at the end of step s+1
    static constexpr int UNROLL = 6;                   // steps per loop iteration: slots mod 2 / mod 3 and the accumulator rotation static
    static constexpr int X0_PLANE = (NRX0 * PITCH + 255) / 256 * 256, M_PLANE = (NRM * PITCH + 255) / 256 * 256;
    static constexpr int X1_PLANE = (NRX1 * PITCH + 255) / 256 * 256;
    static constexpr int OUT_PLANE = GW * 16, OUT_SLOT = 4 * OUT_PLANE;       // staging rows are indexed by GRID column too
    static constexpr int X0_OFF = 0, M1_OFF = X0_OFF + 4 * X0_PLANE, X1_OFF = M1_OFF + 4 * M_PLANE, M2_OFF = X1_OFF + 4 * X1_PLANE;
    static constexpr int OUT_OFF = M2_OFF + 4 * M_PLANE, LDS_BYTES = OUT_OFF + NRO * OUT_SLOT;
    static constexpr int LEAD = 12;                    // steps from the first x0 row of a band to the store of its first output row
    static constexpr int NSTAMP = 4;
    static_assert(UNROLL % NRM == 0 && UNROLL % NRO == 0 && UNROLL % NRX1 == 0 && UNROLL % 3 == 0, "static slots");
    static_assert(NRX0 == PD + 2, "x0 ring: rows s-1 (residual), s (conv1a), s+1 (landed), s+2, s+3 (in flight)");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(3 * X0_PLANE + NRX0 * PITCH < 65536, "fragment offsets fit the 16-bit ds offset");
};

struct H3WTile {
    int nrows, b;                // rows of the band, image index
    size_t img;
    int ybase, ystep;            // image row of band-relative row k: ybase + ystep * k (a reversed band walks bottom-up)
    int X0, X1;                  // output columns [X0, X1)
    int G0;                      // image column of grid column 0
    int D0, D1;                  // x0 columns fetched: [D0, D1), inside the grid and the image
    __device__ __forceinline__ int y(const int k) const { return ybase + ystep * k; }
};

// unit t of a launch = (image, band of rows, strip); reverse_tiles: last unit first and bottom-up (consecutive launches
// alternate, so a launch starts on what the previous one wrote last -- still in the 256 MB Infinity Cache; walking up only
// mirrors the vertical taps: the weight images are loaded with dy -> 2 - dy)
__device__ __forceinline__ H3WTile h3w_tile(const FusedH3WArgs& a, const int t)
{
    H3WTile r;
    const int tt = a.reverse_tiles ? a.ntiles - 1 - t : t;
    const int sx = tt % a.nstrips, rest = tt / a.nstrips;
    const int b = rest / a.tiles_y, ty = rest - b * a.tiles_y;
    const int y0 = ty * a.rows_per_tile;
    r.nrows = min(a.rows_per_tile, a.H - y0);
    r.b = b;
    r.img = (size_t)b * a.H * a.W * 64;
    r.ybase = a.reverse_tiles ? y0 + r.nrows - 1 : y0;
    r.ystep = a.reverse_tiles ? -1 : 1;
    r.X0 = sx * H3WGeom::SW;
    r.X1 = min(a.W, r.X0 + H3WGeom::SW);
    r.G0 = min(max(r.X0 - 8, 0), max(a.W - H3WGeom::GW, 0));
    r.D0 = max(0, (r.X0 - 4) & ~7);
    r.D1 = min(a.W, (r.X1 + 4 + 7) & ~7);
    return r;
}

__device__ __forceinline__ int h3w_wimage(const FusedH3WArgs& a, const int i)
{
    return (a.reverse_tiles && i < 12) ? (2 - i / 4) * 4 + i % 4 : i;
}

// ---------------------------------------------------------------------------------------------------------------------
// role A: a block's first convolution (+ activation): ring row -> three vertical-tap contributions -> intermediate ring.
// State across steps: acc[g][3] (intermediate rows of steps s, s-1, s-2 modulo 3).
// ---------------------------------------------------------------------------------------------------------------------
template <int INP>                 // plane stride of the input ring
struct H3WRoleA {
    using Gm = H3WGeom;
    const char* tin;
    char* tmid;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's LDS byte offset in ring slot 0, group 0: pair fragment (hi planes), single fragment
    int wr;                      // lane's LDS byte offset of its 8-byte hi record in mid ring slot 0, group 0 (lo: + 2 planes)
    float inv_s, relu_floor;
    float lane_scale[Gm::G];     // inv_s where the lane's column is inside the image, else 0
    bool skip_last = false;      // timing experiment (H3V_ABLATE & 256): the role's third wave multiplies two groups only

    __device__ __forceinline__ void init(const void* wimg, const FusedH3WArgs& a, const float* aux, const int lane, const int gc0)
    {
        const int q = lane >> 4;
#pragma unroll
        for (int i = 0; i < 13; ++i) w[i] = reinterpret_cast<const h8*>(wimg)[h3w_wimage(a, i) * 64 + lane];
        inv_s = aux[0];
        relu_floor = a.act1_relu ? 0.f : -__builtin_inff();
        // taps dx = 0, 1 of grid column c are ring columns c, c + 1 (ring column = grid column + 1, centre tap dx = 1)
        rp = (q & 1) * INP + (gc0 + (q >> 1)) * 16;
        rs = ((q & 1) + 2 * (q >> 1)) * INP + (gc0 + 2) * 16;
        wr = (q >> 1) * Gm::M_PLANE + (gc0 + 1) * 16 + (q & 1) * 8;          // channels 4q .. 4q+3: plane q>>1, half-record q&1
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void set_tile(const H3WTile& t, const int W, const int gc0)
    {
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) lane_scale[g] = (t.G0 + gc0 + 16 * g < W) ? inv_s : 0.f;
    }

    __device__ __forceinline__ H3VFrag load(const int islot_bytes, const int g) const
    {
        H3VFrag f;
        const char* p = tin + rp + islot_bytes;
        f.ph = *reinterpret_cast<const h8*>(p + g * 256);
        f.pl = *reinterpret_cast<const h8*>(p + g * 256 + 2 * INP);
        f.s = *reinterpret_cast<const h8*>(tin + rs + islot_bytes + g * 256);
        return f;
    }

    __device__ __forceinline__ H3VEpi<true> epilogue(const int g, const int mslot, const f32x4 v, const bool rowok) const
    {
        H3VEpi<true> e;
        e.v = v;
        e.sc = rowok ? lane_scale[g] : 0.f;              // rows / columns outside the image are the next convolution's zero padding
        e.floor_ = relu_floor;
        e.p = tmid + wr + mslot * Gm::PITCH + g * 256;
        e.lo_off = 2 * Gm::M_PLANE;
        return e;
    }

    // the 15 MFMAs of group g; micro-ops of the previous group's epilogue after MFMA 2, 3, ...
    template <int J, class Epi>
    __device__ __forceinline__ void mfmas(const int g, const int a1, const int a2, const H3VFrag& cur, f32x4& c0, Epi* epi)
    {
        if constexpr (J < 15) {
            constexpr int k = J / 3, which = J % 3;
            if (!(H3V_ABLATE & 8) && !(skip_last && g == Gm::G - 1)) {
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
            mfmas<J + 1>(g, a1, a2, cur, c0, epi);
        }
    }

    // step s (s % UNROLL == PH): consumes the ring row at byte offset islot_bytes, completes the intermediate row of slot PH % 2
    template <int PH, class Hook>
    __device__ __forceinline__ void step(const int islot_bytes, const bool rowok, Hook& hook)
    {
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;      // accumulators of intermediate rows s, s-1, s-2
        constexpr int mslot = PH % Gm::NRM;
        H3VFrag cur = load(islot_bytes, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(islot_bytes, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if (g > 0) {
                H3VEpi<true> e = epilogue(g - 1, mslot, acc[g - 1][a2], rowok);
                mfmas<0>(g, a1, a2, cur, c0, &e);
            } else {
                mfmas<0>(g, a1, a2, cur, c0, (H3VEpi<true>*)nullptr);
                hook.after_first_group();
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        H3VEpi<true> e = epilogue(Gm::G - 1, mslot, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
        e.all();
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// role B: a block's second convolution + folded BN + residual.  B1 writes the x1 ring, B2 the staging rows.
// ---------------------------------------------------------------------------------------------------------------------
template <int RESP, int OUTP>      // plane strides of the ring the residual comes from and of the ring / staging row written
struct H3WRoleB {
    using Gm = H3WGeom;
    const char* tmid;
    const char* tres;
    char* tout;
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's LDS byte offset in mid ring slot 0, group 0: pair / single fragments
    int rr;                      // lane's LDS byte offset of the residual operand [x_hi | x_lo] in its ring's slot 0, group 0
    int wo;                      // lane's LDS byte offset of its 8-byte hi record in output slot 0, group 0 (lo: + 2 planes)
    float inv_s2;
    f32x4 shs;                   // folded BN shift of the lane's four channels times s2 (the accumulators' scale)
    float lane_scale[Gm::G];
    bool skip_last = false;      // timing experiment (H3V_ABLATE & 256)

    __device__ __forceinline__ void init(const void* wimg, const FusedH3WArgs& a, const float* aux, const int lane, const int gc0,
                                         const bool out_is_ring)
    {
        const int q = lane >> 4;
#pragma unroll
        for (int i = 0; i < 13; ++i) w[i] = reinterpret_cast<const h8*>(wimg)[h3w_wimage(a, i) * 64 + lane];
        inv_s2 = aux[48];
        shs = *reinterpret_cast<const f32x4*>(aux + 32 + q * 4) * (1.0f / inv_s2);     // inv_s2 is a power of two: exact
        rp = (q & 1) * Gm::M_PLANE + (gc0 + (q >> 1)) * 16;
        rs = ((q & 1) + 2 * (q >> 1)) * Gm::M_PLANE + (gc0 + 2) * 16;
        rr = ((q & 1) + 2 * (q >> 1)) * RESP + (gc0 + 1) * 16;
        wo = (q >> 1) * OUTP + (gc0 + (out_is_ring ? 1 : 0)) * 16 + (q & 1) * 8;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void set_tile(const H3WTile& t, const int W, const int gc0)
    {
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) lane_scale[g] = (t.G0 + gc0 + 16 * g < W) ? inv_s2 : 0.f;
    }

    __device__ __forceinline__ H3VFrag load(const int mslot, const int g) const
    {
        H3VFrag f;
        const int o = mslot * Gm::PITCH + g * 256;
        f.ph = *reinterpret_cast<const h8*>(tmid + rp + o);
        f.pl = *reinterpret_cast<const h8*>(tmid + rp + o + 2 * Gm::M_PLANE);
        f.s = *reinterpret_cast<const h8*>(tmid + rs + o);
        return f;
    }
    __device__ __forceinline__ h8 load_res(const int xslot_bytes, const int g) const
    {
        return *reinterpret_cast<const h8*>(tres + rr + xslot_bytes + g * 256);
    }

    // the accumulator already holds s2 * (scale * conv2 + x + shift) (the shift is the C operand of the row's first MFMA)
    __device__ __forceinline__ H3VEpi<false> epilogue(const int g, const int oslot_bytes, const f32x4 accv, const bool rowok) const
    {
        H3VEpi<false> e;
        e.v = accv;
        e.sc = rowok ? lane_scale[g] : 0.f;
        e.floor_ = 0.f;
        e.p = tout + wo + oslot_bytes + g * 256;
        e.lo_off = 2 * OUTP;
        return e;
    }

    // the 1 + 15 MFMAs of group g; micro-ops of the previous group's epilogue after MFMA 2, 3, ...
    template <int J, class Epi>
    __device__ __forceinline__ void mfmas(const int g, const int a1, const int a2, const H3VFrag& cur, const h8 xr, f32x4& c0, Epi* epi)
    {
        if constexpr (J < 16) {
            if (!(H3V_ABLATE & 4) && !(skip_last && g == Gm::G - 1)) {
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
            mfmas<J + 1>(g, a1, a2, cur, xr, c0, epi);
        }
    }

    // step s (s % UNROLL == PH): consumes the intermediate row completed in step s-1, starts an output row with the residual at
    // byte offset xslot_bytes of its ring, completes the output row of byte offset oslot_bytes
    template <int PH, class Hook>
    __device__ __forceinline__ void step(const int xslot_bytes, const int oslot_bytes, const bool rowok, Hook& hook)
    {
        constexpr int mslot = (PH + 1) % Gm::NRM;                              // (s - 1) mod 2
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;
        H3VFrag cur = load(mslot, 0);
        h8 xr = load_res(xslot_bytes, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            h8 xn;
            if (g + 1 < Gm::G) {
                nx = load(mslot, g + 1);
                xn = load_res(xslot_bytes, g + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0;
            if (g > 0) {
                H3VEpi<false> e = epilogue(g - 1, oslot_bytes, acc[g - 1][a2], rowok);
                mfmas<0>(g, a1, a2, cur, xr, c0, &e);
            } else {
                mfmas<0>(g, a1, a2, cur, xr, c0, (H3VEpi<false>*)nullptr);
                hook.after_first_group();
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) {
                cur = nx;
                xr = xn;
            }
        }
        H3VEpi<false> e = epilogue(Gm::G - 1, oslot_bytes, bf_acc_ready(acc[Gm::G - 1][a2]), rowok);
        e.all();
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Memory work rides on matrix waves (H3WMem<KIND> is the hook of a role's steps; KIND 0 = none).
// KIND 1, the loader: LDS-DMA of the x0 rows.  Wave j of its role moves piece j (grid columns [64j, 64j+64), the third piece 16
// columns) of all four planes: four wave-instructions per step, EVERY step (rows outside the band / the image read the zero
// line), so that the vmcnt of the wait is a constant.
// KIND 2, the storer: the staged output row -> global memory.  Eight 1-KiB pieces (plane, half) per row; wave j moves pieces
// 3j .. 3j+2.  The staging reads are issued at the start of the step, the stores behind the first group's MFMAs.
// Loads and stores sit on DIFFERENT waves: `s_waitcnt vmcnt(N)` counts both and stores retire early (fused_h3v.hip).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef H3W_MEM_ROLES
#define H3W_MEM_ROLES 0          // which roles carry {loader, storer}: 0 = {A2, A1}, 1 = {A1, B1}, 2 = {B1, A1}, 3 = {B2, A1}
#endif
constexpr int h3w_mem_kind(const int role)
{
    constexpr int loader[4] = {2, 0, 1, 3}, storer[4] = {0, 1, 0, 0};
    return role == loader[H3W_MEM_ROLES] ? 1 : (role == storer[H3W_MEM_ROLES] ? 2 : 0);
}

template <int KIND>
struct H3WMem {
    __device__ __forceinline__ H3WMem(const FusedH3WArgs&, char*, const char*, int, unsigned) {}
    __device__ __forceinline__ void set_tile(const H3WTile&, int) {}
    __device__ __forceinline__ void prologue(const H3WTile&) {}
    __device__ __forceinline__ void begin(const H3WTile&, int, int) {}
    __device__ __forceinline__ void after_first_group() {}
    __device__ __forceinline__ void end() {}
    __device__ __forceinline__ void finish() {}
};

template <>
struct H3WMem<1> {
    using Gm = H3WGeom;
    const FusedH3WArgs& a;
    char* tin;
    int piece;
    unsigned plane_g;
    unsigned col_off;            // lane's byte offset inside a plane-row of the image
    bool active;                 // lane's column is fetched: inside [D0, D1) and the piece
    int dslot;                   // (s + PD) mod 5

    __device__ __forceinline__ H3WMem(const FusedH3WArgs& a_, char* tx0, const char*, const int rw, const unsigned pg)
        : a(a_), tin(tx0), piece(rw), plane_g(pg), col_off(0), active(false), dslot(0) {}
    __device__ __forceinline__ void set_tile(const H3WTile& t, const int lane)
    {
        const int c = t.G0 + 64 * piece + lane;
        active = (c >= t.D0) & (c < t.D1) & (64 * piece + lane < Gm::GW);
        col_off = (unsigned)c * 16u;
    }
    // x0 ring row r (band-relative row r - 4) into ring slot `slot`
    __device__ __forceinline__ void dma_row(const H3WTile& t, const int r, const int slot) const
    {
        if (H3V_ABLATE & 1) return;
        const int y = t.y(r - 4);
        const bool ok = (y >= 0) & (y < a.H) & (r < t.nrows + 8);                   // wave-uniform
#pragma unroll
        for (int plane = 0; plane < 4; ++plane) {
            const char* src = ok ? reinterpret_cast<const char*>(a.in) + t.img + (size_t)plane * plane_g + (size_t)y * a.W * 16 + col_off
                                 : reinterpret_cast<const char*>(a.zeros);
            char* dst = tin + plane * Gm::X0_PLANE + slot * Gm::PITCH + (1 + 64 * piece) * 16;
            if (active)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    }
    static constexpr int INFLIGHT = 2 * 4;             // pieces of the two rows younger than the one awaited
    __device__ __forceinline__ void wait_landed() const { __builtin_amdgcn_s_waitcnt(h3_vmcnt((H3V_ABLATE & 1) ? 0 : INFLIGHT)); }
    // rows 0 .. PD-1 requested, row 0 landed
    __device__ __forceinline__ void prologue(const H3WTile& t)
    {
#pragma unroll
        for (int r = 0; r < Gm::PD; ++r) dma_row(t, r, r);
        wait_landed();
        dslot = Gm::PD;
    }
    __device__ __forceinline__ void begin(const H3WTile& t, const int s, int)
    {
        dma_row(t, s + Gm::PD, dslot);
        dslot = h3v_wrap(dslot + 1, Gm::NRX0);
    }
    __device__ __forceinline__ void after_first_group() {}
    __device__ __forceinline__ void end() { wait_landed(); }      // row s+1 (requested two steps ago) has landed
    // the rows requested past the band's end (zero-line reads into dead slots) must not land in the next band's rows
    __device__ __forceinline__ void finish() { __builtin_amdgcn_s_waitcnt(h3_vmcnt(0)); }
};

template <>
struct H3WMem<2> {
    using Gm = H3WGeom;
    const FusedH3WArgs& a;
    const char* tout;
    int first, np;               // pieces [first, first + np)
    unsigned plane_g;
    int st_off[3];               // lane's byte offset inside a staging slot
    unsigned g_off[3];           // lane's byte offset inside the image (plane + column)
    bool okc[3];
    h8 rec[3];
    bool have;
    char* grow;

    __device__ __forceinline__ H3WMem(const FusedH3WArgs& a_, char*, const char* tout_, const int rw, const unsigned pg)
        : a(a_), tout(tout_), first(3 * rw), np(rw < 2 ? 3 : 2), plane_g(pg), have(false), grow(nullptr) {}
    __device__ __forceinline__ void set_tile(const H3WTile& t, const int lane)
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = first + i, plane = p >> 1, c = t.X0 + 64 * (p & 1) + lane;
            okc[i] = (i < np) & (c < t.X1);
            st_off[i] = plane * Gm::OUT_PLANE + (c - t.G0) * 16;
            g_off[i] = (unsigned)plane * plane_g + (unsigned)c * 16u;
        }
    }
    __device__ __forceinline__ void prologue(const H3WTile&) {}
    // step s (phase PH = s mod 6): output row s - 12, staged by B2 in step s-1
    __device__ __forceinline__ void begin(const H3WTile& t, const int s, const int PH)
    {
        const int k = s - Gm::LEAD, oslot = (PH + 1) % Gm::NRO;
        have = (k >= 0) & (k < t.nrows) & !(H3V_ABLATE & 2);                          // wave-uniform
        if (!have) return;
        grow = reinterpret_cast<char*>(a.out) + t.img + (size_t)t.y(k) * a.W * 16;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < np) rec[i] = *reinterpret_cast<const h8*>(tout + oslot * Gm::OUT_SLOT + st_off[i]);
    }
    __device__ __forceinline__ void after_first_group()
    {
        if (!have) return;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (okc[i]) *reinterpret_cast<h8*>(grow + g_off[i]) = rec[i];
    }
    __device__ __forceinline__ void end() {}
    __device__ __forceinline__ void finish() {}
};

// KIND 3, the head storer (last pair of a network, FusedH3WArgs::head_wh): instead of storing the staged row, waves 0 and 1 of the
// role take one pixel per lane (64 columns each), read its 16 channels (hi + lo: four 16-byte records), multiply by the
// premultiplied 16 x 3 head matrix and store tanh(2 h) * 0.51 denormalised, cropped, [rounded to uint8]: 3 bytes instead of 64
// per pixel leave the chip and the head kernel's pass over the activation disappears.
template <>
struct H3WMem<3> {
    using Gm = H3WGeom;
    const FusedH3WArgs& a;
    const char* tout;
    int half;                    // columns [64 half, 64 half + 64) of the strip; >= 2: nothing to do
    int st_off;                  // lane's byte offset inside plane 0 of a staging slot
    int col;                     // lane's image column
    bool okc;

    __device__ __forceinline__ H3WMem(const FusedH3WArgs& a_, char*, const char* tout_, const int rw, const unsigned)
        : a(a_), tout(tout_), half(rw), st_off(0), col(0), okc(false) {}
    __device__ __forceinline__ void set_tile(const H3WTile& t, const int lane)
    {
        col = t.X0 + 64 * half + lane;
        okc = (half < 2) & (col < t.X1) & (col < a.Wo);
        st_off = (col - t.G0) * 16;
    }
    __device__ __forceinline__ void prologue(const H3WTile&) {}
    __device__ __forceinline__ void begin(const H3WTile& t, const int s, const int PH)
    {
        const int k = s - Gm::LEAD, oslot = (PH + 1) % Gm::NRO;
        bool have = (half < 2) & (k >= 0) & (k < t.nrows) & !(H3V_ABLATE & 2);        // wave-uniform
        if (have) have = t.y(k) < a.Ho;
        if (!have) return;
        char* orow = reinterpret_cast<char*>(a.head_out) + ((size_t)t.b * a.Ho + (size_t)t.y(k)) * a.Wo * 3 * (a.head_u8 ? 1 : 4);
        const char* src = tout + oslot * Gm::OUT_SLOT + st_off;
        // the whole head sits HERE, in front of the role's step (not behind its first group's MFMAs): no fragment is live yet
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const h8 hi = *reinterpret_cast<const h8*>(src + hf * Gm::OUT_PLANE), lo = *reinterpret_cast<const h8*>(src + (2 + hf) * Gm::OUT_PLANE);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = (float)hi[c] + (float)lo[c];
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.head_wh + (hf * 8 + c) * 4);
                h0 = fmaf(v, w.x, h0);
                h1 = fmaf(v, w.y, h1);
                h2 = fmaf(v, w.z, h2);
            }
        }
        // an activation left the f16 range somewhere in the split-f16 blocks: inf / NaN propagate to here (0 * inf = NaN: a zero
        // head weight does not hide one)
        const bool finite = fabsf(h0) <= 3.0e38f && fabsf(h1) <= 3.0e38f && fabsf(h2) <= 3.0e38f;
        if (!finite && a.status) atomicOr(a.status, BF_STATUS_F16_RANGE);
        float r[3] = {h0, h1, h2};
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            // tanh(2h) = 1 - 2 / (exp(4h) + 1): a handful of instructions (libm tanhf: ~30)
            float v = (1.0f - 2.0f / (__expf(4.0f * r[o]) + 1.0f)) * 0.51f;
            if (a.denormalize) v = (fminf(fmaxf(v, -0.5f), 0.5f) + 0.5f) * (a.v_max - a.v_min) + a.v_min;
            r[o] = v;
        }
        if (!okc) return;
        if (a.head_u8) {
            unsigned char* p = reinterpret_cast<unsigned char*>(orow) + (size_t)col * 3;
#pragma unroll
            for (int o = 0; o < 3; ++o) p[o] = (unsigned char)fminf(fmaxf(rintf(r[o]), 0.f), 255.f);      // rintf = round-half-even
        } else {
            float* p = reinterpret_cast<float*>(orow) + (size_t)col * 3;
#pragma unroll
            for (int o = 0; o < 3; ++o) p[o] = r[o];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void after_first_group() {}
    __device__ __forceinline__ void end() {}
    __device__ __forceinline__ void finish() {}
};

template <bool HEAD>
__global__ __launch_bounds__(H3WGeom::NT, 3) void fused_block2_h3w_kernel(FusedH3WArgs a)
{
    using Gm = H3WGeom;
    extern __shared__ __attribute__((aligned(16))) char h3w_lds[];
    char* tx0 = h3w_lds + Gm::X0_OFF;                           // [4 planes][5 rows][146 columns][8 f16]
    char* tm1 = h3w_lds + Gm::M1_OFF;                           // [4 planes][2 rows][146][8]
    char* tx1 = h3w_lds + Gm::X1_OFF;                           // [4 planes][3 rows][146][8]
    char* tm2 = h3w_lds + Gm::M2_OFF;                           // [4 planes][2 rows][146][8]
    char* tout = h3w_lds + Gm::OUT_OFF;                         // [2 rows][4 planes][144 columns][8 f16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15;
    const int role = wave / Gm::NR, rw = wave - role * Gm::NR;
    const unsigned plane_g = (unsigned)a.H * (unsigned)a.W * 16u;      // bytes per global plane
    const int gc0 = 16 * Gm::G * rw + n;                        // lane's grid column in its wave's group 0

    // ring columns 0 and 145 are the zero padding at an image edge: cleared once, never written
    for (int i = tid * 16; i < Gm::LDS_BYTES; i += Gm::NT * 16) *reinterpret_cast<f32x4*>(h3w_lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

#if H3V_ABLATE & 32
    unsigned long long stamp_sum[Gm::NSTAMP] = {0, 0, 0, 0}, stamp_prev, real0;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0), "=s"(stamp_prev)::"memory");
#endif

#define H3W_ROW_IN_IMAGE(k) ((t.y(k) >= 0) & (t.y(k) < a.H))
// one band of one strip: R = the matrix role object, M = its memory hook, ACTIVE(s) = the role has a row in step s,
// STEP(PH) = the role's step call.  Every role runs the same number of barriers.
#define H3W_BAND(R, M, ACTIVE, STEP)                                                                          \
    for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {                                               \
        const H3WTile t = h3w_tile(a, ti);                                                                    \
        R.set_tile(t, a.W, gc0);                                                                              \
        M.set_tile(t, lane);                                                                                  \
        M.prologue(t);                                                                                        \
        h3v_barrier();                                       /* x0 row 0 has landed */                        \
        const int nsteps = t.nrows + Gm::LEAD;                                                                \
        int slot5 = 0;                                       /* s mod 5 */                                    \
        for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {                                                     \
            H3W_ONE(0, M, ACTIVE, STEP); H3W_ONE(1, M, ACTIVE, STEP); H3W_ONE(2, M, ACTIVE, STEP);            \
            H3W_ONE(3, M, ACTIVE, STEP); H3W_ONE(4, M, ACTIVE, STEP); H3W_ONE(5, M, ACTIVE, STEP);            \
        }                                                                                                     \
        M.finish();                                                                                           \
        h3v_barrier();                                                                                        \
    }
#define H3W_ONE(PH, M, ACTIVE, STEP)                                                                          \
    do {                                                                                                      \
        const int s = s0 + PH;                                                                                \
        M.begin(t, s, PH);                                                                                    \
        if (ACTIVE(s)) { STEP(PH); }                                                                          \
        else M.after_first_group();                                                                           \
        slot5 = h3v_wrap(slot5 + 1, Gm::NRX0);                                                                \
        H3V_STAMP(0);                                                                                         \
        M.end();                                                                                              \
        H3V_STAMP(1);                                                                                         \
        h3v_barrier();                                                                                        \
        H3V_STAMP(2);                                                                                         \
    } while (0)

    if (role == 0) {
        // ---- A1: conv1a on the x0 ring (dynamic slot s mod 5) -> first intermediate ring
        __builtin_amdgcn_s_setprio(H3W_PRIO_A1);
        H3WRoleA<Gm::X0_PLANE> R;
        R.tin = tx0; R.tmid = tm1;
        R.init(a.w1r[0], a, a.aux[0], lane, gc0);
        R.skip_last = (H3V_ABLATE & 256) && rw == Gm::NR - 1;
        H3WMem<(HEAD && h3w_mem_kind(0) == 2) ? 3 : h3w_mem_kind(0)> M(a, tx0, tout, rw, plane_g);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));               // weight / scale loads
#define H3W_ACTIVE(s) ((s) < t.nrows + 8)
#define H3W_STEP(PH) R.template step<PH>(slot5 * Gm::PITCH, (s >= 2) & H3W_ROW_IN_IMAGE(s - 5), M)
        H3W_BAND(R, M, H3W_ACTIVE, H3W_STEP)
#undef H3W_ACTIVE
#undef H3W_STEP
    } else if (role == 1) {
        // ---- B1: conv2a + residual x0 (ring row s-1) -> x1 ring
        __builtin_amdgcn_s_setprio(H3W_PRIO_B1);
        H3WRoleB<Gm::X0_PLANE, Gm::X1_PLANE> R;
        R.tmid = tm1; R.tres = tx0; R.tout = tx1;
        R.init(a.w2r[0], a, a.aux[0], lane, gc0, true);
        R.skip_last = (H3V_ABLATE & 256) && rw == Gm::NR - 1;
        H3WMem<(HEAD && h3w_mem_kind(1) == 2) ? 3 : h3w_mem_kind(1)> M(a, tx0, tout, rw, plane_g);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
#define H3W_ACTIVE(s) (((s) >= 3) & ((s) < t.nrows + 9))
#define H3W_STEP(PH) R.template step<PH>(h3v_wrap(slot5 + Gm::NRX0 - 1, Gm::NRX0) * Gm::PITCH, (PH % Gm::NRX1) * Gm::PITCH, H3W_ROW_IN_IMAGE(s - 7), M)
        H3W_BAND(R, M, H3W_ACTIVE, H3W_STEP)
#undef H3W_ACTIVE
#undef H3W_STEP
    } else if (role == 2) {
        // ---- A2: conv1b on the x1 ring -> second intermediate ring
        __builtin_amdgcn_s_setprio(H3W_PRIO_A2);
        H3WRoleA<Gm::X1_PLANE> R;
        R.tin = tx1; R.tmid = tm2;
        R.init(a.w1r[1], a, a.aux[1], lane, gc0);
        R.skip_last = (H3V_ABLATE & 256) && rw == Gm::NR - 1;
        H3WMem<(HEAD && h3w_mem_kind(2) == 2) ? 3 : h3w_mem_kind(2)> M(a, tx0, tout, rw, plane_g);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
#define H3W_ACTIVE(s) (((s) >= 6) & ((s) < t.nrows + 10))
#define H3W_STEP(PH) R.template step<PH>(((PH + 2) % Gm::NRX1) * Gm::PITCH, H3W_ROW_IN_IMAGE(s - 9), M)
        H3W_BAND(R, M, H3W_ACTIVE, H3W_STEP)
#undef H3W_ACTIVE
#undef H3W_STEP
    } else {
        // ---- B2: conv2b + residual x1 -> staging rows
        __builtin_amdgcn_s_setprio(H3W_PRIO_B2);
        H3WRoleB<Gm::X1_PLANE, Gm::OUT_PLANE> R;
        R.tmid = tm2; R.tres = tx1; R.tout = tout;
        R.init(a.w2r[1], a, a.aux[1], lane, gc0, false);
        R.skip_last = (H3V_ABLATE & 256) && rw == Gm::NR - 1;
        H3WMem<(HEAD && h3w_mem_kind(3) == 2) ? 3 : h3w_mem_kind(3)> M(a, tx0, tout, rw, plane_g);
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
#define H3W_ACTIVE(s) (((s) >= 9) & ((s) < t.nrows + 11))
#define H3W_STEP(PH) R.template step<PH>(((PH + 1) % Gm::NRX1) * Gm::PITCH, (PH % Gm::NRO) * Gm::OUT_SLOT, true, M)
        H3W_BAND(R, M, H3W_ACTIVE, H3W_STEP)
#undef H3W_ACTIVE
#undef H3W_STEP
    }
#undef H3W_BAND
#undef H3W_ONE
#undef H3W_ROW_IN_IMAGE
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

static int h3w_nstrips(const int W) { return (W + H3WGeom::SW - 1) / H3WGeom::SW; }

// bands: every strip of every image is cut into ceil(H / rows) bands of `rows` rows; one band of one strip = one unit of work
// of a workgroup.  rows is chosen so that the slowest CU (ceil(units / CUs) units of rows + 14 steps each) finishes earliest.
static int h3w_rows_per_tile(const int B, const int H, const int nstrips, const int cus)
{
    int best = H;
    long best_cost = -1;
    for (int ty = 1; ty <= (H + 7) / 8; ++ty) {
        const int rows = (H + ty - 1) / ty;
        if ((H + rows - 1) / rows != ty) continue;
        const long tiles = (long)B * ty * nstrips;
        const long cost = ((tiles + cus - 1) / cus) * (rows + H3WGeom::LEAD + 2);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = rows;
        }
    }
    return best;
}

bool bf_fused_block2_h3w_supports(int H, int W) { return W >= 1 && H >= 1; }

hipError_t bf_launch_fused_block2_h3w(const FusedH3WArgs& args, hipStream_t s)
{
    using Gm = H3WGeom;
    FusedH3WArgs a = args;
    if (!a.zeros || !a.in || (!a.out && !a.head_wh) || !bf_fused_block2_h3w_supports(a.H, a.W)) return hipErrorInvalidValue;
    if ((int64_t)a.H * a.W * 64 >= ((int64_t)1 << 32)) return hipErrorInvalidValue;      // 32-bit in-image offsets
    const int cus = 256;
    a.nstrips = h3w_nstrips(a.W);
    a.rows_per_tile = h3w_rows_per_tile(a.B, a.H, a.nstrips, cus);
    a.tiles_y = (a.H + a.rows_per_tile - 1) / a.rows_per_tile;
    a.ntiles = a.B * a.tiles_y * a.nstrips;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    if (a.head_wh && (!a.head_out || a.Ho < 1 || a.Wo < 1 || a.Ho > a.H || a.Wo > a.W)) return hipErrorInvalidValue;
    void (*kernel)(FusedH3WArgs) = a.head_wh ? fused_block2_h3w_kernel<true> : fused_block2_h3w_kernel<false>;
    const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(kernel), Gm::LDS_BYTES);      // once per device
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Gm::NT), Gm::LDS_BYTES, s, a);
    return hipGetLastError();
}
