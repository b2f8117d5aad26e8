// Backward of one residual block in ONE kernel (bwd_block_h3t_kernel): split-f16 arithmetic, fp32 NHWC tensors, row streaming on
// 128-column strips (the decomposition of fused_block2_h3w_kernel, fused_h3w.hip), T_i RECOMPUTED instead of read.
//
// Reference: tape.gradient (bfcnn/train_loop.py:273-294) through one iteration of the block loop of
// bfcnn/backbone_blocks.py:167-246 -- conv_0, ReLU, conv_1, BatchNormalization, Add -- given g = dL/dA_{i+1}:
//     dc  = k1 g + k2 C_i + k3                      (BatchNorm backward; k from bn_bwd_finalize)          formed on load
//     T   = relu(conv_0 A_i)                        (what the forward pass computed and did not keep)      role F
//     dW1 = T^T (*) dc                              (weight gradient of conv_1)                            role W
//     dT  = dgrad_1(dc) * (T > 0)                                                                          role D2
//     dW0 = A_i^T (*) dT                            (weight gradient of conv_0)                            role W
//     dA' = dgrad_0(dT) + g                         (data gradient + the skip's)                           role D1 + epilogue
//     sum dA', sum dA' * C_{i-1}                    (the two reductions of the NEXT BatchNorm backward)    epilogue
// As two kernels (bwd3x3_h3_kernel<true, MASK>, <false, RES | BNBWD>) plus the forward's T_i that is 9 + 1 tensor passes at the
// ~5 TB/s the chip sustains on mixed streams (g, C_i, T_i read, dT written; A_i, dT, g, C_{i-1} read, dA' written; T_i written
// by the forward pass); here A_i, g, C_i, C_{i-1} are read once and dA' written once: 5 passes.  The price is matrix work: the
// recomputed conv_0 and the halo columns of a strip (144 for 128) -- the matrix pipe was idle in the HBM-bound kernels.
//
// A workgroup owns a strip of 128 output columns and walks down a band of rows, one image row per step.  All five convolution-
// shaped operators work on the same GRID of 144 columns = nine 16-pixel MFMA groups, G0 = max(X0 - 8, 0): dA' on the strip needs
// dT on 1 column more per side, dc and T on 2, A_i on 3.  Twelve waves, three per SIMD:
//   * F   (waves 0-2)  conv_0 + ReLU on the A ring: A row s-3 -> T row s-4 completes -> T ring (split f16)
//   * D2  (waves 3-5)  dgrad_1 on the dc ring: dc row s-4 -> dT row s-5 completes, masked by T row s-5 -> dT ring
//   * D1  (waves 6-8)  dgrad_0 on the dT ring: dT row s-6 -> raw dA' row s-7 completes -> staging ring (fp32)
//   * W   (waves 9-11) both weight gradients with the transposing LDS reads of train_bwd_h3.hip (pixel index along K), own
//     columns and own rows only, K chunks of 32 pixels: wave 9 = dW1 chunks 0-2, wave 10 = dW0 chunks 0-2, wave 11 = chunk 3 of
//     both.  dW1 pairs T row s-5 with dc rows s-4, s-5, s-6 (vertical taps 0, 1, 2); dW0 pairs dT row s-6 with A rows s-7..s-5.
//     Accumulators stay in registers for the whole launch and leave as per-workgroup partials (fixed order: bitwise reproducible).
// The memory instructions ride on the matrix waves, loads and stores on DIFFERENT waves (fused_h3v.hip on why):
//   * F and D2 waves load: one 16-pixel unit of a row per register quad, converted a step later into the hi / lo planes of the
//     A ring (row s-2) and of the dc ring (row s-3: dc = k1 g + k2 c + k3), re-loaded at once with the next row (the in-place
//     prefetch of train_fwd_h3t.hip);
//   * W waves run the epilogue of staged row s-8: + g, the two sums against C_{i-1}, written back to the staging ring;
//   * D1 waves store staged row s-9.
// LDS rings (ring column = grid column + 1; columns 0 and 145 are never written): A 6 rows, dc 4, T 2, dT 2 (split f16, plane
// stride = 128 mod 256 for the transposed reads) + staging 3 rows of fp32 = 157 KB.  ONE barrier per step, nrows + 9 steps per band.
#include "bf_common.h"
#include "h3_core.h"
#include "h3v_core.h"

typedef __fp16 hu_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

// timing-only ablations (tools/ablate_unit.sh train_bwd_h3t H3U_ABLATE ...; results are WRONG when any is set):
// 1 = no weight-gradient work (W waves: no transposed reads, no MFMAs), 2 = the loaders do not re-load (no global loads on F / D2),
// 4 = no epilogue of the staged row (no g / C_{i-1} loads), 8 = no global stores, 16 = no convolution steps on F / D2 / D1,
// 32 = the W waves' operand loads but none of their MFMAs, 64 = no LDS conversion in the loaders (loads only)
// 128 = s_memtime stamps: per wave the cycles of [matrix work | memory duty | barrier] (BwdBlockH3Args::dbg, tools/exp/stamp_bwd_block.py)
#ifndef H3U_ABLATE
#define H3U_ABLATE 0
#endif
#if H3U_ABLATE & 128
#define H3U_STAMP_DECL unsigned long long stamp_sum[4] = {0, 0, 0, 0}, stamp_prev, real0; \
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0), "=s"(stamp_prev)::"memory");
#define H3U_STAMP(k)                                                                                     \
    do {                                                                                                 \
        unsigned long long now_;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        stamp_sum[k] += now_ - stamp_prev;                                                               \
        stamp_prev = now_;                                                                               \
    } while (0)
#define H3U_STAMP_OUT                                                                                    \
    do {                                                                                                 \
        unsigned long long real1;                                                                        \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");                 \
        stamp_sum[3] = real1 - real0;                                                                    \
        if (a.dbg && lane == 0)                                                                          \
            for (int k = 0; k < 4; ++k) a.dbg[((size_t)blockIdx.x * 12 + (threadIdx.x >> 6)) * 8 + k] = stamp_sum[k]; \
    } while (0)
#else
#define H3U_STAMP_DECL
#define H3U_STAMP(k) do { } while (0)
#define H3U_STAMP_OUT do { } while (0)
#endif

// plane stride of a ring: the smallest size >= bytes that is 128 mod 256 (the transposed reads of a half-wave touch planes 0 and 1
// of 8 neighbouring pixels: train_bwd_h3.hip)
constexpr int h3u_plane(const int bytes) { return bytes / 256 * 256 + 128 >= bytes ? bytes / 256 * 256 + 128 : bytes / 256 * 256 + 384; }

struct H3UGeom {
    static constexpr int NG = 9, GW = 16 * NG;         // grid: nine 16-column groups
    static constexpr int G = 3;                        // groups per F / D wave
    static constexpr int NW = 12, NT = 768;
    static constexpr int SW = 128;                     // output columns of a strip
    static constexpr int PITCH = (GW + 2) * 16;        // 2336 bytes per plane-row of a ring
    static constexpr int NRA = 6, NRC = 4, NRT = 2, NRD = 4;               // ring depths (rows)
    static constexpr int UNROLL = 6;                   // steps per loop iteration: accumulator rotation and the 2- / 3-row rings static
    static constexpr int A_PLANE = h3u_plane(NRA * PITCH), C_PLANE = h3u_plane(NRC * PITCH), T_PLANE = h3u_plane(NRT * PITCH), D_PLANE = h3u_plane(NRD * PITCH);
    static constexpr int A_OFF = 0, C_OFF = A_OFF + 4 * A_PLANE, T_OFF = C_OFF + 4 * C_PLANE, D_OFF = T_OFF + 4 * T_PLANE;
    static constexpr int K_OFF = D_OFF + 4 * D_PLANE, LDS_BYTES = K_OFF + 256;    // K: k1 | k2 | k3 (48 floats)
    static constexpr int LEAD = 8;                     // steps of a band beyond its rows: the last row is stored in step nrows + 6, the weight
                                                       // gradient of conv_0 pairs A row nrows (step nrows + 7) with the last dT row
    static_assert(A_PLANE >= NRA * PITCH && C_PLANE >= NRC * PITCH && T_PLANE >= NRT * PITCH && D_PLANE >= NRD * PITCH, "planes");
    static_assert(A_PLANE % 256 == 128 && C_PLANE % 256 == 128 && T_PLANE % 256 == 128 && D_PLANE % 256 == 128, "plane stride");
    static_assert(UNROLL % NRT == 0 && UNROLL % NRA == 0 && UNROLL % 3 == 0, "static slots");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct H3UTile {
    int nrows;
    size_t img;                  // byte offset of the image in an fp32 NHWC tensor of 16 channels
    int ybase, ystep;            // image row of band-relative row k: ybase + ystep * k (a reversed band walks bottom-up)
    int X0, X1;                  // own (output) columns [X0, X1)
    int G0, go;                  // image column of grid column 0; grid column of X0
    __device__ __forceinline__ int y(const int k) const { return ybase + ystep * k; }
};

#ifndef H3U_XCD_ORDER
#define H3U_XCD_ORDER 1
#endif
__device__ __forceinline__ H3UTile h3u_tile(const BwdBlockH3Args& a, const int t)
{
    H3UTile r;
    // XCD-contiguous tile order (train_fwd_h3t.hip): the 16 halo columns two neighbouring strips share and the LEAD halo rows of vertically
    // adjacent bands come out of one XCD's L2 (H3U_XCD_ORDER 0: 934 MB per launch for 671 MB algorithmic)
    const int tp = (H3U_XCD_ORDER && (a.ntiles & 7) == 0) ? (t & 7) * (a.ntiles >> 3) + (t >> 3) : t;
    const int tt = a.reverse ? a.ntiles - 1 - tp : tp;
    const int sx = tt % a.nstrips, rest = tt / a.nstrips;
    const int b = rest / a.tiles_y, ty = rest - b * a.tiles_y;
    const int y0 = ty * a.rows_per_tile;
    r.nrows = min(a.rows_per_tile, a.H - y0);
    r.img = (size_t)b * a.H * a.W * 64;
    r.ybase = a.reverse ? y0 + r.nrows - 1 : y0;
    r.ystep = a.reverse ? -1 : 1;
    r.X0 = sx * H3UGeom::SW;
    r.X1 = min(a.W, r.X0 + H3UGeom::SW);
    r.G0 = max(r.X0 - 8, 0);
    r.go = r.X0 - r.G0;
    return r;
}

__device__ __forceinline__ int h3u_wimage(const BwdBlockH3Args& a, const int i) { return a.reverse ? (2 - i / 4) * 4 + i % 4 : i; }
__device__ __forceinline__ int h3u_mod(const int v, const int n) { return ((v % n) + n) % n; }

// the 15 MFMAs of one 16-pixel group and step (see train_fwd_h3t.hip)
template <int J, class Epi>
__device__ __forceinline__ void h3u_mfmas(const h8 (&w)[13], f32x4& acc2, f32x4& acc1, f32x4& c0, const H3VFrag& cur, Epi* epi)
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
        h3u_mfmas<J + 1>(w, acc2, acc1, c0, cur, epi);
    }
}

// ---- epilogues as micro-ops (h3v_core.h on why) ------------------------------------------------------------------------------
// D2: hi / lo of v * sc as H3VEpi<false>, then AND with the ReLU mask of the recomputed T (T > 0 <=> its f16 hi half is not zero:
// the F role's epilogue never writes a negative zero), then the two 8-byte records
typedef unsigned short hu_us2 __attribute__((ext_vector_type(2)));
struct H3UEpiMask {
    static constexpr int NOPS = 14;
    f32x4 v;
    float sc;
    unsigned h0, h1, l0, l1, t0, t1;          // t0 / t1: the T hi record of the lane's four channels (all ones when there is no ReLU)
    char* p;
    int lo_off;
    template <int I> __device__ __forceinline__ void op()
    {
        if constexpr (I == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v.x), "v"(sc));
        else if constexpr (I == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v.z), "v"(sc));
        else if constexpr (I == 2) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v.y), "v"(sc));
        else if constexpr (I == 3) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v.w), "v"(sc));
        else if constexpr (I == 4) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(v.x), "v"(sc), "v"(h0));
        else if constexpr (I == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(v.z), "v"(sc), "v"(h1));
        else if constexpr (I == 6) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l0) : "v"(v.y), "v"(sc), "v"(h0));
        else if constexpr (I == 7) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l1) : "v"(v.w), "v"(sc), "v"(h1));
        else if constexpr (I == 8) {
            // 0xffff where the half is not zero, else 0: 0 - min(bits, 1) on packed 16-bit lanes
            const hu_us2 one = {1, 1}, zero = {0, 0};
            t0 = __builtin_bit_cast(unsigned, zero - __builtin_elementwise_min(__builtin_bit_cast(hu_us2, t0), one));
        } else if constexpr (I == 9) {
            const hu_us2 one = {1, 1}, zero = {0, 0};
            t1 = __builtin_bit_cast(unsigned, zero - __builtin_elementwise_min(__builtin_bit_cast(hu_us2, t1), one));
        } else if constexpr (I == 10) { h0 &= t0; l0 &= t0; }
        else if constexpr (I == 11) { h1 &= t1; l1 &= t1; }
        else if constexpr (I == 12) *reinterpret_cast<h3v_u2*>(p) = (h3v_u2){h0, h1};
        else *reinterpret_cast<h3v_u2*>(p + lo_off) = (h3v_u2){l0, l1};
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

// D1: the finished row straight from the accumulator: v * sc + g (the skip's gradient), the two sums of the next BatchNorm backward
// (own rows and columns only), one 16-byte store per lane (4 channels of one pixel; 16 lanes = 1 KiB contiguous)
template <bool BNC>
struct H3UEpiOut {
    static constexpr int NOPS = BNC ? 17 : 13;
    f32x4 v, g, b, s1, s2;
    float sc;
    bool own;
    char* p;
    template <int I> __device__ __forceinline__ void op()
    {
        if constexpr (I < 4) v[I] = fmaf(v[I], sc, g[I]);                    // (sc is a power of two: the product is exact, one rounding as v * sc + g)
        else if constexpr (I < 8) g[I - 4] = own ? v[I - 4] : 0.f;           // g now holds the row masked to the strip's own pixels
        else if constexpr (I < 12) s1[I - 8] += g[I - 8];
        else if constexpr (I == 12) { if (own && !(H3U_ABLATE & 8)) *reinterpret_cast<f32x4*>(p) = v; }
        else s2[I - 13] = fmaf(g[I - 13], b[I - 13], s2[I - 13]);
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

// D1's state across steps: the skip gradient g and C_{i-1} of the row that completes NEXT, in the accumulator layout (lane (n, q): pixel
// n of its group, channels 4 q .. 4 q + 3: 16 bytes), the running sums, the addresses of the step
template <bool BNC>
struct H3UOutState {
    static constexpr int G = 3;
    f32x4 eg[G], eb[BNC ? G : 1], s1, s2;
    bool own[G];
    char* prow;                  // lane's store address of group 0 in the completing row
    const char* gnext;           // lane's load address of group 0 of the NEXT row in g (C_{i-1}: + bdelta)
    ptrdiff_t bdelta;
    int goff[G];                 // byte offset of group g from group 0 (columns clamped to the image)
    // UNCONDITIONAL loads, each issued right behind the epilogue that read its registers, one step before the next use (a branch around
    // them makes hipcc wait at the join; requested at the END of the step they had ~700 cycles to land: stamps)
    __device__ __forceinline__ void reload(const int g)
    {
        if (H3U_ABLATE & 4) return;
        eg[g] = *reinterpret_cast<const f32x4*>(gnext + goff[g]);
        if (BNC) eb[g] = *reinterpret_cast<const f32x4*>(gnext + bdelta + goff[g]);
    }
};

// ---- one convolution-shaped role (F, D2, D1): a ring row in, three vertical-tap contributions, one completed row out ------------
// KIND 0 = F (ReLU, split f16 out), 1 = D2 (mask, split f16 out), 2 = D1 (finished rows straight to global memory: step_out)
template <int KIND, int INP>
struct H3UConv {
    using Gm = H3UGeom;
    const char* tin;             // input ring
    char* tdst;                  // output ring
    const char* tmask;           // D2: the T ring
    h8 w[13];
    f32x4 acc[Gm::G][3];
    int rp, rs;                  // lane's byte offset in input-ring slot 0, group 0: pair fragment (hi planes), single fragment
    int wr;                      // lane's byte offset of its record in output slot 0, group 0
    int mr;                      // D2: lane's byte offset of its 8-byte T hi record in T-ring slot 0, group 0
    float inv_s, relu_floor;
    bool relu;
    float lane_scale[Gm::G];

    __device__ __forceinline__ void init(const float* pack, const BwdBlockH3Args& a, const int lane, const int gc0)
    {
        const int q = lane >> 4;
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = reinterpret_cast<const h8*>(pack)[h3u_wimage(a, i) * 64 + lane];
        w[12] = w[0];
        inv_s = pack[BF_H3R_WPACK_FLOATS];
        relu = a.act_relu != 0;
        relu_floor = relu ? 0.f : -__builtin_inff();
        rp = (q & 1) * INP + (gc0 + (q >> 1)) * 16;
        rs = ((q & 1) + 2 * (q >> 1)) * INP + (gc0 + 2) * 16;
        if (KIND == 0) wr = (q >> 1) * Gm::T_PLANE + (gc0 + 1) * 16 + (q & 1) * 8;
        else if (KIND == 1) wr = (q >> 1) * Gm::D_PLANE + (gc0 + 1) * 16 + (q & 1) * 8;
        else wr = 0;
        mr = (q >> 1) * Gm::T_PLANE + (gc0 + 1) * 16 + (q & 1) * 8;
#pragma unroll
        for (int g = 0; g < Gm::G; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void set_tile(const H3UTile& t, const int W, const int gc0)
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
    // step: consumes the input ring row at byte offset islot_bytes; the row that completes goes to byte offset oslot_bytes of the
    // output ring (rowok: inside the image -- rows / columns outside are the next operator's zero padding); D2: mslot_bytes = the
    // T ring row of the completing row
    template <int PH>
    __device__ __forceinline__ void step(const int islot_bytes, const int oslot_bytes, const int mslot_bytes, const bool rowok)
    {
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;
        H3VFrag cur = load(islot_bytes, 0);
        h3v_u2 mrec[Gm::G];
        if (KIND == 1) {
#pragma unroll
            for (int g = 0; g < Gm::G; ++g) {
                mrec[g] = (h3v_u2){0x3c003c00u, 0x3c003c00u};
                if (relu) mrec[g] = *reinterpret_cast<const h3v_u2*>(tmask + mr + mslot_bytes + g * 256);
            }
        }
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(islot_bytes, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (KIND == 0) {
                if (g > 0) {
                    H3VEpi<true> e = epi_f(g - 1, oslot_bytes, acc[g - 1][a2], rowok);
                    h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, &e);
                } else h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, (H3VEpi<true>*)nullptr);
            } else if constexpr (KIND == 1) {
                if (g > 0) {
                    H3UEpiMask e = epi_m(g - 1, oslot_bytes, acc[g - 1][a2], rowok, mrec[g - 1]);
                    h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, &e);
                } else h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, (H3UEpiMask*)nullptr);
            }
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        const f32x4 last = bf_acc_ready(acc[Gm::G - 1][a2]);
        if constexpr (KIND == 0) { H3VEpi<true> e = epi_f(Gm::G - 1, oslot_bytes, last, rowok); e.all(); }
        else if constexpr (KIND == 1) { H3UEpiMask e = epi_m(Gm::G - 1, oslot_bytes, last, rowok, mrec[Gm::G - 1]); e.all(); }
    }
    // D1: the same walk with the output epilogue (st: the skip gradient / C_{i-1} fragments of the completing row, the running sums,
    // the lane's store address of group 0 in that row; rowok: the completing row is one of the band's own)
    template <int PH, bool BNC, class State>
    __device__ __forceinline__ void step_out(const int islot_bytes, State& st, const bool rowok)
    {
        constexpr int a0 = PH % 3, a1 = (PH + 2) % 3, a2 = (PH + 1) % 3;
        H3VFrag cur = load(islot_bytes, 0);
#pragma unroll
        for (int g = 0; g < Gm::G; ++g) {
            H3VFrag nx;
            if (g + 1 < Gm::G) nx = load(islot_bytes, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
            if (g > 0) {
                H3UEpiOut<BNC> e = epi_o<BNC>(g - 1, acc[g - 1][a2], rowok, st);
                h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, &e);
                st.s1 = e.s1;
                st.s2 = e.s2;
                st.reload(g - 1);              // the operands of the NEXT row take the registers this epilogue just read
            } else h3u_mfmas<0>(w, acc[g][a2], acc[g][a1], c0, cur, (H3UEpiOut<BNC>*)nullptr);
            acc[g][a0] = c0;
            if (g + 1 < Gm::G) cur = nx;
        }
        H3UEpiOut<BNC> e = epi_o<BNC>(Gm::G - 1, bf_acc_ready(acc[Gm::G - 1][a2]), rowok, st);
        e.all();
        st.s1 = e.s1;
        st.s2 = e.s2;
        st.reload(Gm::G - 1);
    }
    template <bool BNC, class State>
    __device__ __forceinline__ H3UEpiOut<BNC> epi_o(const int g, const f32x4 v, const bool rowok, const State& st) const
    {
        H3UEpiOut<BNC> e;
        e.v = v;
        e.g = st.eg[g];
        e.b = BNC ? st.eb[g] : (f32x4){0.f, 0.f, 0.f, 0.f};
        e.s1 = st.s1;
        e.s2 = st.s2;
        e.sc = inv_s;
        e.own = rowok && st.own[g];
        e.p = st.prow + g * 1024;
        return e;
    }
    __device__ __forceinline__ H3VEpi<true> epi_f(const int g, const int oslot_bytes, const f32x4 v, const bool rowok) const
    {
        H3VEpi<true> e;
        e.v = v;
        e.sc = rowok ? lane_scale[g] : 0.f;
        e.floor_ = relu_floor;
        e.p = tdst + wr + oslot_bytes + g * 256;
        e.lo_off = 2 * Gm::T_PLANE;
        return e;
    }
    __device__ __forceinline__ H3UEpiMask epi_m(const int g, const int oslot_bytes, const f32x4 v, const bool rowok, const h3v_u2 m) const
    {
        H3UEpiMask e;
        e.v = v;
        e.sc = rowok ? lane_scale[g] : 0.f;
        e.t0 = m[0];
        e.t1 = m[1];
        e.p = tdst + wr + oslot_bytes + g * 256;
        e.lo_off = 2 * Gm::D_PLANE;
        return e;
    }
};

// ---- loads that ride on the F / D2 waves: one UNIT = the 16 pixels x 4 channel quads of grid group `unit` of a ring row; lane l
// owns pixel l >> 2, quad l & 3.  The registers of a unit are converted a step after they were loaded and re-loaded at once.
// UNCONDITIONAL loads from clamped addresses, zeroed at the conversion (a branch around a load makes hipcc wait for vmcnt(0)). ----
template <bool DC>               // false: A unit (x -> A ring); true: dc unit (k1 g + k2 c + k3 -> dc ring)
struct H3ULoadUnit {
    using Gm = H3UGeom;
    f32x4 x, c;
    bool ok;                     // the row in the registers lies inside the image and inside the rows the band needs
    int unit;

    __device__ __forceinline__ void load(const BwdBlockH3Args& a, const H3UTile& t, const int k, const int lane)
    {
        const int y = min(max(t.y(k), 0), a.H - 1);
        const int col = min(t.G0 + 16 * unit + (lane >> 2), a.W - 1);
        const size_t off = t.img + ((size_t)y * a.W + col) * 64 + (lane & 3) * 16;
        x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(DC ? a.g : a.a) + off);
        if (DC) c = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.c) + off);
    }
    __device__ __forceinline__ bool row_ok(const BwdBlockH3Args& a, const H3UTile& t, const int k) const
    {
        const int y = t.y(k);
        return (y >= 0) && (y < a.H) && (k >= (DC ? -2 : -3)) && (k < t.nrows + (DC ? 2 : 3));
    }
    // registers -> ring row at byte offset slot_bytes of `ring` (plane stride PL); then the registers take band row kn
    // k1 | k2 | k3: the BatchNorm-backward coefficients of the lane's channel quad (read from LDS once per step, behind the matrix work:
    // 12 registers that would otherwise be live through it)
    template <int PL>
    __device__ __forceinline__ void commit_and_reload(const BwdBlockH3Args& a, const H3UTile& t, char* ring, const int slot_bytes, const int kn,
                                                      const int lane, const f32x4 k1, const f32x4 k2, const f32x4 k3)
    {
        const int quad = lane & 3, gcol = 16 * unit + (lane >> 2);
        const bool in = ok && (t.G0 + gcol < a.W);
        f32x4 v;
        if (DC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = in ? fmaf(k1[i], x[i], fmaf(k2[i], c[i], k3[i])) : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = in ? x[i] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        ok = row_ok(a, t, kn);
        if (!(H3U_ABLATE & 2)) load(a, t, kn, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (H3U_ABLATE & 64) return;
        h4 hi, lo;
        h3_split(v, hi, lo);
        char* p = ring + (quad >> 1) * PL + slot_bytes + (gcol + 1) * 16 + (quad & 1) * 8;
        *reinterpret_cast<h4*>(p) = hi;
        *reinterpret_cast<h4*>(p + 2 * PL) = lo;
    }
};

// ---- weight gradients: operands through the transposing LDS read (train_bwd_h3.hip) ----------------------------------------------

extern __shared__ __attribute__((aligned(16))) char h3u_lds[];

// operand through the transposing LDS read: lane's 8-byte chunk at LDS offset addr, and the one 16 pixels further
__device__ __forceinline__ h8 h3u_tr(const int addr)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const hu_fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hu_fp16x4*)(h3u_lds + addr));
    const hu_fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hu_fp16x4*)(h3u_lds + addr + 16 * 16));
    const u2 ua = __builtin_bit_cast(u2, a), ub = __builtin_bit_cast(u2, b);
    return __builtin_bit_cast(h8, (u4){ua[0], ua[1], ub[0], ub[1]});
}

// Operands of a weight gradient (pixel index along K, train_bwd_h3.hip): the x side of one 32-pixel chunk of one ring row -- the row
// shifted by dx - 1 pixels for dx = 0, 1, 2, hi and lo image -- and the gradient side of one chunk of one ring row (hi, lo).  One x set
// serves the three gradient rows of its chunk (27 MFMAs): 12 operands where a per-tap-row split needs 24.
struct H3UOpsX {
    h8 ah[3], al[3];
    // xa: lane's LDS offset of the chunk's first pixel in the hi image; xl: offset of the lo image
    __device__ __forceinline__ void load(const int xa, const int xl)
    {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            ah[dx] = h3u_tr(xa + (dx - 1) * 16);
            al[dx] = h3u_tr(xa + xl + (dx - 1) * 16);
        }
    }
};
struct H3UOpsG {
    h8 bh, bl;
    __device__ __forceinline__ void load(const int ga, const int gl)
    {
        bh = h3u_tr(ga);
        bl = h3u_tr(ga + gl);
    }
};
// acc[dx] += x(dx)^T . g, three split products each; !valid: a gradient row outside the band's own rows, multiplied as zeros
// (straight-line code: a branch around the MFMAs cost the register allocator dearly)
__device__ __forceinline__ void h3u_mfma9(const H3UOpsX& x, const H3UOpsG& g, f32x4& a0, f32x4& a1, f32x4& a2, const bool valid)
{
    if (H3U_ABLATE & 32) {                                   // keeps the operand reads live
        a0[0] += (float)x.ah[0][0] + (float)x.ah[1][1] + (float)x.ah[2][2] + (float)x.al[0][3] + (float)x.al[1][4] + (float)x.al[2][5] + (float)g.bh[6] + (float)g.bl[7];
        return;
    }
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const h8 bh = valid ? g.bh : zero, bl = valid ? g.bl : zero;
    a0 = MFMA_H(x.ah[0], bh, a0); a1 = MFMA_H(x.ah[1], bh, a1); a2 = MFMA_H(x.ah[2], bh, a2);
    a0 = MFMA_H(x.al[0], bh, a0); a1 = MFMA_H(x.al[1], bh, a1); a2 = MFMA_H(x.al[2], bh, a2);
    a0 = MFMA_H(x.ah[0], bl, a0); a1 = MFMA_H(x.ah[1], bl, a1); a2 = MFMA_H(x.ah[2], bl, a2);
}

// One UNIT of weight-gradient work = (convolution, chunk) at one step: its x set and its three gradient rows (band-order vertical taps
// 0, 1, 2).  A convolution's operands at chunk 0 (scalars only: arrays of these structs ended up in scratch memory):
struct H3UConvOps {
    int xa, xl, ga0, ga1, ga2, gl;      // lane's LDS offsets (hi images) of chunk 0's first pixel: x row, the three gradient rows; lo-image offsets
    bool v0, v1, v2;                    // the gradient row is one of the band's own
};
// NU units back to back, software-pipelined: the x set of the next unit and the gradient set two rows ahead are requested before the
// MFMAs of the current row (two x sets, two gradient sets in registers).  Unit i: convolution P (CI = 1, accumulators acc1) or Q (CI = 0,
// acc0) at byte offset OFFi from chunk 0.
template <int NU, int C0, int C1, int C2, int OFF0, int OFF1, int OFF2>
__device__ __forceinline__ void h3u_units(const H3UConvOps& P, const H3UConvOps& Q, f32x4 (&acc1)[9], f32x4 (&acc0)[9])
{
#define H3U_U(I) (((I) == 0 ? C0 : ((I) == 1 ? C1 : C2)) ? P : Q)
#define H3U_O(I) ((I) == 0 ? OFF0 : ((I) == 1 ? OFF1 : OFF2))
    H3UOpsX x0, x1;
    H3UOpsG g0, g1;
    x0.load(H3U_U(0).xa + H3U_O(0), H3U_U(0).xl);
    g0.load(H3U_U(0).ga0 + H3U_O(0), H3U_U(0).gl);
    g1.load(H3U_U(0).ga1 + H3U_O(0), H3U_U(0).gl);
    __builtin_amdgcn_sched_barrier(0);
#define H3U_ROW(X, G, UI, DYB, VAL, NEXT)                                                                     \
    do {                                                                                                      \
        constexpr int c_ = UI == 0 ? C0 : (UI == 1 ? C1 : C2);                                                \
        if constexpr (c_) h3u_mfma9(X, G, acc1[3 * (DYB)], acc1[3 * (DYB) + 1], acc1[3 * (DYB) + 2], VAL);    \
        else h3u_mfma9(X, G, acc0[3 * (DYB)], acc0[3 * (DYB) + 1], acc0[3 * (DYB) + 2], VAL);                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        NEXT;                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
    H3U_ROW(x0, g0, 0, 0, H3U_U(0).v0, { g0.load(H3U_U(0).ga2 + H3U_O(0), H3U_U(0).gl); if (NU > 1) x1.load(H3U_U(1).xa + H3U_O(1), H3U_U(1).xl); });
    H3U_ROW(x0, g1, 0, 1, H3U_U(0).v1, { if (NU > 1) g1.load(H3U_U(1).ga0 + H3U_O(1), H3U_U(1).gl); });
    H3U_ROW(x0, g0, 0, 2, H3U_U(0).v2, { if (NU > 1) g0.load(H3U_U(1).ga1 + H3U_O(1), H3U_U(1).gl); });
    if (NU > 1) {
        H3U_ROW(x1, g1, 1, 0, H3U_U(1).v0, { g1.load(H3U_U(1).ga2 + H3U_O(1), H3U_U(1).gl); if (NU > 2) x0.load(H3U_U(2).xa + H3U_O(2), H3U_U(2).xl); });
        H3U_ROW(x1, g0, 1, 1, H3U_U(1).v1, { if (NU > 2) g0.load(H3U_U(2).ga0 + H3U_O(2), H3U_U(2).gl); });
        H3U_ROW(x1, g1, 1, 2, H3U_U(1).v2, { if (NU > 2) g1.load(H3U_U(2).ga1 + H3U_O(2), H3U_U(2).gl); });
    }
    if (NU > 2) {
        H3U_ROW(x0, g0, 2, 0, H3U_U(2).v0, { g0.load(H3U_U(2).ga2 + H3U_O(2), H3U_U(2).gl); });
        H3U_ROW(x0, g1, 2, 1, H3U_U(2).v1, { });
        H3U_ROW(x0, g0, 2, 2, H3U_U(2).v2, { });
    }
#undef H3U_ROW
#undef H3U_U
#undef H3U_O
}

// One function per role, all inlined into the kernel.  (Tried: not inlined, so that the register allocator treats the roles
// separately.  The callee then sees its arguments as per-lane values -- uniform branches became exec-mask loops, the tensor pointers
// generic, every load a flat_load behind vmcnt(0) -- far worse than the few spills of the inlined form, none of which is in a loop.)
#define H3U_IN_IMAGE(k) ((t.y(k) >= 0) && (t.y(k) < a.H))
    // per band: PRO = prologue statements, BODY(PH) = the role's work of step s = s0 + PH; every role runs the same barriers
#define H3U_BAND(PRO, BODY)                                                                                   \
    for (int ti = blockIdx.x; ti < a.ntiles; ti += gridDim.x) {                                               \
        const H3UTile t = h3u_tile(a, ti);                                                                    \
        PRO;                                                                                                  \
        h3_barrier();                                                                                         \
        const int nsteps = t.nrows + Gm::LEAD;                                                                \
        for (int s0 = 0; s0 < nsteps; s0 += Gm::UNROLL) {                                                     \
            BODY(0); BODY(1); BODY(2); BODY(3); BODY(4); BODY(5);                                             \
        }                                                                                                     \
        h3_barrier();                                                                                         \
    }

__device__ __forceinline__ void h3u_role_f(const BwdBlockH3Args& a, const int lane, const int rw)
{
    using Gm = H3UGeom;
    char* ta = h3u_lds + Gm::A_OFF;
    char* tc = h3u_lds + Gm::C_OFF;
    char* tt = h3u_lds + Gm::T_OFF;
    char* td = h3u_lds + Gm::D_OFF;
    const int n = lane & 15, q = lane >> 4;
    const int gc0 = 16 * Gm::G * rw + n;                        // F / D waves: lane's grid column in its wave's group 0
    (void)ta; (void)tc; (void)tt; (void)td; (void)q; (void)gc0;
    H3U_STAMP_DECL
        // ================= F: T = relu(conv_0 A) ; loads: A units 3 rw .. 3 rw + 2, dc unit rw =================
        __builtin_amdgcn_s_setprio(1);
        H3UConv<0, Gm::A_PLANE> R;
        R.tin = ta; R.tdst = tt; R.tmask = nullptr;
        R.init(a.wfwd0, a, lane, gc0);
        H3ULoadUnit<false> ua[3];
        H3ULoadUnit<true> uc;
#pragma unroll
        for (int i = 0; i < 3; ++i) ua[i].unit = 3 * rw + i;
        uc.unit = rw;
        const char* kq = h3u_lds + Gm::K_OFF + (lane & 3) * 16;
#define H3U_PRO_F                                                                                             \
        R.set_tile(t, a.W, gc0);                                                                              \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                       \
            ua[i].ok = ua[i].row_ok(a, t, -3);                                                                \
            ua[i].load(a, t, -3, lane);                                                                       \
            ua[i].template commit_and_reload<Gm::A_PLANE>(a, t, ta, h3u_mod(-3, Gm::NRA) * Gm::PITCH, -2, lane, f32x4{}, f32x4{}, f32x4{}); \
        }                                                                                                     \
        uc.ok = false;                                                                                        \
        uc.load(a, t, -3, lane);
#define H3U_BODY_F(PH)                                                                                        \
        do {                                                                                                  \
            const int s = s0 + PH;                                                                            \
            if (!(H3U_ABLATE & 16) && s < t.nrows + 6)       /* A rows -3 .. nrows+2 */                       \
                R.template step<PH>(h3u_mod(s - 3, Gm::NRA) * Gm::PITCH, (PH % Gm::NRT) * Gm::PITCH, 0, H3U_IN_IMAGE(s - 4)); \
            H3U_STAMP(0);                                                                                     \
            /* BEHIND the matrix work (the loads then have the whole step to land before anything waits for them): A row s-2 -> its     \
               slot, registers <- A row s-1 ; dc row s-3 -> its slot, registers <- row s-2 */                  \
            {                                                                                                 \
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(kq), k2 = *reinterpret_cast<const f32x4*>(kq + 64), k3 = *reinterpret_cast<const f32x4*>(kq + 128); \
                _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                 \
                    ua[i].template commit_and_reload<Gm::A_PLANE>(a, t, ta, h3u_mod(s - 2, Gm::NRA) * Gm::PITCH, s - 1, lane, k1, k2, k3); \
                uc.template commit_and_reload<Gm::C_PLANE>(a, t, tc, h3u_mod(s - 3, Gm::NRC) * Gm::PITCH, s - 2, lane, k1, k2, k3); \
            }                                                                                                 \
            H3U_STAMP(1);                                                                                     \
            h3_barrier();                                                                                     \
            H3U_STAMP(2);                                                                                     \
        } while (0)
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
        H3U_BAND(H3U_PRO_F, H3U_BODY_F)
#undef H3U_PRO_F
#undef H3U_BODY_F
        H3U_STAMP_OUT;
        h3_barrier();
        h3_barrier();
}

__device__ __forceinline__ void h3u_role_d2(const BwdBlockH3Args& a, const int lane, const int rw)
{
    using Gm = H3UGeom;
    char* ta = h3u_lds + Gm::A_OFF;
    char* tc = h3u_lds + Gm::C_OFF;
    char* tt = h3u_lds + Gm::T_OFF;
    char* td = h3u_lds + Gm::D_OFF;
    const int n = lane & 15, q = lane >> 4;
    const int gc0 = 16 * Gm::G * rw + n;                        // F / D waves: lane's grid column in its wave's group 0
    (void)ta; (void)tc; (void)tt; (void)td; (void)q; (void)gc0;
    H3U_STAMP_DECL
        // ================= D2: dT = dgrad_1(dc) * (T > 0) ; loads: dc units 3 + 2 rw, 4 + 2 rw =================
        __builtin_amdgcn_s_setprio(1);
        H3UConv<1, Gm::C_PLANE> R;
        R.tin = tc; R.tdst = td; R.tmask = tt;
        R.init(a.wdg1, a, lane, gc0);
        H3ULoadUnit<true> uc[2];
        uc[0].unit = 3 + 2 * rw;
        uc[1].unit = 4 + 2 * rw;
        const char* kq = h3u_lds + Gm::K_OFF + (lane & 3) * 16;
#define H3U_PRO_D2                                                                                            \
        R.set_tile(t, a.W, gc0);                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
            uc[i].ok = false;                                                                                 \
            uc[i].load(a, t, -3, lane);                                                                       \
        }
#define H3U_BODY_D2(PH)                                                                                       \
        do {                                                                                                  \
            const int s = s0 + PH;                                                                            \
            /* dc row s-4 in; dT row s-5 completes (slot (s-5) mod 4), masked by T row s-5 (slot (PH+1) mod 2) */ \
            if (!(H3U_ABLATE & 16) && (s >= 2) && (s < t.nrows + 6))                                          \
                R.template step<PH>(h3u_mod(s - 4, Gm::NRC) * Gm::PITCH, h3u_mod(s - 5, Gm::NRD) * Gm::PITCH, ((PH + 1) % Gm::NRT) * Gm::PITCH, \
                                    H3U_IN_IMAGE(s - 5));                                                     \
            H3U_STAMP(0);                                                                                     \
            {                                                /* behind the matrix work: dc row s-3 -> its slot, registers <- row s-2 */ \
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(kq), k2 = *reinterpret_cast<const f32x4*>(kq + 64), k3 = *reinterpret_cast<const f32x4*>(kq + 128); \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
                    uc[i].template commit_and_reload<Gm::C_PLANE>(a, t, tc, h3u_mod(s - 3, Gm::NRC) * Gm::PITCH, s - 2, lane, k1, k2, k3); \
            }                                                                                                 \
            H3U_STAMP(1);                                                                                     \
            h3_barrier();                                                                                     \
            H3U_STAMP(2);                                                                                     \
        } while (0)
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
        H3U_BAND(H3U_PRO_D2, H3U_BODY_D2)
#undef H3U_PRO_D2
#undef H3U_BODY_D2
        H3U_STAMP_OUT;
        h3_barrier();
        h3_barrier();
}

template <bool BNC>
__device__ __forceinline__ void h3u_role_d1(const BwdBlockH3Args& a, const int lane, const int rw)
{
    using Gm = H3UGeom;
    char* ta = h3u_lds + Gm::A_OFF;
    char* tc = h3u_lds + Gm::C_OFF;
    char* tt = h3u_lds + Gm::T_OFF;
    char* td = h3u_lds + Gm::D_OFF;
    const int n = lane & 15, q = lane >> 4;
    const int gc0 = 16 * Gm::G * rw + n;                        // F / D waves: lane's grid column in its wave's group 0
    (void)ta; (void)tc; (void)tt; (void)td; (void)q; (void)gc0;
    H3U_STAMP_DECL
        // ================= D1: dA' = dgrad_0(dT) + g, the two sums against C_{i-1}, stored straight from the accumulators =================
        __builtin_amdgcn_s_setprio(1);
        H3UConv<2, Gm::D_PLANE> R;
        R.tin = td; R.tdst = nullptr; R.tmask = nullptr;
        R.init(a.wdg0, a, lane, gc0);
        // the skip gradient g and C_{i-1} of the row that completes in the NEXT step, in the accumulator layout (lane (n, q): pixel n of
        // its group, channels 4 q .. 4 q + 3: 16 bytes): requested one step ahead, right behind the epilogues that consumed the last ones
        H3UOutState<BNC> st;
        st.s1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        st.s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        st.bdelta = BNC ? reinterpret_cast<const char*>(a.bnc) - reinterpret_cast<const char*>(a.g) : 0;
        auto next_row = [&](const H3UTile& t, const int k) {   // lane's address of group 0 of band row k (clamped to the image) in g
            const int y = min(max(t.y(k), 0), a.H - 1);
            return reinterpret_cast<const char*>(a.g) + t.img + ((size_t)y * a.W + min(t.G0 + gc0, a.W - 1)) * 64 + q * 16;
        };
        __builtin_amdgcn_s_waitcnt(h3_vmcnt(0));
#define H3U_PRO_D1                                                                                            \
        R.set_tile(t, a.W, gc0);                                                                              \
        _Pragma("unroll") for (int g = 0; g < Gm::G; ++g) {                                                   \
            st.own[g] = (t.G0 + gc0 + 16 * g >= t.X0) && (t.G0 + gc0 + 16 * g < t.X1);                         \
            st.goff[g] = (min(t.G0 + gc0 + 16 * g, a.W - 1) - min(t.G0 + gc0, a.W - 1)) * 64;                  \
        }                                                                                                     \
        st.gnext = next_row(t, 0);                                                                            \
        _Pragma("unroll") for (int g = 0; g < Gm::G; ++g) st.reload(g);
#define H3U_BODY_D1(PH)                                                                                       \
        do {                                                                                                  \
            const int s = s0 + PH;                                                                            \
            /* dT row s-6 in (slot (s-6) mod 4); row s-7 completes: + g, sums, store */                        \
            if (!(H3U_ABLATE & 16) && (s >= 5) && (s < t.nrows + 7)) {                                        \
                const int k = s - 7;                                                                          \
                const bool rowok = (k >= 0) && (k < t.nrows);                                                 \
                st.prow = reinterpret_cast<char*>(a.out) + t.img + ((size_t)t.y(rowok ? k : 0) * a.W + t.G0 + gc0) * 64 + q * 16; \
                st.gnext = next_row(t, k + 1);                                                                \
                R.template step_out<PH, BNC>(h3u_mod(s - 6, Gm::NRD) * Gm::PITCH, st, rowok);                 \
                H3U_STAMP(0);                                                                                 \
            }                                                                                                 \
            H3U_STAMP(1);                                                                                     \
            h3_barrier();                                                                                     \
            H3U_STAMP(2);                                                                                     \
        } while (0)
        H3U_BAND(H3U_PRO_D1, H3U_BODY_D1)
#undef H3U_PRO_D1
#undef H3U_BODY_D1
        H3U_STAMP_OUT;
        // the two sums: over the 16 pixel lanes that share a channel quad (q = lane >> 4), then over the three waves (fixed order)
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                st.s1[cidx] += __shfl_xor(st.s1[cidx], m);
                st.s2[cidx] += __shfl_xor(st.s2[cidx], m);
            }
        }
        float* sred = reinterpret_cast<float*>(h3u_lds);     // [3 waves][32]; the rings are free after the last band
        if (n == 0) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                sred[rw * 32 + q * 4 + cidx] = st.s1[cidx];
                sred[rw * 32 + 16 + q * 4 + cidx] = st.s2[cidx];
            }
        }
        h3_barrier();
        if (BNC && rw == 0 && lane < 32) a.stats[(size_t)blockIdx.x * 32 + lane] = (sred[lane] + sred[32 + lane]) + sred[64 + lane];
        h3_barrier();
}

// (one instance per wave of the role: with the wave index as a run-time value hipcc merged the three shapes of the step into code that
// spilled 235 registers; apart they need 122 / 116 / 154 and none)
template <int RW>
__device__ __forceinline__ void h3u_role_w(const BwdBlockH3Args& a, const int lane)
{
    using Gm = H3UGeom;
    char* ta = h3u_lds + Gm::A_OFF;
    char* tc = h3u_lds + Gm::C_OFF;
    char* tt = h3u_lds + Gm::T_OFF;
    char* td = h3u_lds + Gm::D_OFF;
    constexpr int rw = RW;
    const int n = lane & 15, q = lane >> 4;
    const int gc0 = 16 * Gm::G * rw + n;                        // F / D waves: lane's grid column in its wave's group 0
    (void)ta; (void)tc; (void)tt; (void)td; (void)q; (void)gc0;
    H3U_STAMP_DECL
        // ================= W: dW1 = T^T dc, dW0 = A^T dT =================
        // K chunks of 32 pixels of the strip's own columns; a wave owns ALL nine taps of its (convolution, chunk) units, so that one set
        // of x operands serves the three gradient rows of a chunk (the kernel is bound by LDS traffic and these transposed reads were
        // half of it when the work was dealt by tap rows):   rw = 0: dW1 chunks 0, 1, 2 ; rw = 1: dW0 chunks 0, 1, 2 ; rw = 2: chunk 3 of both.
        // dW1 at step s: T row r = s-5 (slot (PH+1) mod 2) against dc rows r - dyb + 1; dW0: A row a = s-7 against dT rows a - dyb + 1
        // (dyb = vertical tap in BAND order; a reversed band walks bottom-up and its tap rows are swapped when they are written out).
        f32x4 acc1[9], acc0[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc0[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // transposed-read offset of this lane inside a 32-pixel chunk: pixel 4 (lane >> 4) + ((lane & 15) >> 2), channels 4 j .. 4 j + 3
        // with j = lane & 3: plane j >> 1, half-record j & 1
        const int tr_px = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8;
        const int tr_hi = (lane & 3) >> 1;
#define H3U_PRO_W
#define H3U_BODY_W(PH)                                                                                        \
        do {                                                                                                  \
            const int s = s0 + PH;                                                                            \
            if (!(H3U_ABLATE & 1) && (s >= 4) && (s < t.nrows + 9)) {                                         \
                const int r = s - 5, ar = s - 7;                                                              \
                const int c0 = (t.go + 1) * 16 + tr_px;                                                       \
                const int tb = Gm::T_OFF + tr_hi * Gm::T_PLANE + ((PH + 1) % Gm::NRT) * Gm::PITCH + c0;       \
                const int ab = Gm::A_OFF + tr_hi * Gm::A_PLANE + h3u_mod(ar, Gm::NRA) * Gm::PITCH + c0;       \
                H3UConvOps P, Q;                             /* dW1 / dW0 at chunk 0 */                        \
                P.xa = tb; P.xl = 2 * Gm::T_PLANE; P.gl = 2 * Gm::C_PLANE;                                    \
                Q.xa = ab; Q.xl = 2 * Gm::A_PLANE; Q.gl = 2 * Gm::D_PLANE;                                    \
                P.ga0 = Gm::C_OFF + tr_hi * Gm::C_PLANE + h3u_mod(r + 1, Gm::NRC) * Gm::PITCH + c0; P.v0 = (r + 1 >= 0) && (r + 1 < t.nrows); \
                P.ga1 = Gm::C_OFF + tr_hi * Gm::C_PLANE + h3u_mod(r, Gm::NRC) * Gm::PITCH + c0;     P.v1 = (r >= 0) && (r < t.nrows);         \
                P.ga2 = Gm::C_OFF + tr_hi * Gm::C_PLANE + h3u_mod(r - 1, Gm::NRC) * Gm::PITCH + c0; P.v2 = (r - 1 >= 0) && (r - 1 < t.nrows); \
                Q.ga0 = Gm::D_OFF + tr_hi * Gm::D_PLANE + h3u_mod(ar + 1, Gm::NRD) * Gm::PITCH + c0; Q.v0 = (ar + 1 >= 0) && (ar + 1 < t.nrows); \
                Q.ga1 = Gm::D_OFF + tr_hi * Gm::D_PLANE + h3u_mod(ar, Gm::NRD) * Gm::PITCH + c0;     Q.v1 = (ar >= 0) && (ar < t.nrows);         \
                Q.ga2 = Gm::D_OFF + tr_hi * Gm::D_PLANE + h3u_mod(ar - 1, Gm::NRD) * Gm::PITCH + c0; Q.v2 = (ar - 1 >= 0) && (ar - 1 < t.nrows); \
                if constexpr (RW == 0) h3u_units<3, 1, 1, 1, 0, 512, 1024>(P, Q, acc1, acc0);                 \
                else if constexpr (RW == 1) h3u_units<3, 0, 0, 0, 0, 512, 1024>(P, Q, acc1, acc0);            \
                else h3u_units<2, 1, 0, 0, 1536, 1536, 0>(P, Q, acc1, acc0);                                  \
            }                                                                                                 \
            H3U_STAMP(0);                                                                                     \
            H3U_STAMP(1);                                                                                     \
            h3_barrier();                                                                                     \
            H3U_STAMP(2);                                                                                     \
        } while (0)
        H3U_BAND(H3U_PRO_W, H3U_BODY_W)
#undef H3U_PRO_W
#undef H3U_BODY_W
        H3U_STAMP_OUT;
        // ---- per-workgroup partials.  Weight gradients: D[ci = 4 q + j][co = n] per lane and tap; band order -> image order of the
        // vertical taps; dW1 = wave 0 + wave 2, dW0 = wave 1 + wave 2 through LDS (fixed order: bitwise reproducible) ----
        float* red = reinterpret_cast<float*>(h3u_lds) + 128;    // [4][2304]: dW1 of wave 0, dW1 of wave 2, dW0 of wave 1, dW0 of wave 2 (behind the D1 sums)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dyb = tap / 3, dx = tap % 3;
            const int ti = (a.reverse ? 2 - dyb : dyb) * 3 + dx;
            const f32x4 v1 = bf_acc_ready(acc1[tap]), v0 = bf_acc_ready(acc0[tap]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rw != 1) red[(rw == 0 ? 0 : 1) * 2304 + ti * 256 + (4 * q + j) * 16 + n] = v1[j];
                if (rw != 0) red[(rw == 1 ? 2 : 3) * 2304 + ti * 256 + (4 * q + j) * 16 + n] = v0[j];
            }
        }
        h3_barrier();
        for (int i = rw * 64 + lane; i < 2304; i += 192) {
            a.wpartial1[(size_t)blockIdx.x * 2304 + i] = red[i] + red[2304 + i];
            a.wpartial0[(size_t)blockIdx.x * 2304 + i] = red[2 * 2304 + i] + red[3 * 2304 + i];
        }
        h3_barrier();
}

#undef H3U_BAND
#undef H3U_IN_IMAGE

template <bool BNC>
__global__ __launch_bounds__(H3UGeom::NT, 3) void bwd_block_h3t_kernel(BwdBlockH3Args a)
{
    using Gm = H3UGeom;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave / 3, rw = wave - role * 3;

    // bn_bwd_finalize_kernel in the prologue (BwdBlockH3Args::fin_partial): every workgroup sums the [fin_nblk][32] partials of the launch
    // before (fp64, fixed order) and forms k1 | k2 | k3 itself; workgroup 0 writes d gamma.  The LDS used is cleared right below.
    float kk[3] = {0.f, 0.f, 0.f};
    if (a.fin_partial) {
        double* red = reinterpret_cast<double*>(h3u_lds);       // [24][32]
        {
            const int ch = tid & 31, stripe = tid >> 5;
            double sum = 0.0;
            for (int r = stripe; r < a.fin_nblk; r += Gm::NT / 32) sum += (double)a.fin_partial[(size_t)r * 32 + ch];
            red[stripe * 32 + ch] = sum;
        }
        __syncthreads();
        if (tid < 16) {
            double sdy = 0.0, sdyc = 0.0;
            for (int k = 0; k < Gm::NT / 32; ++k) { sdy += red[k * 32 + tid]; sdyc += red[k * 32 + 16 + tid]; }
            const double mean = a.fin_meaninv[tid], inv = a.fin_meaninv[16 + tid], g = a.fin_gamma[tid], count = a.fin_count;
            const double sdyx = (sdyc - mean * sdy) * inv;      // sum dy * xhat
            if (blockIdx.x == 0) a.fin_dgamma[tid] = (float)sdyx;
            const double mdy = sdy / count, mdyx = sdyx / count;
            kk[0] = (float)(g * inv);
            kk[1] = (float)(-g * inv * inv * mdyx);
            kk[2] = (float)(-g * inv * mdy + g * inv * inv * mean * mdyx);
        }
        __syncthreads();
    }
    // ring columns 0 and 145 are the zero padding at an image edge: cleared once, never written
    for (int i = tid * 16; i < Gm::LDS_BYTES; i += Gm::NT * 16) *reinterpret_cast<f32x4*>(h3u_lds + i) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (a.fin_partial) {
        if (tid < 16) {
            float* K = reinterpret_cast<float*>(h3u_lds + Gm::K_OFF);
            K[tid] = kk[0];
            K[16 + tid] = kk[1];
            K[32 + tid] = kk[2];
        }
    } else if (tid < 48) {
        reinterpret_cast<float*>(h3u_lds + Gm::K_OFF)[tid] = a.coef[tid];
    }
    __syncthreads();
    if (role == 0) h3u_role_f(a, lane, rw);
    else if (role == 1) h3u_role_d2(a, lane, rw);
    else if (role == 2) h3u_role_d1<BNC>(a, lane, rw);
    else if (rw == 0) h3u_role_w<0>(a, lane);
    else if (rw == 1) h3u_role_w<1>(a, lane);
    else h3u_role_w<2>(a, lane);
}

static int h3u_nstrips(const int W) { return (W + H3UGeom::SW - 1) / H3UGeom::SW; }

// bands: every strip of every image is cut into ceil(H / rows) bands; one band of one strip = one unit of work of a workgroup
static int h3u_rows_per_tile(const int B, const int H, const int nstrips, const int cus)
{
    int best = H;
    long best_cost = -1;
    for (int ty = 1; ty <= (H + 7) / 8; ++ty) {
        const int rows = (H + ty - 1) / ty;
        if ((H + rows - 1) / rows != ty) continue;
        const long tiles = (long)B * ty * nstrips;
        const long cost = ((tiles + cus - 1) / cus) * (rows + H3UGeom::LEAD + 2);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = rows;
        }
    }
    return best;
}

bool bf_bwd_block_h3t_supports(int H, int W) { return W >= 1 && H >= 1; }

// workgroups (= rows of wpartial0 / wpartial1 / stats) a launch uses
int bf_bwd_block_h3t_grid(int B, int H, int W)
{
    const int ns = h3u_nstrips(W);
    const int rows = h3u_rows_per_tile(B, H, ns, 256);
    const long tiles = (long)B * ((H + rows - 1) / rows) * ns;
    return (int)(tiles < 256 ? tiles : 256);
}

hipError_t bf_launch_bwd_block_h3t(const BwdBlockH3Args& args, hipStream_t s)
{
    using Gm = H3UGeom;
    BwdBlockH3Args a = args;
    if (!bf_bwd_block_h3t_supports(a.H, a.W) || !a.a || !a.g || !a.c || (!a.coef && !a.fin_partial) || !a.wfwd0 || !a.wdg1 || !a.wdg0 || !a.out ||
        !a.wpartial0 || !a.wpartial1)
        return hipErrorInvalidValue;
    if (a.fin_partial && (a.fin_partial == a.stats || a.fin_nblk <= 0 || !(a.fin_count > 0.0) || !a.fin_gamma || !a.fin_meaninv || !a.fin_dgamma))
        return hipErrorInvalidValue;
    if (a.out == a.a || a.out == a.g || a.out == a.c || (a.bnc && (!a.stats || a.out == a.bnc))) return hipErrorInvalidValue;
    const int cus = 256;
    a.nstrips = h3u_nstrips(a.W);
    a.rows_per_tile = h3u_rows_per_tile(a.B, a.H, a.nstrips, cus);
    a.tiles_y = (a.H + a.rows_per_tile - 1) / a.rows_per_tile;
    a.ntiles = a.B * a.tiles_y * a.nstrips;
    const int grid = a.ntiles < cus ? a.ntiles : cus;
    void (*kernel)(BwdBlockH3Args) = a.bnc ? bwd_block_h3t_kernel<true> : bwd_block_h3t_kernel<false>;
    const hipError_t e = bf_set_max_lds(reinterpret_cast<const void*>(kernel), Gm::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Gm::NT), Gm::LDS_BYTES, s, a);
    return hipGetLastError();
}
