// Internal declarations shared by the gfx950 kernels and the C-ABI engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/bfcnn_hip.h"
#include "../../include/bfcnn_hip_debug.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Eight wait states that cannot move away from the accumulator they protect (the asm takes it as an in/out operand), put in
// front of an epilogue's first read of a just-finished MFMA accumulator.  hipcc pads "MFMA writes VGPR -> VALU reads it"
// in straight-line code, but with a uniform branch or an EXEC-masked select between the two it padded only the
// fall-through path (found in fused_h3.hip: one wait state on the taken path, stale halves of the accumulator).
__device__ __forceinline__ f32x4 bf_acc_ready(f32x4 acc)
{
    asm volatile("s_nop 7" : "+v"(acc));
    return acc;
}


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device).  A process-wide "done" flag per kernel would
// leave the second device of a process without the attribute (launch failure there), and is not thread-safe.
#include <mutex>
#include <unordered_set>
inline hipError_t bf_set_max_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::unordered_set<uint64_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t key = (uint64_t)(uintptr_t)kernel * 131u + (uint64_t)(dev + 1);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert(key);
    return e;
}

// ------------------------------------------------------------------------------------------
// Compact split-planar activations ("h3c": 48 instead of 64 bytes per pixel).  The hi planes stay f16; a lo plane holds the
// OCP fp8 (e4m3) number nearest to lo * 2^12, 8 bytes per pixel: lo is the rounding residual of hi (|lo| <= 2^-11 |x|), three
// mantissa bits of it keep the pair at 15-16 significant bits.  Sized on the CPU before it was built
// (tools/exp/emulate_f16x3.py, storage fp8lo): 5.8e-6 normalised MAE through 1x18 against the fp64 oracle, max 1 LSB (bar 1e-4;
// 2.2e-7 with f16 lo planes).  v_cvt_scalef32_pk_fp8_f16 does not saturate (|lo| * 2^12 > 448 would become NaN: |x| > 224), so
// lo is clamped first; a clamped lo leaves x at f16 precision, and inf / NaN still travel in the hi plane to the status check.
// Semantics of the two conversions probed in tools/exp/fp8_probe.hip: encode = fp8(src / scale) into the selected 16-bit half,
// decode = fp8 * scale from the selected half; element 0 in the low byte.
// ------------------------------------------------------------------------------------------
typedef _Float16 bf_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf_h2 __attribute__((ext_vector_type(2)));
typedef short bf_s2 __attribute__((ext_vector_type(2)));
typedef unsigned bf_u2 __attribute__((ext_vector_type(2)));
#define BF_H3C_SCALE (1.0f / 4096.0f)
__device__ __forceinline__ bf_u2 bf_h3c_encode8(const bf_h8 v)
{
    const bf_h2 lim = {(_Float16)(448.0f / 4096.0f), (_Float16)(448.0f / 4096.0f)};
    bf_h2 p[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_min(__builtin_elementwise_max(p[i], -lim), lim);
    bf_s2 r0 = {0, 0}, r1 = {0, 0};
    r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, p[0], BF_H3C_SCALE, false);
    r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, p[1], BF_H3C_SCALE, true);
    r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, p[2], BF_H3C_SCALE, false);
    r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, p[3], BF_H3C_SCALE, true);
    bf_u2 out;
    out[0] = ((unsigned)(unsigned short)r0[0]) | ((unsigned)(unsigned short)r0[1] << 16);
    out[1] = ((unsigned)(unsigned short)r1[0]) | ((unsigned)(unsigned short)r1[1] << 16);
    return out;
}
__device__ __forceinline__ bf_h8 bf_h3c_decode8(const bf_u2 b)
{
    const bf_h2 a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b[0], BF_H3C_SCALE, false);
    const bf_h2 a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b[0], BF_H3C_SCALE, true);
    const bf_h2 a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b[1], BF_H3C_SCALE, false);
    const bf_h2 a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(b[1], BF_H3C_SCALE, true);
    return (bf_h8){a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}

// exact (erf) GELU = v * Phi(v) with one transcendental and no branch: Phi(|v|) = 1 - h, Phi(-|v|) = h, h = 0.5 (1 - erf(a)) =
// 2^(-a Q(a) - 1), a = |v| / sqrt 2, Q a degree-7 polynomial fitted to -log2(1 - erf(a)) / a on [0, 4] (a Q(a) keeps growing beyond:
// h -> 0).  |erf error| <= 1.1e-7 in fp32, GELU within 1.2e-7 |v| of 0.5 v (1 + erf(v / sqrt 2)) over [-12, 12] (erff: the same
// class of error with a two-branch polynomial that doubled the time of the MLP kernels).
__device__ __forceinline__ float bf_gelu(const float v)
{
    const float a = fabsf(v) * 0.70710678f;
    float q = fmaf(4.22452448e-05f, a, -4.19236813e-04f);
    q = fmaf(q, a, 1.40204388e-03f);
    q = fmaf(q, a, 9.21467334e-04f);
    q = fmaf(q, a, -2.83861113e-02f);
    q = fmaf(q, a, 1.48544322e-01f);
    q = fmaf(q, a, 9.18402423e-01f);
    q = fmaf(q, a, 1.62790967f);
    const float h = __builtin_amdgcn_exp2f(fmaf(-a, q, -1.f));
    return v * (v >= 0.f ? 1.f - h : h);
}

#define BF_C 16             // feature channels of the MFMA path (filters == 16)
#define BF_WPACK_FLOATS (36 * 64)   // one 3x3 16->16 kernel as MFMA A-operand register images

// epilogue stages of the 3x3 C16 convolution: [scale+shift] -> [ReLU] -> [mask] -> [+residual]
// (general form of SURVEY.md 8a "epilogue note"); STATS = per-channel sum / sum-of-squares of
// the raw convolution output for training-mode batch norm.
// BNBWD = per-channel sum of the FINAL output v and of v * bnc (the two reductions of the next BatchNorm backward,
// train_ops.hip bn_bwd_reduce_kernel) in the same [grid][32] partial format as STATS.
enum { EPI_RELU = 1, EPI_AFFINE = 2, EPI_RES = 4, EPI_MASK = 8, EPI_STATS = 16, EPI_BNBWD = 32 };

struct ConvArgs {
    const float* in;      // [B,H,W,16]
    float* out;           // [B,H,W,16]
    const float* wpack;   // [36][64]
    const float* scale;   // [16]  (EPI_AFFINE)
    const float* shift;   // [16]
    const float* res;     // [B,H,W,16] (EPI_RES)
    const float* mask;    // [B,H,W,16] (EPI_MASK): out = mask > 0 ? out : 0
    float* stats;         // [grid][32] (EPI_STATS, EPI_BNBWD)
    const float* bnc;     // [B,H,W,16] (EPI_BNBWD): raw convolution output c of the BatchNorm whose backward comes next
    int B, H, W;
    int reverse;          // conv3x3_h3_kernel: tiles from the last to the first (Infinity Cache reuse across launches)
    // conv3x3_h3_kernel, "affine + add on load" (pre_c != nullptr): the convolution's input is y = in + pre_scale * pre_c +
    // pre_shift (the BatchNorm apply + skip Add of the block in front), formed while the tile is staged and written to pre_out
    const float* pre_c; const float* pre_scale; const float* pre_shift; float* pre_out;
};

struct FusedBlockArgs {
    const float* in;      // [B,H,W,16] block input x
    float* out;           // [B,H,W,16] x + scale*conv2(relu(conv1 x)) + shift
    const float* w1pack;  // [36][64]
    const float* w2pack;  // [36][64]
    const float* scale;   // [16] folded BN scale
    const float* shift;   // [16] folded BN shift
    int B, H, W;
    int tiles_x, tiles_y, ntiles;
    int act1_relu;        // activation of conv1 (1 = relu, 0 = linear)
    const float* zeros;   // >= 64 B of zeros, 16-B aligned (source of out-of-image elements for the LDS-DMA variant)
    unsigned long long* dbg;  // diagnostic builds only (per-wave phase cycle sums), else NULL
};

// split-f16 ("f16x3") fused block on the f16 matrix cores (fused_h3.hip).  Activations are "split-planar":
// per image 4 planes [H][W][8 x f16] = hi(c0..7), hi(c8..15), lo(c0..7), lo(c8..15), value = hi + lo.
#define BF_H3_WPACK_FLOATS (10 * 64 * 4)                      // ten A-operand register images of 16 B per lane
#define BF_H3R_WPACK_FLOATS (13 * 64 * 4)                     // row-streaming kernel: 12 A-operand images + s2 * identity
// per block: w1, w2 (group kernel), aux (1/s1 | scale/s2 | shift | pad), w1r, w2r (row-streaming kernel)
#define BF_H3_BLOCK_FLOATS (2 * BF_H3_WPACK_FLOATS + 64 + 2 * BF_H3R_WPACK_FLOATS)
struct FusedH3Args {
    const void* in;       // split-planar block input x
    void* out;            // split-planar x + scale*conv2(act(conv1 x)) + shift
    const void* w1;       // [10][64] x 16 B
    const void* w2;
    const void* w1r;      // [13][64] x 16 B (row-streaming kernel)
    const void* w2r;
    const float* aux;     // [0..15] 1/s1, [16..31] scale/s2 (group kernel), [32..47] shift, [48..63] 1/s2r (row kernel)
    int B, H, W;
    int tiles_x, tiles_y, ntiles;   // filled in by the launcher
    int rows_per_tile;    // full-row streaming kernel (fused_h3v.hip): rows per band, filled in by the launcher
    int reverse_tiles;    // full-row streaming kernel: walk the bands last to first (see forward_common)
    int variant;          // kernel selection: < 0 = library default (bf_set_h3_variant), else as bf_set_option("h3_variant")
    int act1_relu;
    const void* zeros;    // >= 64 B of zeros, 16-B aligned (source of out-of-image elements)
    void* dump;           // >= 1024 B writable scratch (sink of out-of-image stores)
    unsigned long long* dbg;  // diagnostic builds only (per-wave phase cycle sums), else NULL
    // last block with the LINEAR denoiser head folded into its epilogue (head_wh != NULL; row-streaming kernel, 3 output
    // channels): the block output is not written, out3 = [round, u8](denormalise(tanh(2 y . wh) * 0.51)) cropped to [Ho,Wo]
    const float* head_wh;  // [16][4] premultiplied head_conv0 . head_conv1 (pack_edges_kernel), or NULL
    void* head_out;        // u8 or f32 [B,Ho,Wo,3]
    int head_u8, Ho, Wo, denormalize;
    float v_min, v_max;
    int* status;           // |= BF_STATUS_F16_RANGE when a block output is not finite
    int compact = 0;       // full-row streaming kernel only: in / out are compact split-planar (fp8 lo planes, see bf_h3c_encode8)
};
hipError_t bf_launch_fused_block_h3(const FusedH3Args& a, hipStream_t s);
// true when bf_launch_fused_block_h3 would run the full-row streaming kernel for these arguments (the one kernel that reads and
// writes the compact layout)
bool       bf_fused_block_h3_is_streaming(const FusedH3Args& a);
bool       bf_fused_block_h3_use_pairs(const FusedH3Args& a);   // two blocks per launch: default selection with >= 4 096 strip rows of >= 24-row images, or a forced streaming variant
const char* bf_fused_block_h3_kernel_name(const FusedH3Args& a);
const char* bf_fused_block_kernel_name();                      // conv3x3_c16.hip: the exact-fp32 fused block
// library default of FusedH3Args::variant (handle-less debug entries): 4 = full-row streaming kernel where it applies
// (W <= 256), 1 = row-streaming tile kernel, 0 / 2 / 3 = earlier tile kernels (A/B only)
void       bf_set_h3_variant(int v);
hipError_t bf_launch_fused_block_h3v(const FusedH3Args& a, hipStream_t s);      // fused_h3v.hip
bool       bf_fused_block_h3v_supports(int H, int W);
// two residual blocks per launch on 128-column strips (fused_h3w.hip); same weight images / aux / layout as FusedH3Args
struct FusedH3WArgs {
    const void* in;       // split-planar x0
    void* out;            // split-planar x2 = block_b(block_a(x0)); must not alias in
    const void* w1r[2];   // [13][64] x 16 B row-streaming images of the first convolution of block a, b
    const void* w2r[2];
    const float* aux[2];  // per block: [0..15] 1/s1, [32..47] shift, [48..63] 1/s2r
    int B, H, W;
    int nstrips, tiles_y, ntiles, rows_per_tile;    // filled in by the launcher
    int reverse_tiles;    // walk the units last to first and the rows bottom-up
    int act1_relu;
    const void* zeros;    // >= 64 B of zeros, 16-B aligned (source of rows outside the image)
    unsigned long long* dbg;  // diagnostic builds only (per-wave phase cycle sums), else NULL
    // last pair of a network with a LINEAR denoiser head of 3 output channels folded into the launch (head_wh != NULL): x2 is
    // not written; head_out = [round, u8](denormalise(tanh(2 x2 . wh) * 0.51)) cropped to [Ho, Wo] (bfcnn/model.py:297-342,
    // utilities.py:435-443, 755-764, module_denoiser.py:71-73)
    const float* head_wh;  // [16][4] premultiplied head_conv0 . head_conv1 (pack_edges_kernel), or NULL
    void* head_out;        // u8 or f32 [B,Ho,Wo,3]
    int head_u8, Ho, Wo, denormalize;
    float v_min, v_max;
    int* status;           // |= BF_STATUS_F16_RANGE when a block output is not finite; may be NULL
};
hipError_t bf_launch_fused_block2_h3w(const FusedH3WArgs& a, hipStream_t s);
bool       bf_fused_block2_h3w_supports(int H, int W);
hipError_t bf_launch_pack_h3(const float* params, const float* state, int64_t p_blocks, int64_t p_stride, float* dst,
                             int64_t d_stride, int layers, int use_bn, float eps, const float* ext_scale,
                             const float* ext_shift, hipStream_t s);
// training: single 3x3 C16 convolution on fp32 NHWC with the split-f16 arithmetic (same ConvArgs / epilogue flags / grid
// as bf_launch_conv3x3_c16; a.wpack = one BF_H3_TRAIN_PACK_FLOATS pack of bf_launch_pack_h3_train)
#define BF_H3_TRAIN_PACK_FLOATS (BF_H3R_WPACK_FLOATS + 64)    // 13 A-operand images (12 used) + 1/s
#define BF_TRAIN_PACK_STRIDE BF_H3_TRAIN_PACK_FLOATS           // per-convolution slot of the training pack area (>= BF_WPACK_FLOATS)
hipError_t bf_launch_conv3x3_h3(const ConvArgs& a, int epi, hipStream_t s);
hipError_t bf_launch_wgrad3x3_h3(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W, hipStream_t s);

// weight gradient + data gradient (+ the BatchNorm backward in front of them) of one convolution in one kernel (train_bwd_h3.hip)
struct BwdH3Args {
    const float* x;        // [B,H,W,16] input of the convolution (and, with EPI_MASK, the ReLU mask: x > 0)
    const float* g;        // gradient at the convolution's output -- or, with coef, at its BatchNorm's output
    const float* c;        // raw convolution output = BatchNorm input (coef != nullptr)
    const float* coef;     // [48] k1 | k2 | k3 of bn_bwd_finalize: dc = k1 g + k2 c + k3 ; nullptr: g is used as it is
    const float* wpack;    // data-gradient pack (pack_h3_train)
    float* out;            // data gradient; may alias res, must not alias x / g / c
    const float* res;      // EPI_RES
    const float* bnc;      // EPI_BNBWD: input of the BatchNorm whose backward comes next (sums of out, out * bnc -> stats)
    float* wpartial;       // [grid][2304]
    float* stats;          // [grid][32]
    int B, H, W, reverse;  // reverse: walk the tiles from the last to the first (Infinity Cache reuse across launches)
    int dbuf;              // 1: the 512-thread kernel with double-buffered LDS images (one workgroup per CU, 256 partial rows)
    int* grid_out;         // if not NULL: number of partial rows (workgroups) the launch wrote
    int tiles_x, tiles_y, ntiles;               // filled in by the launcher
};
// backward of BOTH convolutions of a [3,3] block (second one followed by a BatchNorm, ReLU between them) in one kernel
// (train_bwd2_h3.hip): dT never leaves the CU
struct Bwd2H3Args {
    const float* t;        // [B,H,W,16] input of the second convolution = relu(conv1 A) (also the ReLU mask)
    const float* a;        // [B,H,W,16] block input A (input of the first convolution)
    const float* dy;       // gradient at the block's output (= at the BatchNorm's output, = the skip's gradient)
    const float* c;        // raw output of the second convolution = BatchNorm input
    const float* coef;     // [48] k1 | k2 | k3 of bn_bwd_finalize: dc = k1 dy + k2 c + k3
    const float* wpack2;   // data-gradient packs (pack_h3_train) of the second / first convolution
    const float* wpack1;
    const float* bnc;      // input of the BatchNorm of the block in front (sums of out, out * bnc -> stats) or NULL
    float* out;            // dA' = dgrad1(dT) + dy; must not alias any input
    float* wpartial2;      // [grid][2304] each
    float* wpartial1;
    float* stats;          // [grid][32] (bnc != NULL)
    int B, H, W, reverse;
    int* grid_out;         // if not NULL: number of partial rows (workgroups) the launch wrote
    int tiles_x, tiles_y, ntiles;               // filled in by the launcher
};
// training-mode forward of one [3,3] block in one kernel (train_fwd_h3t.hip): A_i = x + pre_scale * pre_c + pre_shift formed on
// load (pre_c != NULL; written to a_out), T_i = act(conv_0 A_i) kept in LDS (written to t_out when not NULL), C_i = conv_1 T_i
// written to c_out, per-channel sum / sum of squares of C_i to stats[grid][32] (the format of EPI_STATS)
struct FwdBlockH3Args {
    const float* x;        // [B,H,W,16] A_{i-1} (pre_c != NULL) or the block input A_i itself
    const float* pre_c;    // raw output of the previous block's last convolution, or NULL
    const float* pre_scale; const float* pre_shift;     // [16] each: that block's BatchNorm as scale / shift
    float* a_out;          // A_i (pre_c != NULL)
    float* t_out;          // T_i or NULL
    float* c_out;          // C_i
    const float* wpack0;   // forward packs (pack_h3_train) of conv_0 / conv_1
    const float* wpack1;
    float* stats;          // [bf_fwd_block_h3t_grid][32]
    int B, H, W, reverse, act_relu;
    int tiles_y, ntiles, rows_per_tile;         // filled in by the launcher
    // round 4: the BatchNorm finalisation of the block in FRONT (bn_finalize_kernel: 6 us + two kernel boundaries per block) done by every
    // workgroup in its prologue instead: fin_partial = that block's [fin_nblk][32] sums (a DIFFERENT buffer than `stats`), scale | shift |
    // mean | 1 / sigma and the moving statistics written by workgroup 0.  NULL: pre_scale / pre_shift are read as given.
    const float* fin_partial;
    int fin_nblk;
    double fin_count;
    const float* fin_gamma;
    float *fin_mm, *fin_mv;        // moving mean / variance [16]
    float fin_eps, fin_momentum;
    float* fin_scale;              // [32] scale | shift (out)
    float* fin_meaninv;            // [32] mean | 1 / sigma (out)
};
bool       bf_fwd_block_h3t_supports(int H, int W);
int        bf_fwd_block_h3t_grid(int B, int H, int W);
hipError_t bf_launch_fwd_block_h3t(const FwdBlockH3Args& a, hipStream_t s);
// backward of one [3,3] block (BatchNorm behind conv_1, [ReLU] between the convolutions) in one kernel with T RECOMPUTED from A
// (train_bwd_h3t.hip): dc = k1 g + k2 c + k3 ; T = act(conv_0 a) ; dW1 = T^T dc ; dT = dgrad_1(dc) * (T > 0) ; dW0 = a^T dT ;
// out = dgrad_0(dT) + g ; stats = sums of out and out * bnc (bnc != NULL)
struct BwdBlockH3Args {
    const float* a;        // [B,H,W,16] block input A_i
    const float* g;        // gradient at the block's output
    const float* c;        // raw output of conv_1 = BatchNorm input
    const float* coef;     // [48] k1 | k2 | k3 of bn_bwd_finalize
    const float* wfwd0;    // forward pack of conv_0 (pack_h3_train)
    const float* wdg1;     // data-gradient packs of conv_1 / conv_0
    const float* wdg0;
    const float* bnc;      // input of the BatchNorm of the block in front, or NULL
    float* out;            // must not alias a, g, c, bnc
    float* wpartial1;      // [bf_bwd_block_h3t_grid][2304]: weight gradient of conv_1
    float* wpartial0;      //                                 ... of conv_0
    float* stats;          // [grid][32] (bnc != NULL)
    int B, H, W, reverse, act_relu;
    int nstrips, tiles_y, ntiles, rows_per_tile;       // filled in by the launcher
    unsigned long long* dbg;   // diagnostic builds only (per-wave phase cycle sums), else NULL
    // round 4: bn_bwd_finalize_kernel in the prologue of every workgroup: fin_partial = [fin_nblk][32] sums (sum g | sum g * c) the
    // launch before this one wrote (a DIFFERENT buffer than `stats`); k1 | k2 | k3 go to LDS, d gamma is written by workgroup 0.
    // NULL: `coef` is read as given.
    const float* fin_partial;
    int fin_nblk;
    double fin_count;
    const float* fin_gamma;
    const float* fin_meaninv;      // [32] mean | 1 / sigma of this block's BatchNorm
    float* fin_dgamma;             // [16] (out)
};
bool       bf_bwd_block_h3t_supports(int H, int W);
int        bf_bwd_block_h3t_grid(int B, int H, int W);
hipError_t bf_launch_bwd_block_h3t(const BwdBlockH3Args& a, hipStream_t s);
int        bf_bwd2_h3_grid(int B, int H, int W);
hipError_t bf_launch_bwd2_h3(const Bwd2H3Args& a, hipStream_t s);
int        bf_bwd3x3_h3_grid(int B, int H, int W);                 // partial rows of the 256-thread kernel (the larger count: sizing)
int        bf_bwd3x3_h3_grid_ex(int B, int H, int W, int dbuf);    // partial rows a launch writes
hipError_t bf_launch_bwd3x3_h3(const BwdH3Args& a, int epi, float* dw, hipStream_t s);
hipError_t bf_launch_reduce_wgrad_slots(const float* slots, int64_t slot_floats, int nblk, float* out, int64_t p_stride, int layers,
                                        int nconv, int unit, hipStream_t s);
// nconv convolutions per block (forward + data-gradient pack each); unit: floats from one convolution kernel of a block to the
// next behind the first (2320 with BatchNorm gammas in between, else 2304)
hipError_t bf_launch_pack_h3_train(const float* params, int64_t p_blocks, int64_t p_stride, float* dst, int layers, int nconv,
                                   int unit, hipStream_t s);
hipError_t bf_launch_h3_from_f32(const float* x, void* y, int B, int H, int W, hipStream_t s);
hipError_t bf_launch_h3_to_f32(const void* y, float* x, int B, int H, int W, hipStream_t s);

// ---- launchers (each returns hipGetLastError()) -------------------------------------------
hipError_t bf_launch_conv3x3_c16(const ConvArgs& a, int epi, hipStream_t s);
int        bf_conv3x3_c16_grid(int B, int H, int W);
hipError_t bf_launch_fused_block(const FusedBlockArgs& a, hipStream_t s);
void       bf_set_fused_tile(int variant);   // A/B of tile geometries (process-wide, tools/ only)
hipError_t bf_launch_pack_conv(const float* w_hwio, float* wpack, int transpose_flip, hipStream_t s);
hipError_t bf_launch_wgrad3x3_c16(const float* x, const float* dy, float* partial, float* dw,
                                  int B, int H, int W, hipStream_t s);
int        bf_wgrad_grid(int B, int H, int W);

// base convolution k x k, Cin -> 16 on the normalised input (u8 or f32 source) with virtual
// power-of-two padding: source image is [B,Hs,Ws,Cin]; activations are [B,H,W,16] with H>=Hs.
struct BaseConvArgs {
    const void* in; float* out; const float* w;  // w: [k,k,cin,16] HWIO
    int B, Hs, Ws, H, W, cin, k, in_is_u8, act_relu;
    float v_min, v_max;
    int out_split;        // 1: write split-planar f16 hi/lo (input of the f16x3 blocks) instead of fp32 NHWC; 2: compact (fp8 lo planes)
    int* status;          // inference: forward status word, zeroed here (first kernel of a forward); may be NULL
};
hipError_t bf_launch_base_conv(const BaseConvArgs& a, hipStream_t s);
bool       bf_base_conv_rows_supports(const BaseConvArgs& a);          // base_rows.hip: u8, 3x3x3 -> 16, [0, 255], split-planar out
hipError_t bf_launch_base_conv_rows(const BaseConvArgs& a, hipStream_t s);
void       bf_set_base_conv_rows(int on);     // 1 (default): the row-streaming matrix-core kernel where it applies; 0: vector kernel (A/B)
// dW[k,k,cin,16] = sum xn (x) dy ; partial = [grid][k*k*cin*16]
hipError_t bf_launch_base_wgrad(const float* in_f32, const float* dy, float* partial, float* dw,
                                int B, int H, int W, int cin, int k, float v_min, float v_max, hipStream_t s);
int        bf_base_wgrad_grid(int B, int H, int W);

struct HeadArgs {
    const float* feat;    // [B,H,W,16]
    const float* w0;      // [16,hf]
    const float* w1;      // [hf,cout]
    const float* wh;      // [16,4] premultiplied (linear head) or NULL
    void* out;            // u8 or f32 [B,Ho,Wo,cout]
    int B, H, W, Ho, Wo, hf, cout, act, out_is_u8, denormalize;
    float v_min, v_max, leaky_alpha;
    int feat_split;       // 1: feat is split-planar f16 hi/lo; 2: compact split-planar (fp8 lo planes)
    int* status;          // |= BF_STATUS_F16_RANGE when a split-planar feature is not finite (f16 overflow upstream); may be NULL
};
#define BF_STATUS_BYTES 2048  // tail of the inference workspace: [0] status word, [1024, 2048) store sink of out-of-image lanes

hipError_t bf_launch_head(const HeadArgs& a, hipStream_t s);

struct HeadTrainArgs {
    const float* feat; const float* wh;      // [16,4]
    const float* gt; float* pred;            // [B,H,W,cout]; pred may be NULL
    float* dfeat;                            // [B,H,W,16]
    float* partial;                          // [grid][80]: M[16][4] (64), sums: |e|, hinge |e|, relu(e)^2, relu(e, hinge, cutoff^2)^2
    const float* dextra;                     // [B,H,W,cout] added to dL/dpred (RMSE / SSIM terms, loss_terms.hip) or NULL
    int B, H, W, cout, denormalize;
    float v_min, v_max, hinge, cutoff, dscale; // dscale = mae_multiplier*depth_weight/numel
    float dfeat_scale;                         // power of two on the dfeat output only (gradient scaling of the split-f16
                                               // backward, engine.hip bf_train_step); the head's own gradients (M) are not scaled
};
hipError_t bf_launch_head_train(const HeadTrainArgs& a, int grid, hipStream_t s);
// RMSE / SSIM loss terms (loss_terms.hip): additive dL/dpred from the prediction and the head's per-image sums
hipError_t bf_launch_loss_extra(const float* pred, const float* gt, int B, int H, int W, int C, const float* head_partial,
                                int blocks_per_image, float hinge, float cutoff, float mse_multiplier, float ssim_multiplier,
                                float depth_weight, float max_val, float* maps, float* ssim_partial, float* coef, float* scal,
                                float* dextra, hipStream_t s);
hipError_t bf_launch_loss_extra_finalize(const float* scal, int B, int H, int W, int C, float mse_multiplier, float ssim_multiplier,
                                         float depth_weight, float* losses, hipStream_t s);
int        bf_head_train_grid(int B, int H, int W);

// elementwise / reductions
hipError_t bf_launch_bn_finalize(const float* partial, int nblk, double count, const float* gamma,
                                 float* moving_mean, float* moving_var, float eps, float momentum,
                                 float* scale, float* shift, float* mean_inv /*[32]*/, double* stage1 /*[64*32] or NULL*/,
                                 hipStream_t s);
hipError_t bf_launch_affine_add(const float* x, const float* c, const float* scale, const float* shift,
                                float* y, int64_t npix, hipStream_t s);   // y = x + scale*c + shift
hipError_t bf_launch_affine_act(const float* c, const float* scale, const float* shift, float* y, int relu, int64_t npix,
                                hipStream_t s);                           // y = [relu](scale*c + shift)
hipError_t bf_launch_bn_bwd_reduce(const float* dy, const float* c, float* partial, int64_t npix, int grid, hipStream_t s);
hipError_t bf_launch_bn_bwd_finalize(const float* partial, int nblk, double count, const float* gamma,
                                     const float* mean_inv, float* coef /*[48]: k1,k2,k3*/, float* dgamma,
                                     double* stage1 /*[64*32] or NULL*/, hipStream_t s);
hipError_t bf_launch_bn_bwd_apply(const float* dy, const float* c, const float* coef, float* dc, int64_t npix, hipStream_t s);
hipError_t bf_launch_reduce_partials(const float* partial, int nblk, int width, float* out, float scale, hipStream_t s);
hipError_t bf_launch_zero(float* p, int64_t n, hipStream_t s);
