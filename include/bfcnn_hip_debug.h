/* Diagnostic entry points of libbfcnn_hip.so: ONE kernel of the engine at a time on fp32 NHWC tensors, through the same C ABI
 * conventions as bfcnn_hip.h (plain pointers and sizes, caller-owned buffers, `stream` = hipStream_t or NULL, 0 / negative BF_E*
 * return codes).  Users: tests/ (every kernel is held to the fp64 oracle on its own), tools/ (stamps, ablations, A/B) and
 * bench.py --mode train (live timing of the dominant backward kernel).  NOT part of the drop-in boundary: a host that replaces
 * the reference's path binds bfcnn_hip.h only. */
#ifndef BFCNN_HIP_DEBUG_H
#define BFCNN_HIP_DEBUG_H
#include "bfcnn_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* single 3x3 16->16 convolution with epilogue flags (1 relu, 2 affine, 4 residual, 8 mask,
 * 16 stats); transpose_flip = 1 runs the data-gradient form.  wpack_scratch = 2*2304 + 64 floats. */
int bf_debug_conv3x3(const float* in, const float* w_hwio, float* out, const float* scale, const float* shift,
                     const float* res, const float* mask, float* stats, float* wpack_scratch,
                     int batch, int height, int width, int epi, int transpose_flip, void* stream);
int bf_debug_conv3x3_grid(int batch, int height, int width);
int bf_debug_fused_block(const float* in, const float* w1_hwio, const float* w2_hwio, const float* scale,
                         const float* shift, float* out, float* wpack_scratch,
                         int batch, int height, int width, int act1_relu, void* stream);
/* the split-f16 fused block on fp32 NHWC tensors (converts in and out); scratch = the float count below */
int64_t bf_debug_fused_block_h3_scratch_floats(int batch, int height, int width);
int bf_debug_fused_block_h3(const float* in, const float* w1_hwio, const float* w2_hwio, const float* scale,
                            const float* shift, float* out, float* scratch,
                            int batch, int height, int width, int act1_relu, void* stream);
/* two consecutive split-f16 fused blocks in ONE launch (fused_block2_h3w_kernel) on fp32 NHWC tensors: w_hwio = [4][3][3][16][16]
   (conv1 and conv2 of block a, then of block b), scale / shift = [2][16]; reverse = 1 walks the bands bottom-up */
int64_t bf_debug_fused_block2_h3_scratch_floats(int batch, int height, int width);
int bf_debug_fused_block2_h3(const float* in, const float* w_hwio, const float* scale, const float* shift, float* out,
                             float* scratch, int batch, int height, int width, int act1_relu, int reverse, void* stream);
/* kernel the handle-less entry above launches (a handle's own choice is bf_set_option "h3_variant"): 4 full-row streaming
   (falls back to 1 beyond 256 columns), 1 row-streaming tiles, 0 / 2 / 3 earlier tile kernels; < 0 = library default */
int bf_debug_set_h3_variant(int variant);
int64_t bf_debug_conv3x3_h3_scratch_floats(void);
int bf_debug_conv3x3_h3(const float* in, const float* w_hwio, float* out, const float* res, const float* mask, float* stats,
                        float* scratch, int batch, int height, int width, int epi, int transpose_flip, void* stream);
int64_t bf_debug_wgrad_partial_floats(int batch, int height, int width);
int bf_debug_wgrad3x3(const float* x, const float* dy, float* partial, float* dw,
                      int batch, int height, int width, void* stream);
int bf_debug_wgrad3x3_h3(const float* x, const float* dy, float* partial, float* dw,
                         int batch, int height, int width, void* stream);
/* split-f16 training kernels of bf_train_step, one at a time (tests, bench.py's live roofline):
   conv3x3_h3 with the BatchNorm apply + skip Add of the block in front formed on load (y = in + pre_scale * pre_c + pre_shift
   -> pre_out; out = [relu] conv(y)); and the fused backward of one convolution: dw = x^T g', dx = dgrad(g') [* (x > 0) with
   epi 8 | + res with epi 4 | + sums of dx and dx * bnc with epi 4 + 32], g' = coef[0:16] * g + coef[16:32] * c + coef[32:48]
   when coef is not NULL (the BatchNorm backward of bfcnn/backbone_blocks.py:214-240's BatchNormalization). */
int bf_debug_conv3x3_h3_pre(const float* in, const float* pre_c, const float* pre_scale, const float* pre_shift, float* pre_out,
                            const float* w_hwio, float* out, float* scratch, int batch, int height, int width, int relu,
                            int reverse, void* stream);
/* the training-mode forward of one [3,3] block in one kernel (train_fwd_h3t.hip; bfcnn/backbone_blocks.py:174-246 under training=True):
   a_out = x + pre_scale * pre_c + pre_shift (pre_c not NULL), t_out = [relu] conv_0(a) (t_out not NULL), c_out = conv_1(t),
   stats[32] = per-channel sum | sum of squares of c_out.  width <= 256. */
int64_t bf_debug_fwd_block_h3t_scratch_floats(int batch, int height, int width);
int bf_debug_fwd_block_h3t(const float* x, const float* pre_c, const float* pre_scale, const float* pre_shift, const float* w0_hwio,
                           const float* w1_hwio, float* a_out, float* t_out, float* c_out, float* stats, float* scratch, int batch,
                           int height, int width, int relu, int reverse, void* stream);
/* the backward of one [3,3] block in one kernel, T recomputed from the block input (train_bwd_h3t.hip; tape.gradient of
   bfcnn/train_loop.py:273-294 through bfcnn/backbone_blocks.py:174-246): dc = coef[0:16] * g + coef[16:32] * c + coef[32:48],
   T = [relu] conv_0(a), dw1 = T^T dc, dT = dgrad_1(dc) [* (T > 0)], dw0 = a^T dT, out = dgrad_0(dT) + g, stats[32] = per-channel sums
   of out | out * bnc (bnc not NULL).  reverse: bit 0 = bands bottom-up, bit 1 = the kernel alone (weights packed by an earlier call on the
   same scratch, partials not reduced: live timing of the launch). */
int64_t bf_debug_bwd_block_h3t_scratch_floats(int batch, int height, int width);
int bf_debug_bwd_block_h3t(const float* a, const float* g, const float* c, const float* coef, const float* w0_hwio, const float* w1_hwio,
                           const float* bnc, float* out, float* dw1, float* dw0, float* stats, float* scratch, int batch, int height,
                           int width, int relu, int reverse, void* stream);
int64_t bf_debug_bwd3x3_h3_scratch_floats(int batch, int height, int width);
int bf_debug_bwd3x3_h3_grid(int batch, int height, int width);
int bf_debug_bwd3x3_h3_grid_ex(int batch, int height, int width, int dbuf);    /* partial rows written; `reverse` bit 1 of the call below = dbuf */
int bf_debug_bwd3x3_h3(const float* x, const float* g, const float* c, const float* coef, const float* w_hwio, float* out,
                       const float* res, const float* bnc, float* dw, float* stats, float* scratch, int batch, int height,
                       int width, int epi, int reverse, int repack, void* stream);
int bf_debug_mfma_probe(const float* a, const float* b, float* d, void* stream);
/* bf_upsample2x on C % 4 != 0 maps: 1 (default) the row-walking 16-byte kernel, 0 the 4-byte row kernel (same bits; A/B and tests) */
int bf_debug_set_upsample_band(int on);

#ifdef __cplusplus
}
#endif
#endif /* BFCNN_HIP_DEBUG_H */
