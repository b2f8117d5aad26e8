/*
 * bfcnn_hip.h -- C ABI of the MI355X (gfx950) engine for the bfcnn resnet-denoiser hot path.
 *
 * The reference (NikolasMarkou/blind_image_denoising, bfcnn 3.2.0) is pure Python on
 * TensorFlow/Keras: it has no FFI of its own.  These entry points are what a binding for
 * its hot path replaces; every declaration cites the reference interface (file:line under
 * the reference root) whose behaviour it reproduces.  The reference-side stub a maintainer
 * would add is shown in INTEGRATION.md (ctypes, because the reference host is Python).
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (a stream is passed as void*
 *     = hipStream_t, NULL = the null stream);
 *   - every tensor pointer is DEVICE memory on the handle's device unless the name ends in
 *     `_host`; activations are NHWC, contiguous; kernels are HWIO ([kh,kw,cin,cout]) exactly
 *     as keras stores them;
 *   - the caller owns every buffer, including the workspace (size from bf_workspace_bytes);
 *     the library never allocates device memory, never synchronises and spawns no threads:
 *     all work is enqueued on the given stream (graph-capturable);
 *   - return value 0 = BF_OK, negative = error; text via bf_last_error().
 */
#ifndef BFCNN_HIP_H
#define BFCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFCNN_ABI_VERSION 1

typedef struct bf_engine* bf_handle;

enum bf_status {
    BF_OK = 0,
    BF_EINVAL = -1,       /* bad argument (reference raises ValueError)            */
    BF_EUNSUPPORTED = -2, /* config outside the hot path (NotImplementedError)     */
    BF_EWORKSPACE = -3,   /* workspace / packed buffer too small or misaligned     */
    BF_EHIP = -4          /* HIP runtime error on launch                           */
};

enum bf_activation { BF_ACT_LINEAR = 0, BF_ACT_RELU = 1, BF_ACT_LEAKY_RELU = 2 };
enum bf_regularizer { BF_REG_NONE = 0, BF_REG_L1 = 1, BF_REG_L2 = 2 };
enum bf_mode { BF_MODE_INFERENCE = 0, BF_MODE_TRAIN = 1 };
enum bf_tensor_kind { BF_KIND_CONV = 0, BF_KIND_GAMMA = 1, BF_KIND_MOVING_MEAN = 2, BF_KIND_MOVING_VAR = 3 };

/* The arguments of bfcnn/backbone_resnet.py:19-50 (builder) and bfcnn/model.py:251-275
 * (model_denoiser_builder) that the hot path uses.  The host parses the reference's
 * pipeline JSON (bfcnn/utilities.py:59-96) into this struct. */
typedef struct bf_resnet_desc {
    int32_t struct_size;      /* = sizeof(bf_resnet_desc)                                   */
    int32_t in_channels;      /* input_shape[-1], 1..4                                       */
    int32_t filters;          /* `filters`; the MFMA path is built for 16                    */
    int32_t kernel_size;      /* base conv kernel_size: 1,3,5,7                              */
    int32_t no_layers;        /* number of residual blocks                                   */
    int32_t block_convs;      /* len(block_kernels); 2 (fused, trainable) | 1 | 3 (inference)  */
    int32_t block_kernel;     /* block_kernels[*]; 3                                         */
    int32_t activation;       /* bf_activation of the first block conv                       */
    int32_t base_activation;  /* bf_activation of base conv and last block conv (linear)     */
    int32_t use_bn;           /* BatchNormalization(scale=True, center=False) after conv2    */
    int32_t head_filters;     /* denoiser `filters` (default 32), <= 64                      */
    int32_t head_activation;  /* denoiser `activation` (default linear)                      */
    int32_t out_channels;     /* denoiser `output_channels`, 1..4                            */
    int32_t denormalize;      /* 1: always denormalise; 0: literal snapshot graph (model.py:110-116) */
    int32_t reg_base;         /* bf_regularizer: kernel_regularizer of the base conv         */
    int32_t reg_block;        /* block_regularizer                                           */
    int32_t reg_head;         /* denoiser kernel_regularizer                                 */
    float v_min, v_max;       /* value_range                                                 */
    float bn_eps, bn_momentum;/* constants.py:9,11                                           */
    float leaky_alpha;        /* slope when an activation is BF_ACT_LEAKY_RELU               */
} bf_resnet_desc;

/* bfcnn/loss.py:162-179 + the depth weight of train_loop.py:284-285. */
typedef struct bf_loss_desc {
    int32_t struct_size;
    float hinge, cutoff;
    float mae_multiplier;
    float mse_multiplier;     /* > 0: + rmse(hinge, cutoff^2) * this (loss.py:228-235)       */
    float ssim_multiplier;    /* > 0: + (1 - mean tf.image.ssim(7x7, max 255)) * this (:219-227) */
    float regularization;
    float depth_weight;
} bf_loss_desc;

/* losses written by bf_train_step, device float[BF_LOSS_COUNT] (keys of constants.py:35-50). */
enum bf_loss_slot {
    BF_LOSS_TOTAL = 0,          /* p_total_loss (train_loop.py:299-301)                      */
    BF_LOSS_DENOISER_TOTAL = 1, /* denoiser_loss[total_loss]                                 */
    BF_LOSS_MAE = 2,            /* mae_loss (no hinge)                                       */
    BF_LOSS_MSE = 3,            /* mse_loss = rmse (no hinge)                                */
    BF_LOSS_SSIM = 4,           /* ssim_loss = 1 - mean ssim (0 when the term is off)        */
    BF_LOSS_REGULARIZATION = 5, /* model_loss[regularization_loss]                           */
    BF_LOSS_MODEL_TOTAL = 6,    /* model_loss[total_loss]                                    */
    BF_LOSS_GRAD_NORM = 7,      /* global L2 norm of the last gradient given to bf_adam_step */
    BF_LOSS_COUNT = 8
};

typedef struct bf_tensor_info {
    char name[64];
    int64_t offset;           /* element offset in the flat params (or state) buffer         */
    int32_t rank;
    int32_t shape[4];
    int32_t kind;             /* bf_tensor_kind                                              */
    int32_t regularizer;      /* bf_regularizer                                              */
} bf_tensor_info;

/* ---- lifetime ----------------------------------------------------------------------- */

int bf_abi_version(void);

/* model_builder(config) (bfcnn/model.py:58-162): validates the description and builds the
 * launch plan.  No device memory is touched.  *out = NULL on failure; bf_last_error(NULL)
 * then holds the reason. */
int bf_create(const bf_resnet_desc* desc, bf_handle* out);
void bf_destroy(bf_handle h);
const char* bf_last_error(bf_handle h);

/* ---- parameter inventory (keras trainable_variables / non-trainable BN stats) -------- */

int64_t bf_param_count(bf_handle h);   /* trainable floats: conv kernels + BN gammas         */
int64_t bf_state_count(bf_handle h);   /* BN moving_mean / moving_variance floats            */
int bf_tensor_count(bf_handle h, int state);
int bf_tensor_at(bf_handle h, int state, int index, bf_tensor_info* out);

/* ---- inference ---------------------------------------------------------------------- */

int64_t bf_packed_bytes(bf_handle h);
int64_t bf_workspace_bytes(bf_handle h, int mode, int batch, int height, int width);

/* Re-lays the weights for the kernels (MFMA operand images, BN folded to scale/shift with
 * the inference formula of keras BatchNormalization: gamma*(x-moving_mean)*rsqrt(var+eps)).
 * Call after every change of params/state. */
int bf_pack_inference(bf_handle h, const float* params, const float* state, void* packed, void* stream);

/* DenoiserModule.__call__ (bfcnn/module_denoiser.py:46-75): uint8 [B,H,W,C] -> uint8 [B,H,W,Cout]:
 * cast, pad_to_power_of_2 (utilities.py:736-751), hydra, remove_padding, round-half-even, cast. */
int bf_forward_u8(bf_handle h, const void* packed, const uint8_t* in, uint8_t* out,
                  int batch, int height, int width, void* workspace, int64_t workspace_bytes, void* stream);

/* DenoiserModule(cast_to_uint8=False).__call__ (bfcnn/module_denoiser.py:66-75 without the cast branch): the same chain,
 * float32 [B,H,W,Cout] out, not rounded. */
int bf_forward_u8_f32(bf_handle h, const void* packed, const uint8_t* in, float* out,
                      int batch, int height, int width, void* workspace, int64_t workspace_bytes, void* stream);

/* hydra(x, training=False) (bfcnn/model.py:91-151; test_step train_loop.py:253-257):
 * float32 [B,H,W,C] in value_range -> float32 [B,H,W,Cout].  No power-of-two padding. */
int bf_forward_f32(bf_handle h, const void* packed, const float* in, float* out,
                   int batch, int height, int width, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training ----------------------------------------------------------------------- */

/* train_step_single_gpu (bfcnn/train_loop.py:259-312) for the single-output resnet hydra:
 * training-mode forward (BN batch statistics, moving stats updated in `state`), L1 loss with
 * hinge/cutoff (loss.py:40-65,190-247) * depth_weight + regularisation (loss.py:181-187),
 * gradients of the total w.r.t. every trainable variable into `grads` (flat, same layout as
 * params).  `predictions` may be NULL.  `losses` = device float[BF_LOSS_COUNT]. */
int bf_train_step(bf_handle h, const float* params, float* state, const float* gt, const float* noisy,
                  int batch, int height, int width, const bf_loss_desc* loss,
                  float* predictions, float* grads, float* losses,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* apply_grads (bfcnn/train_loop.py:314-321) with the optimizer of bfcnn/optimizer.py:190-206:
 * keras-2.13 Adam, optional global_clipnorm (<=0 disables), grads pre-scaled by grad_scale
 * (1/world_size after a sum all-reduce).  `iterations` = optimizer.iterations before the step;
 * `lr` = schedule(iterations) evaluated by the host.  losses[BF_LOSS_GRAD_NORM] receives the
 * (scaled) global norm when `losses` is not NULL.  scratch = device float[>=2] (may alias
 * the workspace). */
int bf_adam_step(bf_handle h, float* params, const float* grads, float* m, float* v,
                 int64_t iterations, float lr, float beta_1, float beta_2, float epsilon,
                 float global_clipnorm, float grad_scale, float* losses, float* scratch, void* stream);

/* The same with keras' other two clipping modes (optimizer.py:165-169): clipnorm = tf.clip_by_norm on every gradient tensor,
 * clipvalue = clip_by_value; precedence as keras 2.13: clipnorm, else global_clipnorm, else clipvalue (<= 0 disables each).
 * tensor_offsets = device int64[n_tensors + 1]: offsets of the trainable tensors in the flat vector, last = parameter
 * count (bf_tensor_info order); tensor_scratch = device float[n_tensors]. */
int bf_adam_step_ex(bf_handle h, float* params, const float* grads, float* m, float* v,
                    int64_t iterations, float lr, float beta_1, float beta_2, float epsilon,
                    float global_clipnorm, float clipnorm, float clipvalue, const int64_t* tensor_offsets, int n_tensors,
                    float* tensor_scratch, float grad_scale, float* losses, float* scratch, void* stream);

/* ---- pyramid / resampling (bfcnn/pyramid.py, upsampling.py, downsampling.py) ---------- */

/* AveragePooling2D(pool_size=(kh,kw), strides=2, padding="same") (pyramid.py:266-270,374-378). */
int bf_avgpool_s2_same(const float* in, float* out, int batch, int height, int width, int channels,
                       int kh, int kw, void* stream);
/* tf.nn.avg_pool2d(2x2, stride 2, VALID) [+clip 0..255][+round] (utilities.py:642-672). */
int bf_avgpool2_valid(const float* in, float* out, int batch, int height, int width, int channels,
                      int clip_values, int round_values, void* stream);
/* UpSampling2D(2, "bilinear"|"nearest") (pyramid.py:319-325; upsampling.py:65,105):
 * out = up(in) [+ add][- sub...]: out = alpha*up(in) + beta*other (other may be NULL). */
int bf_upsample2x(const float* in, const float* other, float* out, int batch, int height, int width,
                  int channels, int bilinear, float alpha, float beta, void* stream);
/* one level of the Laplacian split in ONE kernel (bfcnn/pyramid.py:374-385): down = AveragePooling2D((kh,kw), strides 2, same)(in)
 * [B,H/2,W/2,C] and lap = in - UpSampling2D(2, bilinear)(down) [B,H,W,C]; x is read once (2.25 n floats of traffic instead of the
 * 3.5 n of bf_avgpool_s2_same + bf_upsample2x) and the results are bitwise those of the two calls.  Returns BF_EUNSUPPORTED
 * without launching anything for shapes it does not take (odd H or W, W*C not a multiple of 4, C a multiple of 4, kh not in
 * {3,5,7}, unaligned tensors): the caller then makes the two calls. */
int bf_laplacian_split(const float* in, float* down, float* lap, int batch, int height, int width, int channels, int kh, int kw,
                       void* stream);
/* x[:, ::2, ::2, :] (downsampling.py:61). */
int bf_strided_slice2(const float* in, float* out, int batch, int height, int width, int channels, void* stream);

/* prepare_data_fn of bfcnn/dataset.py:126-239 on a device batch [B,H,W,C] of floats in value range (not in place):
 *   out_clean = round(flip(in))                               (geometric_augmentation_fn :131-159, tf.round :234)
 *   out_noisy = round(out_clean * (1 + mult_std * tn) + add_std * tn')   (noise_augmentation_fn :161-230)
 * tn, tn' = standard normals truncated at +-2 (tf.random.truncated_normal re-picks outside 2 sigma), drawn per element
 * from Philox4x32-10 (counter = element index, key = seed).  The per-BATCH random choices of the reference -- the two
 * flips (flip_mask bit 0 left-right, bit 1 up-down), whether each noise is applied (std = 0 disables it) and the
 * standard deviations ~ U[min,max] -- are the caller's (blind_image_denoising_amd.dataset draws them on the host).
 * out_clean may be NULL. */
int bf_noise_augment(const float* in, float* out_clean, float* out_noisy, int batch, int height, int width, int channels,
                     int flip_mask, float mult_std, float add_std, uint64_t seed, void* stream);

/* ---- unet_laplacian backbone operators (BASELINE.json configs[4]) ------------------------------
 * What bfcnn/backbone_unet_laplacian.py:281-606 builds from keras layers, as fp32 NHWC device operators; the host
 * (blind_image_denoising_amd/unet_laplacian.py) chains them as the reference builder chains its layers.
 * All pointers are 16-byte aligned device buffers; activation codes: BF_ACT_LINEAR 0, BF_ACT_RELU 1, 2 leaky relu
 * (slope `alpha`), 3 gelu (erf form), 4 tanh.  Unsupported channel counts return BF_EUNSUPPORTED. */

/* weights of a 1x1 Conv2D [cin][cout] (HWIO) -> matrix-core operand order; cin, cout multiples of 16. */
int bf_op_pack_pointwise(const float* w, float* wp, int cin, int cout, void* stream);
/* Conv2D 1x1, use_bias=False (utilities.py:196): out = res + mult * act(in . w); mult [cout] / res [npix][cout] may be NULL. */
int bf_op_pointwise(const float* in, float* out, const float* wp, const float* mult, const float* res, int64_t npix,
                    int cin, int cout, int act, float alpha, void* stream);
/* Conv2D kh x kw, strides s, padding="same" (utilities.py:196): out = res + act(conv(in) + bias), bias [cout] (a folded
 * BatchNorm shift) / res may be NULL; wp = kh*kw
 * tap matrices [cin][cout], each packed by bf_op_pack_pointwise, tap-major; cin, cout in {32, 64, 128}.  Serves the
 * "conv2d" downsample (2x2 stride 2, downsampling.py:45-55) and the 3x3 convolution of upsample_bilinear_conv2d /
 * upsample_nearest_conv2d (upsampling.py:52-72). */
int bf_op_conv2d(const float* in, float* out, const float* wp, const float* res, const float* bias, int batch, int height,
                 int width, int cin, int cout, int kh, int kw, int stride, int act, float alpha, void* stream);
/* DepthwiseConv2D k x k, depth_multiplier m (output channel c*m + j), padding="same", + bias[C*m] (folded BatchNorm shift,
 * may be NULL) + activation (backbone_resnet.py:165-176, block_depthwise); w [k][k][C][m]. */
int bf_op_dwconv_mult(const float* in, float* out, const float* w, const float* bias, int batch, int height, int width,
                      int channels, int multiplier, int k, int act, float alpha, void* stream);
/* bf_op_dwconv_mult followed by a 1x1 convolution cin*m -> cout (wp packed by bf_op_pack_pointwise) in one kernel:
 * out = res + act2(act1(dw(in) + bias1) . w + bias2); the cin*m-wide tensor is never written (the bottleneck tail of the
 * shipped resnet config: depthwise 3x3 x4 -> BN -> ReLU -> grouped 1x1 -> BN -> Add). */
int bf_op_dwmult_pointwise(const float* in, float* out, const float* wd, const float* bias1, int act1, float alpha1,
                           const float* wp, const float* bias2, int act2, float alpha2, const float* res, int batch, int height,
                           int width, int cin, int multiplier, int k, int cout, void* stream);
/* The whole bottleneck block of the shipped resnet config (resnet_color_1x6_bn_32x128x32_1x3x1.json; backbone_resnet.py:149-178,
 * backbone_blocks.py:160-243) in one kernel, its two 1x1 convolutions on the f16 matrix cores with split-f16 operands (hi + lo, three
 * products, fp32 accumulation -- DESIGN.md 4.2):
 *   out = [x +] act2(act1(depthwise3x3_x4(act0(x . w0 + shift0)) + shift1) . w2 + shift2)        32 -> 32 -> 128 -> 32 channels
 * replaces bf_op_pointwise followed by bf_op_dwmult_pointwise.  w0 [32][32], wd [3][3][32][4], w2 [128][32] (grouped 1x1 as a block-
 * diagonal dense matrix), BatchNorm scales folded into wd / w2 and offsets passed as shift0 [32] / shift1 [128] / shift2 [32] (or NULL).
 * Activations: 0 linear, 1 relu, 2 leaky relu (0 <= alpha <= 1); BF_EUNSUPPORTED for others.  out != x. */
int64_t bf_op_bneck_h3_pack_bytes(void);
int bf_op_pack_bneck_h3(const float* w0, const float* wd, const float* w2, void* packed, void* stream);
int bf_op_bneck_block_h3(const float* x, float* out, const void* packed, const float* shift0, int act0, float alpha0,
                         const float* shift1, int act1, float alpha1, const float* shift2, int act2, float alpha2, int add_res,
                         int batch, int height, int width, void* stream);
/* MaxPooling2D(2, 2, padding="same") (downsampling.py:56-58). */
int bf_op_maxpool2(const float* in, float* out, int batch, int height, int width, int channels, void* stream);
/* The same 1x1 convolution with the epilogues of AdditiveAttentionGate (custom_layers.py:805-832):
 * mode 0 = bf_op_pointwise;  mode 1: out = act(in . w + res);  mode 2: out = res * sigmoid(4 * mult * (in . w)) + add;
 * mode 3: out = act(in . w + mult) + res (mult = per-channel bias: a convolution with a folded BatchNorm). */
int bf_op_pointwise_ex(const float* in, float* out, const float* wp, const float* mult, const float* res, const float* add,
                       int64_t npix, int cin, int cout, int act, float alpha, int mode, void* stream);
/* ConvNextBlock conv_2 -> activation -> conv_3 -> ChannelLearnableMultiplier -> Add(skip, .) (custom_layers.py:990-1008;
 * backbone_unet_laplacian.py:351-354): out = skip + mult * (act(in . w1) . w2), w1 [C][4C], w2 [4C][C] packed as above. */
int bf_op_convnext_mlp(const float* in, const float* skip, float* out, const float* w1p, const float* w2p, const float* mult,
                       int64_t npix, int channels, int act, float alpha, void* stream);
/* The same operator on the f16 matrix cores with split-f16 operands (hi + lo f16 pairs, three products, fp32
 * accumulation: ~22 mantissa bits; needs |activation| < 65504): w1 [C][4C], w2 [4C][C] fp32 are packed once into
 * bf_op_mlp_h3_pack_bytes(C) bytes; C = 32 or 64. */
int64_t bf_op_mlp_h3_pack_bytes(int channels);
int bf_op_pack_mlp_h3(const float* w1, const float* w2, void* packed, int channels, void* stream);
int bf_op_convnext_mlp_h3(const float* in, const float* skip, float* out, const void* packed, const float* mult,
                          int64_t npix, int channels, int act, float alpha, void* stream);
/* A whole ConvNextBlock whose depthwise convolution is 1x1 (decoder_kernel_size 1) plus the residual Add, one kernel:
 * out = x + mult * (act(LayerNormalization(x * dw) * ln_gamma . w1) . w2); dw [C], ln_gamma [C] or NULL. */
int bf_op_convnext_block1_h3(const float* x, float* out, const float* dw, const float* ln_gamma, float eps, const void* packed,
                             const float* mult, int64_t npix, int channels, int act, float alpha, void* stream);
/* The first decoder block of a level with the node in front of it formed while its pixels are loaded (32 channels, 1x1 depthwise):
 * x = enc + act_up(UpSampling2D(2, "bilinear")(low)); out = x + ConvNextBlock(x)  (backbone_unet_laplacian.py:438-568, upsampling.py:80-90:
 * replaces bf_op_upsample_act_add followed by bf_op_convnext_block1_h3, bit for bit).  enc, out [B, OH, OW, 32]; low [B, OH/2, OW/2, 32];
 * OH, OW even; act_up 0 linear / 1 relu / 2 leaky relu (alpha_up).  BF_EUNSUPPORTED for other channel counts or odd sizes. */
int bf_op_convnext_block1_up_h3(const float* enc, const float* low, float* out, const float* dw, const float* ln_gamma, float eps,
                                const void* packed, const float* mult, int batch, int out_height, int out_width, int channels, int act,
                                float alpha, int act_up, float alpha_up, void* stream);
/* A CHAIN of 1..3 pixel-wise ConvNext blocks (1x1 depthwise: decoder_kernel_size 1, 32 channels) in one kernel -- the `width` decoder
 * blocks of a level (backbone_unet_laplacian.py:538-560) without a round trip through memory between them:
 *   for b in 0 .. nblocks-1:  x = x + mult[b] * (act(LayerNormalization(x * dw[b]) * gamma[b] . w1[b]) . w2[b])
 * low != NULL: the first block's input is x + act_up(UpSampling2D(2, "bilinear")(low)) (x = the encoder's skip map [B, OH, OW, 32], low
 * [B, OH/2, OW/2, 32]; OH, OW even) -- bf_op_convnext_block1_up_h3 followed by nblocks - 1 times bf_op_convnext_block1_h3.  The arrays
 * are HOST arrays of nblocks device pointers; packed[b] from bf_op_pack_mlp_h3_chain (the operand of bf_op_pack_mlp_h3 with the first
 * kernel's rows in the order the chain kernel holds a pixel's channels; same size); gamma[b] / mult[b] may be NULL.  out may alias x. */
int bf_op_pack_mlp_h3_chain(const float* w1, const float* w2, void* packed, int channels, void* stream);
int bf_op_convnext_chain32_h3(const float* x, const float* low, float* out, int nblocks, const void* const* packed,
                              const float* const* dw, const float* const* ln_gamma, const float* const* mult, float eps, int batch,
                              int out_height, int out_width, int act, float alpha, int act_up, float alpha_up, void* stream);
/* A whole encoder ConvNextBlock (k x k depthwise, k = 3 or 5, 32 channels) plus the residual Add, one kernel
 * (custom_layers.py:975-1008; backbone_unet_laplacian.py:336-354):
 * out = x + mult * (act(LayerNormalization(DepthwiseConv2D_kxk(x)) * ln_gamma . w1) . w2); dw [k][k][C]; out != x. */
int bf_op_convnext_block_h3(const float* x, float* out, const float* dw, int k, const float* ln_gamma, float eps,
                            const void* packed, const float* mult, int batch, int height, int width, int channels, int act,
                            float alpha, void* stream);
/* A/B switch between kernel variants of one operator (process-wide; tests and tools only): key "enc32":
 * 2 = wave-specialised encoder block kernel with the producers' row walk unrolled (default), 1 = wave-specialised,
 * 0 = the single-role one, 3 = two workgroups per CU, 4 = 2 with two consumer waves per SIMD (k = 3 only; k = 5 runs 2). */
int bf_op_set_variant(const char* key, int value);
/* DepthwiseConv2D k x k (SAME, zero pad; w [k][k][C]; k = 0: none) -> LayerNormalization(center=False, epsilon) * gamma
 * (ln_gamma NULL: none) -> activation   (custom_layers.py:979-988; backbone_unet_laplacian.py:355-360). */
int bf_op_dwconv_ln(const float* in, float* out, const float* w, const float* ln_gamma, int batch, int height, int width,
                    int channels, int k, float eps, int act, float alpha, void* stream);
/* Laplacian split between levels (backbone_unet_laplacian.py:366-386): smooth = AveragePooling2D(k, strides 1, same) or,
 * with gauss [k][k], GaussianFilter; lap = in - smooth; down = smooth[:, ::2, ::2, :] (downsampling.py:61). */
int bf_op_smooth_split(const float* in, float* lap, float* down, const float* gauss, int batch, int height, int width,
                       int channels, int k, int down_stride /* 2: the slice above; 1: the whole smooth map */, void* stream);
/* The same split with the level's output normalisation in front, one kernel (backbone_unet_laplacian.py:355-386):
 * y = act(LayerNormalization(in) * ln_gamma) (ln_gamma NULL: y = act(in)) is never written; lap = y - smooth(y),
 * down = smooth(y)[:, ::2, ::2, :]; k = 3 or 5. */
int bf_op_norm_smooth_split(const float* in, const float* ln_gamma, float eps, int act, float alpha, const float* gauss,
                            float* lap, float* down, int batch, int height, int width, int channels, int k, void* stream);
/* out = other + act(UpSampling2D(2, "bilinear")(in)) (upsampling.py:80-102 + the decoder Add). */
int bf_op_upsample_act_add(const float* in, const float* other, float* out, int batch, int height, int width, int channels,
                           int act, float alpha, void* stream);
/* tf.image.resize(BILINEAR, antialias=False), half-pixel centres (custom_layers.py:1328-1334, 1351-1357). */
int bf_op_resize_bilinear(const float* in, float* out, int batch, int height, int width, int channels, int out_height,
                          int out_width, void* stream);
/* keras.layers.Attention(use_scale=False, score_mode="dot") on [query, value, key] (custom_layers.py:1345): [B][T][A], A = 32,
 * out = softmax(q k^T) v per sequence.  batch = number of sequences (images, or image rows for rank-4 inputs: keras then
 * attends along the last-but-one axis only); any length (16 queries per wave, keys walked 16 at a time). */
int bf_op_attention(const float* q, const float* v, const float* k, float* out, int batch, int tokens, int channels, void* stream);
/* The same with q / v / k rows `ld` floats apart (ld >= channels, a multiple of 4): the three projections written side by
 * side by ONE 1x1 convolution with 3 * channels outputs (q = base, the others at base + channels, base + 2 * channels). */
int bf_op_attention_ld(const float* q, const float* v, const float* k, float* out, int batch, int tokens, int channels, int ld,
                       void* stream);
/* first Conv2D k x k cin(<=4) -> cout on the (optionally) normalised image; the [Hs,Ws] source (u8 or f32) is zero-padded
 * to [H,W] before normalisation as pad_to_power_of_2 does (utilities.py:736-751; model.py:100-102). */
int bf_op_first_conv(const void* in, int in_is_u8, float* out, const float* w, int batch, int src_height, int src_width,
                     int height, int width, int cin, int cout, int k, int normalize, float v_min, float v_max, int act,
                     float alpha, void* stream);
/* The same for the one shape the unet_laplacian builder emits (5 x 5, 3 -> 32) on the f16 matrix cores with split-f16
 * operands (three products, fp32 accumulation: the arithmetic of bf_op_convnext_mlp_h3). */
int bf_op_first_conv_h3(const void* in, int in_is_u8, float* out, const float* w, int batch, int src_height, int src_width,
                        int height, int width, int normalize, float v_min, float v_max, int act, float alpha, void* stream);
/* The same for kernel sizes 3, 5 and 7 (the base convolution of the resnet configs: kernel_size 7). */
int bf_op_first_conv_h3k(const void* in, int in_is_u8, float* out, const float* w, int batch, int src_height, int src_width,
                         int height, int width, int k, int normalize, float v_min, float v_max, int act, float alpha, void* stream);
/* last Conv2D 1x1 of a denoiser head + tanh(2x)*0.51 [+ denormalise][+ round, uint8], cropped to [Ho,Wo]
 * (model.py:321-342, 136-139; module_denoiser.py:62-73). */
int bf_op_head_out(const float* in, const float* w, void* out, int out_is_u8, int batch, int height, int width, int out_height,
                   int out_width, int head_filters, int cout, int denormalize, float v_min, float v_max, int* status, void* stream);
/* A whole denoiser head in one kernel: [LayerNormalization(in) * ln_gamma (the backbone's output normalisation,
 * backbone_unet_laplacian.py:547-551; NULL: none)] -> Conv2D 1x1 cin -> 32 (w0p packed by bf_op_pack_pointwise) ->
 * activation -> Conv2D 1x1 32 -> cout (w1 [32][cout]) -> tanh(2x)*0.51 [-> denormalise][-> round, uint8], cropped. */
int bf_op_head_fused(const float* in, const float* ln_gamma, float eps, const float* w0p, int act, float alpha, const float* w1,
                     void* out, int out_is_u8, int batch, int height, int width, int out_height, int out_width, int cin,
                     int head_filters, int cout, int denormalize, float v_min, float v_max, int* status, void* stream);
/* The same head with its first 1x1 on the f16 matrix cores (split-f16 operands, three products, fp32 accumulation) for cin = 32 / 64;
 * same arguments, same operand from bf_op_pack_pointwise; BF_EUNSUPPORTED for other channel counts. */
int bf_op_head_fused_h3(const float* in, const float* ln_gamma, float eps, const float* w0p, int act, float alpha, const float* w1,
                        void* out, int out_is_u8, int batch, int height, int width, int out_height, int out_width, int cin, int hf,
                        int cout, int denormalize, float v_min, float v_max, int* status, void* stream);
/* status (both heads above; may be NULL): int32 on the device, |= BF_STATUS_F16_RANGE when the value in front of the tanh is
 * not finite -- an activation left the f16 range inside a split-f16 operator upstream; clear it with bf_op_fill32. */
int bf_op_fill32(void* p, int value, int64_t n, void* stream);
/* Conv2DTranspose k x k, stride s, padding "same", no bias (upsample_type "conv2d_transpose": bfcnn/upsampling.py:37-48,
 * utilities.py:200-202): in [B,H,W,cin] -> out [B,H*s,W*s,cout]; w [k,k,cout,cin] (keras kernel layout); act as
 * bf_op_pointwise. */
int bf_op_conv2d_transpose(const float* in, const float* w, float* out, int batch, int height, int width, int cin, int cout,
                           int k, int stride, int act, float alpha, void* stream);
/* mult[c] = tanh(relu(1 + w[c])) (ChannelLearnableMultiplier, custom_layers.py:304-306). */
int bf_op_channel_multiplier(const float* w, float* mult, int n, void* stream);

/* bf_adam_step_ex for a flat parameter vector no handle describes (models assembled from bf_op_*: unet_laplacian) */
int bf_op_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t iterations, float lr, float beta_1,
                    float beta_2, float epsilon, float global_clipnorm, float clipnorm, float clipvalue,
                    const int64_t* tensor_offsets, int n_tensors, float* tensor_scratch, float grad_scale, float* losses,
                    float* scratch, void* stream);

/* ---- backward primitives of the operator library (train_prims.hip) --------------------------------------------------------
 * What `unet_laplacian` training needs beyond the forward operators (bfcnn/train_loop.py:259-312 with the multi-output hydra;
 * tf.GradientTape does this in the reference): exact fp32, NHWC, reductions through caller-supplied scratch (fixed order).
 * act codes as bf_op_pointwise (0 linear, 1 relu, 2 leaky relu(alpha), 3 exact-erf gelu). */
/* dx = dy * act'(.) ; ref = the activation's input, or (ref_is_output, relu / leaky relu only) its output */
int bf_op_act_bwd(const float* ref, const float* dy, float* dx, int64_t n, int act, float alpha, int ref_is_output, void* stream);
/* 1x1 convolution weight gradient dW[cin][cout] = sum_p x[p][cin] dy[p][cout]; scratch >= cin*cout floats (more = more splits) */
int bf_op_matmul_wgrad(const float* x, const float* dy, float* dw, int64_t npix, int cin, int cout, float* scratch,
                       int64_t scratch_floats, void* stream);
/* DepthwiseConv2D k x k (same) weight gradient dw[k][k][C]; scratch >= k*k*C floats */
int bf_op_dwconv_wgrad(const float* x, const float* dy, float* dw, int batch, int height, int width, int channels, int k,
                       float* scratch, int64_t scratch_floats, void* stream);
/* LayerNormalization(center=False) backward: dx and dgamma from the layer's input x; scratch >= C floats */
int bf_op_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, int64_t npix, int channels,
                        float eps, float* scratch, int64_t scratch_floats, void* stream);
/* out = res + t * m[c] * s[b] (ChannelLearnableMultiplier value m, StochasticDepth sample scale s, skip Add; each optional) */
int bf_op_scale_add(const float* res, const float* t, const float* m, const float* sample_scale, float* out, int batch, int64_t hw,
                    int channels, void* stream);
int bf_op_scale_add_bwd(const float* t, const float* m, const float* sample_scale, const float* dy, float* dt, float* dm, int batch,
                        int64_t hw, int channels, float* scratch, int64_t scratch_floats, void* stream);
/* ChannelLearnableMultiplier m = tanh(relu(1 + w)) (custom_layers.py:304-306): dw from dm */
int bf_op_multiplier_bwd(const float* w, const float* dm, float* dw, int n, void* stream);
/* adjoint of bf_op_smooth_split (gauss NULL: AveragePooling2D with the in-bounds divisor): dx from (dlap, ddown) */
int bf_op_smooth_split_bwd(const float* dlap, const float* ddown, const float* gauss, float* dx, int batch, int height, int width,
                           int channels, int k, void* stream);
/* the same with ddown the gradient of the full-resolution smooth map (down_stride 1: conv2d / maxpool down-sampling) or of
   smooth[:, ::2, ::2] (2); averaging or Gaussian windows of any size k <= 7 (TF same padding: the extra tap after for even k) */
int bf_op_smooth_split_bwd_ex(const float* dlap, const float* ddown, const float* gauss, float* dx, int batch, int height, int width,
                              int channels, int k, int down_stride, void* stream);
/* adjoint of MaxPooling2D(2, 2, same) (bfcnn/downsampling.py:56-68): dx [B,H,W,C] = dy [B,ceil(H/2),ceil(W/2),C] at each window's
   first maximum, 0 elsewhere */
int bf_op_maxpool2_bwd(const float* x, const float* dy, float* dx, int batch, int height, int width, int channels, void* stream);
/* adjoint of UpSampling2D(2, bilinear | nearest): dx [B,H,W,C] from dy [B,2H,2W,C] */
int bf_op_upsample2x_bwd(const float* dy, float* dx, int batch, int height, int width, int channels, int bilinear, void* stream);
/* k x k convolution (same, stride 1) weight gradient for the first convolution; x = the raw image, normalised as the forward does */
int bf_op_conv2d_wgrad(const void* x, int x_is_u8, const float* dy, float* dw, int batch, int height, int width, int cin, int cout,
                       int k, int normalize, float v_min, float v_max, float* scratch, int64_t scratch_floats, void* stream);
/* last stage of a denoiser head backward (model.py:321-342): h [npix,hf] activated hidden layer, w1 [hf,cout], dL/dpred ->
 * dh, dw1 (tanh(2x)*0.51, clip, denormalise differentiated) */
int bf_op_head_out_bwd(const float* h, const float* w1, const float* dpred, float* dh, float* dw1, int64_t npix, int head_filters,
                       int cout, int denormalize, float v_min, float v_max, float* scratch, int64_t scratch_floats, void* stream);
/* denoiser_loss of ONE output scale (loss.py:190-247) and d(total * depth_weight)/dpred; losses[BF_LOSS_COUNT] */
int64_t bf_op_denoiser_loss_scratch_floats(int batch, int height, int width, int channels);
int bf_op_denoiser_loss(const float* pred, const float* gt, int batch, int height, int width, int channels, const bf_loss_desc* loss,
                        float* dpred, float* losses, float* scratch, int64_t scratch_floats, void* stream);
/* dot-product attention for training: out = (softmax(q k^T) * pscale) v with P = softmax kept ([B,T,T]; pscale = dropout
 * keep-mask / keep-probability or NULL), and its backward (dS_scratch [B,T,T]) */
int bf_op_attention_train(const float* q, const float* v, const float* k, const float* pscale, float* out, float* P, int batch,
                          int tokens, int channels, void* stream);
int bf_op_attention_bwd(const float* q, const float* v, const float* k, const float* pscale, const float* P, const float* dout,
                        float* dq, float* dv, float* dk, float* dS_scratch, int batch, int tokens, int channels, void* stream);
/* adjoint of bf_op_resize_bilinear; scratch: B*H*out_width*C floats */
int bf_op_resize_bilinear_bwd(const float* dy, float* dx, int batch, int height, int width, int channels, int out_height,
                              int out_width, float* scratch, void* stream);
/* regularisers: value[0] += term, grad += grad_scale * d(term)/dw (bf_op_reg_elementwise: grad may be NULL, the value alone).
 * kind BF_REG_L1 / BF_REG_L2 with coefficient coef;
 * SoftOrthonormalConstraintRegularizer (regularizers.py:283-338) on a 1x1 kernel [cin][cout], scratch 2*cout*cout floats */
int bf_op_reg_elementwise(const float* w, float* grad, int64_t n, int kind, float coef, float grad_scale, float* value, void* stream);
int bf_op_reg_soft_orthonormal(const float* w, float* grad, int cin, int cout, float lambda, float l1, float l2, float grad_scale,
                               float* value, float* scratch, void* stream);
/* the same with grad NULL allowed (the value alone) and mask_diagonal 1 = SoftOrthogonalConstraintRegularizer (regularizers.py:208-280:
   the terms on W^T W with its diagonal zeroed); w is any [rows][cout] matrix: an HWIO kernel [k,k,cin,cout] with rows = k k cin */
int bf_op_reg_soft_orthogonal_ex(const float* w, float* grad, int cin, int cout, float lambda, float l1, float l2, float grad_scale,
                                 float* value, float* scratch, int mask_diagonal, void* stream);
/* weight re-layouts for the data gradients: spatial flip of [k][k][inner]; transpose of [a][b] */
/* ---- training-mode operators of the resnet builder outside the 16-filter 3x3 engine (csrc/train_generic.hip) ----
   BatchNormalization(center=False) with batch statistics, bfcnn/utilities.py:204-206 under training=True: y = act(gamma (x - mean)
   / sqrt(var + eps)); save [2C] = mean | 1 / sqrt(var + eps) for the backward; moving statistics (may be NULL) updated with
   `momentum` and the Bessel-corrected variance (fused-kernel semantics).  channels must divide 256.  Backward: dx, dgamma from dy
   (the gradient at the BatchNorm's output, in front of the activation). */
int64_t bf_op_bn_train_scratch_floats(int channels);
int bf_op_bn_train_fwd(const float* x, const float* gamma, float* y, float* save, float* moving_mean, float* moving_var, int64_t npix,
                       int channels, float eps, float momentum, int act, float alpha, float* scratch, int64_t scratch_floats,
                       void* stream);
int bf_op_bn_train_bwd(const float* x, const float* gamma, const float* save, const float* dy, float* dx, float* dgamma, int64_t npix,
                       int channels, float* scratch, int64_t scratch_floats, void* stream);
/* channel gate of add_gates (bfcnn/backbone_blocks.py:199-208): g = hard_sigmoid(relu(mean_hw(x) W0) W1), out = x * g [+ res];
   w0 [C][C8], w1 [C8][C] (keras Dense kernels, no bias), C8 <= 32; save: bf_op_gate_save_floats floats kept for the backward.
   Backward: dy = gradient at x * g; dx = gradient at x through the multiply AND the mean; dw0 / dw1 overwritten. */
int64_t bf_op_gate_save_floats(int batch, int channels, int squeeze);
int64_t bf_op_gate_scratch_floats(int batch, int channels);
int bf_op_gate_fwd(const float* x, const float* w0, const float* w1, const float* res, float* out, float* save, int batch, int64_t hw,
                   int channels, int squeeze, float* scratch, int64_t scratch_floats, void* stream);
int bf_op_gate_bwd(const float* x, const float* w0, const float* w1, const float* save, const float* dy, float* dx, float* dw0,
                   float* dw1, int batch, int64_t hw, int channels, int squeeze, float* scratch, int64_t scratch_floats, void* stream);
/* the gate's relatives on the same kernels.  m = mean_hw(sel) [B, sel_channels]; h = act0(m W0 + b0) (act0 1 relu, 2 leaky relu
   with alpha0; biases may be NULL); p = h W1 + b1; g = F(p) with mode 0 hard_sigmoid(p), 1 sigmoid(p), 2 hard_sigmoid(2.5 - relu(p)),
   3 sigmoid(2.5 - relu(p)); out = x g [+ x2 (1 - g)] [+ res].
     squeeze_and_excite_block (bfcnn/backbone_blocks.py:251-313): sel = x, act0 2 / alpha0 0.1, mode 1 (default), 0 (hard_sigmoid_version),
       2 (+ learn_to_turn_off);  selector_block, scale_type GLOBAL (custom_layers_selector.py:268-283, 316-330): sel = the selector
       layer, x = input_1, x2 = input_2, mode 2 (HARD) / 3 (SOFT).
   bf_op_selector_mix: the per-pixel form of the same mix for scale_type LOCAL (u >= 0 the up-sampled selector map):
     s = F(2.5 - u), out = x1 s + x2 (1 - s).  bf_op_avgpool_same: AveragePooling2D(pool, strides, padding "same"), any sizes,
     divisor = taps inside the image (custom_layers_selector.py:196-201). */
int64_t bf_op_channel_gate_save_floats(int batch, int sel_channels, int channels, int squeeze);
int bf_op_channel_gate_ex(const float* sel, const float* x, const float* x2, const float* res, float* out, const float* w0,
                          const float* b0, const float* w1, const float* b1, float* save, int batch, int64_t hw, int sel_channels,
                          int channels, int squeeze, int act0, float alpha0, int mode, float* scratch, int64_t scratch_floats,
                          void* stream);
int bf_op_dense2(const float* in, const float* w0, const float* b0, const float* w1, const float* b1, float* out, int64_t n,
                 int in_channels, int channels, int squeeze, int act0, float alpha0, int mode, void* stream);   /* rows of [n][in_channels]; mode 4 = relu */
int bf_op_selector_mix(const float* x1, const float* x2, const float* u, float* out, int64_t n, int soft, void* stream);
int bf_op_avgpool_same(const float* in, float* out, int batch, int height, int width, int channels, int pool_h, int pool_w,
                       int stride_h, int stride_w, void* stream);
/* add_concat_input (bfcnn/backbone_resnet.py:277-279): out [B,H,W,out_channels] = feat [channels] | normalise(x) [in_channels] | 0 ...,
   x the raw image [B,Hs,Ws,in_channels] (u8 or f32), zero-padded to [H,W] before normalisation as the first convolution sees it */
int bf_op_concat_input(const float* feat, const void* x, int x_is_u8, float* out, int batch, int height, int width, int src_height,
                       int src_width, int channels, int in_channels, int out_channels, float v_min, float v_max, void* stream);
/* selector_block's optional pre-filters (bfcnn/custom_layers_selector.py:160-185; utilities.py:566-620).  local_normalization =
   bf_op_avgpool_same (strides 1) + bf_op_center_scale (var NULL: out = (x - mean)^2; else out = (x - mean) / sqrt(var + eps));
   global_normalization = bf_op_bn_train_fwd per sample with gamma 1; lowpass / highpass: out = x (1 - tanh(a x)^b) / x tanh(a x)^b */
int bf_op_center_scale(const float* x, const float* mean, const float* var, float* out, int64_t n, float eps, void* stream);
int bf_op_pass_filter(const float* x, float* out, int64_t n, float a, int b, int highpass, void* stream);
/* their adjoints: dx of the pass filters; for y = (x - m) / sqrt(v + eps): dd = dy / sqrt(v + eps) and dv = d y / d v (bf_op_center_scale_bwd),
   then out = dd + 2 (x - m) t with t = the pooling's adjoint of dv (bf_op_center_sq_bwd); dx = out - pooling adjoint of out */
int bf_op_pass_filter_bwd(const float* x, const float* dy, float* dx, int64_t n, float a, int b, int highpass, void* stream);
int bf_op_center_scale_bwd(const float* x, const float* mean, const float* var, const float* dy, float* dd, float* dv, int64_t n, float eps,
                           void* stream);
int bf_op_center_sq_bwd(const float* x, const float* mean, const float* t, const float* dd, float* out, int64_t n, void* stream);
/* selector_block in training: adjoints of bf_op_selector_mix (dx1, dx2, du from dy), bf_op_avgpool_same (dx [B,H,W,C] from the pooled
   map's gradient; accumulate != 0: added to dx), bf_op_dense2 in its selector form (no biases, act0 = leaky ReLU alpha0, final ReLU:
   din, dw0 [in_channels][squeeze], dw1 [squeeze][channels]; scratch: bf_op_dense2_bwd_scratch_floats), and the channel slice
   dst[r][0:channels] = src[r][offset:offset+channels] that undoes bf_op_concat_channels */
int bf_op_selector_mix_bwd(const float* x1, const float* x2, const float* u, const float* dy, float* dx1, float* dx2, float* du, int64_t n,
                           int soft, void* stream);
int bf_op_avgpool_same_bwd(const float* dpooled, float* dx, int batch, int height, int width, int channels, int pool_h, int pool_w,
                           int stride_h, int stride_w, int accumulate, void* stream);
int64_t bf_op_dense2_bwd_scratch_floats(int64_t n, int channels, int squeeze);
int bf_op_dense2_bwd(const float* in, const float* w0, const float* w1, const float* dout, float* din, float* dw0, float* dw1, int64_t n,
                     int in_channels, int channels, int squeeze, float alpha0, float* scratch, int64_t scratch_floats, void* stream);
int bf_op_slice_channels(const float* src, float* dst, int64_t rows, int src_channels, int offset, int channels, void* stream);
/* selector_block's MIXED / MULTISCALE inputs (custom_layers_selector.py:203-262): keras Concatenate on the channel axis of up to
   three [rows][C*] tensors; the per-sample channel mean of x [B][hw][C] broadcast to out [B][rows_out][C]
   (tf.reduce_mean(x, axis=[1, 2], keepdims=True) added to a zeroed pooled map); scratch: bf_op_gate_scratch_floats(B, C) */
int bf_op_concat_channels(const float* a, const float* b, const float* c, float* out, int64_t rows, int ca, int cb, int cc, void* stream);
int bf_op_channel_mean_broadcast(const float* x, float* out, int batch, int64_t hw, int channels, int64_t rows_out, float* scratch,
                                 int64_t scratch_floats, void* stream);
/* the last stage of AdditiveAttentionGate (bfcnn/custom_layers.py:826-832) with the Add behind it, for training:
   out = enc * sigmoid(4 o) [+ up]; backward: denc = dy * s, do = dy * enc * 4 s (1 - s) */
int bf_op_sigmoid_gate(const float* enc, const float* o, const float* up, float* out, int64_t n, void* stream);
int bf_op_sigmoid_gate_bwd(const float* enc, const float* o, const float* dy, float* denc, float* dout_o, int64_t n, void* stream);
/* ChannelwiseMultiplier / Multiplier of the resnet builder (bfcnn/custom_layers.py:1028-1160; backbone_blocks.py:215-221): the factor
   m[c] = relu(w0[c or 0] + w1) for bf_op_scale_add (nw = C: per channel, nw = 1: one scalar), and d w0 from d m */
int bf_op_relu_shift(const float* w0, int nw, float w1, float* m, int channels, void* stream);
int bf_op_relu_shift_bwd(const float* w0, int nw, float w1, const float* dm, float* dw0, int channels, void* stream);
int bf_op_linear_shift(const float* w0, int nw, float w1, float* m, int channels, void* stream);   /* activation "linear": m = w0 + w1 */
/* the normalise / denormalise layers on their own (bfcnn/model.py:364-430; inside the hydras they are fused into the first
   convolution and the head): inverse 0: (clip(x, v_min, v_max) - v_min) / (v_max - v_min) - 0.5; 1: (clip(x, -.5, .5) + .5) * range + v_min */
int bf_op_normalize(const float* x, float* out, int64_t n, float v_min, float v_max, int inverse, void* stream);
/* layout helpers: out[r][c * m + j] = x[r][c] (a DepthwiseConv2D with depth_multiplier m is a plain depthwise convolution of the
   repeated tensor) and its adjoint out[r][c] = sum_j x[r][c * m + j]; keras Conv2D(groups) kernel [cin / groups][cout] to / from the
   block-diagonal dense [cin][cout] (extract = 1 writes w from dense) */
int bf_op_channel_repeat(const float* x, float* out, int64_t npix, int channels, int m, void* stream);
int bf_op_channel_group_sum(const float* x, float* out, int64_t npix, int channels, int m, void* stream);
int bf_op_group_kernel(float* w, float* dense, int cin, int cout, int groups, int extract, void* stream);
int bf_op_flip_hw(const float* w, float* out, int k, int inner, void* stream);
int bf_op_transpose2d(const float* w, float* out, int a, int b, void* stream);

/* ---- data-parallel exchange (the reference is single-device, bfcnn/train_loop.py:259-321) -----------------
 * ONE sum-all-reduce of the flat fp32 gradient buffer over RCCL per training step; every rank then runs the identical
 * bf_adam_step with grad_scale = 1 / world.  RCCL is bound at run time: BF_EUNSUPPORTED when librccl is not installed.
 *   rank 0: bf_comm_unique_id(id) -> id to every rank over any host channel -> each rank, with ITS GPU current:
 *   bf_comm_init_rank(&comm, world, rank, id) -> per step bf_allreduce_grads(h, grads, n, comm, stream) -> bf_comm_destroy.
 * `comm` is an ncclComm_t; one the host already owns (e.g. from its own ncclCommInitRank) is accepted as well. */
int bf_comm_unique_id(char id_out[128]);
int bf_comm_init_rank(void** comm_out, int world, int rank, const char id[128]);
int bf_comm_destroy(void* comm);
int bf_allreduce_grads(bf_handle h, float* grads, int64_t n, void* comm, void* stream);
const char* bf_comm_last_error(void);

/* y += a * x over n floats (overwrite != 0: y = x): gradient accumulation over micro-batches (bfcnn/train_loop.py:296-310). */
int bf_op_axpy(float* y, const float* x, float a, int overwrite, int64_t n, void* stream);

/* ---- options and diagnostics (not part of the drop-in surface; used by tests/) ------------- */

/* Inference forwards keep a status word in the LAST 2048 bytes of the workspace they are given (ws + ws_bytes - 2048,
 * int32; the rest of that tail is kernel scratch): 0 after a clean forward; bit BF_STATUS_F16_RANGE is set when an activation left the f16 range inside the
 * split-f16 blocks (the output is then not trustworthy: re-run with option "arith" = 0).  Reading it needs a stream
 * synchronisation, which is the caller's decision (the Python host checks it whenever it hands back host arrays). */
#define BF_STATUS_F16_RANGE 1

/* "fused_blocks" = 1 (default): one kernel per residual block; 0: one kernel per convolution.
 * "arith" = 1 (default): fused inference blocks run split-f16 ("f16x3": x = hi + lo in f16, three products,
 *   fp32 accumulation) on the f16 matrix cores, ~fp32 accuracy, needs |activation| < 65504;
 *   0: exact fp32 on the f32 matrix cores.  Training and the unfused path are always exact fp32.
 * "train_arith" = 1 (default): the training convolutions (forward, data gradient, weight gradient) run split-f16 on the
 *   f16 matrix cores; 0: exact fp32.  With it (all default 1, A/B only): "train_fused_bwd" = weight gradient, data gradient
 *   and the BatchNorm-backward apply of a convolution in one kernel; "train_fused_fwd" = a block's BatchNorm apply + skip Add
 *   formed by the next block's first convolution while it stages its tile; "train_zigzag" = consecutive tile kernels walk the
 *   tensors in opposite directions (Infinity Cache reuse).  "train_fused_bwd2" (default 0): [3,3] blocks with BatchNorm and
 *   ReLU run BOTH convolutions' backward in one kernel (6 tensor passes for 9; measured slower than the two kernels).
 * "fused_head" = 1: with split-f16 blocks, a linear denoiser head and 3 output channels, the head (premultiplied 16 x 3
 *   matrix, tanh, denormalise, rounding) runs in the epilogue of the last block: no head kernel, the last block output is
 *   never written; 0 (default): separate head kernel (the two measure within 0.5 % of each other).
 * "h3_pair": 1 (default): wherever the full-row streaming kernel applies, consecutive residual blocks run TWO per launch
 *   (fused_block2_h3w_kernel: the activation between them stays in LDS; an odd block count runs its single block first); 0: one
 *   block per launch.  "h3_pair_head": 1: the last pair launch also runs a linear 3-channel head in its store step (no head
 *   kernel); 0 (default; the two measure the same).  "h3_pair" = 2 and "base_rows" = 2 (A/B and tests only) run the two-block kernel /
 *   the row-streaming base convolution wherever they CAN run instead of where the selection prefers them ("base_rows" = 0: the
 *   tile kernel of the base convolution everywhere; process-wide).
 * "fused_tile" / "h3_variant": kernel variants of the fused blocks (A/B only; negative = default). */
int bf_set_option(bf_handle h, const char* key, int value);

/* with option "timing" = 1 every forward brackets its residual-block launches with two HIP events on
 * the caller's stream (a ring of 256 pairs; setting the option again restarts the window); after the
 * caller has synchronised, this returns the SUM of the elapsed milliseconds of the brackets of all
 * forwards since the option was set (the last 256 at most) and the number of kernel launches inside
 * them (bench.py roofline: average launch duration over the timed region). */
int bf_get_timing(bf_handle h, float* ms, int* launches);
/* name of the kernel that ran most of the residual-block launches of the handle's LAST forward ("" before the first one) and
 * the number of block launches that forward made: what a profile of the run must show, and the key bench.py looks counter
 * traffic up by.  The string is static storage. */
const char* bf_get_block_kernel(bf_handle h, int* launches_per_forward);
/* the kernels that ran the residual blocks of the handle's LAST bf_train_step, as "fwd: <kernels>; bwd: <kernels>" ("" before the
 * first step): what a profile of a training run must show (bench.py's train records carry it).  Valid until the next step. */
const char* bf_get_train_kernels(bf_handle h);

/* The single-kernel diagnostic entries (bf_debug_*: one kernel at a time on fp32 NHWC tensors, used by the parity tests, the
 * profiling tools and bench.py's live roofline of the training kernel) are declared in bfcnn_hip_debug.h; they are exported by
 * the same library and are not part of the drop-in ABI above. */

#ifdef __cplusplus
}
#endif
#endif /* BFCNN_HIP_H */
