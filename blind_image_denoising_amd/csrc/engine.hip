// C-ABI engine: validates the model description, lays out parameters exactly as keras creates
// the trainable variables, and turns DenoiserModule.__call__ / hydra() / train_step_single_gpu /
// apply_grads into stream-ordered launch sequences of the gfx950 kernels.
// The library owns no device memory and never synchronises (include/bfcnn_hip.h).
#include "bf_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

constexpr int BF_TIMING_RING = 256;

struct bf_engine {
    bf_resnet_desc d;
    std::string err;
    std::vector<bf_tensor_info> tensors, states;
    int64_t n_params = 0, n_state = 0;
    // parameter offsets (floats)
    int64_t p_base = 0, p_blocks = 0, p_block_stride = 0, p_head0 = 0, p_head1 = 0;
    int64_t n_base = 0;
    // packed-inference layout (floats)
    int64_t k_base = 0, k_blocks = 0, k_block_stride = 0, k_w0 = 0, k_w1 = 0, k_wh = 0, k_zero = 0, k_h3 = 0, k_total = 0;
    int fused_blocks = 1;
    int fused_head = 0;      // 1: split-f16 path, linear head, 3 output channels: head folded into the last block's epilogue
                             // (measured 5.49 vs 5.51 ms per batch of 128: the longer epilogue of the last block costs what the
                             // head kernel saves, so it stays an option)
    int h3_zigzag = 1;              // alternate the band order of consecutive split-f16 blocks (Infinity Cache reuse)
    int h3_variant = -1;            // split-f16 block kernel: < 0 = library default (bf_set_h3_variant), else that variant
    int h3_compact = 0;             // 1: full-row streaming kernel keeps the activations between the launches in the compact layout
    int h3_pair = 1;                // 1: where the streaming kernel applies, consecutive blocks run two per launch (fused_h3w.hip)
    int h3_pair_head = 0;           // 1: the last pair launch also runs a linear 3-channel head (no head kernel, the last activation is
                                    // never written).  Off by default: measured equal (4.437 vs 4.435 ms per batch of 128): the ~250
                                    // vector instructions per 64 pixels cost the issue-bound launch what the head kernel's pass costs
    int block_launches = 0;         // launches of the last forward's residual blocks (bf_get_timing)
    const char* block_kernel = "";  // name of the kernel that ran most of them
    std::string train_kernels;      // the block kernels of the last bf_train_step (bf_get_train_kernels)
                                    // (fp8 lo planes, 48 B per pixel; bf_common.h): +5 % images/s for 6e-6 instead of 2e-7 normalised MAE
    // arithmetic of the fused inference blocks: 1 = split-f16 on the f16 matrix cores (fused_h3.hip, needs
    // |activation| < 65504), 0 = exact fp32 on the f32 matrix cores (conv3x3_c16.hip)
    int arith = 1;
    // arithmetic of the training convolutions (forward + data gradient): 1 = split-f16 on the f16 matrix cores (default),
    // 0 = exact fp32 on the f32 matrix cores; the weight gradients follow the same switch
    int train_arith = 1;
    int train_zigzag = 1;           // split-f16 training: consecutive kernels walk their tiles in opposite directions
    int train_fused_fwd = 1;        // split-f16 training: BatchNorm apply + skip Add of block i formed while block i+1's first convolution stages its tile
    int train_bwd_dbuf = 0;         // fused backward kernel: 512-thread form with double-buffered LDS images (A/B: 10 % slower)
    int train_fwd_block = 1;        // [3,3] blocks with BatchNorm + ReLU, W <= 256: the whole training forward of a block in ONE row-streaming
                                    // kernel (train_fwd_h3t.hip; T kept in LDS unless the backward pass reads it).  1 = where a forward holds
                                    // enough rows (bf_train_step), 2 = wherever it can run (tests), 0 = the two convolution kernels
    int train_bwd_block = 1;        // [3,3] blocks with BatchNorm + ReLU: the whole backward of a block in ONE row-streaming kernel that recomputes
                                    // T from the block input (train_bwd_h3t.hip: 5 tensor passes for 9, and the forward pass need not write T).
                                    // 1 = where a step holds enough strip rows, 2 = wherever it can run (tests), 0 = one kernel per convolution
    int train_fold_finalize = 1;    // block kernels: the BatchNorm finalisation kernels between the blocks (bn_finalize / bn_bwd_finalize, ~6 us +
                                    // two kernel boundaries each, 34 per step of 1x18) run in the prologue of the next block kernel instead
    int train_fused_bwd = 1;        // split-f16 training: weight + data gradient (+ BatchNorm backward) of a convolution in one kernel
    int train_fused_bwd2 = 0;       // [3,3] blocks with BatchNorm and ReLU: BOTH convolutions' backward in one kernel (bwd2_h3_kernel: 6 tensor
                                    // passes for 9, but 348 us against 131 + 110: one workgroup per CU and a recomputed halo -- DESIGN 4.3)
    // optional HIP-event bracket around the residual-block launches of a forward (bench.py roofline)
    int timing = 0;
    // ring of event pairs: one pair per timed forward since the option was (re)set, BF_TIMING_RING forwards at most
    std::vector<hipEvent_t> ev;
    int64_t n_timed = 0;
    int timed_launches = 0;
};

static thread_local std::string g_create_error;

static int fail(bf_handle h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

static int hip_fail(bf_handle h, hipError_t e, const char* what)
{
    return fail(h, BF_EHIP, "%s: %s", what, hipGetErrorString(e));
}

#define BF_HIP(call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(h, e__, what); } while (0)

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

extern "C" int bf_abi_version(void) { return BFCNN_ABI_VERSION; }

extern "C" const char* bf_last_error(bf_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void add_tensor(std::vector<bf_tensor_info>& v, const char* name, int64_t& off, int rank, int s0, int s1, int s2,
                       int s3, int kind, int reg)
{
    bf_tensor_info t;
    memset(&t, 0, sizeof(t));
    snprintf(t.name, sizeof(t.name), "%s", name);
    t.offset = off;
    t.rank = rank;
    t.shape[0] = s0; t.shape[1] = s1; t.shape[2] = s2; t.shape[3] = s3;
    t.kind = kind;
    t.regularizer = reg;
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) n *= t.shape[i];
    off += n;
    v.push_back(t);
}

// model_builder (bfcnn/model.py:58-162) -> backbone_resnet.builder (backbone_resnet.py:19-298)
// argument checks, restricted to the configurations the gfx950 kernels are built for.
extern "C" int bf_create(const bf_resnet_desc* d, bf_handle* out)
{
    if (out) *out = nullptr;
    if (!d || !out) return fail(nullptr, BF_EINVAL, "bf_create: desc/out must not be NULL");
    if (d->struct_size != (int32_t)sizeof(bf_resnet_desc))
        return fail(nullptr, BF_EINVAL, "bf_create: struct_size %d != %d (ABI mismatch)", d->struct_size, (int)sizeof(bf_resnet_desc));
    if (d->no_layers < 0) return fail(nullptr, BF_EINVAL, "no_layers must be >= 0");              // backbone_blocks.py:122
    if (d->block_convs <= 0) return fail(nullptr, BF_EINVAL, "len(block_kernels) must be >= 0 "); // backbone_resnet.py:111
    if (d->block_convs > 3) return fail(nullptr, BF_EINVAL, "len(block_kernels) must be <= 3");   // backbone_resnet.py:113
    if (d->in_channels != 1 && d->in_channels != 3)
        return fail(nullptr, BF_EUNSUPPORTED, "in_channels %d: only 1 and 3 are built", d->in_channels);
    if (d->filters != BF_C) return fail(nullptr, BF_EUNSUPPORTED, "filters %d: the MFMA path is built for 16", d->filters);
    if (d->kernel_size != 1 && d->kernel_size != 3 && d->kernel_size != 5 && d->kernel_size != 7)
        return fail(nullptr, BF_EUNSUPPORTED, "base kernel_size %d: only 1,3,5,7", d->kernel_size);
    // [3,3] is the fused / trainable path; [3] and [3,3,3] (conv1 no-BN+act, conv2 BN+act, conv3 BN+linear, backbone_blocks.py:
    // 174-213) run inference through the single-convolution kernel with the general epilogue [affine][ReLU][+residual]
    if (d->block_kernel != 3)
        return fail(nullptr, BF_EUNSUPPORTED, "block_kernels must be 3x3 (got %d convs of %d)", d->block_convs, d->block_kernel);
    if (d->activation != BF_ACT_RELU && d->activation != BF_ACT_LINEAR)
        return fail(nullptr, BF_EUNSUPPORTED, "block activation must be relu or linear");
    if (d->base_activation != BF_ACT_LINEAR)
        return fail(nullptr, BF_EUNSUPPORTED, "base_activation must be linear");
    if (d->head_filters < 1 || d->head_filters > 64) return fail(nullptr, BF_EUNSUPPORTED, "head filters must be in 1..64");
    if (d->out_channels < 1 || d->out_channels > 4) return fail(nullptr, BF_EUNSUPPORTED, "output_channels must be in 1..4");
    if (!(d->v_max > d->v_min)) return fail(nullptr, BF_EINVAL, "value_range must have max > min");

    bf_engine* h = new bf_engine();
    h->d = *d;
    const int k = d->kernel_size, cin = d->in_channels, hf = d->head_filters, co = d->out_channels;
    int64_t off = 0;
    h->p_base = off;
    add_tensor(h->tensors, "base/kernel", off, 4, k, k, cin, BF_C, BF_KIND_CONV, d->reg_base);
    h->n_base = off;
    h->p_blocks = off;
    const int nb = d->block_convs;
    h->p_block_stride = nb * 2304 + (d->use_bn ? (nb - 1) * 16 : 0);
    char name[64];
    for (int i = 0; i < d->no_layers; ++i) {
        for (int j = 0; j < nb; ++j) {             // keras creation order: conv j, then its BN gamma (first conv has no BN)
            snprintf(name, sizeof(name), "block%d/conv%d/kernel", i, j);
            add_tensor(h->tensors, name, off, 4, 3, 3, BF_C, BF_C, BF_KIND_CONV, d->reg_block);
            if (j >= 1 && d->use_bn) {
                snprintf(name, sizeof(name), "block%d/bn%d/gamma", i, j);
                add_tensor(h->tensors, name, off, 1, BF_C, 0, 0, 0, BF_KIND_GAMMA, BF_REG_NONE);
            }
        }
    }
    h->p_head0 = off;
    add_tensor(h->tensors, "head/conv0/kernel", off, 4, 1, 1, BF_C, hf, BF_KIND_CONV, d->reg_head);
    h->p_head1 = off;
    add_tensor(h->tensors, "head/conv1/kernel", off, 4, 1, 1, hf, co, BF_KIND_CONV, d->reg_head);
    h->n_params = off;
    int64_t soff = 0;
    if (d->use_bn) {
        for (int i = 0; i < d->no_layers; ++i)
            for (int j = 1; j < nb; ++j) {
                snprintf(name, sizeof(name), "block%d/bn%d/moving_mean", i, j);
                add_tensor(h->states, name, soff, 1, BF_C, 0, 0, 0, BF_KIND_MOVING_MEAN, BF_REG_NONE);
                snprintf(name, sizeof(name), "block%d/bn%d/moving_variance", i, j);
                add_tensor(h->states, name, soff, 1, BF_C, 0, 0, 0, BF_KIND_MOVING_VAR, BF_REG_NONE);
            }
    }
    h->n_state = soff;
    // packed layout
    int64_t ko = 0;
    h->k_base = ko; ko += align_up(h->n_base, 64);
    // per block: nb weight images, then one folded (scale, shift) pair per convolution (identity for the BN-less first one)
    h->k_blocks = ko; h->k_block_stride = nb == 2 ? 2 * BF_WPACK_FLOATS + 32 : nb * (BF_WPACK_FLOATS + 32);
    ko += h->k_block_stride * d->no_layers;
    h->k_w0 = ko; ko += align_up(16 * hf, 64);
    h->k_w1 = ko; ko += align_up(hf * co, 64);
    h->k_wh = ko; ko += 64;
    h->k_zero = ko; ko += 64;
    h->k_h3 = ko; ko += (int64_t)BF_H3_BLOCK_FLOATS * d->no_layers;
    h->k_total = ko;
    *out = h;
    return BF_OK;
}

extern "C" void bf_destroy(bf_handle h)
{
    if (!h) return;
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
}
extern "C" int64_t bf_param_count(bf_handle h) { return h ? h->n_params : -1; }
extern "C" int64_t bf_state_count(bf_handle h) { return h ? h->n_state : -1; }
extern "C" int bf_tensor_count(bf_handle h, int state) { return h ? (int)(state ? h->states.size() : h->tensors.size()) : -1; }

extern "C" int bf_tensor_at(bf_handle h, int state, int index, bf_tensor_info* out)
{
    if (!h || !out) return BF_EINVAL;
    const auto& v = state ? h->states : h->tensors;
    if (index < 0 || index >= (int)v.size()) return fail(h, BF_EINVAL, "tensor index %d out of range", index);
    *out = v[index];
    return BF_OK;
}

extern "C" int bf_set_option(bf_handle h, const char* key, int value)
{
    if (!h || !key) return BF_EINVAL;
    if (!strcmp(key, "fused_blocks")) { h->fused_blocks = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "fused_tile")) { bf_set_fused_tile(value); return BF_OK; }
    if (!strcmp(key, "h3_variant")) { h->h3_variant = value; return BF_OK; }      // per handle; < 0 = library default
    if (!strcmp(key, "h3_zigzag")) { h->h3_zigzag = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "h3_compact")) { h->h3_compact = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "h3_pair")) { h->h3_pair = value == 2 ? 2 : (value ? 1 : 0); return BF_OK; }
    if (!strcmp(key, "base_rows")) { bf_set_base_conv_rows(value); return BF_OK; }       // process-wide (A/B only)
    if (!strcmp(key, "h3_pair_head")) { h->h3_pair_head = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "fused_head")) { h->fused_head = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_zigzag")) { h->train_zigzag = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_fused_fwd")) { h->train_fused_fwd = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_bwd_dbuf")) { h->train_bwd_dbuf = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_fused_bwd")) { h->train_fused_bwd = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_bwd_block")) { h->train_bwd_block = value < 0 ? 1 : (value > 2 ? 2 : value); return BF_OK; }
    if (!strcmp(key, "train_fwd_block")) { h->train_fwd_block = value < 0 ? 1 : (value > 2 ? 2 : value); return BF_OK; }
    if (!strcmp(key, "train_fused_bwd2")) { h->train_fused_bwd2 = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_fold_finalize")) { h->train_fold_finalize = value ? 1 : 0; return BF_OK; }
    if (!strcmp(key, "train_arith")) { h->train_arith = value < 0 ? 1 : (value ? 1 : 0); return BF_OK; }
    if (!strcmp(key, "arith")) { h->arith = value < 0 ? 1 : (value ? 1 : 0); return BF_OK; }
    if (!strcmp(key, "timing")) {
        h->timing = value ? 1 : 0;
        h->n_timed = 0;                             // (re)setting the option restarts the measurement window
        if (h->timing && h->ev.empty()) {
            h->ev.resize(2 * BF_TIMING_RING, nullptr);
            for (hipEvent_t& e : h->ev)
                if (hipEventCreate(&e) != hipSuccess) return fail(h, BF_EHIP, "hipEventCreate failed");
        }
        return BF_OK;
    }
    return fail(h, BF_EINVAL, "unknown option [%s]", key);
}

// elapsed ms between the events bracketing the residual-block launches of the LAST forward and
// the number of kernel launches in that bracket.  The caller must have synchronised the stream.
extern "C" int bf_get_timing(bf_handle h, float* ms, int* launches)
{
    if (!h || !ms || !launches) return BF_EINVAL;
    if (!h->timing || h->ev.empty()) return fail(h, BF_EINVAL, "timing option is off");
    const int64_t n = h->n_timed < BF_TIMING_RING ? h->n_timed : BF_TIMING_RING;
    if (n == 0) return fail(h, BF_EINVAL, "no forward has run since the timing option was set");
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        float one = 0.f;
        hipError_t e = hipEventElapsedTime(&one, h->ev[2 * i], h->ev[2 * i + 1]);
        if (e != hipSuccess) return hip_fail(h, e, "hipEventElapsedTime");
        total += one;
    }
    *ms = (float)total;
    *launches = (int)(h->timed_launches * n);
    return BF_OK;
}

// name of the kernel that ran most of the residual-block launches of the last forward, and how many launches one forward made
extern "C" const char* bf_get_block_kernel(bf_handle h, int* launches_per_forward)
{
    if (!h) return "";
    if (launches_per_forward) *launches_per_forward = h->block_launches;
    return h->block_kernel;
}

// the kernels that ran the residual blocks of the handle's LAST bf_train_step: "fwd: <kernels>; bwd: <kernels>" ("" before the first)
extern "C" const char* bf_get_train_kernels(bf_handle h) { return h ? h->train_kernels.c_str() : ""; }

extern "C" int64_t bf_packed_bytes(bf_handle h) { return h ? h->k_total * 4 : -1; }

// ------------------------------------------------------------------------------------------
// pow2 target of pad_to_power_of_2 (bfcnn/utilities.py:736-751): the reference evaluates
// 2^ceil(log(n)/log(2)) in float32; for every n <= 4096 that equals the exact next power of two
// (checked in tests/test_oracle_properties.py), which is what is computed here.
// ------------------------------------------------------------------------------------------
static int pow2_target(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ---- tiny pack kernels ---------------------------------------------------------------------
__global__ void pack_all_convs_kernel(const float* __restrict__ params, int64_t p_blocks, int64_t p_stride, float* __restrict__ dst,
                                      int64_t d_stride, int with_dgrad, int nconv, int unit)
{
    // blockIdx.x = layer * (with_dgrad ? 2 : 1) * nconv + which ; which < nconv: forward pack of convolution `which`, else the
    // data-gradient pack of convolution which - nconv (convolution j of a block at j * 2304, + (j - 1) * 16 behind gammas)
    const int per = (with_dgrad ? 2 : 1) * nconv;
    const int layer = blockIdx.x / per, which = blockIdx.x % per;
    const int cj = which % nconv;
    const float* w = params + p_blocks + layer * p_stride + (cj == 0 ? 0 : 2304 + (int64_t)(cj - 1) * unit);
    float* o = dst + layer * d_stride + which * (with_dgrad ? (int64_t)BF_TRAIN_PACK_STRIDE : (int64_t)BF_WPACK_FLOATS);
    const int tf = which / nconv;
    for (int idx = threadIdx.x; idx < BF_WPACK_FLOATS; idx += blockDim.x) {
        const int i = idx >> 6, l = idx & 63;
        const int tap = i >> 2, kk = i & 3;
        const int cin = 4 * (l >> 4) + kk, cout = l & 15;
        o[idx] = tf ? w[((8 - tap) * 16 + cout) * 16 + cin] : w[(tap * 16 + cin) * 16 + cout];
    }
}

// folded inference BN: keras BatchNormalization(training=False): gamma*(x-mean)*rsqrt(var+eps)
__global__ void fold_bn_kernel(const float* __restrict__ params, const float* __restrict__ state, int64_t p_blocks, int64_t p_stride,
                               float* __restrict__ packed, int64_t k_blocks, int64_t k_stride, int layers, int use_bn, float eps)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= layers * 16) return;
    const int layer = i / 16, c = i % 16;
    float sc = 1.f, sh = 0.f;
    if (use_bn) {
        const float g = params[p_blocks + layer * p_stride + 4608 + c];
        const float mean = state[layer * 32 + c], var = state[layer * 32 + 16 + c];
        sc = g / sqrtf(var + eps);
        sh = -sc * mean;
    }
    float* o = packed + k_blocks + layer * k_stride + 2 * BF_WPACK_FLOATS;
    o[c] = sc;
    o[16 + c] = sh;
}

__global__ void pack_edges_kernel(const float* __restrict__ params, float* __restrict__ packed, int64_t p_base, int64_t n_base,
                                  int64_t p_head0, int64_t p_head1, int hf, int co, int64_t k_base, int64_t k_w0, int64_t k_w1,
                                  int64_t k_wh)
{
    for (int i = threadIdx.x; i < n_base; i += blockDim.x) packed[k_base + i] = params[p_base + i];
    for (int i = threadIdx.x; i < 16 * hf; i += blockDim.x) packed[k_w0 + i] = params[p_head0 + i];
    for (int i = threadIdx.x; i < hf * co; i += blockDim.x) packed[k_w1 + i] = params[p_head1 + i];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 2, o = threadIdx.x & 3;
        float s = 0.f;
        if (o < co)
            for (int j = 0; j < hf; ++j) s = fmaf(params[p_head0 + c * hf + j], params[p_head1 + j * co + o], s);
        packed[k_wh + threadIdx.x] = s;
        packed[k_wh + 64 + threadIdx.x] = 0.f;     // k_zero line (directly behind k_wh)
    }
}

// block_kernels of length 1 or 3: per block [nb weight images][nb x (scale16, shift16)]
__global__ void pack_generic_blocks_kernel(const float* __restrict__ params, const float* __restrict__ state, int64_t p_blocks,
                                           int64_t p_stride, float* __restrict__ dst, int64_t d_stride, int nb, int use_bn, float eps)
{
    const int layer = blockIdx.x / nb, j = blockIdx.x % nb;
    const int64_t conv_off = (int64_t)j * 2304 + (use_bn && j >= 2 ? (j - 1) * 16 : 0);   // gamma j-1 sits before conv j (j >= 2)
    const float* w = params + p_blocks + layer * p_stride + conv_off;
    float* o = dst + layer * d_stride + (int64_t)j * BF_WPACK_FLOATS;
    for (int idx = threadIdx.x; idx < BF_WPACK_FLOATS; idx += blockDim.x) {
        const int i = idx >> 6, l = idx & 63;
        const int tap = i >> 2, kk = i & 3;
        const int cin = 4 * (l >> 4) + kk, cout = l & 15;
        o[idx] = w[(tap * 16 + cin) * 16 + cout];
    }
    if (threadIdx.x < 16) {
        const int c = threadIdx.x;
        float sc = 1.f, sh = 0.f;
        if (use_bn && j >= 1) {
            const float g = w[2304 + c];                                   // gamma follows its convolution
            const float* st = state + ((int64_t)layer * (nb - 1) + (j - 1)) * 32;
            sc = g / sqrtf(st[16 + c] + eps);
            sh = -sc * st[c];
        }
        float* aff = dst + layer * d_stride + (int64_t)nb * BF_WPACK_FLOATS + j * 32;
        aff[c] = sc;
        aff[16 + c] = sh;
    }
}

extern "C" int bf_pack_inference(bf_handle h, const float* params, const float* state, void* packed, void* stream)
{
    if (!h) return BF_EINVAL;
    if (!params || !packed || (h->n_state > 0 && !state)) return fail(h, BF_EINVAL, "bf_pack_inference: NULL buffer");
    if ((uintptr_t)packed % 16) return fail(h, BF_EWORKSPACE, "packed buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float* pk = (float*)packed;
    const bf_resnet_desc& d = h->d;
    if (d.no_layers > 0 && d.block_convs != 2) {
        hipLaunchKernelGGL(pack_generic_blocks_kernel, dim3(d.no_layers * d.block_convs), dim3(256), 0, s, params, state, h->p_blocks,
                           h->p_block_stride, pk + h->k_blocks, h->k_block_stride, d.block_convs, d.use_bn, d.bn_eps);
        BF_HIP(hipGetLastError(), "pack_generic_blocks");
    } else if (d.no_layers > 0) {
        hipLaunchKernelGGL(pack_all_convs_kernel, dim3(d.no_layers * 2), dim3(256), 0, s, params, h->p_blocks, h->p_block_stride,
                           pk + h->k_blocks, h->k_block_stride, 0, 2, 2320);
        BF_HIP(hipGetLastError(), "pack_all_convs");
        hipLaunchKernelGGL(fold_bn_kernel, dim3((d.no_layers * 16 + 255) / 256), dim3(256), 0, s, params, state, h->p_blocks,
                           h->p_block_stride, pk, h->k_blocks, h->k_block_stride, d.no_layers, d.use_bn, d.bn_eps);
        BF_HIP(hipGetLastError(), "fold_bn");
        BF_HIP(bf_launch_pack_h3(params, state, h->p_blocks, h->p_block_stride, pk + h->k_h3, BF_H3_BLOCK_FLOATS, d.no_layers,
                                 d.use_bn, d.bn_eps, nullptr, nullptr, s), "pack_h3");
    }
    hipLaunchKernelGGL(pack_edges_kernel, dim3(1), dim3(256), 0, s, params, pk, h->p_base, h->n_base, h->p_head0, h->p_head1,
                       d.head_filters, d.out_channels, h->k_base, h->k_w0, h->k_w1, h->k_wh);
    BF_HIP(hipGetLastError(), "pack_edges");
    return BF_OK;
}

// ------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------
struct TrainLayout {
    int64_t wpack, wh, bn_scale, bn_meaninv, coef, stage1, partial, wslots, acts, extra, total;   // float offsets
    int64_t act_floats, partial_floats, wslot_floats;
};

static int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

static TrainLayout train_layout(bf_handle h, int B, int H, int W)
{
    TrainLayout L;
    const int N = h->d.no_layers;
    int64_t o = 0;
    const int nb = h->d.block_convs;                       // convolutions per block: forward + data-gradient pack each
    L.wpack = o; o += (int64_t)N * 2 * nb * BF_TRAIN_PACK_STRIDE;
    L.wh = o; o += 64;
    L.bn_scale = o; o += (int64_t)N * (nb > 1 ? nb - 1 : 1) * 32 + 32;
    L.bn_meaninv = o; o += (int64_t)N * (nb > 1 ? nb - 1 : 1) * 32 + 32;
    L.coef = o; o += 64;
    L.stage1 = o; o += 64 * 32 * 2;            // doubles
    int64_t pf = (int64_t)bf_conv3x3_c16_grid(B, H, W) * 32;
    pf = max64(pf, 4096 * 32);
    pf = max64(pf, (int64_t)bf_wgrad_grid(B, H, W) * 2304);
    pf = max64(pf, (int64_t)bf_bwd3x3_h3_grid(B, H, W) * (2304 + 32));
    pf = max64(pf, (int64_t)bf_fwd_block_h3t_grid(B, H, W) * 32);
    pf = max64(pf, (int64_t)bf_bwd_block_h3t_grid(B, H, W) * (2304 + 64));           // + two sets of BatchNorm sums in turn
    pf = max64(pf, (int64_t)bf_base_wgrad_grid(B, H, W) * h->n_base);
    pf = max64(pf, (int64_t)bf_head_train_grid(B, H, W) * 80);
    L.partial_floats = align_up(pf, 64);
    L.partial = o; o += L.partial_floats + 256;     // +256: reduced head sums / scratch
    // one weight-gradient partial slot per block convolution (fused backward kernel): summed by ONE launch at the end of the step
    L.wslot_floats = max64(bf_bwd3x3_h3_grid(B, H, W), bf_bwd_block_h3t_grid(B, H, W)) * 2304;
    L.wslots = o; o += L.wslot_floats * N * nb;
    o = align_up(o, 64);
    L.act_floats = (int64_t)B * H * W * 16;
    // A_0..A_N, per block and convolution j >= 1 its input T_j and its raw output C_j, dA + two more gradient buffers (the
    // fused backward kernel reads its operands with a halo, so it never writes over one of them)
    L.acts = o; o += L.act_floats * ((int64_t)N * (2 * (nb - 1) + 1) + 4);
    // RMSE / SSIM loss terms (loss_terms.hip): prediction, extra gradient, three window maps (4 channels at most), partials
    L.extra = o; o += (int64_t)B * H * W * 4 * 5 + 4096 + align_up(B, 64) + 64;
    L.total = o;
    return L;
}

extern "C" int64_t bf_workspace_bytes(bf_handle h, int mode, int B, int H, int W)
{
    if (!h || B <= 0 || H <= 0 || W <= 0) return -1;
    if (mode == BF_MODE_INFERENCE) {
        const int Hp = pow2_target(H), Wp = pow2_target(W);      // u8 path pads; f32 path needs <= this
        return (int64_t)B * Hp * Wp * 16 * 4 * 3 + BF_STATUS_BYTES;
    }
    return train_layout(h, B, H, W).total * 4;
}

// ------------------------------------------------------------------------------------------
// inference
// ------------------------------------------------------------------------------------------
static int forward_common(bf_handle h, const float* pk, const void* in, int in_is_u8, void* out, int out_is_u8, int B, int Hs,
                          int Ws, int H, int W, void* ws, int64_t ws_bytes, hipStream_t s)
{
    const bf_resnet_desc& d = h->d;
    const int64_t act_bytes = (int64_t)B * H * W * 16 * 4;
    if (!ws || (uintptr_t)ws % 16) return fail(h, BF_EWORKSPACE, "workspace must be a 16-byte aligned device buffer");
    if (ws_bytes < act_bytes * 3 + BF_STATUS_BYTES)
        return fail(h, BF_EWORKSPACE, "workspace too small: %lld < %lld bytes", (long long)ws_bytes,
                    (long long)(act_bytes * 3 + BF_STATUS_BYTES));
    int* status = (int*)((char*)ws + (ws_bytes - BF_STATUS_BYTES) / 4 * 4);
    float* buf[3] = {(float*)ws, (float*)((char*)ws + act_bytes), (float*)((char*)ws + 2 * act_bytes)};

    BaseConvArgs ba;
    ba.in = in; ba.out = buf[0]; ba.w = pk + h->k_base;
    ba.B = B; ba.Hs = Hs; ba.Ws = Ws; ba.H = H; ba.W = W; ba.cin = d.in_channels; ba.k = d.kernel_size;
    ba.in_is_u8 = in_is_u8; ba.act_relu = d.base_activation == BF_ACT_RELU;
    ba.v_min = d.v_min; ba.v_max = d.v_max;
    // split-f16 blocks keep the activations split-planar between base conv and head (same bytes as fp32)
    const int h3 = h->fused_blocks && h->arith == 1 && d.no_layers > 0 && d.block_convs == 2;
    // the head epilogue exists in the row-streaming tile kernel only: asking for it selects that kernel for the last block
    const bool head_in_block = h3 && h->fused_head && d.head_activation == BF_ACT_LINEAR && d.out_channels == 3;
    // compact layout (48 instead of 64 bytes per pixel between the launches): when every block runs the full-row streaming kernel
    bool compact = false;
    if (h3 && h->h3_compact && !head_in_block) {
        FusedH3Args probe;
        memset(&probe, 0, sizeof(probe));
        probe.B = B; probe.H = H; probe.W = W; probe.variant = h->h3_variant;
        compact = bf_fused_block_h3_is_streaming(probe);
    }
    ba.out_split = h3 ? (compact ? 2 : 1) : 0;
    ba.status = status;
    BF_HIP(bf_launch_base_conv(ba, s), "base_conv");

    int cur = 0;
    const int64_t tslot = h->n_timed % BF_TIMING_RING;
    if (h->timing) BF_HIP(hipEventRecord(h->ev[2 * tslot], s), "hipEventRecord");
    // two blocks per launch (fused_h3w.hip) wherever the one-block streaming kernel would run: x1 and both intermediate
    // activations stay in LDS, 128 instead of 256 bytes per pixel through HBM for a pair
    bool pair_ok = false;
    if (h3 && h->h3_pair && !compact) {
        FusedH3Args probe;
        memset(&probe, 0, sizeof(probe));
        probe.B = B; probe.H = H; probe.W = W; probe.variant = h->h3_variant;
        pair_ok = (h->h3_pair == 2 || bf_fused_block_h3_use_pairs(probe)) && bf_fused_block2_h3w_supports(H, W);   // 2: A/B only
    }
    int launches = 0;
    // which kernel ran how many of the blocks: noted AT each launch site (bf_get_block_kernel reports the one that ran the most)
    const char* ran_name[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int ran_count[6] = {0, 0, 0, 0, 0, 0};
    auto ran = [&](const char* name, int n) {
        for (int k = 0; k < 6; ++k) {
            if (ran_name[k] == nullptr) ran_name[k] = name;
            if (!strcmp(ran_name[k], name)) { ran_count[k] += n; return; }
        }
    };
    // an odd block count runs its single block FIRST, so that the last launch is a pair and can carry the head
    const bool head_in_pair = pair_ok && !head_in_block && h->h3_pair_head && d.no_layers >= 2 &&
                              d.head_activation == BF_ACT_LINEAR && d.out_channels == 3;
    for (int i = 0; i < d.no_layers; ++i) {
        const float* blk = pk + h->k_blocks + i * h->k_block_stride;
        if (pair_ok && i + 1 < d.no_layers && !(head_in_block && i + 1 == d.no_layers - 1) && !(i == 0 && (d.no_layers & 1))) {
            FusedH3WArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.in = buf[cur]; fa.out = buf[cur ^ 1];
            for (int b = 0; b < 2; ++b) {
                const float* aux = pk + h->k_h3 + (int64_t)(i + b) * BF_H3_BLOCK_FLOATS + 2 * BF_H3_WPACK_FLOATS;
                fa.aux[b] = aux; fa.w1r[b] = aux + 64; fa.w2r[b] = aux + 64 + BF_H3R_WPACK_FLOATS;
            }
            fa.B = B; fa.H = H; fa.W = W;
            fa.reverse_tiles = h->h3_zigzag ? (launches & 1) : 0;
            fa.act1_relu = d.activation == BF_ACT_RELU; fa.zeros = pk + h->k_zero; fa.dbg = nullptr;
            if (head_in_pair && i + 2 == d.no_layers) {          // last pair: the linear head rides in its store step
                fa.head_wh = pk + h->k_wh; fa.head_out = out; fa.head_u8 = out_is_u8; fa.Ho = Hs; fa.Wo = Ws;
                fa.denormalize = d.denormalize; fa.v_min = d.v_min; fa.v_max = d.v_max; fa.status = status;
            }
            BF_HIP(bf_launch_fused_block2_h3w(fa, s), "fused_block2_h3w");
            cur ^= 1;
            ++i;
            ++launches;
            ran("fused_block2_h3w_kernel", 2);                 // (weighted by the blocks a launch runs)
            continue;
        }
        launches += d.block_convs != 2 ? d.block_convs : ((h3 || h->fused_blocks) ? 1 : 2);
        if (d.block_convs != 2) {
            // general block: conv1 (no BN) + act, [conv2 + BN + act,] conv_last + BN + linear, + skip (backbone_blocks.py:174-242;
            // a one-convolution block is conv (no BN, linear) + skip: the last activation is forced to base_activation)
            const int nbk = d.block_convs;
            const float* aff = blk + (int64_t)nbk * BF_WPACK_FLOATS;
            int src = cur;
            for (int j = 0; j < nbk; ++j) {
                const bool last = j == nbk - 1;
                int dst = (src + 1) % 3;
                if (dst == cur) dst = (dst + 1) % 3;                 // the block input stays alive for the skip
                ConvArgs ca;
                memset(&ca, 0, sizeof(ca));
                ca.in = buf[src]; ca.out = buf[dst]; ca.wpack = blk + (int64_t)j * BF_WPACK_FLOATS; ca.B = B; ca.H = H; ca.W = W;
                ca.scale = aff + j * 32; ca.shift = ca.scale + 16; ca.res = buf[cur];
                const bool relu = !last && d.activation == BF_ACT_RELU;
                int epi = (j >= 1 ? EPI_AFFINE : 0) | (relu ? EPI_RELU : 0) | (last ? EPI_RES : 0);
                BF_HIP(bf_launch_conv3x3_c16(ca, epi, s), "block conv");
                ran("conv3x3_c16_kernel", 1);
                src = dst;
            }
            cur = src;
        } else if (h3) {
            const float* b3 = pk + h->k_h3 + (int64_t)i * BF_H3_BLOCK_FLOATS;
            FusedH3Args fa;
            fa.in = buf[cur]; fa.out = buf[cur ^ 1];
            fa.w1 = b3; fa.w2 = b3 + BF_H3_WPACK_FLOATS; fa.aux = b3 + 2 * BF_H3_WPACK_FLOATS;
            fa.w1r = fa.aux + 64; fa.w2r = fa.aux + 64 + BF_H3R_WPACK_FLOATS;
            fa.B = B; fa.H = H; fa.W = W; fa.tiles_x = fa.tiles_y = fa.ntiles = 0; fa.rows_per_tile = 0; fa.variant = h->h3_variant;
            // consecutive blocks walk the batch in opposite directions: a block starts on the bands the previous one wrote
            // last, which are the ones still in the 256 MB Infinity Cache
            fa.reverse_tiles = h->h3_zigzag ? ((launches - 1) & 1) : 0;
            fa.act1_relu = d.activation == BF_ACT_RELU; fa.zeros = pk + h->k_zero; fa.dump = (char*)status + 1024; fa.dbg = nullptr;
            fa.head_wh = nullptr; fa.head_out = nullptr; fa.head_u8 = 0; fa.Ho = fa.Wo = 0; fa.denormalize = 0;
            fa.v_min = fa.v_max = 0.f; fa.status = nullptr; fa.compact = compact;
            if (head_in_block && i == d.no_layers - 1) {          // last block: linear head in its epilogue, no head kernel
                fa.variant = 1;
                fa.head_wh = pk + h->k_wh; fa.head_out = out; fa.head_u8 = out_is_u8; fa.Ho = Hs; fa.Wo = Ws;
                fa.denormalize = d.denormalize; fa.v_min = d.v_min; fa.v_max = d.v_max; fa.status = status;
            }
            BF_HIP(bf_launch_fused_block_h3(fa, s), "fused_block_h3");
            ran(bf_fused_block_h3_kernel_name(fa), 1);
            cur ^= 1;
        } else if (h->fused_blocks) {
            FusedBlockArgs fa;
            fa.in = buf[cur]; fa.out = buf[cur ^ 1];
            fa.w1pack = blk; fa.w2pack = blk + BF_WPACK_FLOATS;
            fa.scale = blk + 2 * BF_WPACK_FLOATS; fa.shift = fa.scale + 16;
            fa.B = B; fa.H = H; fa.W = W; fa.tiles_x = fa.tiles_y = fa.ntiles = 0;
            fa.act1_relu = d.activation == BF_ACT_RELU; fa.dbg = nullptr; fa.zeros = pk + h->k_zero;
            BF_HIP(bf_launch_fused_block(fa, s), "fused_block");
            ran(bf_fused_block_kernel_name(), 1);
            cur ^= 1;
        } else {
            // unfused: T = act(conv1 x) ; y = x + scale*conv2(T) + shift
            const int t = (cur + 1) % 3, y = (cur + 2) % 3;
            ConvArgs ca;
            memset(&ca, 0, sizeof(ca));
            ca.in = buf[cur]; ca.out = buf[t]; ca.wpack = blk; ca.B = B; ca.H = H; ca.W = W;
            BF_HIP(bf_launch_conv3x3_c16(ca, d.activation == BF_ACT_RELU ? EPI_RELU : 0, s), "conv1");
            ca.in = buf[t]; ca.out = buf[y]; ca.wpack = blk + BF_WPACK_FLOATS;
            ca.scale = blk + 2 * BF_WPACK_FLOATS; ca.shift = ca.scale + 16; ca.res = buf[cur];
            BF_HIP(bf_launch_conv3x3_c16(ca, EPI_AFFINE | EPI_RES, s), "conv2");
            ran("conv3x3_c16_kernel", 2);
            cur = y;
        }
    }
    if (h->timing) {
        BF_HIP(hipEventRecord(h->ev[2 * tslot + 1], s), "hipEventRecord");
        h->timed_launches = launches;
        ++h->n_timed;
    }
    h->block_launches = launches;
    h->block_kernel = "";
    for (int k = 0, best = 0; k < 6 && ran_name[k]; ++k)
        if (ran_count[k] > best) { best = ran_count[k]; h->block_kernel = ran_name[k]; }
    if (head_in_block || head_in_pair) return BF_OK;
    HeadArgs ha;
    ha.feat = buf[cur];
    ha.w0 = pk + h->k_w0; ha.w1 = pk + h->k_w1;
    ha.wh = d.head_activation == BF_ACT_LINEAR ? pk + h->k_wh : nullptr;
    ha.out = out;
    ha.B = B; ha.H = H; ha.W = W; ha.Ho = Hs; ha.Wo = Ws; ha.hf = d.head_filters; ha.cout = d.out_channels;
    ha.act = d.head_activation; ha.out_is_u8 = out_is_u8; ha.denormalize = d.denormalize;
    ha.v_min = d.v_min; ha.v_max = d.v_max; ha.leaky_alpha = d.leaky_alpha;
    ha.feat_split = h3 ? (compact ? 2 : 1) : 0;
    ha.status = status;
    BF_HIP(bf_launch_head(ha, s), "head");
    return BF_OK;
}

static int check_dims(bf_handle h, const void* packed, const void* in, const void* out, int B, int H, int W)
{
    if (!packed || !in || !out) return fail(h, BF_EINVAL, "NULL tensor pointer");
    if (B <= 0 || H <= 0 || W <= 0) return fail(h, BF_EINVAL, "batch/height/width must be positive (got %d,%d,%d)", B, H, W);
    if ((int64_t)B * H * W * 16 >= ((int64_t)1 << 40)) return fail(h, BF_EINVAL, "tensor too large");
    return BF_OK;
}

extern "C" int bf_forward_u8(bf_handle h, const void* packed, const uint8_t* in, uint8_t* out, int B, int H, int W, void* ws,
                             int64_t ws_bytes, void* stream)
{
    if (!h) return BF_EINVAL;
    int rc = check_dims(h, packed, in, out, B, H, W);
    if (rc) return rc;
    return forward_common(h, (const float*)packed, in, 1, out, 1, B, H, W, pow2_target(H), pow2_target(W), ws, ws_bytes,
                          (hipStream_t)stream);
}

// DenoiserModule(cast_to_uint8=False): the same chain as bf_forward_u8 without the final round + cast
extern "C" int bf_forward_u8_f32(bf_handle h, const void* packed, const uint8_t* in, float* out, int B, int H, int W, void* ws,
                                 int64_t ws_bytes, void* stream)
{
    if (!h) return BF_EINVAL;
    int rc = check_dims(h, packed, in, out, B, H, W);
    if (rc) return rc;
    return forward_common(h, (const float*)packed, in, 1, out, 0, B, H, W, pow2_target(H), pow2_target(W), ws, ws_bytes,
                          (hipStream_t)stream);
}

extern "C" int bf_forward_f32(bf_handle h, const void* packed, const float* in, float* out, int B, int H, int W, void* ws,
                              int64_t ws_bytes, void* stream)
{
    if (!h) return BF_EINVAL;
    int rc = check_dims(h, packed, in, out, B, H, W);
    if (rc) return rc;
    return forward_common(h, (const float*)packed, in, 0, out, 0, B, H, W, H, W, ws, ws_bytes, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// training
// ------------------------------------------------------------------------------------------
// reduces the head partials and writes the head gradients + the data-term losses
//   partial rows: [0,64) M | 64 sum|e| | 65 hinge sum | 66 per-block sum e^2 (blocks of one image are contiguous)
__global__ __launch_bounds__(1024) void head_finalize_kernel(const float* __restrict__ partial, int nblk, int blocks_per_image, int B,
                                                             double numel, double per_image, const float* __restrict__ w0,
                                                             const float* __restrict__ w1, int hf, int co, float* __restrict__ g0,
                                                             float* __restrict__ g1, float* __restrict__ losses,
                                                             float mae_multiplier, float depth_weight)
{
    constexpr int NS = 15;                     // 66 columns x 15 row stripes = 990 threads
    __shared__ double M[64];
    __shared__ double sums[2];
    __shared__ double rm[1024];
    __shared__ double part[NS][66];
    const int tid = threadIdx.x;
    // (one thread per column walked all B*64 rows alone: 540 us per step; 3 stripes on 198 threads: 61 us), fixed order
    if (tid < 66 * NS) {
        const int col = tid % 66, stripe = tid / 66;
        double s = 0.0;
        // loads issued eight at a time (a rolled load -> add loop waits out one L2 round trip per row); same add order
        int r = stripe;
        for (; r + 7 * NS < nblk; r += 8 * NS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + NS * u) * 80 + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < nblk; r += NS) s += (double)partial[(size_t)r * 80 + col];
        part[stripe][col] = s;
    }
    __syncthreads();
    if (tid < 66) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) s += part[k][tid];
        if (tid < 64) M[tid] = s; else sums[tid - 64] = s;
    }
    // rmse: mean over images of sqrt(mean_sq + DEFAULT_EPSILON)   (loss.py:92-113, constants.py:7)
    double acc = 0.0;
    for (int b = tid; b < B; b += 1024) {
        double sq = 0.0;
        int k = 0;
        for (; k + 7 < blocks_per_image; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b * blocks_per_image + k + u) * 80 + 66];
#pragma unroll
            for (int u = 0; u < 8; ++u) sq += (double)v[u];
        }
        for (; k < blocks_per_image; ++k) sq += (double)partial[(size_t)(b * blocks_per_image + k) * 80 + 66];
        acc += sqrt(sq / per_image + 1e-3);
    }
    rm[tid] = acc;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) rm[tid] += rm[tid + st];
        __syncthreads();
    }
    // dW0[c][j] = sum_o M[c][o] * W1[j][o] ; dW1[j][o] = sum_c W0[c][j] * M[c][o]
    for (int i = tid; i < 16 * hf; i += 1024) {
        const int c = i / hf, j = i % hf;
        double s = 0.0;
        for (int o = 0; o < co; ++o) s += M[c * 4 + o] * (double)w1[j * co + o];
        g0[i] = (float)s;
    }
    for (int i = tid; i < hf * co; i += 1024) {
        const int j = i / co, o = i % co;
        double s = 0.0;
        for (int c = 0; c < 16; ++c) s += (double)w0[c * hf + j] * M[c * 4 + o];
        g1[i] = (float)s;
    }
    if (tid == 0) {
        const double mae_actual = sums[0] / numel;
        const double mae_loss = mae_multiplier > 0.f ? sums[1] / numel : 0.0;
        losses[BF_LOSS_MAE] = (float)mae_actual;
        losses[BF_LOSS_MSE] = (float)(rm[0] / (double)B);
        losses[BF_LOSS_SSIM] = 0.f;
        losses[BF_LOSS_DENOISER_TOTAL] = (float)(mae_loss * mae_multiplier);
        losses[BF_LOSS_TOTAL] = (float)(mae_loss * mae_multiplier * depth_weight);     // + model loss added by reg kernel
    }
}

// regularisers (keras "l1" -> 0.01*sum|w|, "l2" -> 0.01*sum w^2; bfcnn/loss.py:181-187):
// adds d(reg*regularization)/dw to grads; per-workgroup fp64 partial sums (fixed order), finished by
// regularizer_finalize_kernel.  (One workgroup walking all 84 k parameters alone took 50 us of a step.)
constexpr int REG_GRID = 64;
__global__ __launch_bounds__(1024) void regularizer_kernel(const float* __restrict__ params, float* __restrict__ grads, int64_t n,
                                                           int64_t n_base, int64_t p_blocks, int64_t p_stride, int64_t p_head0,
                                                           int reg_base, int reg_block, int reg_head, float regularization,
                                                           double* __restrict__ wg_sums, int unit)
{
    __shared__ double red[1024];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)REG_GRID * 1024) {
        int reg;
        if (i < n_base) reg = reg_base;
        else if (i >= p_head0) reg = reg_head;
        else {      // block: conv0 [2304], then per further convolution its kernel [2304] and (with BatchNorm) its gamma [16]
            const unsigned r = (unsigned)(i - p_blocks) % (unsigned)p_stride;
            reg = (r < 2304u || ((r - 2304u) % (unsigned)unit) < 2304u) ? reg_block : BF_REG_NONE;
        }
        const float w = params[i];
        if (reg == BF_REG_L1) {
            acc += 0.01 * fabs((double)w);
            grads[i] = grads[i] + regularization * 0.01f * (w > 0.f ? 1.f : (w < 0.f ? -1.f : 0.f));
        } else if (reg == BF_REG_L2) {
            acc += 0.01 * (double)w * (double)w;
            grads[i] = grads[i] + regularization * 0.02f * w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) wg_sums[blockIdx.x] = red[0];
}

__global__ void regularizer_finalize_kernel(const double* __restrict__ wg_sums, float regularization, float* __restrict__ losses)
{
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int k = 0; k < REG_GRID; ++k) r += wg_sums[k];
        losses[BF_LOSS_REGULARIZATION] = (float)r;
        losses[BF_LOSS_MODEL_TOTAL] = (float)(r * regularization);
        losses[BF_LOSS_TOTAL] = losses[BF_LOSS_TOTAL] + (float)(r * regularization);
        losses[BF_LOSS_GRAD_NORM] = 0.f;
    }
}

__global__ void premultiply_head_kernel(const float* __restrict__ w0, const float* __restrict__ w1, int hf, int co, float* __restrict__ wh)
{
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 2, o = threadIdx.x & 3;
        float s = 0.f;
        if (o < co)
            for (int j = 0; j < hf; ++j) s = fmaf(w0[c * hf + j], w1[j * co + o], s);
        wh[threadIdx.x] = s;
    }
}

__global__ void scale_range_kernel(float* __restrict__ p, int64_t n, float f)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] *= f;
}

__global__ void fill_identity_affine_kernel(float* scale_shift)
{
    if (threadIdx.x < 16) { scale_shift[threadIdx.x] = 1.f; scale_shift[16 + threadIdx.x] = 0.f; }
}

extern "C" int bf_train_step(bf_handle h, const float* params, float* state, const float* gt, const float* noisy, int B, int H,
                             int W, const bf_loss_desc* loss, float* predictions, float* grads, float* losses, void* ws,
                             int64_t ws_bytes, void* stream)
{
    if (!h) return BF_EINVAL;
    const bf_resnet_desc& d = h->d;
    if (!params || !gt || !noisy || !loss || !grads || !losses || (h->n_state > 0 && !state))
        return fail(h, BF_EINVAL, "bf_train_step: NULL argument");
    if (loss->struct_size != (int32_t)sizeof(bf_loss_desc)) return fail(h, BF_EINVAL, "bf_loss_desc struct_size mismatch");
    if (B <= 0 || H <= 0 || W <= 0) return fail(h, BF_EINVAL, "batch/height/width must be positive");
    const bool extra_terms = loss->ssim_multiplier > 0.f || loss->mse_multiplier > 0.f;     // use_ssim / use_mse (loss.py:174-179)
    if (loss->ssim_multiplier > 0.f && (H < 7 || W < 7)) return fail(h, BF_EINVAL, "SSIM needs images of at least 7x7");
    if (loss->ssim_multiplier > 0.f && !d.denormalize)
        return fail(h, BF_EUNSUPPORTED, "SSIM term (max_val 255) is built for the denormalised hydra output");
    if (d.head_activation != BF_ACT_LINEAR) return fail(h, BF_EUNSUPPORTED, "training is built for the linear denoiser head");
    if (d.block_convs < 1 || d.block_convs > 3) return fail(h, BF_EUNSUPPORTED, "training is built for blocks of 1 to 3 convolutions (got %d)", d.block_convs);
    if (d.out_channels != d.in_channels) return fail(h, BF_EINVAL, "gt/prediction channel mismatch");
    const TrainLayout L = train_layout(h, B, H, W);
    if (!ws || (uintptr_t)ws % 16) return fail(h, BF_EWORKSPACE, "workspace must be a 16-byte aligned device buffer");
    if (ws_bytes < L.total * 4) return fail(h, BF_EWORKSPACE, "workspace too small: %lld < %lld bytes", (long long)ws_bytes,
                                            (long long)(L.total * 4));
    hipStream_t s = (hipStream_t)stream;
    float* w = (float*)ws;
    const int N = d.no_layers;
    const int nb = d.block_convs;                   // block: conv_0 [+ act] , conv_j + BN [+ act] (j >= 1), last one linear, + skip
    const int unit = d.use_bn ? 2320 : 2304;        // floats from convolution kernel j >= 1 of a block to the next (gamma in between)
    const int64_t npix = (int64_t)B * H * W;
    const double count = (double)npix;
    float* partial = w + L.partial;
    double* stage1 = reinterpret_cast<double*>(w + L.stage1);     // (offset is a multiple of 2 floats: 8-byte aligned)
    auto ACT = [&](int64_t i) { return w + L.acts + i * L.act_floats; };
    // buffer map: A_i = ACT(i) (i = 0..N: block inputs / outputs) ; T(i,j) = input of convolution j >= 1 of block i (the
    // activated output of convolution j-1) ; C(i,j) = raw output of convolution j >= 1 (in front of its BatchNorm) ; dA
    auto A = [&](int i) { return ACT(i); };
    auto T = [&](int i, int j) { return ACT(N + 1 + (int64_t)i * (nb - 1) + (j - 1)); };
    auto C = [&](int i, int j) { return ACT(N + 1 + (int64_t)N * (nb - 1) + (int64_t)i * (nb - 1) + (j - 1)); };
    float* dA = ACT(N + 1 + 2 * (int64_t)N * (nb - 1));
    auto conv_off = [&](int j) { return j == 0 ? (int64_t)0 : 2304 + (int64_t)(j - 1) * unit; };     // inside a block's parameters
    auto bn_idx = [&](int i, int j) { return (int64_t)i * (nb - 1) + (j - 1); };                       // BatchNorm of convolution j >= 1

    const int h3t = h->train_arith == 1;
    if (N > 0) {
        if (h3t) {
            BF_HIP(bf_launch_pack_h3_train(params, h->p_blocks, h->p_block_stride, w + L.wpack, N, nb, unit, s), "pack_h3_train");
        } else {
            hipLaunchKernelGGL(pack_all_convs_kernel, dim3(N * 2 * nb), dim3(256), 0, s, params, h->p_blocks, h->p_block_stride,
                               w + L.wpack, (int64_t)2 * nb * BF_TRAIN_PACK_STRIDE, 1, nb, unit);
            BF_HIP(hipGetLastError(), "pack_all_convs");
        }
    }
    // every tile kernel of the step reads what the one before it wrote: alternate the walking direction (train_zigzag)
    int launch_no = 0;
    auto next_reverse = [&]() { return h->train_zigzag ? (launch_no++ & 1) : 0; };
    auto conv = [&](ConvArgs& ca, int epi) {
        if (!h3t) return bf_launch_conv3x3_c16(ca, epi, s);
        ca.reverse = next_reverse();
        return bf_launch_conv3x3_h3(ca, epi, s);
    };
    auto wgrad = [&](const float* xx, const float* dyy, float* dw) {
        return h3t ? bf_launch_wgrad3x3_h3(xx, dyy, partial, dw, B, H, W, s) : bf_launch_wgrad3x3_c16(xx, dyy, partial, dw, B, H, W, s);
    };
    hipLaunchKernelGGL(premultiply_head_kernel, dim3(1), dim3(64), 0, s, params + h->p_head0, params + h->p_head1, d.head_filters,
                       d.out_channels, w + L.wh);
    BF_HIP(hipGetLastError(), "premultiply_head");

    // ---- forward, training mode (hydra(noisy, training=True), train_loop.py:249-251, 277) ----
    BaseConvArgs ba;
    ba.in = noisy; ba.out = A(0); ba.w = params + h->p_base;
    ba.B = B; ba.Hs = H; ba.Ws = W; ba.H = H; ba.W = W; ba.cin = d.in_channels; ba.k = d.kernel_size; ba.in_is_u8 = 0;
    ba.act_relu = 0; ba.v_min = d.v_min; ba.v_max = d.v_max; ba.out_split = 0; ba.status = nullptr;
    BF_HIP(bf_launch_base_conv(ba, s), "base_conv");
    const int conv_grid = bf_conv3x3_c16_grid(B, H, W);
    const bool relu = d.activation == BF_ACT_RELU;
    bool pending_affine = false;
    // whole blocks in one kernel (train_fwd_h3t.hip): [3,3] blocks, BatchNorm on the second convolution, the split-f16 arithmetic,
    // images up to 256 columns, and a forward of enough rows that its bands (rows + 6 steps each) keep 256 workgroups busy
    const bool fwd_block = h3t && h->train_fwd_block && h->train_fused_fwd && nb == 2 && d.use_bn && bf_fwd_block_h3t_supports(H, W) &&
                           (h->train_fwd_block == 2 || (int64_t)B * H >= 4096);
    // the whole backward of a block in one kernel that RECOMPUTES T_i from A_i (train_bwd_h3t.hip): same kind of block, any width
    const int64_t bwd_strips = (W + 127) / 128;
    const bool bwd_block = h3t && h->train_bwd_block && h->train_fused_bwd && nb == 2 && d.use_bn && bf_bwd_block_h3t_supports(H, W) &&
                           (h->train_bwd_block == 2 || (int64_t)B * H * bwd_strips >= 8192);
    const bool need_t = !bwd_block;                         // the per-convolution backward kernels read T_i
    for (int i = 0; i < N; ++i) {
        const float* wp = w + L.wpack + (int64_t)i * 2 * nb * BF_TRAIN_PACK_STRIDE;        // forward packs 0..nb-1, then data-gradient packs
        if (fwd_block) {
            // A_i = A_{i-1} + bn(C_{i-1}) on load ; T_i = act(conv_0 A_i) ; C_i = conv_1 T_i + its batch statistics.
            // train_fold_finalize: the BatchNorm finalisation of block i - 1 runs in THIS launch's prologue (every workgroup sums that
            // block's partials itself; two partial buffers in turn), so a forward is one launch per block instead of two
            const int fgrid = bf_fwd_block_h3t_grid(B, H, W);
            const int64_t pp = ((int64_t)fgrid * 32 + 63) / 64 * 64;
            const bool fold = h->train_fold_finalize != 0;
            float* part_i = fold ? partial + (i & 1) * pp : partial;
            FwdBlockH3Args fa;
            memset(&fa, 0, sizeof(fa));
            fa.B = B; fa.H = H; fa.W = W; fa.reverse = next_reverse(); fa.act_relu = relu;
            fa.x = A(i);
            if (pending_affine) {
                fa.x = A(i - 1); fa.pre_c = C(i - 1, 1); fa.a_out = A(i);
                fa.pre_scale = w + L.bn_scale + bn_idx(i - 1, 1) * 32; fa.pre_shift = fa.pre_scale + 16;
                if (fold) {
                    fa.fin_partial = partial + ((i - 1) & 1) * pp; fa.fin_nblk = fgrid; fa.fin_count = count;
                    fa.fin_gamma = params + h->p_blocks + (i - 1) * h->p_block_stride + conv_off(1) + 2304;
                    fa.fin_mm = state + bn_idx(i - 1, 1) * 32; fa.fin_mv = fa.fin_mm + 16;
                    fa.fin_eps = d.bn_eps; fa.fin_momentum = d.bn_momentum;
                    fa.fin_scale = w + L.bn_scale + bn_idx(i - 1, 1) * 32; fa.fin_meaninv = w + L.bn_meaninv + bn_idx(i - 1, 1) * 32;
                }
                pending_affine = false;
            }
            fa.t_out = need_t ? T(i, 1) : nullptr; fa.c_out = C(i, 1);
            fa.wpack0 = wp; fa.wpack1 = wp + BF_TRAIN_PACK_STRIDE; fa.stats = part_i;
            BF_HIP(bf_launch_fwd_block_h3t(fa, s), "fwd_block_h3t");
            if (fold && i + 1 < N) {
                pending_affine = true;                              // block i + 1 finalises this BatchNorm itself
                continue;
            }
            float* scale = w + L.bn_scale + bn_idx(i, 1) * 32;
            BF_HIP(bf_launch_bn_finalize(part_i, fgrid, count, params + h->p_blocks + i * h->p_block_stride + conv_off(1) + 2304,
                                         state + bn_idx(i, 1) * 32, state + bn_idx(i, 1) * 32 + 16, d.bn_eps, d.bn_momentum, scale,
                                         scale + 16, w + L.bn_meaninv + bn_idx(i, 1) * 32, stage1, s), "bn_finalize");
            if (i + 1 < N) pending_affine = true;
            else BF_HIP(bf_launch_affine_add(A(i), C(i, 1), scale, scale + 16, A(i + 1), npix, s), "affine_add");
            continue;
        }
        for (int j = 0; j < nb; ++j) {
            const bool last = j == nb - 1, bn = j >= 1 && d.use_bn;
            ConvArgs ca;
            memset(&ca, 0, sizeof(ca));
            ca.B = B; ca.H = H; ca.W = W;
            ca.in = j == 0 ? A(i) : T(i, j); ca.wpack = wp + (int64_t)j * BF_TRAIN_PACK_STRIDE;
            if (j == 0 && pending_affine) {
                // A(i) = A(i-1) + scale * C(i-1, last) + shift has not been formed yet: this convolution does it on load
                ca.in = A(i - 1); ca.pre_c = C(i - 1, nb - 1); ca.pre_out = A(i);
                ca.pre_scale = w + L.bn_scale + bn_idx(i - 1, nb - 1) * 32; ca.pre_shift = ca.pre_scale + 16;
                pending_affine = false;
            }
            if (bn) {
                // conv -> BatchNorm (batch statistics ride in the convolution's epilogue) -> [activation | + skip]
                float* scale = w + L.bn_scale + bn_idx(i, j) * 32;
                ca.out = C(i, j); ca.stats = partial;
                BF_HIP(conv(ca, EPI_STATS), "conv + statistics");
                BF_HIP(bf_launch_bn_finalize(partial, conv_grid, count, params + h->p_blocks + i * h->p_block_stride + conv_off(j) + 2304,
                                             state + bn_idx(i, j) * 32, state + bn_idx(i, j) * 32 + 16, d.bn_eps, d.bn_momentum, scale,
                                             scale + 16, w + L.bn_meaninv + bn_idx(i, j) * 32, stage1, s), "bn_finalize");
                // block i+1's conv_0 forms A(i+1) on load.  (The head kernel doing the same for the last block was tried: its register
                // count went past 256, one wave per SIMD, +105 us in the head for the 79 us of affine_add.)
                if (last && h3t && h->train_fused_fwd && i + 1 < N && nb >= 2) pending_affine = true;
                else if (last) BF_HIP(bf_launch_affine_add(A(i), C(i, j), scale, scale + 16, A(i + 1), npix, s), "affine_add");
                else BF_HIP(bf_launch_affine_act(C(i, j), scale, scale + 16, T(i, j + 1), relu, npix, s), "affine_act");
            } else if (last) {
                // no BatchNorm on the block's last convolution (one-convolution block, or use_bn off): linear, + skip
                ca.out = A(i + 1); ca.res = A(i);
                BF_HIP(conv(ca, EPI_RES), "conv + skip");
            } else {
                ca.out = T(i, j + 1);
                BF_HIP(conv(ca, relu ? EPI_RELU : 0), "conv + activation");
            }
        }
    }

    // ---- head forward + loss + head backward ---------------------------------------------------
    const double numel = (double)npix * d.out_channels;
    HeadTrainArgs ta;
    ta.feat = A(N); ta.wh = w + L.wh;
    ta.gt = gt; ta.pred = predictions; ta.dfeat = dA; ta.partial = partial; ta.dextra = nullptr;
    ta.B = B; ta.H = H; ta.W = W; ta.cout = d.out_channels; ta.denormalize = d.denormalize;
    ta.v_min = d.v_min; ta.v_max = d.v_max; ta.hinge = loss->hinge; ta.cutoff = loss->cutoff;
    ta.dscale = loss->mae_multiplier > 0.f ? (float)((double)loss->mae_multiplier * loss->depth_weight / numel) : 0.f;
    // Gradient scaling of the split-f16 backward.  dL/dprediction is O(1 / numel): 5e-8 at 32 x 256 x 256 x 3.  The data-
    // and weight-gradient kernels split every dy into two f16 numbers while they stage it; below 2^-14 the split keeps an
    // ABSOLUTE floor of 2^-25, so unscaled gradients lost most of their bits -- the larger the batch the more (bf_train_step
    // against itself on a batch that repeats two images: weight gradients 0.3 % off at 32 x 64 x 64, 12 % at 32 x 256 x 256;
    // tools/exp/train_batch_rep.py).  The head hands the blocks dfeat * S, S the power of two next to numel / (multiplier *
    // depth_weight); every backward operator is linear in dy, and the block / base gradients are multiplied by 1 / S (exact)
    // before the regularisers are added.  The exact-fp32 arithmetic runs with S = 1 as before.
    float grad_unscale = 1.0f;
    ta.dfeat_scale = 1.0f;
    if (h3t) {
        const double per = (loss->mae_multiplier > 0.f ? (double)loss->mae_multiplier : 1.0) * (loss->depth_weight > 0.f ? loss->depth_weight : 1.0) / numel;
        int ex = 0;
        (void)frexp(1.0 / per, &ex);
        ex = ex - 1 < 0 ? 0 : (ex - 1 > 40 ? 40 : ex - 1);
        ta.dfeat_scale = ldexpf(1.0f, ex);
        grad_unscale = ldexpf(1.0f, -ex);
    }
    const int hgrid = bf_head_train_grid(B, H, W);
    float* scal = nullptr;
    if (extra_terms) {
        // pass A: prediction + per-image sums; then the additive gradient of the RMSE / SSIM terms; pass B below adds it
        const int64_t pe = npix * d.out_channels;
        float* ex = w + L.extra;
        float* predbuf = predictions ? predictions : ex;
        float *dextra = ex + pe, *maps = ex + 2 * pe, *ssim_partial = ex + 5 * (int64_t)npix * 4;
        float* coef = ssim_partial + 4096;
        scal = coef + align_up(B, 64);
        ta.pred = predbuf;
        BF_HIP(bf_launch_head_train(ta, hgrid, s), "head_train (prediction pass)");
        BF_HIP(bf_launch_loss_extra(predbuf, gt, B, H, W, d.out_channels, partial, hgrid / B, loss->hinge, loss->cutoff,
                                    loss->mse_multiplier > 0.f ? loss->mse_multiplier : 0.f,
                                    loss->ssim_multiplier > 0.f ? loss->ssim_multiplier : 0.f, loss->depth_weight, 255.0f, maps,
                                    ssim_partial, coef, scal, dextra, s), "loss_extra");
        ta.pred = nullptr;
        ta.dextra = dextra;
    }
    BF_HIP(bf_launch_head_train(ta, hgrid, s), "head_train");
    hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, hgrid, hgrid / B, B, numel,
                       (double)H * W * d.out_channels, params + h->p_head0, params + h->p_head1, d.head_filters, d.out_channels,
                       grads + h->p_head0, grads + h->p_head1, losses, loss->mae_multiplier, loss->depth_weight);
    BF_HIP(hipGetLastError(), "head_finalize");
    if (extra_terms)
        BF_HIP(bf_launch_loss_extra_finalize(scal, B, H, W, d.out_channels, loss->mse_multiplier > 0.f ? loss->mse_multiplier : 0.f,
                                             loss->ssim_multiplier > 0.f ? loss->ssim_multiplier : 0.f, loss->depth_weight, losses, s),
               "loss_extra_finalize");

    // ---- backward through the blocks -----------------------------------------------------------
    // g = dL/d(block output) arrives in dA.  Per convolution j = nb-1 .. 0: [BatchNorm backward: g -> dc, dgamma] ; weight
    // gradient from (input of conv j, dc) ; data gradient through conv j -- for j >= 1 written over T(i,j) with the ReLU mask
    // of the activation that produced T(i,j), for j = 0 added to dA (the skip).
    int64_t n4 = npix * 4;
    int bgrid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    const bool fused_bwd = h3t && h->train_fused_bwd;
    // gradient buffers: dA (the head's output) and two spares; the fused kernel ping-pongs between them
    float* const gbuf[3] = {dA, ACT(N + 2 + 2 * (int64_t)N * (nb - 1)), ACT(N + 3 + 2 * (int64_t)N * (nb - 1))};
    // both convolutions of a block in one launch: [3,3] blocks, BatchNorm on the second convolution, ReLU between them
    const bool fused_bwd2 = fused_bwd && h->train_fused_bwd2 && nb == 2 && d.use_bn && relu;
    h->train_kernels = std::string("fwd: ") + (fwd_block ? "fwd_block_h3t_kernel" : h3t ? (h->train_fused_fwd && nb >= 2 && d.use_bn ? "conv3x3_h3_kernel<.., PRE> + conv3x3_h3_kernel" : "conv3x3_h3_kernel")
                                                   : "conv3x3_c16_kernel")
                       + "; bwd: " + (bwd_block ? "bwd_block_h3t_kernel" : fused_bwd2 ? "bwd2_h3_kernel" : fused_bwd ? "bwd3x3_h3_kernel<true, 8> + bwd3x3_h3_kernel<false, 36>"
                                                 : h3t ? "wgrad3x3_h3_kernel + conv3x3_h3_kernel" : "wgrad3x3_c16_kernel + conv3x3_c16_kernel");
    const int bwd_grid = bwd_block ? bf_bwd_block_h3t_grid(B, H, W) : fused_bwd2 ? bf_bwd2_h3_grid(B, H, W) : bf_bwd3x3_h3_grid_ex(B, H, W, h->train_bwd_dbuf);
    float* bwd_stats = partial + (int64_t)bwd_grid * 2304;
    // train_fold_finalize with the block backward kernel: launch i reads the sums launch i + 1 wrote and finalises them in its prologue,
    // so the sums go to two buffers in turn (both behind the weight-gradient slots inside `partial`)
    const bool bfold = bwd_block && h->train_fold_finalize != 0 && (int64_t)bwd_grid * 2304 + 2 * (int64_t)bwd_grid * 32 <= L.partial_floats;
    auto bstats = [&](int i) { return bfold ? bwd_stats + (int64_t)(i & 1) * bwd_grid * 32 : bwd_stats; };
    for (int i = N - 1; i >= 0; --i) {
        const float* wp = w + L.wpack + (int64_t)i * 2 * nb * BF_TRAIN_PACK_STRIDE;
        float* gblk = grads + h->p_blocks + i * h->p_block_stride;
        const float* g = dA;
        for (int j = nb - 1; j >= 0; --j) {
            const bool last = j == nb - 1, bn = j >= 1 && d.use_bn;
            const float* dy = g;
            if (bn) {
                // sum dy, sum dy*c: for the block's last BatchNorm they come from the data-gradient kernel of the block above
                // when it produced dA (split-f16 path: its epilogue accumulates them), else from the reduction kernel
                const bool fused_sums = h3t && last && i < N - 1;
                if (!fused_sums) BF_HIP(bf_launch_bn_bwd_reduce(g, C(i, j), partial, npix, bgrid, s), "bn_bwd_reduce");
                if (!(bfold && fused_sums))
                BF_HIP(bf_launch_bn_bwd_finalize(fused_sums && fused_bwd ? bstats(i + 1) : partial,
                                                 fused_sums ? (fused_bwd ? bwd_grid : conv_grid) : bgrid, count,
                                                 params + h->p_blocks + i * h->p_block_stride + conv_off(j) + 2304,
                                                 w + L.bn_meaninv + bn_idx(i, j) * 32, w + L.coef, gblk + conv_off(j) + 2304, stage1, s),
                       "bn_bwd_finalize");
                if (!fused_bwd) {
                    BF_HIP(bf_launch_bn_bwd_apply(g, C(i, j), w + L.coef, C(i, j), npix, s), "bn_bwd_apply");
                    dy = C(i, j);
                }
            }
            if (bwd_block) {
                // one row-streaming kernel for the whole block, T recomputed from A(i): dc = k1 g + k2 c + k3 ; T = act(conv_0 A) ;
                // dw1 = T^T dc ; dT = dgrad_1(dc) * (T > 0) ; dw0 = A^T dT ; dA' = dgrad_0(dT) + g [+ the sums of the BatchNorm in front]
                BwdBlockH3Args fa;
                memset(&fa, 0, sizeof(fa));
                fa.B = B; fa.H = H; fa.W = W; fa.act_relu = relu; fa.reverse = next_reverse();
                fa.a = A(i); fa.g = g; fa.c = C(i, 1); fa.coef = w + L.coef;
                if (bfold && i < N - 1) {                           // the sums came from launch i + 1: finalised in this launch's prologue
                    fa.fin_partial = bstats(i + 1); fa.fin_nblk = bwd_grid; fa.fin_count = count;
                    fa.fin_gamma = params + h->p_blocks + i * h->p_block_stride + conv_off(1) + 2304;
                    fa.fin_meaninv = w + L.bn_meaninv + bn_idx(i, 1) * 32;
                    fa.fin_dgamma = gblk + conv_off(1) + 2304;
                }
                fa.wfwd0 = wp; fa.wdg0 = wp + (int64_t)nb * BF_TRAIN_PACK_STRIDE; fa.wdg1 = wp + (int64_t)(nb + 1) * BF_TRAIN_PACK_STRIDE;
                fa.wpartial1 = w + L.wslots + ((int64_t)i * nb + 1) * L.wslot_floats;
                fa.wpartial0 = w + L.wslots + ((int64_t)i * nb + 0) * L.wslot_floats;
                fa.stats = bstats(i);
                if (i > 0) fa.bnc = C(i - 1, nb - 1);
                float* out = nullptr;
                for (int k = 0; k < 3 && !out; ++k)
                    if (gbuf[k] != g) out = gbuf[k];
                fa.out = out;
                // option "timing": one HIP-event pair around EVERY launch of this kernel (ring of BF_TIMING_RING pairs; bf_get_timing
                // returns their sum and count: bench.py's live roofline of the training step, measured inside real steps)
                const int64_t tslot = h->timing ? h->n_timed % BF_TIMING_RING : 0;
                if (h->timing) BF_HIP(hipEventRecord(h->ev[2 * tslot], s), "hipEventRecord");
                BF_HIP(bf_launch_bwd_block_h3t(fa, s), "bwd_block_h3t");
                if (h->timing) {
                    BF_HIP(hipEventRecord(h->ev[2 * tslot + 1], s), "hipEventRecord");
                    h->timed_launches = 1;
                    ++h->n_timed;
                }
                g = out;
                dA = out;
                break;                                              // both convolutions done
            }
            if (fused_bwd2) {
                // one kernel for the whole block: dc = k1 g + k2 c + k3 ; dw2 = T^T dc ; dT = dgrad2(dc) * (T > 0) (LDS only) ;
                // dw1 = A^T dT ; dA' = dgrad1(dT) + g [+ the sums of the BatchNorm in front]
                Bwd2H3Args fa;
                memset(&fa, 0, sizeof(fa));
                fa.B = B; fa.H = H; fa.W = W;
                fa.t = T(i, 1); fa.a = A(i); fa.dy = g; fa.c = C(i, 1); fa.coef = w + L.coef;
                fa.wpack2 = wp + (int64_t)(nb + 1) * BF_TRAIN_PACK_STRIDE; fa.wpack1 = wp + (int64_t)nb * BF_TRAIN_PACK_STRIDE;
                fa.wpartial2 = w + L.wslots + ((int64_t)i * nb + 1) * L.wslot_floats;
                fa.wpartial1 = w + L.wslots + ((int64_t)i * nb + 0) * L.wslot_floats;
                fa.stats = bwd_stats; fa.reverse = next_reverse();
                if (i > 0) fa.bnc = C(i - 1, nb - 1);
                float* out = nullptr;
                for (int k = 0; k < 3 && !out; ++k)
                    if (gbuf[k] != g) out = gbuf[k];
                fa.out = out;
                BF_HIP(bf_launch_bwd2_h3(fa, s), "bwd2_h3");
                g = out;
                dA = out;
                break;                                              // both convolutions done
            }
            if (fused_bwd) {
                // one kernel: [dc = k1 g + k2 c + k3] ; dw = x^T dc ; dx = dgrad(dc) [* mask | + skip]
                BwdH3Args fa;
                memset(&fa, 0, sizeof(fa));
                fa.B = B; fa.H = H; fa.W = W;
                fa.x = j == 0 ? A(i) : T(i, j);
                fa.g = g;
                if (bn) { fa.c = C(i, j); fa.coef = w + L.coef; }
                fa.wpack = wp + (int64_t)(nb + j) * BF_TRAIN_PACK_STRIDE;
                fa.wpartial = w + L.wslots + ((int64_t)i * nb + j) * L.wslot_floats; fa.stats = bwd_stats; fa.reverse = next_reverse(); fa.dbuf = h->train_bwd_dbuf;
                float* out = nullptr;
                for (int k = 0; k < 3 && !out; ++k)
                    if (gbuf[k] != g && gbuf[k] != dA) out = gbuf[k];
                int epi;
                if (j > 0) {
                    epi = relu ? EPI_MASK : 0;
                } else {
                    fa.res = dA;
                    if (g != dA) out = dA;                          // in place over the skip gradient (read at the same element only)
                    epi = EPI_RES;
                    if (d.use_bn && nb >= 2 && i > 0) { fa.bnc = C(i - 1, nb - 1); epi |= EPI_BNBWD; }
                }
                fa.out = out;
                BF_HIP(bf_launch_bwd3x3_h3(fa, epi, nullptr, s), "bwd3x3_h3");
                g = out;
                if (j == 0) dA = out;                               // (one-convolution block: another buffer than before)
                continue;
            }
            BF_HIP(wgrad(j == 0 ? A(i) : T(i, j), dy, gblk + conv_off(j)), "wgrad");
            ConvArgs ca;
            memset(&ca, 0, sizeof(ca));
            ca.B = B; ca.H = H; ca.W = W;
            ca.in = dy; ca.wpack = wp + (int64_t)(nb + j) * BF_TRAIN_PACK_STRIDE;
            if (j > 0) {
                ca.out = T(i, j); ca.mask = T(i, j);
                BF_HIP(conv(ca, relu ? EPI_MASK : 0), "dgrad");
                g = T(i, j);
            } else {
                ca.out = dA; ca.res = dA;
                if (h3t && d.use_bn && nb >= 2 && i > 0) {       // dA becomes dy of block i-1's last BatchNorm: its sums ride along
                    ca.bnc = C(i - 1, nb - 1); ca.stats = partial;
                    BF_HIP(conv(ca, EPI_RES | EPI_BNBWD), "dgrad + skip");
                } else {
                    BF_HIP(conv(ca, EPI_RES), "dgrad + skip");
                }
            }
        }
    }
    if (fused_bwd && N > 0)
        BF_HIP(bf_launch_reduce_wgrad_slots(w + L.wslots, L.wslot_floats, bwd_grid, grads + h->p_blocks, h->p_block_stride, N, nb, unit, s),
               "reduce_wgrad_slots");
    BF_HIP(bf_launch_base_wgrad(noisy, dA, partial, grads + h->p_base, B, H, W, d.in_channels, d.kernel_size, d.v_min, d.v_max, s),
           "base_wgrad");
    if (grad_unscale != 1.0f) {
        // base + block gradients (everything in front of the head's tensors) back to the loss's own scale
        hipLaunchKernelGGL(scale_range_kernel, dim3(64), dim3(256), 0, s, grads, h->p_head0, grad_unscale);
        BF_HIP(hipGetLastError(), "grad_unscale");
    }
    hipLaunchKernelGGL(regularizer_kernel, dim3(REG_GRID), dim3(1024), 0, s, params, grads, h->n_params, h->n_base, h->p_blocks,
                       h->p_block_stride, h->p_head0, d.reg_base, d.reg_block, d.reg_head, loss->regularization, stage1,
                       d.use_bn ? 2320 : 2304);
    hipLaunchKernelGGL(regularizer_finalize_kernel, dim3(1), dim3(64), 0, s, stage1, loss->regularization, losses);
    BF_HIP(hipGetLastError(), "regularizer");
    return BF_OK;
}

// ------------------------------------------------------------------------------------------
// Adam (keras 2.13, bfcnn/optimizer.py:190-206) with global_clipnorm (tf.clip_by_global_norm)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void grad_norm_kernel(const float* __restrict__ g, int64_t n, float grad_scale, float* scratch,
                                                         float* losses)
{
    __shared__ double red[1024];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double v = (double)g[i] * grad_scale;
        acc += v * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        scratch[0] = (float)sqrt(red[0]);
        if (losses) losses[BF_LOSS_GRAD_NORM] = scratch[0];
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float alpha, float beta_1, float beta_2,
                                                   float epsilon, float clip, float grad_scale, const float* __restrict__ scratch)
{
    float factor = grad_scale;
    if (clip > 0.f) {
        const float norm = scratch[0];
        factor *= clip / fmaxf(norm, clip);
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * factor;
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta_1);
        const float vi = v[i] + (gi * gi - v[i]) * (1.0f - beta_2);
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (mi * alpha) / (sqrtf(vi) + epsilon);
    }
}

// per-tensor clipping (keras clipnorm = tf.clip_by_norm on every gradient tensor, optimizer.py:165-169): one workgroup per
// tensor sums its squares in a fixed order; factor = c / max(norm, c)
__global__ __launch_bounds__(256) void tensor_clip_factor_kernel(const float* __restrict__ g, const int64_t* __restrict__ offs,
                                                                 float grad_scale, float clipnorm, float* __restrict__ factor)
{
    __shared__ double red[256];
    const int64_t a = offs[blockIdx.x], b = offs[blockIdx.x + 1];
    double acc = 0.0;
    for (int64_t i = a + threadIdx.x; i < b; i += 256) {
        const double x = (double)g[i] * grad_scale;
        acc += x * x;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) factor[blockIdx.x] = clipnorm / fmaxf((float)sqrt(red[0]), clipnorm);
}

__global__ __launch_bounds__(256) void adam_tensor_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const int64_t* __restrict__ offs,
                                                          const float* __restrict__ factor, float clipvalue, float alpha, float beta_1,
                                                          float beta_2, float epsilon, float grad_scale)
{
    const int64_t a = offs[blockIdx.x], b = offs[blockIdx.x + 1];
    const float f = grad_scale * (factor ? factor[blockIdx.x] : 1.f);
    for (int64_t i = a + threadIdx.x; i < b; i += 256) {
        float gi = g[i] * f;
        if (clipvalue > 0.f) gi = fminf(fmaxf(gi, -clipvalue), clipvalue);
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta_1);
        const float vi = v[i] + (gi * gi - v[i]) * (1.0f - beta_2);
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (mi * alpha) / (sqrtf(vi) + epsilon);
    }
}

// bf_adam_step with keras' other two clipping modes.  Precedence as keras 2.13 (_clip_gradients): clipnorm (per tensor), else
// global_clipnorm, else clipvalue.  tensor_offsets = device int64[n_tensors + 1] (offsets of the trainable tensors in the flat
// vector, last = n_params), tensor_scratch = device float[n_tensors]; both only read when clipnorm or clipvalue is on.
static int adam_core(bf_handle h, int64_t n, float* params, const float* grads, float* m, float* v, int64_t iterations, float lr,
                     float beta_1, float beta_2, float epsilon, float global_clipnorm, float grad_scale, float* losses, float* scratch,
                     void* stream)
{
    if (!params || !grads || !m || !v || !scratch || n <= 0) return fail(h, BF_EINVAL, "bf_adam_step: NULL argument");
    if (iterations < 0) return fail(h, BF_EINVAL, "iterations must be >= 0");
    hipStream_t s = (hipStream_t)stream;
    if (global_clipnorm > 0.f || losses) {
        hipLaunchKernelGGL(grad_norm_kernel, dim3(1), dim3(1024), 0, s, grads, n, grad_scale, scratch, losses);
        BF_HIP(hipGetLastError(), "grad_norm");
    }
    const double t = (double)iterations + 1.0;
    const double alpha = (double)lr * sqrt(1.0 - pow((double)beta_2, t)) / (1.0 - pow((double)beta_1, t));
    const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, params, grads, m, v, n, (float)alpha, beta_1, beta_2, epsilon,
                       global_clipnorm, grad_scale, scratch);
    BF_HIP(hipGetLastError(), "adam");
    return BF_OK;
}

static int adam_ex_core(bf_handle h, int64_t n, float* params, const float* grads, float* m, float* v, int64_t iterations, float lr,
                        float beta_1, float beta_2, float epsilon, float global_clipnorm, float clipnorm, float clipvalue,
                        const int64_t* tensor_offsets, int n_tensors, float* tensor_scratch, float grad_scale, float* losses,
                        float* scratch, void* stream)
{
    const bool local = clipnorm > 0.f, by_value = !local && !(global_clipnorm > 0.f) && clipvalue > 0.f;
    if (!local && !by_value)
        return adam_core(h, n, params, grads, m, v, iterations, lr, beta_1, beta_2, epsilon, global_clipnorm, grad_scale, losses, scratch,
                         stream);
    if (!params || !grads || !m || !v || !scratch || !tensor_offsets || n_tensors <= 0 || (local && !tensor_scratch))
        return fail(h, BF_EINVAL, "bf_adam_step_ex: NULL argument");
    if (iterations < 0) return fail(h, BF_EINVAL, "iterations must be >= 0");
    hipStream_t s = (hipStream_t)stream;
    if (losses) {
        hipLaunchKernelGGL(grad_norm_kernel, dim3(1), dim3(1024), 0, s, grads, n, grad_scale, scratch, losses);
        BF_HIP(hipGetLastError(), "grad_norm");
    }
    if (local) {
        hipLaunchKernelGGL(tensor_clip_factor_kernel, dim3(n_tensors), dim3(256), 0, s, grads, tensor_offsets, grad_scale, clipnorm,
                           tensor_scratch);
        BF_HIP(hipGetLastError(), "tensor_clip_factor");
    }
    const double t = (double)iterations + 1.0;
    const double alpha = (double)lr * sqrt(1.0 - pow((double)beta_2, t)) / (1.0 - pow((double)beta_1, t));
    hipLaunchKernelGGL(adam_tensor_kernel, dim3(n_tensors), dim3(256), 0, s, params, grads, m, v, tensor_offsets,
                       local ? tensor_scratch : (const float*)nullptr, by_value ? clipvalue : 0.f, (float)alpha, beta_1, beta_2, epsilon,
                       grad_scale);
    BF_HIP(hipGetLastError(), "adam_tensor");
    return BF_OK;
}

extern "C" int bf_adam_step_ex(bf_handle h, float* params, const float* grads, float* m, float* v, int64_t iterations, float lr,
                               float beta_1, float beta_2, float epsilon, float global_clipnorm, float clipnorm, float clipvalue,
                               const int64_t* tensor_offsets, int n_tensors, float* tensor_scratch, float grad_scale, float* losses,
                               float* scratch, void* stream)
{
    if (!h) return BF_EINVAL;
    return adam_ex_core(h, h->n_params, params, grads, m, v, iterations, lr, beta_1, beta_2, epsilon, global_clipnorm, clipnorm, clipvalue,
                        tensor_offsets, n_tensors, tensor_scratch, grad_scale, losses, scratch, stream);
}

extern "C" int bf_adam_step(bf_handle h, float* params, const float* grads, float* m, float* v, int64_t iterations, float lr,
                            float beta_1, float beta_2, float epsilon, float global_clipnorm, float grad_scale, float* losses,
                            float* scratch, void* stream)
{
    if (!h) return BF_EINVAL;
    return adam_core(h, h->n_params, params, grads, m, v, iterations, lr, beta_1, beta_2, epsilon, global_clipnorm, grad_scale, losses,
                     scratch, stream);
}

// the same update for a flat parameter vector that no bf_handle describes (models assembled from the operator library:
// unet_laplacian); n = number of parameters, everything else as bf_adam_step_ex
extern "C" int bf_op_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t iterations, float lr, float beta_1,
                               float beta_2, float epsilon, float global_clipnorm, float clipnorm, float clipvalue,
                               const int64_t* tensor_offsets, int n_tensors, float* tensor_scratch, float grad_scale, float* losses,
                               float* scratch, void* stream)
{
    return adam_ex_core(nullptr, n, params, grads, m, v, iterations, lr, beta_1, beta_2, epsilon, global_clipnorm, clipnorm, clipvalue,
                        tensor_offsets, n_tensors, tensor_scratch, grad_scale, losses, scratch, stream);
}

// ------------------------------------------------------------------------------------------
// diagnostics used by tests/ (single-kernel entry points; not part of the drop-in surface)
// ------------------------------------------------------------------------------------------
extern "C" int bf_debug_conv3x3(const float* in, const float* w_hwio, float* out, const float* scale, const float* shift,
                                const float* res, const float* mask, float* stats, float* wpack_scratch, int B, int H, int W,
                                int epi, int transpose_flip, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (bf_launch_pack_conv(w_hwio, wpack_scratch, transpose_flip, s) != hipSuccess) return BF_EHIP;
    ConvArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.in = in; ca.out = out; ca.wpack = wpack_scratch; ca.scale = scale; ca.shift = shift; ca.res = res; ca.mask = mask;
    ca.stats = stats; ca.B = B; ca.H = H; ca.W = W;
    return bf_launch_conv3x3_c16(ca, epi, s) == hipSuccess ? BF_OK : BF_EHIP;
}

static unsigned long long* g_fused_dbg = nullptr;
// diagnostic builds (tools/ablate.sh 8): device buffer of 512*8*8 u64 that receives per-wave phase cycle sums
extern "C" int bf_debug_set_fused_dbg(void* buf) { g_fused_dbg = (unsigned long long*)buf; return BF_OK; }

extern "C" int bf_debug_conv3x3_grid(int B, int H, int W) { return bf_conv3x3_c16_grid(B, H, W); }

extern "C" int bf_debug_fused_block(const float* in, const float* w1_hwio, const float* w2_hwio, const float* scale,
                                    const float* shift, float* out, float* wpack_scratch, int B, int H, int W, int act1_relu,
                                    void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (bf_launch_pack_conv(w1_hwio, wpack_scratch, 0, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_pack_conv(w2_hwio, wpack_scratch + BF_WPACK_FLOATS, 0, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_zero(wpack_scratch + 2 * BF_WPACK_FLOATS, 64, s) != hipSuccess) return BF_EHIP;
    FusedBlockArgs fa;
    fa.zeros = wpack_scratch + 2 * BF_WPACK_FLOATS;
    fa.in = in; fa.out = out; fa.w1pack = wpack_scratch; fa.w2pack = wpack_scratch + BF_WPACK_FLOATS; fa.scale = scale;
    fa.shift = shift; fa.B = B; fa.H = H; fa.W = W; fa.tiles_x = fa.tiles_y = fa.ntiles = 0; fa.act1_relu = act1_relu;
    fa.dbg = g_fused_dbg;
    return bf_launch_fused_block(fa, s) == hipSuccess ? BF_OK : BF_EHIP;
}

// split-f16 fused block on fp32 NHWC tensors: convert in, run, convert out.  scratch: float buffer of at least
// 2 * B*H*W*16 + BF_H3_BLOCK_FLOATS + 4608 + 32 + 64 + 128 floats (two split-planar activations, packed weights,
// the two HWIO kernels + gamma-free BN stand-in, zero line, dump line).
extern "C" int64_t bf_debug_fused_block_h3_scratch_floats(int B, int H, int W)
{
    return 2 * (int64_t)B * H * W * 16 + BF_H3_BLOCK_FLOATS + 4608 + 16 + 32 + 64 + 256;
}

extern "C" int bf_debug_fused_block_h3(const float* in, const float* w1_hwio, const float* w2_hwio, const float* scale,
                                       const float* shift, float* out, float* scratch, int B, int H, int W, int act1_relu,
                                       void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    const int64_t act = (int64_t)B * H * W * 16;
    float* xa = scratch;
    float* ya = scratch + act;
    float* pk = ya + act;                          // BF_H3_BLOCK_FLOATS
    float* params = pk + BF_H3_BLOCK_FLOATS;       // [w1 2304][w2 2304][gamma 16]
    float* state = params + 4608 + 16;             // [mean 16][var 16]
    float* zeros = state + 32;                     // 64
    float* dump = zeros + 64;                      // 256
    // the caller's scale / shift stand in for the folded BN (ext_scale / ext_shift of the pack kernel)
    if (hipMemcpyAsync(params, w1_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (hipMemcpyAsync(params + 2304, w2_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_zero(zeros, 64, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_pack_h3(params, state, 0, 4608 + 16, pk, BF_H3_BLOCK_FLOATS, 1, 0, 0.f, scale, shift, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_h3_from_f32(in, xa, B, H, W, s) != hipSuccess) return BF_EHIP;
    FusedH3Args fa;
    fa.in = xa; fa.out = ya; fa.w1 = pk; fa.w2 = pk + BF_H3_WPACK_FLOATS; fa.aux = pk + 2 * BF_H3_WPACK_FLOATS;
    fa.w1r = fa.aux + 64; fa.w2r = fa.aux + 64 + BF_H3R_WPACK_FLOATS;
    fa.B = B; fa.H = H; fa.W = W; fa.tiles_x = fa.tiles_y = fa.ntiles = 0; fa.rows_per_tile = 0; fa.variant = -1; fa.reverse_tiles = 0; fa.act1_relu = act1_relu;
    fa.zeros = zeros; fa.dump = dump; fa.dbg = g_fused_dbg;
    fa.head_wh = nullptr; fa.head_out = nullptr; fa.head_u8 = 0; fa.Ho = fa.Wo = 0; fa.denormalize = 0; fa.v_min = fa.v_max = 0.f;
    fa.status = nullptr;
    if (bf_launch_fused_block_h3(fa, s) != hipSuccess) return BF_EHIP;
    return bf_launch_h3_to_f32(ya, out, B, H, W, s) == hipSuccess ? BF_OK : BF_EHIP;
}

// TWO split-f16 fused blocks in one launch (fused_h3w.hip) on fp32 NHWC tensors: convert in, run, convert out.
// w_hwio = [4][3][3][16][16] (conv1a, conv2a, conv1b, conv2b), scale / shift = [2][16] (block a, b).
extern "C" int64_t bf_debug_fused_block2_h3_scratch_floats(int B, int H, int W)
{
    return 2 * (int64_t)B * H * W * 16 + 2 * (int64_t)BF_H3_BLOCK_FLOATS + 2 * (4608 + 16) + 32 + 64;
}

extern "C" int bf_debug_fused_block2_h3(const float* in, const float* w_hwio, const float* scale, const float* shift, float* out,
                                        float* scratch, int B, int H, int W, int act1_relu, int reverse, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!in || !w_hwio || !scale || !shift || !out || !scratch || B <= 0 || H <= 0 || W <= 0) return BF_EINVAL;
    const int64_t act = (int64_t)B * H * W * 16;
    float* xa = scratch;
    float* ya = scratch + act;
    float* pk = ya + act;                              // 2 x BF_H3_BLOCK_FLOATS
    float* params = pk + 2 * BF_H3_BLOCK_FLOATS;       // 2 x [w1 2304][w2 2304][gamma 16]
    float* state = params + 2 * (4608 + 16);           // [mean 16][var 16] (unused: the caller's scale / shift stand in)
    float* zeros = state + 32;                         // 64
    if (hipMemcpyAsync(params, w_hwio, 4608 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (hipMemcpyAsync(params + 4608 + 16, w_hwio + 4608, 4608 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_zero(zeros, 64, s) != hipSuccess) return BF_EHIP;
    for (int b = 0; b < 2; ++b)
        if (bf_launch_pack_h3(params + b * (4608 + 16), state, 0, 4608 + 16, pk + b * BF_H3_BLOCK_FLOATS, BF_H3_BLOCK_FLOATS, 1, 0,
                              0.f, scale + 16 * b, shift + 16 * b, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_h3_from_f32(in, xa, B, H, W, s) != hipSuccess) return BF_EHIP;
    FusedH3WArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.in = xa; fa.out = ya;
    for (int b = 0; b < 2; ++b) {
        const float* aux = pk + b * BF_H3_BLOCK_FLOATS + 2 * BF_H3_WPACK_FLOATS;
        fa.aux[b] = aux; fa.w1r[b] = aux + 64; fa.w2r[b] = aux + 64 + BF_H3R_WPACK_FLOATS;
    }
    fa.B = B; fa.H = H; fa.W = W; fa.reverse_tiles = reverse ? 1 : 0; fa.act1_relu = act1_relu;
    fa.zeros = zeros; fa.dbg = g_fused_dbg;
    if (bf_launch_fused_block2_h3w(fa, s) != hipSuccess) return BF_EHIP;
    return bf_launch_h3_to_f32(ya, out, B, H, W, s) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_debug_set_h3_variant(int variant)
{
    bf_set_h3_variant(variant);
    return BF_OK;
}

// single split-f16 3x3 convolution on fp32 NHWC (the training convolution); scratch = 4 * BF_H3_TRAIN_PACK floats + 2304
extern "C" int64_t bf_debug_conv3x3_h3_scratch_floats(void) { return 4 * (int64_t)BF_H3_TRAIN_PACK_FLOATS + 2 * 2304 + 16; }
extern "C" int bf_debug_conv3x3_h3(const float* in, const float* w_hwio, float* out, const float* res, const float* mask,
                                   float* stats, float* scratch, int B, int H, int W, int epi, int transpose_flip, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    float* params = scratch + 4 * BF_H3_TRAIN_PACK_FLOATS;        // [w 2304][unused 2304][gamma 16]: one "layer"
    if (hipMemcpyAsync(params, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (hipMemcpyAsync(params + 2304, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_pack_h3_train(params, 0, 4608 + 16, scratch, 1, 2, 2320, s) != hipSuccess) return BF_EHIP;
    ConvArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.in = in; ca.out = out; ca.wpack = scratch + (transpose_flip ? 2 : 0) * BF_H3_TRAIN_PACK_FLOATS;
    ca.res = res; ca.mask = mask; ca.stats = stats; ca.B = B; ca.H = H; ca.W = W;
    return bf_launch_conv3x3_h3(ca, epi, s) == hipSuccess ? BF_OK : BF_EHIP;
}

// conv3x3_h3 with "affine + add on load": y = in + pre_scale * pre_c + pre_shift -> pre_out ; out = [relu] conv(y)
extern "C" int bf_debug_conv3x3_h3_pre(const float* in, const float* pre_c, const float* pre_scale, const float* pre_shift,
                                       float* pre_out, const float* w_hwio, float* out, float* scratch, int B, int H, int W, int relu,
                                       int reverse, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    float* params = scratch + 4 * BF_H3_TRAIN_PACK_FLOATS;
    if (hipMemcpyAsync(params, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (hipMemcpyAsync(params + 2304, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_pack_h3_train(params, 0, 4608 + 16, scratch, 1, 2, 2320, s) != hipSuccess) return BF_EHIP;
    ConvArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.in = in; ca.out = out; ca.wpack = scratch; ca.B = B; ca.H = H; ca.W = W; ca.reverse = reverse;
    ca.pre_c = pre_c; ca.pre_scale = pre_scale; ca.pre_shift = pre_shift; ca.pre_out = pre_out;
    return bf_launch_conv3x3_h3(ca, relu ? EPI_RELU : 0, s) == hipSuccess ? BF_OK : BF_EHIP;
}

// the training-mode forward of one [3,3] block in one kernel (train_fwd_h3t.hip): a_out = x + pre_scale * pre_c + pre_shift (pre_c
// given), t_out = [relu] conv_0(a) (t_out given), c_out = conv_1(t), stats[32] = per-channel sum | sum of squares of c_out.
// scratch: bf_debug_fwd_block_h3t_scratch_floats(B, H, W) floats
extern "C" int64_t bf_debug_fwd_block_h3t_scratch_floats(int B, int H, int W)
{
    return 4 * (int64_t)BF_H3_TRAIN_PACK_FLOATS + 2 * 2304 + 16 + (int64_t)bf_fwd_block_h3t_grid(B, H, W) * 32;
}
extern "C" int bf_debug_fwd_block_h3t(const float* x, const float* pre_c, const float* pre_scale, const float* pre_shift,
                                      const float* w0_hwio, const float* w1_hwio, float* a_out, float* t_out, float* c_out, float* stats,
                                      float* scratch, int B, int H, int W, int relu, int reverse, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!bf_fwd_block_h3t_supports(H, W)) return BF_EUNSUPPORTED;
    float* params = scratch + 4 * BF_H3_TRAIN_PACK_FLOATS;
    if (hipMemcpyAsync(params, w0_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (hipMemcpyAsync(params + 2304, w1_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_pack_h3_train(params, 0, 4608 + 16, scratch, 1, 2, 2320, s) != hipSuccess) return BF_EHIP;
    float* partial = params + 2 * 2304 + 16;
    FwdBlockH3Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.x = x; fa.pre_c = pre_c; fa.pre_scale = pre_scale; fa.pre_shift = pre_shift; fa.a_out = a_out; fa.t_out = t_out; fa.c_out = c_out;
    fa.wpack0 = scratch; fa.wpack1 = scratch + BF_H3_TRAIN_PACK_FLOATS; fa.stats = partial;
    fa.B = B; fa.H = H; fa.W = W; fa.reverse = reverse; fa.act_relu = relu;
    if (bf_launch_fwd_block_h3t(fa, s) != hipSuccess) return BF_EHIP;
    return bf_launch_reduce_partials(partial, bf_fwd_block_h3t_grid(B, H, W), 32, stats, 1.0f, s) == hipSuccess ? BF_OK : BF_EHIP;
}

// the backward of one [3,3] block in one kernel with T recomputed (train_bwd_h3t.hip): dc = k1 g + k2 c + k3 (coef = k1 | k2 | k3),
// T = [relu] conv_0(a), dw1 = T^T dc, dT = dgrad_1(dc) [* (T > 0)], dw0 = a^T dT, out = dgrad_0(dT) + g, stats[32] = sums of out |
// out * bnc (bnc given).  scratch: bf_debug_bwd_block_h3t_scratch_floats(B, H, W) floats
extern "C" int64_t bf_debug_bwd_block_h3t_scratch_floats(int B, int H, int W)
{
    return 4 * (int64_t)BF_H3_TRAIN_PACK_FLOATS + 2 * 2304 + 16 + (int64_t)bf_bwd_block_h3t_grid(B, H, W) * (2 * 2304 + 32);
}
extern "C" int bf_debug_bwd_block_h3t(const float* a_in, const float* g, const float* c, const float* coef, const float* w0_hwio,
                                      const float* w1_hwio, const float* bnc, float* out, float* dw1, float* dw0, float* stats,
                                      float* scratch, int B, int H, int W, int relu, int reverse, void* stream)
{
    // reverse: bit 0 = walk the bands bottom-up; bit 1 = the KERNEL ALONE (weights packed by an earlier call with the same scratch, no
    // reduction of the partials: bench.py's live timing of the launch)
    hipStream_t s = (hipStream_t)stream;
    if (!bf_bwd_block_h3t_supports(H, W)) return BF_EUNSUPPORTED;
    const bool alone = (reverse & 2) != 0;
    reverse &= 1;
    float* params = scratch + 4 * BF_H3_TRAIN_PACK_FLOATS;
    if (!alone) {
        if (hipMemcpyAsync(params, w0_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
        if (hipMemcpyAsync(params + 2304, w1_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
        if (bf_launch_pack_h3_train(params, 0, 4608 + 16, scratch, 1, 2, 2320, s) != hipSuccess) return BF_EHIP;
    }
    const int grid = bf_bwd_block_h3t_grid(B, H, W);
    float* wp1 = params + 2 * 2304 + 16;
    float* wp0 = wp1 + (int64_t)grid * 2304;
    float* st = wp0 + (int64_t)grid * 2304;
    BwdBlockH3Args fa;
    memset(&fa, 0, sizeof(fa));
    fa.a = a_in; fa.g = g; fa.c = c; fa.coef = coef; fa.bnc = bnc; fa.out = out;
    fa.wfwd0 = scratch; fa.wdg0 = scratch + 2 * BF_H3_TRAIN_PACK_FLOATS; fa.wdg1 = scratch + 3 * BF_H3_TRAIN_PACK_FLOATS;
    fa.wpartial1 = wp1; fa.wpartial0 = wp0; fa.stats = st;
    fa.B = B; fa.H = H; fa.W = W; fa.reverse = reverse; fa.act_relu = relu; fa.dbg = g_fused_dbg;
    if (bf_launch_bwd_block_h3t(fa, s) != hipSuccess) return BF_EHIP;
    if (alone) return BF_OK;
    if (bf_launch_reduce_partials(wp1, grid, 2304, dw1, 1.0f, s) != hipSuccess) return BF_EHIP;
    if (bf_launch_reduce_partials(wp0, grid, 2304, dw0, 1.0f, s) != hipSuccess) return BF_EHIP;
    if (bnc && stats && bf_launch_reduce_partials(st, grid, 32, stats, 1.0f, s) != hipSuccess) return BF_EHIP;
    return BF_OK;
}

// the fused backward kernel of one convolution (train_bwd_h3.hip): dw = x^T g', dx = dgrad(g') [* (x > 0) | + res], with
// g' = k1 g + k2 c + k3 when coef is given; stats (EPI_BNBWD): [grid][32] partials of (sum dx, sum dx * bnc).
// scratch: bf_debug_bwd3x3_h3_scratch_floats(B, H, W) floats
extern "C" int64_t bf_debug_bwd3x3_h3_scratch_floats(int B, int H, int W)
{
    return bf_debug_conv3x3_h3_scratch_floats() + (int64_t)bf_bwd3x3_h3_grid(B, H, W) * (2304 + 32);
}
extern "C" int bf_debug_bwd3x3_h3_grid(int B, int H, int W) { return bf_bwd3x3_h3_grid(B, H, W); }
extern "C" int bf_debug_bwd3x3_h3_grid_ex(int B, int H, int W, int dbuf) { return bf_bwd3x3_h3_grid_ex(B, H, W, dbuf); }
extern "C" int bf_debug_bwd3x3_h3(const float* x, const float* g, const float* c, const float* coef, const float* w_hwio, float* out,
                                  const float* res, const float* bnc, float* dw, float* stats, float* scratch, int B, int H, int W,
                                  int epi, int reverse, int repack, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (repack) {
        float* params = scratch + 4 * BF_H3_TRAIN_PACK_FLOATS;
        if (hipMemcpyAsync(params, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
        if (hipMemcpyAsync(params + 2304, w_hwio, 2304 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EHIP;
        if (bf_launch_pack_h3_train(params, 0, 4608 + 16, scratch, 1, 2, 2320, s) != hipSuccess) return BF_EHIP;
    }
    float* partial = scratch + bf_debug_conv3x3_h3_scratch_floats();
    BwdH3Args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.g = g; a.c = c; a.coef = coef; a.wpack = scratch + 2 * BF_H3_TRAIN_PACK_FLOATS; a.out = out; a.res = res; a.bnc = bnc;
    a.wpartial = partial; a.stats = partial + (int64_t)bf_bwd3x3_h3_grid(B, H, W) * 2304;
    a.B = B; a.H = H; a.W = W; a.reverse = reverse & 1; a.dbuf = (reverse >> 1) & 1;       // reverse: bit 0 walk direction, bit 1 kernel form
    if (bf_launch_bwd3x3_h3(a, epi, dw, s) != hipSuccess) return BF_EHIP;
    if (stats && (epi & EPI_BNBWD) &&
        hipMemcpyAsync(stats, a.stats, (size_t)bf_bwd3x3_h3_grid_ex(B, H, W, a.dbuf) * 32 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return BF_EHIP;
    return BF_OK;
}

extern "C" int64_t bf_debug_wgrad_partial_floats(int B, int H, int W) { return (int64_t)bf_wgrad_grid(B, H, W) * 2304; }

extern "C" int bf_debug_wgrad3x3_h3(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W, void* stream)
{
    return bf_launch_wgrad3x3_h3(x, dy, partial, dw, B, H, W, (hipStream_t)stream) == hipSuccess ? BF_OK : BF_EHIP;
}

extern "C" int bf_debug_wgrad3x3(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W, void* stream)
{
    return bf_launch_wgrad3x3_c16(x, dy, partial, dw, B, H, W, (hipStream_t)stream) == hipSuccess ? BF_OK : BF_EHIP;
}

// raw MFMA layout probe: D = A(16x4) * B(4x16) with A[m][k] = a_in[m*4+k], B[k][n] = b_in[k*16+n]
__global__ void mfma_probe_kernel(const float* a_in, const float* b_in, float* d_out)
{
    const int l = threadIdx.x;
    const float a = a_in[(l & 15) * 4 + (l >> 4)];
    const float b = b_in[(l >> 4) * 16 + (l & 15)];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) d_out[((l >> 4) * 4 + j) * 16 + (l & 15)] = acc[j];
}

extern "C" int bf_debug_mfma_probe(const float* a, const float* b, float* d, void* stream)
{
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EHIP;
}
