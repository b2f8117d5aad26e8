"""CPU oracle for the `unet_laplacian` backbone + per-scale denoiser heads (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product path (blind_image_denoising_amd/ fails loudly without its HIP library).

A NumPy fp64 restatement of what the reference builds for `configs/unet_laplacian_v5.json`, inference mode
(Dropout / StochasticDepth / attention dropout are identities):
  bfcnn/backbone_unet_laplacian.py:35-615 (graph), bfcnn/custom_layers.py:838-1022 (ConvNextBlock),
  :218-322 (ChannelLearnableMultiplier), :1205-1379 (ConvolutionalSelfAttention), :133-168 (GaussianFilter),
  bfcnn/upsampling.py:37-116, bfcnn/downsampling.py:37-72, bfcnn/model.py:58-162 (hydra with one head per scale),
  :251-359 (denoiser head), bfcnn/utilities.py:132-224 (conv2d_wrapper: conv -> activation layer).

PINNED BY THE REFERENCE'S OWN ARTIFACTS for the graph revision of its one trained network
(bfcnn/pretrained/unet_laplacian_v5.6: model_hydra.keras + the same network exported as denoiser_model.tflite), with
no TensorFlow involved (tests/test_unet_pretrained.py, fixture tests/golden/unet_v56.npz written by
tests/golden/make_unet_v56_fixture.py):
  * structure: the operator list of the exported graph (tools/exp/tflite_graph.py) gives the dataflow this file follows
    for that revision (spec fields mlp_activation .. upsample_linear);
  * known-answer constants: the GaussianFilter taps, LayerNorm epsilon, head constants, and all 74 weight tensors of the
    exported graph (float constants exactly, int8 kernels through their per-channel scales max|w|/127, which also fixes
    ChannelLearnableMultiplier = tanh(relu(1 + w)) because the converter folded it into conv_3);
  * behaviour: the reference's acceptance test for a trained network (tests/bfcnn/test_pretrained.py: denoised beats
    noisy in PSNR, SSIM and MAE on its KITTI frames) holds for the oracle and for the HIP path with the real weights.
What stays UNPINNED: bit-level TensorFlow numerics (no golden activations exist), and the pieces only the snapshot
builder has (16x16 resized attention, AveragePooling split, in-line output norms, attention gates): those follow
SURVEY.md appendix A; convolutions / resize / attention are cross-checked against torch-CPU fp64 in
tests/test_unet_oracle.py.

One deliberate deviation: ConvolutionalSelfAttention hands the STRING "leaky_relu" to keras.layers.Conv2D
(custom_layers.py:1272-1282 with backbone_unet_laplacian.py:330); Keras 2.13 has no activation of that name
(it raises), Keras >= 2.15 / 3 resolve it to leaky_relu with negative_slope 0.2.  The oracle uses 0.2 and keeps
it a spec field (`attention_alpha`).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import bfcnn_oracle as O

F64 = np.float64
LN_EPS = 1e-3                     # DEFAULT_LN_EPSILON (constants.py:10) == keras LayerNormalization default


def leaky(x, alpha):
    return np.where(x > 0, x, alpha * x)


def act(x, name: str):
    name = (name or "linear").lower().strip()
    if name == "gelu":            # keras "gelu": exact erf form (approximate=False)
        from math import sqrt
        from scipy.special import erf
        return 0.5 * x * (1.0 + erf(x / sqrt(2.0)))
    return O.activation_fwd(x, name)


def conv2d_transpose_same(x: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """keras Conv2DTranspose(kernel_size k, strides s, padding="same", use_bias=False) (upsample_type "conv2d_transpose",
    bfcnn/upsampling.py:37-48 -> utilities.py:200-202); x [B,H,W,cin], w [k,k,cout,cin] (keras kernel layout), out
    [B,H*s,W*s,cout].  The transpose of the stride-s SAME convolution from [H*s,W*s] to [H,W]: every input pixel scatters
    its k x k patch to out[iy*s + i - pb, ix*s + j - pb], pb = max(k - s, 0) // 2 (that convolution's leading pad)."""
    B, H, W, cin = x.shape
    k, s = w.shape[0], int(stride)
    cout = w.shape[2]
    pb = max(k - s, 0) // 2
    full = np.zeros((B, (H - 1) * s + k + s, (W - 1) * s + k + s, cout), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            full[:, i:i + (H - 1) * s + 1:s, j:j + (W - 1) * s + 1:s, :] += np.einsum("bhwc,oc->bhwo", x, w[i, j])
    return full[:, pb:pb + H * s, pb:pb + W * s, :]


def depthwise_same(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """keras DepthwiseConv2D(depth_multiplier=1, padding="same", use_bias=False); w [kh,kw,C,1]
    (custom_layers.py:936; zero padding, cross-correlation)."""
    B, H, W, C = x.shape
    kh, kw = w.shape[:2]
    _, pt, pb = O.same_pads(H, kh, 1)
    _, pl, pr = O.same_pads(W, kw, 1)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros_like(x)
    for i in range(kh):
        for j in range(kw):
            y += xp[:, i:i + H, j:j + W, :] * w[i, j, :, 0]
    return y


def layer_norm(x: np.ndarray, gamma: np.ndarray, eps: float = LN_EPS) -> np.ndarray:
    """keras LayerNormalization(axis=-1, center=False, scale=True): biased variance over the channels."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma


def channel_multiplier(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """ChannelLearnableMultiplier: tanh(relu(1 + w)) * x (custom_layers.py:304-306)."""
    return np.tanh(np.maximum(1.0 + w, 0.0)) * x


def gaussian_kernel_3(kernel_size=(3, 3)) -> np.ndarray:
    """GaussianFilter's fixed kernel: depthwise_gaussian_kernel with nsig = (k-1)/2, cast to fp32
    (custom_layers.py:146, 151-158; utilities.py:272-321)."""
    ax = [np.linspace(-(k - 1) / 2, (k - 1) / 2, k, dtype=np.float64) for k in kernel_size]
    gx, gy = np.meshgrid(ax[0], ax[1])
    g = np.exp(-(gx * gx + gy * gy) / 2.0)
    return (g / g.sum()).astype(np.float32).astype(np.float64)


def max_pool_2x2_same(x: np.ndarray) -> np.ndarray:
    """keras MaxPooling2D(pool_size 2, strides 2, padding="same") (downsampling.py:56-58): padded taps are ignored."""
    B, H, W, C = x.shape
    xp = np.full((B, H + H % 2, W + W % 2, C), -np.inf, dtype=x.dtype)
    xp[:, :H, :W] = x
    return np.maximum(np.maximum(xp[:, 0::2, 0::2], xp[:, 1::2, 0::2]), np.maximum(xp[:, 0::2, 1::2], xp[:, 1::2, 1::2]))


def resize_bilinear(x: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """tf.image.resize(method=BILINEAR, antialias=False): half-pixel centres, src = (dst + 0.5) * in/out - 0.5,
    lower = max(floor(src), 0), upper = min(ceil(src), in - 1), weight = src - floor(src)
    (custom_layers.py:1328-1334, 1351-1357)."""
    def axis_w(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        fl = np.floor(src)
        lo = np.maximum(fl, 0).astype(np.int64)
        hi = np.minimum(np.ceil(src), n_in - 1).astype(np.int64)
        return lo, hi, src - fl
    B, H, W, C = x.shape
    lo, hi, t = axis_w(H, oh)
    x = x[:, lo] * (1.0 - t)[None, :, None, None] + x[:, hi] * t[None, :, None, None]
    lo, hi, t = axis_w(W, ow)
    return x[:, :, lo] * (1.0 - t)[None, None, :, None] + x[:, :, hi] * t[None, None, :, None]


def dot_attention(q: np.ndarray, v: np.ndarray, k: np.ndarray) -> np.ndarray:
    """keras.layers.Attention(use_scale=False, score_mode="dot") on [query, value, key]: softmax(q k^T) v."""
    s = q @ np.swapaxes(k, 1, 2)
    s = s - s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return p @ v


@dataclass(frozen=True)
class UnetLaplacianSpec:
    depth: int = 3
    width: int = 3
    filters: int = 32
    in_channels: int = 3
    encoder_kernel_size: int = 5
    decoder_kernel_size: int = 1
    gaussian_kernel_size: int = 3
    activation: str = "leaky_relu_01"
    upsample_type: str = "upsample_laplacian_conv2d"
    downsample_type: str = "strides"
    filters_level_multiplier: float = 2.0
    use_ln: bool = True
    use_gamma: bool = True
    use_laplacian: bool = True
    use_laplacian_averaging: bool = True
    use_mix_project: bool = False
    use_concat: bool = False                  # the reference builder's default is True (backbone_unet_laplacian.py:52); every shipped config: false
    use_self_attention: bool = True
    use_attention_gates: bool = False
    use_output_normalization: bool = True
    attention_alpha: float = 0.2
    attention_resolution: Tuple[int, int] = (16, 16)
    # graph revision of the one trained archive the reference ships (pretrained/unet_laplacian_v5.6/model_hydra.keras,
    # written by older code than the snapshot builder; structure read from the archive's config.json and from the
    # operator names inside denoiser_model.tflite next to it): the ConvNext MLP has its own activation, the encoder
    # levels end without LayerNorm / activation, the output LayerNorms sit only in front of the heads (the decoder
    # reads the un-normalised maps), attention runs at the deepest level's own resolution (no resize), row by row, with a
    # second LayerNorm between the attention product and the output convolution
    mlp_activation: str = ""                  # "" = `activation`
    level_activation: bool = True
    output_norm_at_heads: bool = False
    attention_full: bool = False
    attention_activation: str = ""            # "" = LeakyReLU(attention_alpha)
    upsample_linear: bool = False             # upsample_laplacian_conv2d as 1x1 -> bilinear x2 without an activation
    head_filters: int = 32
    head_activation: str = "leaky_relu_01"
    out_channels: int = 3
    v_min: float = 0.0
    v_max: float = 255.0

    @staticmethod
    def from_config(model_config: Dict) -> "UnetLaplacianSpec":
        bb, dn = model_config["backbone"], model_config["denoiser"]
        if bb["type"] != "unet_laplacian":
            raise ValueError(bb["type"])
        unsupported = dict(use_bn=False, use_bias=False,
                           use_complex_base=False, multiple_scale_outputs=True)
        for k, v in unsupported.items():
            if bb.get(k, v) != v:
                raise NotImplementedError(f"{k}={bb[k]} is outside the restated graph")
        if bb.get("use_concat", True) and bb.get("use_attention_gates", False):
            raise NotImplementedError("use_concat with use_attention_gates is outside the restated graph")
        vr = bb.get("value_range", [0, 255])
        return UnetLaplacianSpec(
            depth=bb.get("depth", 5), width=max(bb.get("width", 1) or 1, 1), filters=bb.get("filters", 32),
            in_channels=bb["input_shape"][-1], encoder_kernel_size=bb.get("encoder_kernel_size", 5),
            decoder_kernel_size=bb.get("decoder_kernel_size", 3), gaussian_kernel_size=bb.get("gaussian_kernel_size", 3),
            activation=bb.get("activation", "leaky_relu_01"), upsample_type=bb.get("upsample_type", "bilinear"),
            downsample_type=bb.get("downsample_type", "strides"),
            filters_level_multiplier=float(bb.get("filters_level_multiplier", 2.0)),
            use_ln=bb.get("use_ln", True), use_gamma=bb.get("use_gamma", True), use_laplacian=bb.get("use_laplacian", True),
            use_laplacian_averaging=bb.get("use_laplacian_averaging", True), use_mix_project=bb.get("use_mix_project", True),
            use_concat=bb.get("use_concat", True),
            use_self_attention=bb.get("use_self_attention", False),
            use_attention_gates=bb.get("use_attention_gates", False),
            use_output_normalization=bb.get("use_output_normalization", False),
            mlp_activation=bb.get("convnext_activation", ""), level_activation=bb.get("encoder_level_activation", True),
            output_norm_at_heads=bb.get("output_normalization_at_heads", False),
            attention_full=bb.get("attention_full_resolution", False), attention_activation=bb.get("attention_activation", ""),
            upsample_linear=bb.get("upsample_linear", False),
            head_filters=dn.get("filters", 32), head_activation=dn.get("activation", "linear"),
            out_channels=dn.get("output_channels", 3), v_min=float(vr[0]), v_max=float(vr[1]))

    def level_filters(self, d: int) -> int:
        return int(round(self.filters * max(1, self.filters_level_multiplier ** d)))   # backbone_unet_laplacian.py:198-205

    def tensors(self) -> List[Tuple[str, Tuple[int, ...], str]]:
        """(name, shape, kind) of every trainable tensor in graph-construction order; kind in
        conv | depthwise | ln_gamma | multiplier."""
        out = [("base/kernel", (5, 5, self.in_channels, self.filters), "conv")]       # :296-309 (always 5x5)
        A = self.filters                                                               # attention_channels=filters (:329)

        def block(prefix, C, k, attention, cin=None):
            cin = C if cin is None else cin              # first decoder block behind a Concatenate: 2 C channels in, C out
            if attention:
                if self.use_ln:
                    out.append((f"{prefix}/ln/gamma", (C,), "ln_gamma"))
                for n in ("key", "query", "value"):
                    out.append((f"{prefix}/{n}/kernel", (1, 1, C, A), "conv"))
                if self.attention_full and self.use_ln:
                    out.append((f"{prefix}/ln1/gamma", (A,), "ln_gamma"))
                out.append((f"{prefix}/out/kernel", (1, 1, A, C), "conv"))
                out.append((f"{prefix}/gamma/w", (C,), "multiplier"))
                return
            out.append((f"{prefix}/dw/kernel", (k, k, cin, 1), "depthwise"))
            if self.use_ln:
                out.append((f"{prefix}/ln/gamma", (cin,), "ln_gamma"))
            out.append((f"{prefix}/pw1/kernel", (1, 1, cin, 4 * C), "conv"))
            out.append((f"{prefix}/pw2/kernel", (1, 1, 4 * C, C), "conv"))
            if self.use_gamma:
                out.append((f"{prefix}/gamma/w", (C,), "multiplier"))

        for d in range(self.depth):
            C = self.level_filters(d)
            for w in range(self.width):
                block(f"enc{d}_{w}", C, self.encoder_kernel_size, self.use_self_attention and d == self.depth - 1)
            if self.use_output_normalization and self.use_ln and (d == self.depth - 1 or not self.output_norm_at_heads):
                out.append((f"enc{d}/out_ln/gamma", (C,), "ln_gamma"))
            if d != self.depth - 1:
                kd = 2 if self.downsample_type == "conv2d" else 1            # downsampling.py:45-72
                out.append((f"down{d}/kernel", (kd, kd, C, self.level_filters(d + 1)), "conv"))
        for d in reversed(range(self.depth - 1)):
            C = self.level_filters(d)
            if self.upsample_type == "upsample_laplacian_conv2d":
                out.append((f"up{d}/kernel", (1, 1, self.level_filters(d + 1), C), "conv"))
            elif self.upsample_type in ("upsample_bilinear_conv2d", "upsample_nearest_conv2d"):
                out.append((f"up{d}/kernel", (3, 3, self.level_filters(d + 1), C), "conv"))   # upsampling.py:52-72
            if self.use_attention_gates:                    # AdditiveAttentionGate.build (custom_layers.py:749-790)
                out.append((f"gate{d}/x/kernel", (1, 1, C, C), "conv"))
                if self.use_ln:
                    out.append((f"gate{d}/x_ln/gamma", (C,), "ln_gamma"))
                out.append((f"gate{d}/y/kernel", (1, 1, C, C), "conv"))
                if self.use_ln:
                    out.append((f"gate{d}/y_ln/gamma", (C,), "ln_gamma"))
                out.append((f"gate{d}/o/kernel", (1, 1, C, C), "conv"))
                out.append((f"gate{d}/scale/w", (C,), "multiplier"))
            cat = 2 * C if self.use_concat else C           # Concatenate([encoder feature, upsampled]) (:516-517)
            if self.use_mix_project:
                out.append((f"mix{d}/kernel", (1, 1, cat, C), "conv"))
                cat = C
            for w in range(self.width):
                block(f"dec{d}_{w}", C, self.decoder_kernel_size, False, cin=cat if w == 0 else C)
            if self.use_output_normalization and self.use_ln:
                out.append((f"dec{d}/out_ln/gamma", (C,), "ln_gamma"))
        for i in range(self.depth):
            out.append((f"head{i}/conv0/kernel", (1, 1, self.level_filters(i), self.head_filters), "conv"))
            out.append((f"head{i}/conv1/kernel", (1, 1, self.head_filters, self.out_channels), "conv"))
        return out

    def param_count(self) -> int:
        return sum(int(np.prod(s)) for _, s, _ in self.tensors())

    def offsets(self) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
        off, o = {}, 0
        for name, shape, _ in self.tensors():
            off[name] = (o, shape)
            o += int(np.prod(shape))
        return off


def canonical_config(depth: int = 3, width: int = 3, filters: int = 32) -> Dict:
    """model section of configs/unet_laplacian_v5.json."""
    return {"model": {
        "backbone": {"type": "unet_laplacian", "input_shape": ["?", "?", 3], "depth": depth, "width": width,
                     "filters": filters, "use_bn": False, "use_ln": True, "use_bias": False, "use_concat": False,
                     "use_gamma": True, "use_complex_base": False, "use_mix_project": False,
                     "use_self_attention": True, "use_attention_gates": False, "use_output_normalization": True,
                     "encoder_kernel_size": 5, "decoder_kernel_size": 1, "multiple_scale_outputs": True,
                     "activation": "leaky_relu_01", "use_soft_orthonormal_regularization": True,
                     "kernel_initializer": "glorot_normal", "kernel_regularizer": "l2",
                     "upsample_type": "upsample_laplacian_conv2d", "downsample_type": "strides",
                     "depth_drop_rate": 0.5, "convolutional_self_attention_dropout_rate": 0.25},
        "denoiser": {"filters": 32, "use_bn": False, "use_ln": False, "use_bias": False,
                     "activation": "leaky_relu_01", "output_channels": 3, "kernel_regularizer": "l2",
                     "kernel_initializer": "glorot_normal"}}}


def init_params(spec: UnetLaplacianSpec, seed: int = 42, nontrivial: bool = True) -> np.ndarray:
    """glorot_normal kernels; LN gamma 1 and multiplier w ~ truncated_normal(0, 0.01) as keras creates them
    (custom_layers.py:271), or - `nontrivial` - spread out so that a wrong gamma / multiplier shows in the output."""
    rng = np.random.default_rng(seed)
    parts = []
    for name, shape, kind in spec.tensors():
        if kind in ("conv", "depthwise"):
            a = O.glorot_normal(shape, rng)
        elif kind == "ln_gamma":
            a = rng.uniform(0.6, 1.4, shape) if nontrivial else np.ones(shape)
        else:
            a = rng.uniform(-1.2, 0.5, shape) if nontrivial else np.clip(rng.normal(0, 0.01, shape), -0.02, 0.02)
        parts.append(np.asarray(a, np.float32).ravel())
    return np.concatenate(parts)


def _views(spec: UnetLaplacianSpec, flat: np.ndarray, dtype=F64) -> Dict[str, np.ndarray]:
    flat = np.asarray(flat)
    return {n: flat[o:o + int(np.prod(s))].reshape(s).astype(dtype) for n, (o, s) in spec.offsets().items()}


def backbone_forward(spec: UnetLaplacianSpec, P: Dict[str, np.ndarray], xn: np.ndarray) -> List[np.ndarray]:
    """normalised input [B,H,W,3] -> [full-res, 1/2, 1/4, ...] feature maps (backbone_unet_laplacian.py:281-606)."""
    a = spec.activation
    conv = O.conv2d_same

    def convnext(prefix, x):
        t = depthwise_same(x, P[f"{prefix}/dw/kernel"])                 # conv_1, linear (custom_layers.py:979-988)
        if spec.use_ln:
            t = layer_norm(t, P[f"{prefix}/ln/gamma"])
        t = act(conv(t, P[f"{prefix}/pw1/kernel"]), spec.mlp_activation or a)   # conv_2 + activation (:991-993)
        t = conv(t, P[f"{prefix}/pw2/kernel"])                          # conv_3, linear (:1000-1002)
        if spec.use_gamma:
            t = channel_multiplier(t, P[f"{prefix}/gamma/w"])
        return t

    def attention(prefix, x):
        B, H, W, C = x.shape
        rh, rw = (H, W) if spec.attention_full else spec.attention_resolution
        t = x if spec.attention_full else resize_bilinear(x, rh, rw)
        if spec.use_ln:
            t = layer_norm(t, P[f"{prefix}/ln/gamma"])
        qkv_act = (lambda z: act(z, spec.attention_activation)) if spec.attention_activation else \
                  (lambda z: leaky(z, spec.attention_alpha))
        if spec.attention_full:
            # the archive's graph (operator list of denoiser_model.tflite): keras Attention fed the rank-4 maps without a
            # reshape, i.e. one sequence per image ROW (scores [B,H,W,W]), and handed over in the order [query, key, value]
            # where keras reads [query, value, key]: scores = query_conv . value_conv^T, output = softmax . key_conv
            q, v, k = (qkv_act(conv(t, P[f"{prefix}/{n}/kernel"])).reshape(B * rh, rw, -1) for n in ("query", "key", "value"))
        else:
            q, v, k = (qkv_act(conv(t, P[f"{prefix}/{n}/kernel"])).reshape(B, rh * rw, -1) for n in ("query", "value", "key"))
        t = dot_attention(q, v, k).reshape(B, rh, rw, -1)
        if spec.attention_full:
            if spec.use_ln:
                t = layer_norm(t, P[f"{prefix}/ln1/gamma"])
        else:
            t = resize_bilinear(t, H, W)
        t = conv(t, P[f"{prefix}/out/kernel"])                           # output_activation "linear" (:331)
        return channel_multiplier(t, P[f"{prefix}/gamma/w"])             # use_gamma=True fixed (:326)

    x = act(conv(xn, P["base/kernel"]), a)
    nodes = {}
    for d in range(spec.depth):
        for w in range(spec.width):
            if spec.use_self_attention and d == spec.depth - 1:
                x = x + attention(f"enc{d}_{w}", x)
            else:
                x = x + convnext(f"enc{d}_{w}", x)                      # Add (:351-354), StochasticDepth = identity
        if spec.use_output_normalization and spec.use_ln and not spec.output_norm_at_heads:
            x = layer_norm(x, P[f"enc{d}/out_ln/gamma"])                 # keras default epsilon 1e-3 (:359)
        if spec.level_activation:
            x = act(x, a)                                                # :360
        nodes[d] = x
        if d != spec.depth - 1:
            if spec.use_laplacian or spec.use_laplacian_averaging:
                k = spec.gaussian_kernel_size
                if spec.use_laplacian_averaging:
                    smooth = O.avg_pool_same(x, (k, k), 1)               # :369-374
                else:
                    g = gaussian_kernel_3((k, k))
                    smooth = depthwise_same(x, np.repeat(g[:, :, None, None], x.shape[-1], axis=2))
                nodes[d] = x - smooth
                x = smooth
            if spec.downsample_type == "strides":
                x = act(conv(O.strided_slice_2x(x), P[f"down{d}/kernel"]), a)             # downsampling.py:60-72
            elif spec.downsample_type == "conv2d":
                x = act(conv(x, P[f"down{d}/kernel"], stride=2), a)                       # 2x2, strides 2, same (:45-55)
            elif spec.downsample_type == "maxpool":
                x = act(conv(max_pool_2x2_same(x), P[f"down{d}/kernel"]), a)              # :56-68
            else:
                raise ValueError(spec.downsample_type)
    outs = {spec.depth - 1: nodes[spec.depth - 1]}                       # nodes_output[(depth-1, 1)] (:434)
    for d in reversed(range(spec.depth - 1)):
        low = outs[d + 1]
        if spec.upsample_type == "upsample_laplacian_conv2d":
            if a == "linear" or spec.upsample_linear:                    # upsampling.py:80-90
                up = O.upsample_bilinear_2x(conv(low, P[f"up{d}/kernel"]))
            else:                                                        # :91-102
                up = act(conv(O.upsample_bilinear_2x(low), P[f"up{d}/kernel"]), a)
        elif spec.upsample_type == "upsample_bilinear_conv2d":              # upsampling.py:52-63
            up = act(conv(O.upsample_bilinear_2x(low), P[f"up{d}/kernel"]), a)
        elif spec.upsample_type == "upsample_nearest_conv2d":               # :64-76
            up = act(conv(O.upsample_nearest_2x(low), P[f"up{d}/kernel"]), a)
        elif spec.upsample_type == "bilinear":
            up = O.upsample_bilinear_2x(low)
        elif spec.upsample_type in ("nn", "nearest"):
            up = O.upsample_nearest_2x(low)
        else:
            raise NotImplementedError(spec.upsample_type)
        enc = nodes[d]
        if spec.use_attention_gates:                                     # :497-509; AdditiveAttentionGate.call (custom_layers.py:805-832)
            yg = conv(layer_norm(enc, P[f"gate{d}/y_ln/gamma"]) if spec.use_ln else enc, P[f"gate{d}/y/kernel"])
            xg = conv(layer_norm(up, P[f"gate{d}/x_ln/gamma"]) if spec.use_ln else up, P[f"gate{d}/x/kernel"])
            o = channel_multiplier(conv(leaky(xg + yg, 0.1), P[f"gate{d}/o/kernel"]), P[f"gate{d}/scale/w"])
            enc = enc * (1.0 / (1.0 + np.exp(-4.0 * o)))
        x = np.concatenate([enc, up], axis=-1) if spec.use_concat else enc + up      # Concatenate / Add (:516-519)
        if spec.use_mix_project:
            x = act(conv(x, P[f"mix{d}/kernel"]), a)
        for w in range(spec.width):
            y = convnext(f"dec{d}_{w}", x)
            x = x + y if y.shape[-1] == x.shape[-1] else y               # the skip only where the channel counts agree (:557-560)
        if spec.use_output_normalization and spec.use_ln and not spec.output_norm_at_heads:
            x = layer_norm(x, P[f"dec{d}/out_ln/gamma"])
        outs[d] = x
    if spec.use_output_normalization and spec.use_ln and spec.output_norm_at_heads:
        last = spec.depth - 1
        outs = {d: layer_norm(outs[d], P[f"enc{d}/out_ln/gamma" if d == last else f"dec{d}/out_ln/gamma"]) for d in outs}
    return [outs[d] for d in range(spec.depth)]                          # deepest-last after the double reverse (:569-588)


def hydra_forward(spec: UnetLaplacianSpec, params: np.ndarray, x: np.ndarray, dtype=F64) -> List[np.ndarray]:
    """hydra(x): normalise -> backbone -> head_i -> denormalise, one output per scale, full resolution first
    (model.py:100-142, 297-342)."""
    P = _views(spec, params, dtype)
    xn = O.layer_normalize(x.astype(dtype), spec.v_min, spec.v_max)
    outs = []
    for i, f in enumerate(backbone_forward(spec, P, xn)):
        h = act(O.conv2d_same(f, P[f"head{i}/conv0/kernel"]), spec.head_activation)
        h = O.conv2d_same(h, P[f"head{i}/conv1/kernel"])
        outs.append(O.layer_denormalize(np.tanh(2.0 * h) * 0.51, spec.v_min, spec.v_max))
    return outs


def denoiser_module_call(spec: UnetLaplacianSpec, params, image_u8: np.ndarray, cast_to_uint8: bool = True):
    """DenoiserModule.__call__ (module_denoiser.py:46-75): cast, pad to a power of two, hydra, first output, crop,
    round half to even, cast."""
    x = image_u8.astype(F64)
    xp, ph, pw = O.pad_to_power_of_2(x)
    y = O.remove_padding(hydra_forward(spec, params, xp)[0], ph, pw)
    if not cast_to_uint8:
        return y
    return np.clip(O.round_half_even(y), 0, 255).astype(np.uint8)
