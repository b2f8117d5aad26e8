"""CPU oracle for resnet backbones outside the 16-filter 3x3 family: per-block kernel sizes / filters, depthwise
convolutions with a depth multiplier and grouped convolutions, as in the one resnet config the reference ships
(`configs/resnet_color_1x6_bn_32x128x32_1x3x1_128x128_depthwise_l1_relu.json`), and the channel gate of `add_gates`
(backbone_blocks.py:199-208).  This file is the inference forward; the training step (batch-statistics BatchNorm, loss,
regularisers, gradients by autograd) is oracle/resnet_generic_torch.py, whose forward is pinned against this one.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).  NumPy fp64 restatement of
  bfcnn/backbone_resnet.py:36-298 (builder: conv params per block position, last activation = base_activation),
  bfcnn/backbone_blocks.py:163-246 (first conv without BN, second / third with BN, Add),
  bfcnn/utilities.py:132-224 (conv2d_wrapper: conv -> BN -> activation; depth_multiplier => DepthwiseConv2D),
  bfcnn/model.py:58-162, 251-359 (normalise, head 1x1 -> 1x1 -> tanh(2x)*0.51, denormalise).
Parity unpinned beyond structure for the same reason as the other oracles (TensorFlow 2.13.1 absent, no golden vectors)."""
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from . import bfcnn_oracle as O

F64 = np.float64
BN_EPS = 1e-3


def depthwise_mult_same(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """keras DepthwiseConv2D(depth_multiplier=m, padding="same"): w [kh,kw,C,m], output channel c*m + j."""
    B, H, W, C = x.shape
    kh, kw, _, m = w.shape
    _, pt, pb = O.same_pads(H, kh, 1)
    _, pl, pr = O.same_pads(W, kw, 1)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros((B, H, W, C, m), dtype=np.result_type(x.dtype, w.dtype))
    for i in range(kh):
        for j in range(kw):
            y += xp[:, i:i + H, j:j + W, :, None] * w[i, j][None, None, None]
    return y.reshape(B, H, W, C * m)


def grouped_conv_same(x: np.ndarray, w: np.ndarray, groups: int) -> np.ndarray:
    """keras Conv2D(groups=g): w [kh,kw,cin/g,cout]; group i maps input channels [i cin/g, (i+1) cin/g) to output channels
    [i cout/g, (i+1) cout/g)."""
    if groups == 1:
        return O.conv2d_same(x, w)
    cin_g, cout_g = w.shape[2], w.shape[3] // groups
    return np.concatenate([O.conv2d_same(x[..., g * cin_g:(g + 1) * cin_g], w[..., g * cout_g:(g + 1) * cout_g])
                           for g in range(groups)], axis=-1)


def hard_sigmoid(x):
    """keras 2.13 hard_sigmoid: 0 below -2.5, 1 above 2.5, 0.2 x + 0.5 in between"""
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def gate(x: np.ndarray, w0: np.ndarray, w1: np.ndarray) -> np.ndarray:
    """backbone_blocks.py:199-208: y = mean over H, W -> Dense(relu, no bias) -> Dense(hard_sigmoid, no bias); x * y"""
    y = np.maximum(x.mean(axis=(1, 2)) @ w0, 0.0)
    y = hard_sigmoid(y @ w1)
    return x * y[:, None, None, :]


def avgpool_same(x: np.ndarray, pool: Tuple[int, int], stride: Tuple[int, int]) -> np.ndarray:
    """keras AveragePooling2D(pool, strides, padding="same"): TF SAME padding, divisor = taps inside the image"""
    B, H, W, C = x.shape
    OH, OW = -(-H // stride[0]), -(-W // stride[1])
    pt = max((OH - 1) * stride[0] + pool[0] - H, 0) // 2
    pl = max((OW - 1) * stride[1] + pool[1] - W, 0) // 2
    out = np.zeros((B, OH, OW, C), dtype=x.dtype)
    for oy in range(OH):
        y0, y1 = max(oy * stride[0] - pt, 0), min(oy * stride[0] - pt + pool[0], H)
        for ox in range(OW):
            x0, x1 = max(ox * stride[1] - pl, 0), min(ox * stride[1] - pl + pool[1], W)
            out[:, oy, ox] = x[:, y0:y1, x0:x1].mean(axis=(1, 2))
    return out


def squeeze_and_excite_block(x, w0, b0, w1, b1, hard_sigmoid_version=False, learn_to_turn_off=False):
    """backbone_blocks.py:251-313: GlobalAvgPool -> 1x1 conv (+bias) -> LeakyReLU(0.1) -> 1x1 conv (+bias) -> sigmoid, or
    [2.5 - relu] -> hard_sigmoid; x * gate.  w0 [C, Cs], w1 [Cs, C]; biases may be None (use_bias=False)."""
    m = x.mean(axis=(1, 2))
    h = m @ w0 + (0.0 if b0 is None else b0)
    h = np.where(h > 0, h, 0.1 * h)
    p = h @ w1 + (0.0 if b1 is None else b1)
    if hard_sigmoid_version:
        if learn_to_turn_off:
            p = 2.5 - np.maximum(p, 0.0)
        g = hard_sigmoid(p)
    else:
        g = 1.0 / (1.0 + np.exp(-p))
    return x * g[:, None, None, :]


SELECTOR_GLOBAL_LEAKY = 0.2      # Dense(activation="leaky_relu") does not resolve in Keras 2.13; Keras >= 2.15 gives slope 0.2


def selector_prefilter(sel, flags, pre_w, pool_size):
    """custom_layers_selector.py:160-185 with utilities.py:566-620: Conv2D 1x1 (linear) -> global_normalization -> local_normalization
    -> lowpass_filter(a=4, b=4) -> highpass_filter(a=4, b=4), each optional, in this order"""
    x = sel
    if "use_conv1x1_selector" in flags:
        x = x @ pre_w.reshape(pre_w.shape[-2], pre_w.shape[-1])
    if "use_global_normalization" in flags:
        m = x.mean(axis=(1, 2), keepdims=True)
        x = (x - m) / np.sqrt(((x - m) ** 2).mean(axis=(1, 2), keepdims=True) + 1e-3)
    if "use_local_normalization" in flags:
        m = avgpool_same(x, pool_size, (1, 1))
        x = (x - m) / np.sqrt(avgpool_same((x - m) ** 2, pool_size, (1, 1)) + 1e-3)
    if "use_lowpass" in flags:
        x = (1.0 - np.tanh(4.0 * x) ** 4.0) * x
    if "use_highpass" in flags:
        x = np.tanh(4.0 * x) ** 4.0 * x
    return x


def selector_block(x1, x2, sel, w0, w1, scale_type="local", activation_type="hard", pool_size=(32, 32), strides_size=None):
    """custom_layers_selector.py:81-330, scale types LOCAL (the default), MULTISCALE, MIXED and GLOBAL, no optional pre-filters:
    LOCAL: AveragePooling2D(pool, strides = pool / 4, same) -> 1x1 conv leaky_relu (0.3, activation_wrapper) -> 1x1 conv relu ->
    UpSampling2D(strides, bilinear); GLOBAL: mean -> Dense leaky_relu -> Dense relu; then s = F(2.5 - x), x1 s + x2 (1 - s)."""
    from . import unet_oracle as U
    strides_size = strides_size or (pool_size[0] // 4, pool_size[1] // 4)
    st = scale_type.lower()
    if st in ("local", "multiscale", "mixed"):
        if st == "local":
            u = avgpool_same(sel, pool_size, strides_size)
        elif st == "multiscale":                               # three window sizes side by side (:203-232)
            u = np.concatenate([avgpool_same(sel, (pool_size[0] // 2, pool_size[1] // 2), strides_size),
                                avgpool_same(sel, pool_size, strides_size),
                                avgpool_same(sel, (pool_size[0] * 2, pool_size[1] * 2), strides_size)], axis=-1)
        else:                                                  # local means next to the image's global mean (:284-299)
            loc = avgpool_same(sel, pool_size, strides_size)
            u = np.concatenate([loc, np.zeros_like(loc) + sel.mean(axis=(1, 2), keepdims=True)], axis=-1)
        u = u @ w0.reshape(w0.shape[-2], w0.shape[-1])
        u = np.where(u > 0, u, 0.3 * u)
        u = np.maximum(u @ w1.reshape(w1.shape[-2], w1.shape[-1]), 0.0)
        u = U.resize_bilinear(u, u.shape[1] * strides_size[0], u.shape[2] * strides_size[1])
    elif st == "global":
        u = sel.mean(axis=(1, 2)) @ w0
        u = np.where(u > 0, u, SELECTOR_GLOBAL_LEAKY * u)
        u = np.maximum(u @ w1, 0.0)[:, None, None, :]
    else:
        raise ValueError(scale_type)
    p = 2.5 - u
    s = hard_sigmoid(p) if activation_type.lower() == "hard" else 1.0 / (1.0 + np.exp(-p))
    return x1 * s + x2 * (1.0 - s)


@dataclass(frozen=True)
class GenericResnetSpec:
    filters: int
    kernel_size: int
    no_layers: int
    block_kernels: Tuple[int, ...]
    block_filters: Tuple[int, ...]
    block_depthwise: Tuple[int, ...]
    block_groups: Tuple[int, ...]
    block_activation: Tuple[str, ...]
    base_activation: str = "linear"
    use_bn: bool = True
    in_channels: int = 3
    head_filters: int = 32
    head_activation: str = "linear"
    out_channels: int = 3
    v_min: float = 0.0
    v_max: float = 255.0
    add_gates: bool = False
    kernel_regularizer: str = "l1"
    block_regularizer: Tuple[str, ...] = ()
    head_regularizer: str = "l2"
    selector: Tuple = ()                  # () or (scale_type, activation_type, compress channels, pool, stride, regulariser)
    add_initial_bn: bool = False          # BatchNormalization behind the base convolution (backbone_resnet.py:264-265)
    add_final_bn: bool = False            # ... behind the last block (:274-275)
    add_channelwise_scaling: bool = False   # ChannelwiseMultiplier closing every block and the backbone (:236-238, 282-283)
    add_learnable_multiplier: bool = False  # Multiplier, likewise (:240-242, 286-287)
    dropout_rate: float = -1.0            # RandomOnOff on every block's branch (:231-235): training only
    add_concat_input: bool = False        # Concatenate([features, the backbone's (normalised) input]) ahead of the closing layers (:277-279)

    @staticmethod
    def from_config(model_config: Dict) -> "GenericResnetSpec":
        bb, dn = model_config["backbone"], model_config["denoiser"]
        bk = tuple(bb.get("block_kernels", [3, 3]))
        act = bb.get("activation", "relu")
        ba = list(bb.get("block_activation") or [act] * len(bk))
        base_act = bb.get("base_activation", "linear")
        ba[-1] = base_act                                            # backbone_resnet.py:178
        vr = bb.get("value_range", [0, 255])
        return GenericResnetSpec(
            filters=bb.get("filters", 32), kernel_size=bb.get("kernel_size", 3), no_layers=bb["no_layers"], block_kernels=bk,
            block_filters=tuple(bb.get("block_filters", [bb.get("filters", 32)] * len(bk))),
            block_depthwise=tuple(bb.get("block_depthwise") or [-1] * len(bk)),
            block_groups=tuple(bb.get("block_groups") or [1] * len(bk)), block_activation=tuple(ba),
            base_activation=base_act, use_bn=bb.get("use_bn", True), in_channels=bb["input_shape"][-1],
            head_filters=dn.get("filters", 32), head_activation=dn.get("activation", "linear"),
            out_channels=dn.get("output_channels", 3), v_min=float(vr[0]), v_max=float(vr[1]),
            add_gates=bool(bb.get("add_gates", False)), kernel_regularizer=bb.get("kernel_regularizer", "l1"),
            block_regularizer=tuple(bb.get("block_regularizer") or [bb.get("kernel_regularizer", "l1")] * len(bk)),
            head_regularizer=dn.get("kernel_regularizer", "l2"),
            selector=GenericResnetSpec._selector(bb),
            add_initial_bn=bool(bb.get("add_initial_bn", False)), add_final_bn=bool(bb.get("add_final_bn", False)),
            add_channelwise_scaling=bool(bb.get("add_channelwise_scaling", False)),
            add_learnable_multiplier=bool(bb.get("add_learnable_multiplier", False)), dropout_rate=float(bb.get("dropout_rate", -1)),
            add_concat_input=bool(bb.get("add_concat_input", False)))

    @staticmethod
    def _selector(bb) -> Tuple:
        sp = bb.get("selector_params")
        if sp is None:
            return ()
        pool = tuple(int(v) for v in sp.get("pool_size", (32, 32)))
        stride = tuple(int(v) for v in sp.get("strides_size", (pool[0] / 4, pool[1] / 4)))
        filters = bb.get("filters", 32)
        return (str(sp.get("scale_type", "local")).lower(), str(sp.get("activation_type", "hard")).lower(),
                max(1, int(round(filters * sp.get("filters_compress_ratio", 0.25)))), pool, stride,
                sp.get("kernel_regularizer", "l1"),                       # custom_layers_selector.py:88: the selector layers' regulariser
                tuple(k for k in ("use_conv1x1_selector", "use_global_normalization", "use_local_normalization", "use_lowpass",
                                  "use_highpass") if sp.get(k, False)))   # optional pre-filters (:160-185)

    def gate_channels(self) -> int:
        """backbone_blocks.py:131-141: the second convolution's filters, or filters x depth_multiplier for a depthwise one"""
        if len(self.block_kernels) < 2:
            raise ValueError("add_gates needs a second convolution (gate_no_filters)")
        return self.block_filters[1] if self.block_depthwise[1] == -1 else self.block_filters[0] * self.block_depthwise[1]

    def tensors(self) -> List[Tuple[str, Tuple[int, ...], str]]:
        """(name, shape, kind) in graph-construction order; kind in conv | depthwise | bn_gamma | dense | channelwise | multiplier.
        ChannelwiseMultiplier / Multiplier (custom_layers.py:1028-1160): the trainable `w0` (zeros at creation); their
        non-trainable `w1` stays at its creation value `multiplier` = 1.0 and is a constant here."""
        k = self.kernel_size
        out = [("base/kernel", (k, k, self.in_channels, self.filters), "conv")]
        if self.add_initial_bn:
            out.append(("initial_bn/gamma", (self.filters,), "bn_gamma"))
        for i in range(self.no_layers):
            cin = self.filters
            for j, (kk, cf, dm, g) in enumerate(zip(self.block_kernels, self.block_filters, self.block_depthwise, self.block_groups)):
                if dm != -1:
                    out.append((f"block{i}/conv{j}/kernel", (kk, kk, cin, dm), "depthwise"))
                    cout = cin * dm
                else:
                    out.append((f"block{i}/conv{j}/kernel", (kk, kk, cin // g, cf), "conv"))
                    cout = cf
                if j >= 1 and self.use_bn:
                    out.append((f"block{i}/bn{j}/gamma", (cout,), "bn_gamma"))
                if j == 1 and self.add_gates:                      # created right after the second convolution (:199-208)
                    gc = self.gate_channels()
                    out.append((f"block{i}/gate/dense0/kernel", (gc, max(int(gc / 8), 2)), "dense"))
                    out.append((f"block{i}/gate/dense1/kernel", (max(int(gc / 8), 2), gc), "dense"))
                cin = cout
            if self.add_channelwise_scaling:                       # backbone_blocks.py:215-221, ahead of the Add / selector
                out.append((f"block{i}/channelwise/w0", (cin,), "channelwise"))
            if self.add_learnable_multiplier:
                out.append((f"block{i}/multiplier/w0", (1,), "multiplier"))
            if self.selector:                                      # the selector's two layers close the block (backbone_blocks.py:227-239)
                cs, cc = self.block_filters[0] if self.block_depthwise[0] == -1 else self.filters * self.block_depthwise[0], self.selector[2]
                if "use_conv1x1_selector" in self.selector[6]:
                    out.append((f"block{i}/selector/pre/kernel", (1, 1, cs, self.filters), "conv"))
                    cs = self.filters
                if self.selector[0] != "global":
                    cs *= {"local": 1, "mixed": 2, "multiscale": 3}[self.selector[0]]
                    out.append((f"block{i}/selector/conv0/kernel", (1, 1, cs, cc), "conv"))
                    out.append((f"block{i}/selector/conv1/kernel", (1, 1, cc, self.filters), "conv"))
                else:
                    out.append((f"block{i}/selector/dense0/kernel", (cs, cc), "dense"))
                    out.append((f"block{i}/selector/dense1/kernel", (cc, self.filters), "dense"))
        if self.add_final_bn:
            out.append(("final_bn/gamma", (self.filters,), "bn_gamma"))
        cf = self.filters + (self.in_channels if self.add_concat_input else 0)
        if self.add_channelwise_scaling:
            out.append(("channelwise/w0", (cf,), "channelwise"))
        if self.add_learnable_multiplier:
            out.append(("multiplier/w0", (1,), "multiplier"))
        out.append(("head/conv0/kernel", (1, 1, cf, self.head_filters), "conv"))
        out.append(("head/conv1/kernel", (1, 1, self.head_filters, self.out_channels), "conv"))
        return out

    def state_tensors(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        for name, shape, kind in self.tensors():
            if kind == "bn_gamma":
                base = name[:-len("/gamma")]
                out += [(base + "/moving_mean", shape), (base + "/moving_variance", shape)]
        return out


def init_params(spec: GenericResnetSpec, seed: int = 42) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    params = []
    for name, shape, kind in spec.tensors():
        if kind == "bn_gamma":
            params.append(rng.uniform(0.5, 1.5, shape))
        elif kind in ("channelwise", "multiplier"):                  # zeros at creation; spread out so that a wrong one shows
            params.append(rng.uniform(-1.2, 0.6, shape) if kind == "channelwise" else rng.uniform(-0.5, 0.5, shape))
        else:
            params.append(O.glorot_normal(shape, rng) if len(shape) == 4 else O.glorot_normal((1, 1) + tuple(shape), rng).reshape(shape))
    state = []
    for name, shape in spec.state_tensors():
        state.append(rng.normal(0, 0.1, shape) if name.endswith("mean") else rng.uniform(0.5, 1.5, shape))
    cat = lambda parts: np.concatenate([np.asarray(p, np.float32).ravel() for p in parts]) if parts else np.zeros(0, np.float32)
    return cat(params), cat(state)


def _views(items, flat, dtype):
    out, o = {}, 0
    for item in items:
        name, shape = item[0], item[1]
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[o:o + n]).reshape(shape).astype(dtype)
        o += n
    return out


def hydra_forward(spec: GenericResnetSpec, params: np.ndarray, state: np.ndarray, x: np.ndarray, dtype=F64) -> np.ndarray:
    P = _views(spec.tensors(), params, dtype)
    S = _views(spec.state_tensors(), state, dtype)
    xn = O.layer_normalize(x.astype(dtype), spec.v_min, spec.v_max)
    f = O.activation_fwd(O.conv2d_same(xn, P["base/kernel"]), spec.base_activation)
    scaled = lambda t, name: t * np.maximum(P[name] + 1.0, 0.0)        # activation "relu" of (w0 + w1), w1 = 1 (backbone_resnet.py:190-202)
    if spec.add_initial_bn:
        f = O.bn_infer(f, P["initial_bn/gamma"], S["initial_bn/moving_mean"], S["initial_bn/moving_variance"], BN_EPS)
    for i in range(spec.no_layers):
        t = f
        for j, (dm, g, a) in enumerate(zip(spec.block_depthwise, spec.block_groups, spec.block_activation)):
            w = P[f"block{i}/conv{j}/kernel"]
            t = depthwise_mult_same(t, w) if dm != -1 else grouped_conv_same(t, w, g)
            if j >= 1 and spec.use_bn:                                   # first conv: bn_params=None (backbone_blocks.py:174-179)
                base = f"block{i}/bn{j}"
                t = O.bn_infer(t, P[base + "/gamma"], S[base + "/moving_mean"], S[base + "/moving_variance"], BN_EPS)
            t = O.activation_fwd(t, a)
            if j == 0:
                first = t                                                # x_1st_conv: the selector layer
            if j == 1 and spec.add_gates:
                t = gate(t, P[f"block{i}/gate/dense0/kernel"], P[f"block{i}/gate/dense1/kernel"])
        if spec.add_channelwise_scaling:
            t = scaled(t, f"block{i}/channelwise/w0")
        if spec.add_learnable_multiplier:
            t = scaled(t, f"block{i}/multiplier/w0")
        # RandomOnOff (dropout_rate): Dropout is the identity outside training
        if spec.selector:
            kind = "dense" if spec.selector[0] == "global" else "conv"
            if spec.selector[6]:
                first = selector_prefilter(first, spec.selector[6], P.get(f"block{i}/selector/pre/kernel"), spec.selector[3])
            f = selector_block(f, t, first, P[f"block{i}/selector/{kind}0/kernel"], P[f"block{i}/selector/{kind}1/kernel"],
                               spec.selector[0], spec.selector[1], spec.selector[3], spec.selector[4])
        else:
            f = t + f
    if spec.add_final_bn:
        f = O.bn_infer(f, P["final_bn/gamma"], S["final_bn/moving_mean"], S["final_bn/moving_variance"], BN_EPS)
    if spec.add_concat_input:
        f = np.concatenate([f, xn], axis=-1)
    if spec.add_channelwise_scaling:
        f = scaled(f, "channelwise/w0")
    if spec.add_learnable_multiplier:
        f = scaled(f, "multiplier/w0")
    h = O.activation_fwd(O.conv2d_same(f, P["head/conv0/kernel"]), spec.head_activation)
    h = O.conv2d_same(h, P["head/conv1/kernel"])
    return O.layer_denormalize(np.tanh(2.0 * h) * 0.51, spec.v_min, spec.v_max)


def denoiser_module_call(spec: GenericResnetSpec, params, state, image_u8: np.ndarray, cast_to_uint8: bool = True):
    xp, ph, pw = O.pad_to_power_of_2(image_u8.astype(F64))
    y = O.remove_padding(hydra_forward(spec, params, state, xp), ph, pw)
    return np.clip(O.round_half_even(y), 0, 255).astype(np.uint8) if cast_to_uint8 else y


def shipped_config() -> Dict:
    """model section of configs/resnet_color_1x6_bn_32x128x32_1x3x1_128x128_depthwise_l1_relu.json (backbone values as
    shipped; the denoiser section of that file carries only regulariser / initialiser settings)."""
    return {"backbone": {"type": "resnet", "filters": 32, "no_layers": 6, "kernel_size": 7, "block_kernels": [1, 3, 1],
                         "block_filters": [32, 128, 32], "block_depthwise": [-1, 4, -1], "block_regularizer": ["l1", "l1", "l1"],
                         "block_activation": ["relu", "relu", "linear"], "block_groups": [1, 1, 2], "value_range": [0, 255],
                         "batchnorm": True, "activation": "relu", "add_final_bn": False, "input_shape": ["?", "?", 3],
                         "kernel_regularizer": "l1", "kernel_initializer": "glorot_normal"},
            "denoiser": {"output_channels": 3, "kernel_regularizer": "l1", "kernel_initializer": "glorot_normal"}}
