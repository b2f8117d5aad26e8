"""Resnet backbones outside the 16-filter 3x3 family (bfcnn/backbone_resnet.py:36-298): per-position kernel sizes and
filters, depthwise convolutions with a depth multiplier, grouped convolutions -- e.g. the config the reference ships,
`resnet_color_1x6_bn_32x128x32_1x3x1_..._depthwise` (1x1 32->32, depthwise 3x3 x4, grouped 1x1 128->32), and the channel gate
of `add_gates` (backbone_blocks.py:199-208).  This file is inference (training: resnet_generic_train.py), on
the operator library of csrc/unet_ops.hip: first convolution, 1x1 / k x k matrix-core convolutions with the BatchNorm
folded (scale into the weights at pack time, shift as the epilogue bias), depthwise-with-multiplier kernel, fused head.
The 16-filter 3x3 family keeps its own engine (`HydraModel`, fused split-f16 blocks, training)."""
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as N
from . import unet_laplacian as UL

BN_EPSILON = 1e-3          # DEFAULT_BN_EPSILON (bfcnn/constants.py:9)


def channel_gate(x: torch.Tensor, w0: torch.Tensor, w1: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """the `add_gates` gate (backbone_blocks.py:199-208): x * hard_sigmoid(relu(mean_hw(x) @ w0) @ w1) [+ res]"""
    B, H, W, Cc = x.shape
    L = N.lib()
    out = torch.empty_like(x)
    save = torch.empty(int(L.bf_op_gate_save_floats(B, Cc, int(w0.shape[1]))), dtype=torch.float32, device=x.device)
    scratch = torch.empty(int(L.bf_op_gate_scratch_floats(B, Cc)) + 2, dtype=torch.float32, device=x.device)
    N.check(L.bf_op_gate_fwd(N.ptr(x), N.ptr(w0), N.ptr(w1), N.ptr(res), N.ptr(out), N.ptr(save), B, H * W, Cc, int(w0.shape[1]),
                             N.ptr(scratch), scratch.numel(), N.stream_ptr(x)), None, "bf_op_gate_fwd")
    return out


def scale_add(res: Optional[torch.Tensor], t: torch.Tensor, m: Optional[torch.Tensor], s: Optional[torch.Tensor] = None) -> torch.Tensor:
    """res + t * m[channel] * s[sample] (each of res / m / s optional)"""
    B, Cc = t.shape[0], t.shape[-1]
    out = torch.empty_like(t)
    N.check(N.lib().bf_op_scale_add(N.ptr(res), N.ptr(t), N.ptr(m), N.ptr(s), N.ptr(out), B, t.numel() // (B * Cc), Cc, N.stream_ptr(t)),
            None, "bf_op_scale_add")
    return out


def _gate_ex(sel, x, x2, res, w0, b0, w1, b1, act0, alpha0, mode):
    B, H, W, Cc = x.shape
    Cs, C8 = sel.shape[-1], int(w0.shape[1])
    L = N.lib()
    out = torch.empty_like(x)
    save = torch.empty(int(L.bf_op_channel_gate_save_floats(B, Cs, Cc, C8)), dtype=torch.float32, device=x.device)
    scratch = torch.empty(int(L.bf_op_gate_scratch_floats(B, Cs)) + 2, dtype=torch.float32, device=x.device)
    N.check(L.bf_op_channel_gate_ex(N.ptr(sel), N.ptr(x), N.ptr(x2), N.ptr(res), N.ptr(out), N.ptr(w0), N.ptr(b0), N.ptr(w1), N.ptr(b1),
                                    N.ptr(save), B, H * W, Cs, Cc, C8, act0, float(alpha0), mode, N.ptr(scratch), scratch.numel(),
                                    N.stream_ptr(x)), None, "bf_op_channel_gate_ex")
    return out


def squeeze_and_excite_block(x: torch.Tensor, w0: torch.Tensor, b0: Optional[torch.Tensor], w1: torch.Tensor, b1: Optional[torch.Tensor],
                             hard_sigmoid_version: bool = False, learn_to_turn_off: bool = False) -> torch.Tensor:
    """bfcnn/backbone_blocks.py:251-313 on a [B,H,W,C] device tensor: GlobalAvgPool -> 1x1 conv (w0 [C, Cs], + b0) -> LeakyReLU(0.1) ->
    1x1 conv (w1 [Cs, C], + b1) -> sigmoid | hard_sigmoid | hard_sigmoid(2.5 - relu(.)); x * gate.  (No builder of the snapshot
    calls it; it is offered as the operator it is.)"""
    mode = (2 if learn_to_turn_off else 0) if hard_sigmoid_version else 1
    return _gate_ex(x, x, None, None, w0, b0, w1, b1, 2, 0.1, mode)


SELECTOR_EPSILON = 1e-3          # DEFAULT_EPSILON (bfcnn/constants.py:7) of global_normalization / local_normalization
SELECTOR_GLOBAL_LEAKY = 0.2      # Dense(activation="leaky_relu"): unresolvable in Keras 2.13, slope 0.2 from Keras 2.15 on


def _selector_prefilter(sel: torch.Tensor, pre: Dict, pre_w: Optional[torch.Tensor], pool, Ct: int) -> torch.Tensor:
    """the optional stages in front of the pooling (custom_layers_selector.py:160-185): Conv2D 1x1 linear -> global_normalization ->
    local_normalization(pool) -> lowpass_filter(4, 4) -> highpass_filter(4, 4) (utilities.py:566-620)"""
    L = N.lib()
    x = sel
    if pre.get("conv1x1"):
        x = UL.pointwise(x, pre_w, Ct)
    B, H, W, C = x.shape
    if pre.get("gn"):            # per sample and channel (x - mean) / sqrt(var + 1e-3): BatchNorm's batch-statistics forward on one sample
        ones, mm, mv = (torch.ones(C, dtype=torch.float32, device=x.device) for _ in range(3))
        save = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        scratch = torch.empty(int(L.bf_op_bn_train_scratch_floats(C)) + 2, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        for b in range(B):
            xb, ob = x[b], out[b]
            N.check(L.bf_op_bn_train_fwd(N.ptr(xb), N.ptr(ones), N.ptr(ob), N.ptr(save), N.ptr(mm), N.ptr(mv), H * W, C, SELECTOR_EPSILON, 0.0, 0,
                                         0.0, N.ptr(scratch), scratch.numel(), N.stream_ptr(x)), None, "bf_op_bn_train_fwd")
        x = out
    if pre.get("ln"):
        def pooled(t):
            o = torch.empty_like(t)
            N.check(L.bf_op_avgpool_same(N.ptr(t), N.ptr(o), B, H, W, C, pool[0], pool[1], 1, 1, N.stream_ptr(t)), None, "bf_op_avgpool_same")
            return o
        mean = pooled(x)
        sq = torch.empty_like(x)
        N.check(L.bf_op_center_scale(N.ptr(x), N.ptr(mean), None, N.ptr(sq), x.numel(), SELECTOR_EPSILON, N.stream_ptr(x)), None, "bf_op_center_scale")
        var = pooled(sq)
        out = torch.empty_like(x)
        N.check(L.bf_op_center_scale(N.ptr(x), N.ptr(mean), N.ptr(var), N.ptr(out), x.numel(), SELECTOR_EPSILON, N.stream_ptr(x)), None,
                "bf_op_center_scale")
        x = out
    for key, high in (("lp", 0), ("hp", 1)):
        if pre.get(key):
            out = torch.empty_like(x)
            N.check(L.bf_op_pass_filter(N.ptr(x), N.ptr(out), x.numel(), 4.0, 4, high, N.stream_ptr(x)), None, "bf_op_pass_filter")
            x = out
    return x


def selector_block(x1: torch.Tensor, x2: torch.Tensor, sel: torch.Tensor, w0: torch.Tensor, w1: torch.Tensor, scale_type: str = "local",
                   activation_type: str = "hard", pool=(32, 32), stride=(8, 8), compress: Optional[int] = None,
                   pre: Optional[Dict] = None, pre_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """selector_block (bfcnn/custom_layers_selector.py:81-330), all four scale types: x1 * s + x2 * (1 - s) with
    s = hard_sigmoid | sigmoid (2.5 - u), u >= 0 computed from the selector layer (through its optional pre-filters)."""
    soft = activation_type == "soft"
    if pre:
        sel = _selector_prefilter(sel, pre, pre_w, pool, int(x1.shape[-1]))
    if scale_type == "global":
        return _gate_ex(sel, x1, x2, None, w0, None, w1, None, 2, SELECTOR_GLOBAL_LEAKY, 3 if soft else 2)
    B, H, W, Cs = sel.shape
    if H % stride[0] or W % stride[1]:
        raise ValueError(f"selector_block LOCAL: the image ({H}x{W}) must be a multiple of the strides {stride} (UpSampling2D restores it)")
    L = N.lib()
    OH, OW = H // stride[0], W // stride[1]

    def pool_map(ph, pw):
        out_ = torch.empty((B, OH, OW, Cs), dtype=torch.float32, device=sel.device)
        N.check(L.bf_op_avgpool_same(N.ptr(sel), N.ptr(out_), B, H, W, Cs, ph, pw, stride[0], stride[1], N.stream_ptr(sel)), None,
                "bf_op_avgpool_same")
        return out_
    pooled = pool_map(pool[0], pool[1])
    if scale_type in ("multiscale", "mixed"):
        if scale_type == "multiscale":          # pool / 2 | pool | 2 pool, same strides (custom_layers_selector.py:203-232)
            parts = [pool_map(pool[0] // 2, pool[1] // 2), pooled, pool_map(pool[0] * 2, pool[1] * 2)]
        else:                                   # local means | the global mean on the same grid (:284-299)
            glob = torch.empty_like(pooled)
            scratch = torch.empty(int(L.bf_op_gate_scratch_floats(B, Cs)) + 2, dtype=torch.float32, device=sel.device)
            N.check(L.bf_op_channel_mean_broadcast(N.ptr(sel), N.ptr(glob), B, H * W, Cs, OH * OW, N.ptr(scratch), scratch.numel(),
                                                   N.stream_ptr(sel)), None, "bf_op_channel_mean_broadcast")
            parts = [pooled, glob]
        cat = torch.empty((B, OH, OW, Cs * len(parts)), dtype=torch.float32, device=sel.device)
        N.check(L.bf_op_concat_channels(N.ptr(parts[0]), N.ptr(parts[1]), N.ptr(parts[2]) if len(parts) > 2 else None, N.ptr(cat),
                                        B * OH * OW, Cs, Cs, Cs if len(parts) > 2 else 0, N.stream_ptr(sel)), None, "bf_op_concat_channels")
        pooled, Cs = cat, Cs * len(parts)
    Ct = x1.shape[-1]
    u = torch.empty((B, OH, OW, Ct), dtype=torch.float32, device=sel.device)
    N.check(L.bf_op_dense2(N.ptr(pooled), N.ptr(w0), None, N.ptr(w1), None, N.ptr(u), B * OH * OW, Cs, Ct, int(w0.shape[1]), 2, 0.3, 4,
                           N.stream_ptr(sel)), None, "bf_op_dense2")
    up = UL.resize_bilinear(u, H, W)                                        # UpSampling2D(strides, bilinear): half-pixel centres
    out = torch.empty_like(x1)
    N.check(L.bf_op_selector_mix(N.ptr(x1), N.ptr(x2), N.ptr(up), N.ptr(out), x1.numel(), int(soft), N.stream_ptr(x1)), None,
            "bf_op_selector_mix")
    return out


class GenericResnetHydra:
    multi_output = False
    auto_exact_fallback = False

    class _Desc:
        def __init__(self, cin, cout):
            self.in_channels, self.out_channels = cin, cout
            self.denormalize = 1

    def __init__(self, config: Dict, device=None, seed: Optional[int] = None):
        bb, dn = config["backbone"], config["denoiser"]
        self.config = config
        if bb.get("use_bias", False):
            raise NotImplementedError("resnet: use_bias is outside the built graph")
        self.add_concat_input = bool(bb.get("add_concat_input", False))                # backbone_resnet.py:277-279
        # add_gelu / add_gradient_dropout / add_mean_sigma_normalization: the builder turns them into gelu_params /
        # gradient_dropout_params / mean_sigma_params (backbone_resnet.py:207-223), which resnet_blocks_full accepts and never reads
        # (backbone_blocks.py:118-125 sets use_mean_sigma / use_gradient_dropout, the block body :164-243 uses neither; gelu_params
        # lands in **kwargs): the graph is the same with or without them.
        self.add_initial_bn = bool(bb.get("add_initial_bn", False))                    # backbone_resnet.py:264-265
        self.add_final_bn = bool(bb.get("add_final_bn", False))                        # :274-275
        self.add_channelwise = bool(bb.get("add_channelwise_scaling", False))          # :236-238, 282-283; backbone_blocks.py:215-217
        self.add_multiplier = bool(bb.get("add_learnable_multiplier", False))          # :240-242, 286-287; backbone_blocks.py:219-221
        self.dropout_rate = float(bb.get("dropout_rate", -1))                          # RandomOnOff (:231-235): identity at inference
        if self.dropout_rate != -1 and not 0.0 <= self.dropout_rate < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        if bb.get("base_conv_params") is not None:
            raise NotImplementedError("resnet: base_conv_params is outside the built graph")
        self.selector = None
        sp = bb.get("selector_params")
        if sp is not None:
            # selector_block (custom_layers_selector.py:81-330) in place of the skip Add (backbone_blocks.py:227-239)
            # optional pre-filters on the selector layer, in the reference's order (custom_layers_selector.py:160-185)
            pre = dict(conv1x1=bool(sp.get("use_conv1x1_selector", False)), gn=bool(sp.get("use_global_normalization", False)),
                       ln=bool(sp.get("use_local_normalization", False)), lp=bool(sp.get("use_lowpass", False)),
                       hp=bool(sp.get("use_highpass", False)))
            st, at = str(sp.get("scale_type", "local")).strip().lower(), str(sp.get("activation_type", "hard")).strip().lower()
            if st not in ("local", "global", "mixed", "multiscale"):
                raise KeyError(st.upper())                                          # ScaleType[...] (custom_layers_selector.py:45)
            if at not in ("hard", "soft"):
                raise KeyError(at.upper())
            pool = tuple(int(v) for v in sp.get("pool_size", (32, 32)))
            stride = tuple(int(v) for v in sp.get("strides_size", (pool[0] / 4, pool[1] / 4)))
            self.selector = dict(scale_type=st, activation_type=at, pool=pool, stride=stride,
                                 compress=max(1, int(round(int(bb.get("filters", 32)) * sp.get("filters_compress_ratio", 0.25)))),
                                 pre=pre if any(pre.values()) else None)
        if dn.get("use_bias", False) or dn.get("use_bn", False) or dn.get("use_ln", False):
            raise NotImplementedError("denoiser head: use_bias / use_bn / use_ln are outside the built graph")
        self.filters = int(bb.get("filters", 32))
        self.kernel_size = int(bb.get("kernel_size", 3))
        self.no_layers = int(bb["no_layers"])
        self.block_kernels = list(bb.get("block_kernels", [3, 3]))
        nb = len(self.block_kernels)
        if nb <= 0:
            raise ValueError("len(block_kernels) must be >= 0 ")                     # backbone_resnet.py:110-113
        if nb > 3:
            raise ValueError("len(block_kernels) must be <= 3")
        self.block_filters = list(bb.get("block_filters", [self.filters] * nb))
        self.block_depthwise = list(bb.get("block_depthwise") or [-1] * nb)
        self.block_groups = list(bb.get("block_groups") or [1] * nb)
        act = bb.get("activation", "relu")
        self.block_activation = list(bb.get("block_activation") or [act] * nb)
        if not (len(self.block_filters) == len(self.block_depthwise) == len(self.block_groups) == len(self.block_activation) == nb):
            raise ValueError("len(block_filters) must == len(block_kernels)")         # :116-126
        self.base_activation = bb.get("base_activation", "linear")
        self.block_activation[-1] = self.base_activation                              # :178
        self.use_bn = bool(bb.get("use_bn", True))
        self.add_gates = bool(bb.get("add_gates", False))
        if self.add_gates and nb < 2:
            raise ValueError("don't know what to do here")                           # backbone_blocks.py:131-141 (gate_no_filters)
        self.in_channels = int(bb["input_shape"][-1])
        vr = bb.get("value_range", [0, 255])
        self.v_min, self.v_max = float(vr[0]), float(vr[1])
        self.head_filters = int(dn.get("filters", 32))
        self.head_activation = dn.get("activation", "linear")
        self.out_channels = int(dn.get("output_channels", 3))
        for a in self.block_activation + [self.base_activation, self.head_activation]:
            UL._act(a)
        # channel bookkeeping + what the operators cover
        ok_c = (32, 64, 128)
        cin = self.filters
        if cin not in ok_c:
            raise NotImplementedError(f"resnet: filters={cin} (32 / 64 / 128 are built here; 16 has its own engine)")
        for kk, cf, dm, g in zip(self.block_kernels, self.block_filters, self.block_depthwise, self.block_groups):
            cout = cin * dm if dm != -1 else cf
            if cout not in ok_c or (dm == -1 and (cin % g or cf % g)):
                raise NotImplementedError(f"resnet: a block convolution {cin}->{cout} (groups {g}) is outside the built operators")
            cin = cout
        if cin != self.filters:
            raise ValueError(f"the last block convolution must produce {self.filters} channels for the residual Add (got {cin})")
        if self.head_filters != 32 or self.kernel_size > 7:
            raise NotImplementedError("resnet: head filters must be 32 and the base kernel at most 7x7")
        self.desc = self._Desc(self.in_channels, self.out_channels)
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self._inventory, self._state_inventory = self._build_inventory()
        self.n_params = sum(int(np.prod(s)) for _, s, _ in self._inventory)
        self.n_state = sum(int(np.prod(s)) for _, s in self._state_inventory)
        self.params = torch.from_numpy(self._initial_values(seed)).to(self.device)
        st = np.concatenate([np.zeros(s, np.float32).ravel() if n.endswith("mean") else np.ones(s, np.float32).ravel()
                             for n, s in self._state_inventory]) if self._state_inventory else np.zeros(0, np.float32)
        self.state = torch.from_numpy(st).to(self.device)
        self.fuse_bottleneck = 1                 # see set_option
        self.arith = 1
        self._packed = None

    # -- inventory ---------------------------------------------------------------------------
    def _build_inventory(self):
        k = self.kernel_size
        out = [("base/kernel", (k, k, self.in_channels, self.filters), "conv")]
        state = []
        if self.add_initial_bn:
            out.append(("initial_bn/gamma", (self.filters,), "bn_gamma"))
            state += [("initial_bn/moving_mean", (self.filters,)), ("initial_bn/moving_variance", (self.filters,))]
        for i in range(self.no_layers):
            cin = self.filters
            for j, (kk, cf, dm, g) in enumerate(zip(self.block_kernels, self.block_filters, self.block_depthwise, self.block_groups)):
                if dm != -1:
                    out.append((f"block{i}/conv{j}/kernel", (kk, kk, cin, dm), "depthwise"))
                    cout = cin * dm
                else:
                    out.append((f"block{i}/conv{j}/kernel", (kk, kk, cin // g, cf), "conv"))
                    cout = cf
                if j >= 1 and self.use_bn:              # the first convolution of a block has no BN (backbone_blocks.py:174-179)
                    out.append((f"block{i}/bn{j}/gamma", (cout,), "bn_gamma"))
                    state += [(f"block{i}/bn{j}/moving_mean", (cout,)), (f"block{i}/bn{j}/moving_variance", (cout,))]
                if j == 1 and self.add_gates:           # two bias-free Dense layers, created right behind the second convolution
                    c8 = max(int(cout / 8), 2)
                    out.append((f"block{i}/gate/dense0/kernel", (cout, c8), "dense"))
                    out.append((f"block{i}/gate/dense1/kernel", (c8, cout), "dense"))
                cin = cout
            # ChannelwiseMultiplier / Multiplier (custom_layers.py:1028-1160): x * relu(w0 + w1); the trainable w0 (zeros at
            # creation) is a parameter, the non-trainable w1 keeps its creation value `multiplier` = 1.0 and is a constant here
            if self.add_channelwise:
                out.append((f"block{i}/channelwise/w0", (cin,), "channelwise"))
            if self.add_multiplier:
                out.append((f"block{i}/multiplier/w0", (1,), "multiplier"))
            if self.selector:
                cs = self.block_filters[0] if self.block_depthwise[0] == -1 else self.filters * self.block_depthwise[0]
                cc = self.selector["compress"]
                if (self.selector["pre"] or {}).get("conv1x1"):       # Conv2D 1x1, linear, to the target filters: created first (:162-169)
                    out.append((f"block{i}/selector/pre/kernel", (1, 1, cs, self.filters), "conv"))
                    cs = self.filters
                if self.selector["scale_type"] != "global":
                    cs *= {"local": 1, "mixed": 2, "multiscale": 3}[self.selector["scale_type"]]
                    out.append((f"block{i}/selector/conv0/kernel", (1, 1, cs, cc), "conv"))
                    out.append((f"block{i}/selector/conv1/kernel", (1, 1, cc, self.filters), "conv"))
                else:
                    out.append((f"block{i}/selector/dense0/kernel", (cs, cc), "dense"))
                    out.append((f"block{i}/selector/dense1/kernel", (cc, self.filters), "dense"))
        if self.add_final_bn:
            out.append(("final_bn/gamma", (self.filters,), "bn_gamma"))
            state += [("final_bn/moving_mean", (self.filters,)), ("final_bn/moving_variance", (self.filters,))]
        cf = self.filters + (self.in_channels if self.add_concat_input else 0)   # Concatenate([x, backbone input]) ahead of the closing layers
        if self.add_channelwise:
            out.append(("channelwise/w0", (cf,), "channelwise"))
        if self.add_multiplier:
            out.append(("multiplier/w0", (1,), "multiplier"))
        out.append(("head/conv0/kernel", (1, 1, cf, self.head_filters), "conv"))
        out.append(("head/conv1/kernel", (1, 1, self.head_filters, self.out_channels), "conv"))
        return out, state

    @property
    def trainable_variables(self):
        o, res = 0, []
        for name, shape, kind in self._inventory:
            res.append((name, shape, kind, o))
            o += int(np.prod(shape))
        return res

    @property
    def non_trainable_variables(self):
        o, res = 0, []
        for name, shape in self._state_inventory:
            res.append((name, shape, o))
            o += int(np.prod(shape))
        return res

    def count_params(self) -> int:
        return self.n_params

    def _initial_values(self, seed) -> np.ndarray:
        from .model import glorot_normal
        rng = np.random.default_rng(seed)
        init = lambda s, kind: np.ones(s) if kind == "bn_gamma" else (np.zeros(s) if kind in ("channelwise", "multiplier") else
            (glorot_normal((1, 1) + tuple(s), rng).reshape(s) if kind == "dense" else glorot_normal(s, rng)))
        return np.concatenate([np.asarray(init(s, kind), np.float32).ravel() for _, s, kind in self._inventory])

    def get_weights(self):
        return self.params.detach().cpu().numpy(), self.state.detach().cpu().numpy()

    def set_weights(self, params: np.ndarray, state: Optional[np.ndarray] = None):
        params = np.ascontiguousarray(params, np.float32).ravel()
        if params.size != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {params.size}")
        self.params.copy_(torch.from_numpy(params))
        if state is not None:
            state = np.ascontiguousarray(state, np.float32).ravel()
            if state.size != self.n_state:
                raise ValueError(f"expected {self.n_state} state values, got {state.size}")
            self.state.copy_(torch.from_numpy(state))
        self.mark_dirty()

    def mark_dirty(self):
        """parameters or moving statistics changed in place (optimizer / training step): drop the folded operands"""
        self._packed = None
        self.version = getattr(self, "version", 0) + 1

    def set_option(self, key: str, value: int):
        """fuse_bottleneck: 1 (default) a block of the shape 1x1 32 -> 32, depthwise 3x3 x4, 1x1 128 -> 32 runs as one kernel with split-f16
        GEMMs (bf_op_bneck_block_h3); 0: the fp32 operators (bf_op_pointwise, bf_op_dwmult_pointwise).
        arith: 1 (default) the base convolution (k x k, 3 -> 32, k = 3 / 5 / 7) on the f16 matrix cores with split-f16 operands
        (bf_op_first_conv_h3k) and the fused bottleneck block where it applies; 0: every product in exact fp32."""
        if key not in ("fuse_bottleneck", "arith") or int(value) not in (0, 1):
            raise ValueError(f"unknown option {key}={value}")
        setattr(self, key, int(value))
        self.version = getattr(self, "version", 0) + 1

    def check_status(self, raise_on_overflow: bool = True) -> bool:
        return True

    # -- packing (host arithmetic on the weights only: BatchNorm folding, block-diagonal grouped kernels) ----------------
    def _pack(self):
        if self._packed is not None:
            return self._packed
        w, st = self.get_weights()
        W = {n: w[o:o + int(np.prod(s))].reshape(s).astype(np.float64) for n, s, _, o in self.trainable_variables}
        S = {n: st[o:o + int(np.prod(s))].reshape(s).astype(np.float64) for n, s, o in self.non_trainable_variables}
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        P = {"base": dev(W["base/kernel"])}
        homogeneous = lambda a: UL._act(a)[0] in (0, 1, 2)                 # act(s z) = s act(z) for s >= 0: linear, relu, leaky relu

        def bn_affine(base):
            sc = W[base + "/gamma"] / np.sqrt(S[base + "/moving_variance"] + BN_EPSILON)
            return sc, -sc * S[base + "/moving_mean"]

        def end_scale(prefix, n):
            """relu(w0 + 1) of the ChannelwiseMultiplier times that of the Multiplier behind it: a factor >= 0 per channel"""
            sc = np.ones(n)
            if self.add_channelwise:
                sc = sc * np.maximum(W[prefix + "channelwise/w0"] + 1.0, 0.0)
            if self.add_multiplier:
                sc = sc * np.maximum(W[prefix + "multiplier/w0"] + 1.0, 0.0)
            return sc
        if self.add_initial_bn:                                           # per-channel affine behind the base convolution's activation
            sc, sh = bn_affine("initial_bn")
            P["initial_bn"] = (dev(sc.reshape(1, 1, -1, 1)), dev(sh))
        nb_ = len(self.block_kernels)
        for i in range(self.no_layers):
            cin = self.filters
            raw = []                                                      # (kind, folded fp32 kernel, shift) per convolution
            for j, (kk, cf, dm, g) in enumerate(zip(self.block_kernels, self.block_filters, self.block_depthwise, self.block_groups)):
                k = W[f"block{i}/conv{j}/kernel"]
                cout = cin * dm if dm != -1 else cf
                scale, shift = np.ones(cout), None
                if j >= 1 and self.use_bn:              # inference BN folded: y = gamma (x - mean) / sqrt(var + eps), center=False
                    base = f"block{i}/bn{j}"
                    scale = W[base + "/gamma"] / np.sqrt(S[base + "/moving_variance"] + BN_EPSILON)
                    shift = -scale * S[base + "/moving_mean"]
                if j == nb_ - 1 and (self.add_channelwise or self.add_multiplier):
                    # the block's closing multipliers are >= 0 and its last activation is base_activation: folded into the last
                    # convolution's scale / shift when that commutes (and no gate sits between), else applied as their own pass
                    es = end_scale(f"block{i}/", cout)
                    # (a one-convolution block with a selector: that convolution's output is also the selector layer, which the
                    # reference takes BEFORE the multipliers -- backbone_blocks.py:174-179 vs 215-221 -- so nothing may ride on it)
                    if homogeneous(self.block_activation[j]) and not (self.add_gates and j == 1) and not (self.selector and nb_ == 1):
                        scale, shift = scale * es, (None if shift is None else shift * es)
                    else:
                        P[f"b{i}scale"] = dev(es)
                if dm != -1:
                    kf = (k * scale.reshape(cin, dm)[None, None]).reshape(kk, kk, cin * dm)
                    P[f"b{i}c{j}"] = ("dw", dev(kf.reshape(kk, kk, cin, dm)), None if shift is None else dev(shift))
                    raw.append(("dw", kf.reshape(kk, kk, cin, dm), shift))
                else:
                    dense = np.zeros((kk, kk, cin, cf))     # grouped convolution as a block-diagonal dense one
                    ci_g, co_g = cin // g, cf // g
                    for gi in range(g):
                        dense[:, :, gi * ci_g:(gi + 1) * ci_g, gi * co_g:(gi + 1) * co_g] = k[:, :, :, gi * co_g:(gi + 1) * co_g]
                    dense = dense * scale[None, None, None, :]
                    packed = UL.pack_pointwise(dev(dense[0, 0])) if kk == 1 else UL.pack_conv(dev(dense))
                    P[f"b{i}c{j}"] = ("pw" if kk == 1 else "conv", packed, None if shift is None else dev(shift))
                    raw.append(("pw" if kk == 1 else "conv", dense, shift))
                if j == 1 and self.add_gates:
                    P[f"b{i}gate"] = (dev(W[f"block{i}/gate/dense0/kernel"]), dev(W[f"block{i}/gate/dense1/kernel"]))
                cin = cout
            if ([r[0] for r in raw] == ["pw", "dw", "pw"] and raw[0][1].shape[2:] == (32, 32) and raw[1][1].shape == (3, 3, 32, 4)
                    and raw[2][1].shape[2:] == (128, 32) and not self.add_gates and not self.selector and f"b{i}scale" not in P
                    and all(a in ("linear", "relu") or a.startswith("leaky") for a in self.block_activation[:3])):
                # the shipped bottleneck shape: the whole block as one kernel
                P[f"b{i}bneck"] = (UL.pack_bneck_h3(dev(raw[0][1][0, 0]), dev(raw[1][1]), dev(raw[2][1][0, 0])),
                                   tuple(None if r[2] is None else dev(r[2]) for r in raw))
            if self.selector:
                kind = "dense" if self.selector["scale_type"] == "global" else "conv"
                w0, w1 = W[f"block{i}/selector/{kind}0/kernel"], W[f"block{i}/selector/{kind}1/kernel"]
                P[f"b{i}sel"] = (dev(w0.reshape(w0.shape[-2], w0.shape[-1])), dev(w1.reshape(w1.shape[-2], w1.shape[-1])))
                if f"block{i}/selector/pre/kernel" in W:
                    P[f"b{i}selpre"] = UL.pack_pointwise(dev(W[f"block{i}/selector/pre/kernel"][0, 0]))
        w_head0 = W["head/conv0/kernel"][0, 0]
        cf = w_head0.shape[0]
        if self.add_final_bn and not self.add_concat_input:               # BN, then the closing multipliers: one per-channel affine
            sc, sh = bn_affine("final_bn")
            es = end_scale("", self.filters)
            P["final_affine"] = (dev((sc * es).reshape(1, 1, -1, 1)), dev(sh * es))
        else:
            if self.add_final_bn:                                         # the input joins behind the BN: the multipliers cannot ride on it
                sc, sh = bn_affine("final_bn")
                P["final_affine"] = (dev(sc.reshape(1, 1, -1, 1)), dev(sh))
            if self.add_channelwise or self.add_multiplier:               # no shift: the head's first 1x1 absorbs the factor (rows of W)
                w_head0 = w_head0 * end_scale("", cf)[:, None]
        if self.add_concat_input:                                         # zero rows up to the channel count the head kernel takes
            self._head_cin = next(c for c in (32, 64, 128, 256) if c >= cf)
            w_head0 = np.concatenate([w_head0, np.zeros((self._head_cin - cf, w_head0.shape[1]))], axis=0)
        P["head0"] = UL.pack_pointwise(dev(w_head0))
        P["head1"] = dev(W["head/conv1/kernel"])
        self._packed = P
        return P

    # -- forward -----------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("resnet inference needs the GPU: there is no CPU execution path")

    def _features(self, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
        P = self._pack()
        f = UL.first_conv(x, P["base"], H, W, self.base_activation, True, self.v_min, self.v_max, arith=self.arith)
        if self.add_initial_bn:
            f = UL.dwconv_mult(f, P["initial_bn"][0], P["initial_bn"][1])
        nb = len(self.block_kernels)
        fused = {(32, 4, 32, 3), (32, 2, 32, 3), (64, 2, 64, 3), (32, 4, 64, 3), (32, 1, 32, 3), (64, 1, 64, 3)}   # built instances
        for i in range(self.no_layers):
            if self.arith and self.fuse_bottleneck and f"b{i}bneck" in P:
                packed, (s0, s1, s2) = P[f"b{i}bneck"]
                f = UL.bneck_block_h3(f, packed, s0, self.block_activation[0], s1, self.block_activation[1], s2, self.block_activation[2])
                continue
            t = f
            first = None
            j = 0
            tail = P.get(f"b{i}scale")                       # closing multipliers that could not be folded: own pass, carries the Add
            while j < nb:
                kind, wp, shift = P[f"b{i}c{j}"]
                gate_here = self.add_gates and j == 1
                if kind == "dw" and not gate_here and j + 1 < nb and P[f"b{i}c{j + 1}"][0] == "pw" and \
                        (t.shape[-1], int(wp.shape[-1]), self.block_filters[j + 1], int(wp.shape[0])) in fused:
                    # depthwise (+BN, act) and the 1x1 after it (+BN, act, +skip) in one kernel: the wide tensor stays on chip
                    _, wp2, shift2 = P[f"b{i}c{j + 1}"]
                    t = UL.dwmult_pointwise(t, wp, shift, self.block_activation[j], wp2, self.block_filters[j + 1], shift2,
                                            self.block_activation[j + 1],
                                            f if j + 1 == nb - 1 and not self.selector and tail is None else None)
                    j += 2
                    continue
                res = f if j == nb - 1 and not gate_here and not self.selector and tail is None else None   # Add(block output, block input) (:242)
                a = self.block_activation[j]
                if kind == "dw":
                    t = UL.dwconv_mult(t, wp, shift, a)
                    if j == nb - 1 and not gate_here and not self.selector and tail is None:
                        raise NotImplementedError("a depthwise convolution as the last convolution of a block")
                elif kind == "pw":
                    cout = self.block_filters[j]
                    t = UL.pointwise_ex(t, wp, cout, 3, a, mult=shift, res=res)
                else:
                    t = UL.conv2d(t, wp, self.block_filters[j], self.block_kernels[j], 1, a, res=res, bias=shift)
                if gate_here:                                # x * hard_sigmoid(relu(mean(x) W0) W1) [+ skip when the block ends here]
                    t = channel_gate(t, *P[f"b{i}gate"], res=f if j == nb - 1 and not self.selector and tail is None else None)
                if j == 0:
                    first = t                                # x_1st_conv: the selector layer (backbone_blocks.py:229-231)
                j += 1
            if tail is not None:
                t = scale_add(None if self.selector else f, t, tail)
            f = selector_block(f, t, first, *P[f"b{i}sel"], pre_w=P.get(f"b{i}selpre"), **self.selector) if self.selector else t
        if self.add_final_bn:
            f = UL.dwconv_mult(f, P["final_affine"][0], P["final_affine"][1])
        if self.add_concat_input:
            B, Hs, Ws, cin = x.shape
            cat = torch.empty((B, H, W, self._head_cin), dtype=torch.float32, device=f.device)
            N.check(N.lib().bf_op_concat_input(N.ptr(f), N.ptr(x), int(x.dtype == torch.uint8), N.ptr(cat), B, H, W, Hs, Ws, self.filters, cin,
                                               self._head_cin, self.v_min, self.v_max, N.stream_ptr(f)), None, "bf_op_concat_input")
            f = cat
        return f

    def _as_device(self, x):
        was_numpy = isinstance(x, np.ndarray)
        if was_numpy:
            x = torch.from_numpy(np.ascontiguousarray(x))
        if x.dim() != 4 or x.shape[-1] != self.in_channels:
            raise ValueError(f"expected [B,H,W,{self.in_channels}], got {tuple(x.shape)}")
        if x.dtype != torch.uint8:
            x = x.to(torch.float32)
        return x.to(self.device).contiguous(), was_numpy

    def __call__(self, x, training: bool = False):
        if training:
            raise NotImplementedError("hydra(x, training=True) on its own is not built here; use train_loop's train_step_single_gpu")
        self._require_gpu()
        x, was_numpy = self._as_device(x)
        B, H, W, _ = x.shape
        P = self._pack()
        out = UL.head_fused(self._features(x, H, W), None, P["head0"], self.head_activation, P["head1"], H, W, False, True,
                            self.v_min, self.v_max, arith=self.arith)
        if was_numpy:
            torch.cuda.synchronize(self.device)
            return out.cpu().numpy()
        return out

    def predict(self, x):
        return self(x)

    def infer_u8(self, image: torch.Tensor, cast_to_uint8: bool = True) -> torch.Tensor:
        from .utilities import next_power_of_2
        self._require_gpu()
        B, Hs, Ws, _ = image.shape
        H, W = next_power_of_2(Hs), next_power_of_2(Ws)
        P = self._pack()
        return UL.head_fused(self._features(image, H, W), None, P["head0"], self.head_activation, P["head1"], Hs, Ws, bool(cast_to_uint8), True,
                             self.v_min, self.v_max, arith=self.arith)
