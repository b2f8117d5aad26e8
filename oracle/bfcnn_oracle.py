"""
CPU oracle for the bfcnn resnet-denoiser hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy (float64 by default) restatement of the algorithm the
reference implements with TensorFlow/Keras 2.13 ops.  It is imported only by
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`;
the product path (`blind_image_denoising_amd`) never imports it and fails loudly
when the HIP library is missing.

PARITY STATUS
  * conv / batch-norm / head / loss / Adam numerics: **parity unpinned** by the
    reference itself.  TensorFlow is not installed in the build image
    (`import tensorflow` -> ModuleNotFoundError, an ordinary Python error, nothing
    was refused), the reference ships no golden tensors for these ops, and its
    pretrained resnet weights are not in the snapshot.  The restatement is pinned
    instead by an independent second implementation (PyTorch-CPU float64,
    `tests/test_oracle_vs_torch.py`) and by algebraic known-answer tests.
  * pyramid split/merge: pinned by the reference's own round-trip identity
    `mean(abs(inverse(pyramid(x)) - x)) < 1e-7` (tests/bfcnn/test_pyramid.py:22-409)
    on the reference's own fixture image (images/test/etc/lena.jpg).
  * shape contracts: tests/bfcnn/test_model_denoiser.py:19-70.

Every function cites the reference file:line (relative to /root/reference) it
follows.  TF/Keras op semantics restated here are listed in SURVEY.md appendix A.
"""
from __future__ import annotations

import copy
import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# bfcnn/constants.py:7-13
DEFAULT_EPSILON = 1e-3
DEFAULT_BN_EPSILON = 1e-3
DEFAULT_BN_MOMENTUM = 0.995

F64 = np.float64


# ---------------------------------------------------------------------------
# elementary TF op restatements
# ---------------------------------------------------------------------------

def same_pads(n: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF "SAME" padding: returns (out, pad_before, pad_after); extra pad goes
    to the bottom/right (keras Conv2D / AveragePooling2D padding="same";
    call sites bfcnn/utilities.py:196, bfcnn/pyramid.py:266-270)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    before = total // 2
    return out, before, total - before


def conv2d_same(x: np.ndarray, w: np.ndarray, stride: int = 1) -> np.ndarray:
    """keras.layers.Conv2D(padding="same", use_bias=False) on NHWC input with a
    HWIO kernel: cross-correlation, zero padding (bfcnn/utilities.py:196)."""
    B, H, W, C = x.shape
    kh, kw, ci, co = w.shape
    assert ci == C, (ci, C)
    oh, pt, pb = same_pads(H, kh, stride)
    ow, pl, pr = same_pads(W, kw, stride)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    y = np.zeros((B, oh, ow, co), dtype=np.result_type(x.dtype, w.dtype))
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (oh - 1) * stride + 1:stride,
                       j:j + (ow - 1) * stride + 1:stride, :]
            y += patch @ w[i, j]
    return y


def conv2d_same_grad_input(dy: np.ndarray, w: np.ndarray) -> np.ndarray:
    """d(conv2d_same stride 1, odd k)/dx = SAME cross-correlation of dy with the
    spatially flipped, in/out-transposed kernel (what tf.GradientTape yields for
    bfcnn/train_loop.py:302-304)."""
    wt = np.ascontiguousarray(np.transpose(w[::-1, ::-1, :, :], (0, 1, 3, 2)))
    return conv2d_same(dy, wt)


def conv2d_same_grad_kernel(x: np.ndarray, dy: np.ndarray, kh: int, kw: int) -> np.ndarray:
    """d(conv2d_same stride 1)/dw: dW[i,j,ci,co] = sum_{b,y,x} xpad[b,y+i,x+j,ci] dy[b,y,x,co]."""
    B, H, W, C = x.shape
    co = dy.shape[-1]
    _, pt, pb = same_pads(H, kh, 1)
    _, pl, pr = same_pads(W, kw, 1)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    dw = np.zeros((kh, kw, C, co), dtype=np.result_type(x.dtype, dy.dtype))
    dyf = dy.reshape(-1, co)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + H, j:j + W, :].reshape(-1, C)
            dw[i, j] = patch.T @ dyf
    return dw


def activation_fwd(x: np.ndarray, name: Optional[str]) -> np.ndarray:
    """bfcnn/utilities.py:229-267 (activation_wrapper); only the activations the
    hot path can reach are restated."""
    name = (name or "linear").lower().strip()
    if name == "linear":
        return x
    if name == "relu":
        return np.maximum(x, 0.0)
    if name in ("leakyrelu", "leaky_relu"):
        return np.where(x > 0, x, 0.3 * x)
    if name in ("leakyrelu_01", "leaky_relu_01"):
        return np.where(x > 0, x, 0.1 * x)
    if name in ("leaky_relu_001", "leakyrelu_001"):
        return np.where(x > 0, x, 0.01 * x)
    if name == "tanh":
        return np.tanh(x)
    if name == "gelu":                     # keras "gelu": the exact erf form (approximate=False)
        from scipy.special import erf
        return 0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))
    raise NotImplementedError(f"activation [{name}] is outside the hot path")


def activation_bwd(pre: np.ndarray, dy: np.ndarray, name: Optional[str]) -> np.ndarray:
    name = (name or "linear").lower().strip()
    if name == "linear":
        return dy
    if name == "relu":
        return dy * (pre > 0)
    if name in ("leakyrelu", "leaky_relu"):
        return dy * np.where(pre > 0, 1.0, 0.3)
    if name in ("leakyrelu_01", "leaky_relu_01"):
        return dy * np.where(pre > 0, 1.0, 0.1)
    if name in ("leaky_relu_001", "leakyrelu_001"):
        return dy * np.where(pre > 0, 1.0, 0.01)
    if name == "tanh":
        return dy * (1.0 - np.tanh(pre) ** 2)
    raise NotImplementedError(name)


def layer_normalize(x: np.ndarray, v_min: float, v_max: float) -> np.ndarray:
    """bfcnn/utilities.py:449-461."""
    return (np.clip(x, v_min, v_max) - v_min) / (v_max - v_min) - 0.5


def layer_denormalize(y: np.ndarray, v_min: float, v_max: float) -> np.ndarray:
    """bfcnn/utilities.py:435-443 (+ clip_normalized_tensor :23-36)."""
    return (np.clip(y, -0.5, 0.5) + 0.5) * (v_max - v_min) + v_min


def pow2_target(n: int) -> int:
    """bfcnn/utilities.py:736-751: 2**ceil(log(n)/log(2)) evaluated in float32
    like the reference does (tf.math.log on a float32 cast)."""
    v = np.float32(np.log(np.float32(n))) / np.float32(np.log(np.float32(2.0)))
    return int(2 ** int(np.ceil(np.float32(v))))


def pad_to_power_of_2(x: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """bfcnn/utilities.py:736-751: zero pad bottom/right."""
    H, W = x.shape[1], x.shape[2]
    ph, pw = pow2_target(H) - H, pow2_target(W) - W
    return np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0))), ph, pw


def remove_padding(x: np.ndarray, ph: int, pw: int) -> np.ndarray:
    """bfcnn/utilities.py:755-764."""
    return x[:, :x.shape[1] - ph, :x.shape[2] - pw, :]


def round_half_even(x: np.ndarray) -> np.ndarray:
    """tf.round (bfcnn/module_denoiser.py:72)."""
    return np.rint(x)


# ---------------------------------------------------------------------------
# batch normalisation (keras BatchNormalization(scale=True, center=False))
# ---------------------------------------------------------------------------

def bn_infer(x, gamma, mean, var, eps=DEFAULT_BN_EPSILON, beta=None):
    """Inference mode: gamma*(x-moving_mean)*rsqrt(moving_var+eps) (+beta).
    Config built at bfcnn/backbone_resnet.py:129-135, applied utilities.py:207-208."""
    y = gamma * (x - mean) / np.sqrt(var + eps)
    return y if beta is None else y + beta


def bn_train(x, gamma, mov_mean, mov_var, eps=DEFAULT_BN_EPSILON,
             momentum=DEFAULT_BN_MOMENTUM, beta=None):
    """Training mode: batch mean / biased batch variance over (N,H,W); moving
    stats updated with momentum; keras' fused kernel feeds the Bessel-corrected
    variance into the moving-variance update.  Returns (y, cache, new_mean, new_var)."""
    n = x.shape[0] * x.shape[1] * x.shape[2]
    mu = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mu) * inv
    y = gamma * xhat
    if beta is not None:
        y = y + beta
    unbiased = var * (n / max(n - 1, 1))
    new_mean = mov_mean * momentum + mu * (1.0 - momentum)
    new_var = mov_var * momentum + unbiased * (1.0 - momentum)
    return y, (xhat, inv, mu, var), new_mean, new_var


def bn_train_bwd(dy, gamma, cache):
    """Standard BN backward w.r.t. x and gamma (no beta, center=False)."""
    xhat, inv, _, _ = cache
    n = dy.shape[0] * dy.shape[1] * dy.shape[2]
    dgamma = (dy * xhat).sum(axis=(0, 1, 2))
    dsum = dy.sum(axis=(0, 1, 2))
    dx = (gamma * inv) * (dy - dsum / n - xhat * (dgamma / n))
    return dx, dgamma, dsum


# ---------------------------------------------------------------------------
# model spec (canonical resnet_color_1xN_bn_16x3x3)
# ---------------------------------------------------------------------------

def _fix_shape(shape):
    """bfcnn/utilities.py:89-96 (input_shape_fixer)."""
    return [None if s in ("?", "", "-1") else s for s in shape]


@dataclass
class ResnetSpec:
    """The subset of bfcnn/backbone_resnet.py:19-50 + bfcnn/model.py:251-275
    arguments that the hot path uses."""
    in_channels: int = 3
    filters: int = 16
    kernel_size: int = 3
    no_layers: int = 6
    block_kernels: Tuple[int, ...] = (3, 3)
    block_filters: Tuple[int, ...] = (16, 16)
    activation: str = "relu"
    base_activation: str = "linear"
    use_bn: bool = True
    use_bias: bool = False
    kernel_regularizer: Optional[str] = "l1"
    block_regularizer: Optional[Tuple[Optional[str], ...]] = None
    head_filters: int = 32
    head_activation: str = "linear"
    out_channels: int = 3
    head_regularizer: Optional[str] = "l2"
    v_min: float = 0.0
    v_max: float = 255.0
    bn_eps: float = DEFAULT_BN_EPSILON
    bn_momentum: float = DEFAULT_BN_MOMENTUM
    denormalize: bool = True   # False = literal single-output snapshot graph (model.py:110-116)

    @staticmethod
    def from_config(model_config: Dict, strict_snapshot: bool = False) -> "ResnetSpec":
        """model_config = config["model"] of a pipeline JSON (bfcnn/model.py:58-66)."""
        bb = copy.deepcopy(model_config["backbone"])
        hd = copy.deepcopy(model_config.get("denoiser", {}))
        if bb.get("type", "resnet").strip().lower() != "resnet":
            raise ValueError("oracle restates the resnet backbone only")
        shape = _fix_shape(list(bb.get("input_shape", [None, None, 1])))
        vr = bb.get("value_range", (0, 255))
        bk = tuple(bb.get("block_kernels", [3, 3]))
        bf = tuple(bb.get("block_filters", [32, 32]))
        if len(bk) != len(bf) or not (1 <= len(bk) <= 3):
            raise ValueError("block_kernels/block_filters length")  # backbone_resnet.py:111-118
        kr = bb.get("kernel_regularizer", "l1")
        br = bb.get("block_regularizer") or [kr] * len(bk)
        return ResnetSpec(
            in_channels=int(shape[-1]), filters=int(bb["filters"]),
            kernel_size=int(bb["kernel_size"]), no_layers=int(bb["no_layers"]),
            block_kernels=bk, block_filters=bf,
            activation=bb.get("activation", "relu"),
            base_activation=bb.get("base_activation", "linear"),
            use_bn=bool(bb.get("use_bn", True)), use_bias=bool(bb.get("use_bias", False)),
            kernel_regularizer=kr, block_regularizer=tuple(br),
            head_filters=int(hd.get("filters", 32)),
            head_activation=hd.get("activation", "linear"),
            out_channels=int(hd.get("output_channels", 3)),
            head_regularizer=hd.get("kernel_regularizer", "l2"),
            v_min=float(vr[0]), v_max=float(vr[1]),
            denormalize=not strict_snapshot)

    # -- parameter inventory, in keras variable-creation order ------------
    def tensors(self) -> List[Tuple[str, Tuple[int, ...], str, Optional[str]]]:
        """[(name, shape, kind, regulariser)], kind in {conv, gamma}.  Order = the
        order keras creates trainable variables: base conv, then per block
        conv1, conv2, bn-gamma[, conv3, bn-gamma], then the two head convs."""
        F = self.filters
        k = self.kernel_size
        out = [("base/kernel", (k, k, self.in_channels, F), "conv", self.kernel_regularizer)]
        nb = len(self.block_kernels)
        breg = self.block_regularizer or tuple([self.kernel_regularizer] * nb)
        for i in range(self.no_layers):
            cin = F
            for j in range(nb):
                kk, cf = self.block_kernels[j], self.block_filters[j]
                out.append((f"block{i}/conv{j}/kernel", (kk, kk, cin, cf), "conv", breg[j]))
                if j >= 1 and self.use_bn:
                    out.append((f"block{i}/bn{j}/gamma", (cf,), "gamma", None))
                cin = cf
        out.append(("head/conv0/kernel", (1, 1, F, self.head_filters), "conv", self.head_regularizer))
        out.append(("head/conv1/kernel", (1, 1, self.head_filters, self.out_channels), "conv",
                    self.head_regularizer))
        return out

    def state_tensors(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """Non-trainable BN moving statistics: per BN, moving_mean then moving_variance."""
        out = []
        if not self.use_bn:
            return out
        for i in range(self.no_layers):
            for j in range(1, len(self.block_kernels)):
                cf = self.block_filters[j]
                out.append((f"block{i}/bn{j}/moving_mean", (cf,)))
                out.append((f"block{i}/bn{j}/moving_variance", (cf,)))
        return out

    def param_count(self) -> int:
        return sum(int(np.prod(s)) for _, s, _, _ in self.tensors())

    def state_count(self) -> int:
        return sum(int(np.prod(s)) for _, s in self.state_tensors())

    def offsets(self) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
        off, o = {}, 0
        for name, shape, _, _ in self.tensors():
            off[name] = (o, shape)
            o += int(np.prod(shape))
        return off

    def state_offsets(self) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
        off, o = {}, 0
        for name, shape in self.state_tensors():
            off[name] = (o, shape)
            o += int(np.prod(shape))
        return off


def canonical_config(no_layers: int = 6, filters: int = 16, kernel_size: int = 3) -> Dict:
    """`resnet_color_1xN_bn_16x3x3` as derived in SURVEY.md section 8 (no JSON for it
    ships; naming by analogy with configs/resnet_color_1x6_bn_32x128x32_1x3x1_*.json)."""
    return {
        "model": {
            "backbone": {
                "type": "resnet", "input_shape": ["?", "?", 3], "filters": filters,
                "kernel_size": kernel_size, "no_layers": no_layers,
                "block_kernels": [3, 3], "block_filters": [filters, filters],
                "activation": "relu", "use_bn": True, "use_bias": False,
                "kernel_regularizer": "l1", "kernel_initializer": "glorot_normal",
                "value_range": [0, 255]},
            "denoiser": {"use_bias": False, "output_channels": 3, "kernel_regularizer": "l2",
                         "kernel_initializer": "glorot_normal"}},
        "train": {"optimizer": {"type": "ADAM", "gradient_clipping_by_norm": 1.0,
                                "schedule": {"type": "exponential_decay",
                                             "config": {"decay_rate": 0.9, "decay_steps": 40000,
                                                        "learning_rate": 0.001}}}},
        "loss": {"hinge": 0.5, "cutoff": 255.0, "mae_multiplier": 1.0, "ssim_multiplier": 0.0,
                 "mse_multiplier": 0.0, "regularization": 0.01},
    }


def glorot_normal(shape: Sequence[int], rng: np.random.Generator) -> np.ndarray:
    """keras "glorot_normal" (bfcnn/backbone_resnet.py:36): truncated normal
    (resample outside 2 sigma), sigma = sqrt(2/(fan_in+fan_out))/0.87962566103423978."""
    kh, kw, ci, co = shape
    fan_in, fan_out = kh * kw * ci, kh * kw * co
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    w = rng.standard_normal(shape)
    bad = np.abs(w) > 2.0
    while bad.any():
        w[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(w) > 2.0
    return (w * std).astype(np.float32)


def init_params(spec: ResnetSpec, seed: int = 42, nontrivial_bn: bool = True
                ) -> Tuple[np.ndarray, np.ndarray]:
    """Flat float32 (params, state).  Weights: glorot normal; gamma = 1.  With
    nontrivial_bn the moving stats are drawn (mean ~ N(0,0.05^2), var ~ U[0.5,1.5])
    so that the folded BN shift is exercised; otherwise keras defaults (0, 1)."""
    rng = np.random.default_rng(seed)
    parts = []
    for _, shape, kind, _ in spec.tensors():
        parts.append(glorot_normal(shape, rng).ravel() if kind == "conv"
                     else np.ones(shape, np.float32))
    params = np.concatenate(parts).astype(np.float32)
    sparts = []
    for name, shape in spec.state_tensors():
        if name.endswith("moving_mean"):
            sparts.append((rng.standard_normal(shape) * 0.05 if nontrivial_bn
                           else np.zeros(shape)).astype(np.float32))
        else:
            sparts.append((rng.uniform(0.5, 1.5, shape) if nontrivial_bn
                           else np.ones(shape)).astype(np.float32))
    state = np.concatenate(sparts).astype(np.float32) if sparts else np.zeros(0, np.float32)
    return params, state


def _views(spec: ResnetSpec, flat: np.ndarray, dtype=F64) -> Dict[str, np.ndarray]:
    return {n: flat[o:o + int(np.prod(s))].reshape(s).astype(dtype)
            for n, (o, s) in spec.offsets().items()}


def _state_views(spec: ResnetSpec, flat: np.ndarray, dtype=F64) -> Dict[str, np.ndarray]:
    return {n: flat[o:o + int(np.prod(s))].reshape(s).astype(dtype)
            for n, (o, s) in spec.state_offsets().items()}


# ---------------------------------------------------------------------------
# hydra forward  (normalise -> backbone -> head -> [denormalise])
# ---------------------------------------------------------------------------

def hydra_forward(spec: ResnetSpec, params: np.ndarray, state: np.ndarray, x: np.ndarray,
                  training: bool = False, dtype=F64, want_cache: bool = False):
    """bfcnn/model.py:91-151 (hydra) = normalizer (model.py:364-394) ->
    resnet backbone (backbone_resnet.py:250-298; blocks backbone_blocks.py:167-246;
    conv order conv->BN->activation utilities.py:196-215) -> denoiser head
    (model.py:297-342) -> denormalise (model.py:136-139; see ResnetSpec.denormalize).

    x: float [B,H,W,C] in value range.  Returns y (and, if training, new_state; and a
    cache for the backward pass when want_cache)."""
    P = _views(spec, params, dtype)
    S = _state_views(spec, state, dtype)
    new_state = {k: v.copy() for k, v in S.items()}
    cache = {"blocks": []}
    x = x.astype(dtype)
    xn = layer_normalize(x, spec.v_min, spec.v_max)
    cache["clip_in"] = (x >= spec.v_min) & (x <= spec.v_max)
    cache["xn"] = xn
    pre = conv2d_same(xn, P["base/kernel"])
    f = activation_fwd(pre, spec.base_activation)        # base conv: no BN (backbone_resnet.py:258-262)
    cache["base_pre"] = pre
    nb = len(spec.block_kernels)
    for i in range(spec.no_layers):
        blk = {"x": f, "convs": []}
        t = f
        for j in range(nb):
            last = j == nb - 1
            act = spec.base_activation if last else spec.activation   # backbone_resnet.py:178
            c = conv2d_same(t, P[f"block{i}/conv{j}/kernel"])
            ent = {"in": t, "conv": c, "act": act}
            if j >= 1 and spec.use_bn:                  # first conv has bn_params=None (backbone_blocks.py:174-179)
                g = P[f"block{i}/bn{j}/gamma"]
                mk, vk = f"block{i}/bn{j}/moving_mean", f"block{i}/bn{j}/moving_variance"
                if training:
                    c, bnc, nm, nv = bn_train(c, g, S[mk], S[vk], spec.bn_eps, spec.bn_momentum)
                    new_state[mk], new_state[vk] = nm, nv
                    ent["bn"] = bnc
                else:
                    c = bn_infer(c, g, S[mk], S[vk], spec.bn_eps)
            ent["pre_act"] = c
            t = activation_fwd(c, act)
            blk["convs"].append(ent)
        f = t + f                                         # Add (backbone_blocks.py:242)
        cache["blocks"].append(blk)
    cache["feat"] = f
    h0p = conv2d_same(f, P["head/conv0/kernel"])
    h0 = activation_fwd(h0p, spec.head_activation)
    h1 = conv2d_same(h0, P["head/conv1/kernel"])
    p = np.tanh(2.0 * h1) * 0.51                          # model.py:342
    cache.update(h0p=h0p, h0=h0, h1=h1, p=p)
    y = layer_denormalize(p, spec.v_min, spec.v_max) if spec.denormalize else p
    if training:
        flat_state = (np.concatenate([new_state[n].ravel() for n, _ in spec.state_tensors()])
                      if spec.state_tensors() else np.zeros(0))
        return (y, flat_state, cache) if want_cache else (y, flat_state)
    return (y, cache) if want_cache else y


def denoiser_module_call(spec: ResnetSpec, params, state, image_u8: np.ndarray,
                         cast_to_uint8: bool = True, dtype=F64) -> np.ndarray:
    """bfcnn/module_denoiser.py:46-75: uint8 -> f32 -> pad_to_power_of_2 -> hydra ->
    remove_padding -> round(half-even) -> uint8."""
    if image_u8.dtype != np.uint8 or image_u8.ndim != 4:
        raise ValueError("input must be a rank-4 uint8 tensor")   # input_signature :43-45
    x = image_u8.astype(dtype)
    xp, ph, pw = pad_to_power_of_2(x)
    y = hydra_forward(spec, params, state, xp, training=False, dtype=dtype)
    y = remove_padding(y, ph, pw)
    if cast_to_uint8:
        y = round_half_even(y)
        return np.clip(y, 0, 255).astype(np.uint8)
    return y


# ---------------------------------------------------------------------------
# losses (bfcnn/loss.py) and the training step (bfcnn/train_loop.py:259-312)
# ---------------------------------------------------------------------------

def keras_relu(x, threshold=0.0, max_value=None):
    """keras.activations.relu(x, threshold, max_value): x if x > threshold else 0, then
    min(., max_value) (call site bfcnn/loss.py:53-57)."""
    y = np.where(x > threshold, x, 0.0)
    if max_value is not None:
        y = np.minimum(y, max_value)
    return y


def mae_diff(error, hinge=0.0, cutoff=255.0):
    """bfcnn/loss.py:40-65 (the stray trailing comma at :57 only wraps d in a 1-tuple;
    the result is still the global mean)."""
    d = keras_relu(np.abs(error), hinge, cutoff)
    return d.mean(axis=(1, 2, 3)).mean()


def rmse_diff(error, hinge=0.0, cutoff=255.0 * 255.0):
    """bfcnn/loss.py:92-113: relu(error, threshold, max) on the SIGNED error, square,
    per-image mean, sqrt(+DEFAULT_EPSILON), batch mean."""
    d = keras_relu(error, hinge, cutoff) ** 2
    return np.sqrt(d.mean(axis=(1, 2, 3)) + DEFAULT_EPSILON).mean()


def ssim_gauss_kernel(size: int = 7, sigma: float = 1.5) -> np.ndarray:
    """tf.image.ssim's window (_fspecial_gauss): softmax over the 2-D grid of -(x^2 + y^2) / (2 sigma^2)."""
    c = np.arange(size, dtype=F64) - (size - 1) / 2.0
    g = -0.5 * c * c / (sigma * sigma)
    g = g[None, :] + g[:, None]
    e = np.exp(g - g.max())
    return e / e.sum()


def _valid_window_sum(x, g):
    """depthwise_conv2d(x, g, strides 1, padding "VALID") for one window g shared by all channels."""
    k = g.shape[0]
    H, W = x.shape[1] - k + 1, x.shape[2] - k + 1
    out = np.zeros((x.shape[0], H, W, x.shape[3]), dtype=x.dtype)
    for i in range(k):
        for j in range(k):
            out += g[i, j] * x[:, i:i + H, j:j + W, :]
    return out


def _valid_window_sum_transposed(m, g, H, W):
    """adjoint of _valid_window_sum: scatters every window value back over the pixels it covers."""
    k = g.shape[0]
    out = np.zeros((m.shape[0], H, W, m.shape[3]), dtype=m.dtype)
    for i in range(k):
        for j in range(k):
            out[:, i:i + m.shape[1], j:j + m.shape[2], :] += g[i, j] * m
    return out


def ssim_mean_and_grad(gt, pred, max_val: float = 255.0, filter_size: int = 7, filter_sigma: float = 1.5,
                       k1: float = 0.01, k2: float = 0.03):
    """tf.reduce_mean(tf.image.ssim(img1=gt, img2=pred, filter_size=7, max_val=255)) as bfcnn/loss.py:219-226 calls it
    (TF 2.13 image_ops_impl: _ssim_per_channel / _ssim_helper, compensation 1.0, VALID windows, mean over windows then
    channels then batch = one global mean) and its gradient with respect to `pred`."""
    x, y = np.asarray(pred, F64), np.asarray(gt, F64)          # x: the variable
    g = ssim_gauss_kernel(filter_size, filter_sigma)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    a, b = _valid_window_sum(x, g), _valid_window_sum(y, g)
    s, q = _valid_window_sum(x * y, g), _valid_window_sum(x * x + y * y, g)
    nl, dl = 2.0 * a * b + c1, a * a + b * b + c1
    nc, dc = 2.0 * s - 2.0 * a * b + c2, q - a * a - b * b + c2
    lum, cs = nl / dl, nc / dc
    val = (lum * cs).mean()
    n = lum.size
    dS_da = cs * (2.0 * b * dl - nl * 2.0 * a) / (dl * dl) + lum * (-2.0 * b * dc + nc * 2.0 * a) / (dc * dc)
    dS_ds = lum * 2.0 / dc
    dS_dq = -lum * nc / (dc * dc)
    H, W = x.shape[1], x.shape[2]
    grad = (_valid_window_sum_transposed(dS_da, g, H, W) + y * _valid_window_sum_transposed(dS_ds, g, H, W)
            + 2.0 * x * _valid_window_sum_transposed(dS_dq, g, H, W)) / n
    return val, grad


@dataclass
class LossSpec:
    """bfcnn/loss.py:162-179 (defaults as the reference: ssim_multiplier defaults to 1.0)."""
    hinge: float = 0.0
    cutoff: float = 255.0
    mae_multiplier: float = 1.0
    ssim_multiplier: float = 1.0
    mse_multiplier: float = 0.0
    regularization: float = 1.0

    @staticmethod
    def from_config(cfg: Dict) -> "LossSpec":
        return LossSpec(hinge=cfg.get("hinge", 0.0), cutoff=cfg.get("cutoff", 255.0),
                        mae_multiplier=cfg.get("mae_multiplier", 1.0),
                        ssim_multiplier=cfg.get("ssim_multiplier", 1.0),
                        mse_multiplier=cfg.get("mse_multiplier", 0.0),
                        regularization=cfg.get("regularization", 1.0))


def denoiser_loss(ls: LossSpec, gt, pred) -> Dict[str, float]:
    """bfcnn/loss.py:190-247."""
    err = gt - pred
    mae_actual = mae_diff(err, 0.0, 255.0)
    mse_actual = rmse_diff(err, 0.0, 255.0)   # reference passes cutoff=255.0 here (:205-209)
    mae_pl = mae_diff(err, ls.hinge, ls.cutoff) if ls.mae_multiplier > 0 else 0.0
    mse_pl = rmse_diff(err, ls.hinge, ls.cutoff * ls.cutoff) if ls.mse_multiplier > 0 else 0.0
    ssim_loss = 1.0 - ssim_mean_and_grad(gt, pred)[0] if ls.ssim_multiplier > 0 else 0.0      # :219-227
    return {"total_loss": mae_pl * ls.mae_multiplier + mse_pl * ls.mse_multiplier + ssim_loss * ls.ssim_multiplier,
            "mse_loss": mse_actual, "mae_loss": mae_actual, "ssim_loss": ssim_loss}


def _reg_value_grad(w, kind):
    """keras string regularisers: "l1" -> 0.01*sum|w|, "l2" -> 0.01*sum w^2."""
    if kind is None:
        return 0.0, np.zeros_like(w)
    kind = kind.lower()
    if kind == "l1":
        return 0.01 * np.abs(w).sum(), 0.01 * np.sign(w)
    if kind == "l2":
        return 0.01 * (w * w).sum(), 0.02 * w
    if kind == "l1_l2":
        return 0.01 * np.abs(w).sum() + 0.01 * (w * w).sum(), 0.01 * np.sign(w) + 0.02 * w
    raise NotImplementedError(kind)


def model_loss(spec: ResnetSpec, ls: LossSpec, params, dtype=F64) -> Dict[str, float]:
    """bfcnn/loss.py:181-187: tf.add_n(model.losses) * regularization."""
    P = _views(spec, params, dtype)
    reg = sum(_reg_value_grad(P[n], r)[0] for n, _, k, r in spec.tensors() if k == "conv")
    return {"regularization_loss": reg, "total_loss": reg * ls.regularization}


def _relu_sites(spec: ResnetSpec, C):
    """(name, pre-activation array) of every ReLU-type gate of a training forward's cache, in a fixed order"""
    sites = []
    if spec.base_activation in ("relu", "leaky_relu"):
        sites.append(("base", C["base_pre"]))
    for i, blk in enumerate(C["blocks"]):
        for j, ent in enumerate(blk["convs"]):
            if ent["act"] in ("relu", "leaky_relu"):
                sites.append((f"block{i}/conv{j}", ent["pre_act"]))
    if spec.head_activation in ("relu", "leaky_relu"):
        sites.append(("head0", C["h0p"]))
    return sites


def training_step_ties(spec: ResnetSpec, ls: LossSpec, params, state, gt, noisy, rel: float = 2e-6, dtype=F64):
    """Elements of a training step that sit within fp32 rounding of a KINK of the graph (test infrastructure for the GPU parity
    tests: where fp32 and fp64 may legitimately take different sides).  A kink is a point where the backward pass is discontinuous:
    a ReLU gate (pre-activation 0; backbone_blocks.py:174-213 activations), the L1 / RMSE hinge and cutoff thresholds on
    |gt - prediction| (loss.py:40-65: keras relu with threshold / max_value), the denormaliser's clip at +-0.5
    (utilities.py:435-443).  An element counts when its distance to the kink is below `rel` times the largest magnitude of its
    tensor (pre-activations: sums of O(100) fp32 products) or 4 fp32 ulps of the prediction (thresholds on the 0..255 scale).
    Returns a list of (site, flat_index, margin)."""
    pred, _, C = hydra_forward(spec, params, state, noisy, training=True, dtype=dtype, want_cache=True)
    ties = []
    for name, pre in _relu_sites(spec, C):
        tol = rel * max(float(np.abs(pre).max()), 1e-30)
        for idx in np.flatnonzero(np.abs(pre) < tol):
            ties.append((name, int(idx), float(abs(pre.ravel()[idx]))))
    a = np.abs(gt.astype(dtype) - pred)
    ulp4 = 4 * np.spacing(np.maximum(np.abs(pred), 1.0).astype(np.float32)).astype(dtype)
    for name, thr in (("hinge", ls.hinge), ("cutoff", ls.cutoff)):
        if ls.mae_multiplier > 0 or ls.mse_multiplier > 0:
            for idx in np.flatnonzero(np.abs(a - thr) < ulp4):
                ties.append((name, int(idx), float(abs(a.ravel()[idx] - thr))))
    if spec.denormalize:
        p = C["p"]
        for idx in np.flatnonzero(np.abs(np.abs(p) - 0.5) < 4 * 6e-8):
            ties.append(("clip", int(idx), float(abs(abs(p.ravel()[idx]) - 0.5))))
    return ties


def train_step_single_gpu(spec: ResnetSpec, ls: LossSpec, params, state, gt, noisy,
                          depth_weight: float = 1.0, dtype=F64, flips=None):
    """bfcnn/train_loop.py:259-312 for a single-output model: training-mode forward,
    denoiser loss * depth_weight[0] + model loss, gradients w.r.t. every trainable
    variable.  Returns (total_loss, model_loss, [denoiser_loss], predictions, grads_flat,
    new_state_flat).
    flips (tests only): entries of training_step_ties whose gate the BACKWARD pass takes on the other side -- what an fp32
    evaluation may do with an element that sits within rounding of the kink (the forward values are left as they are: the two
    sides differ by less than the tie tolerance there)."""
    P = _views(spec, params, dtype)
    pred, new_state, C = hydra_forward(spec, params, state, noisy, training=True, dtype=dtype,
                                       want_cache=True)
    gt = gt.astype(dtype)
    flip_mask = {}
    if flips:
        sites = dict(_relu_sites(spec, C))
        for name, idx, _ in flips:
            if name in sites:                       # the gate reads the sign of the cached pre-activation: invert it
                v = sites[name].ravel()
                v[idx] = -v[idx] if v[idx] != 0 else 1e-300
            else:
                flip_mask.setdefault(name, np.zeros(pred.size, bool))[idx] = True
    def _flipped(mask, name):
        return mask ^ flip_mask[name].reshape(mask.shape) if name in flip_mask else mask
    dl = denoiser_loss(ls, gt, pred)
    ml = model_loss(spec, ls, params, dtype)
    total = dl["total_loss"] * depth_weight + ml["total_loss"]

    # ---- backward -------------------------------------------------------
    n_el = pred.size
    err = gt - pred
    a = np.abs(err)
    dpred = np.where(_flipped(a > ls.hinge, "hinge") & _flipped(a < ls.cutoff, "cutoff"), -np.sign(err), 0.0) \
        * (ls.mae_multiplier * depth_weight / n_el) if ls.mae_multiplier > 0 else np.zeros_like(pred)
    if ls.mse_multiplier > 0:          # d/dpred of mean_b sqrt(mean(d^2) + eps), d = relu(gt - pred, hinge, cutoff^2)
        d = keras_relu(err, ls.hinge, ls.cutoff * ls.cutoff)
        live = _flipped(err > ls.hinge, "hinge") & (err < ls.cutoff * ls.cutoff)
        per_image = d[0].size
        rm = np.sqrt((d * d).mean(axis=(1, 2, 3)) + DEFAULT_EPSILON)
        dpred = dpred - (ls.mse_multiplier * depth_weight / (pred.shape[0] * per_image)) * (d * live) / rm[:, None, None, None]
    if ls.ssim_multiplier > 0:         # loss term (1 - mean ssim) * multiplier
        dpred = dpred - (ls.ssim_multiplier * depth_weight) * ssim_mean_and_grad(gt, pred)[1]
    if spec.denormalize:
        p = C["p"]
        dp = dpred * (spec.v_max - spec.v_min) * _flipped((p >= -0.5) & (p <= 0.5), "clip")
    else:
        dp = dpred
    G = {}
    dh1 = dp * 0.51 * 2.0 * (1.0 - np.tanh(2.0 * C["h1"]) ** 2)
    G["head/conv1/kernel"] = conv2d_same_grad_kernel(C["h0"], dh1, 1, 1)
    dh0 = conv2d_same_grad_input(dh1, P["head/conv1/kernel"])
    dh0p = activation_bwd(C["h0p"], dh0, spec.head_activation)
    G["head/conv0/kernel"] = conv2d_same_grad_kernel(C["feat"], dh0p, 1, 1)
    df = conv2d_same_grad_input(dh0p, P["head/conv0/kernel"])
    nb = len(spec.block_kernels)
    for i in reversed(range(spec.no_layers)):
        blk = C["blocks"][i]
        dt = df
        for j in reversed(range(nb)):
            ent = blk["convs"][j]
            dc = activation_bwd(ent["pre_act"], dt, ent["act"])
            if "bn" in ent:
                dc, dg, _ = bn_train_bwd(dc, P[f"block{i}/bn{j}/gamma"], ent["bn"])
                G[f"block{i}/bn{j}/gamma"] = dg
            kk = spec.block_kernels[j]
            G[f"block{i}/conv{j}/kernel"] = conv2d_same_grad_kernel(ent["in"], dc, kk, kk)
            dt = conv2d_same_grad_input(dc, P[f"block{i}/conv{j}/kernel"])
        df = df + dt
    dbase = activation_bwd(C["base_pre"], df, spec.base_activation)
    G["base/kernel"] = conv2d_same_grad_kernel(C["xn"], dbase, spec.kernel_size, spec.kernel_size)
    for n, _, k, r in spec.tensors():
        if k == "conv":
            G[n] = G[n] + _reg_value_grad(P[n], r)[1] * ls.regularization
    grads = np.concatenate([G[n].ravel() for n, _, _, _ in spec.tensors()])
    return total, ml, [dl], pred, grads, new_state


# ---------------------------------------------------------------------------
# optimiser (bfcnn/optimizer.py:83-206; keras 2.13 Adam)
# ---------------------------------------------------------------------------

def exponential_decay(lr0: float, decay_steps: float, decay_rate: float, step: float) -> float:
    """keras ExponentialDecay, non-staircase (bfcnn/optimizer.py:107-115)."""
    return lr0 * decay_rate ** (step / decay_steps)


def clip_by_global_norm(g: np.ndarray, clip: Optional[float]) -> Tuple[np.ndarray, float]:
    norm = float(np.sqrt((g.astype(F64) ** 2).sum()))
    if clip is None:
        return g, norm
    return g * (clip / max(norm, clip)), norm


def adam_step(params, grads, m, v, iterations: int, lr: float, beta_1=0.9, beta_2=0.999,
              epsilon=1e-7, global_clipnorm: Optional[float] = None, clipnorm: Optional[float] = None,
              clipvalue: Optional[float] = None, tensor_offsets=None):
    """keras.optimizers.Adam.update_step (Keras 2.13): t = iterations+1,
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    w -= alpha*m/(sqrt(v)+eps).  Gradient clipping first (optimizer.py:165-206), with keras' precedence
    (_clip_gradients): clipnorm = tf.clip_by_norm per tensor (needs `tensor_offsets`, the starts of the tensors in the
    flat vector + its length), else global_clipnorm, else clipvalue."""
    g = grads.astype(F64)
    if clipnorm and clipnorm > 0:
        g = g.copy()
        for a, b in zip(tensor_offsets[:-1], tensor_offsets[1:]):
            nrm = math.sqrt(float((g[a:b] ** 2).sum()))
            g[a:b] *= clipnorm / max(nrm, clipnorm)
    elif global_clipnorm and global_clipnorm > 0:
        g, _ = clip_by_global_norm(g, global_clipnorm)
    elif clipvalue and clipvalue > 0:
        g = np.clip(g, -clipvalue, clipvalue)
    t = iterations + 1
    alpha = lr * math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t)
    m = m + (g - m) * (1.0 - beta_1)
    v = v + (g * g - v) * (1.0 - beta_2)
    params = params - alpha * m / (np.sqrt(v) + epsilon)
    return params, m, v


# ---------------------------------------------------------------------------
# pyramid (bfcnn/pyramid.py) and resampling
# ---------------------------------------------------------------------------

def avg_pool_same(x: np.ndarray, k: Tuple[int, int] = (5, 5), s: int = 2) -> np.ndarray:
    """keras AveragePooling2D(pool_size=k, strides=s, padding="same")
    (bfcnn/pyramid.py:266-270, 374-378): divisor = number of in-bounds taps."""
    B, H, W, C = x.shape
    oh, pt, pb = same_pads(H, k[0], s)
    ow, pl, pr = same_pads(W, k[1], s)
    xp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=x.dtype)
    xp[:, pt:pt + H, pl:pl + W, :] = x
    cp = np.zeros((H + pt + pb, W + pl + pr), dtype=x.dtype)
    cp[pt:pt + H, pl:pl + W] = 1.0
    acc = np.zeros((B, oh, ow, C), dtype=x.dtype)
    cnt = np.zeros((oh, ow), dtype=x.dtype)
    for i in range(k[0]):
        for j in range(k[1]):
            acc += xp[:, i:i + (oh - 1) * s + 1:s, j:j + (ow - 1) * s + 1:s, :]
            cnt += cp[i:i + (oh - 1) * s + 1:s, j:j + (ow - 1) * s + 1:s]
    return acc / cnt[None, :, :, None]


def avg_pool_valid_2x2(x: np.ndarray) -> np.ndarray:
    """tf.nn.avg_pool2d(ksize 2, strides 2, "VALID") (bfcnn/utilities.py:655)."""
    B, H, W, C = x.shape
    h2, w2 = H // 2, W // 2
    x = x[:, :h2 * 2, :w2 * 2, :]
    return 0.25 * (x[:, 0::2, 0::2] + x[:, 1::2, 0::2] + x[:, 0::2, 1::2] + x[:, 1::2, 1::2])


def _up2_axis(x: np.ndarray, axis: int) -> np.ndarray:
    n = x.shape[axis]
    idx = np.arange(n)
    prev = np.take(x, np.clip(idx - 1, 0, n - 1), axis=axis)
    nxt = np.take(x, np.clip(idx + 1, 0, n - 1), axis=axis)
    even = 0.25 * prev + 0.75 * x
    odd = 0.75 * x + 0.25 * nxt
    out = np.stack([even, odd], axis=axis + 1)
    shp = list(x.shape)
    shp[axis] = 2 * n
    return out.reshape(shp)


def upsample_bilinear_2x(x: np.ndarray) -> np.ndarray:
    """keras UpSampling2D(2, "bilinear") = tf.image.resize half-pixel centres, edge clamp
    (bfcnn/pyramid.py:319-325, 380-382, 434-436)."""
    return _up2_axis(_up2_axis(x, 1), 2)


def upsample_nearest_2x(x: np.ndarray) -> np.ndarray:
    """keras UpSampling2D(2, "nearest") (bfcnn/upsampling.py:65,105)."""
    return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)


def strided_slice_2x(x: np.ndarray) -> np.ndarray:
    """x[:, ::2, ::2, :] (bfcnn/downsampling.py:61)."""
    return x[:, ::2, ::2, :]


def gaussian_pyramid(x, levels: int, kernel_size=(5, 5)) -> List[np.ndarray]:
    """bfcnn/pyramid.py:238-283 (also used for type NONE, :486-490)."""
    out = [x]
    for _ in range(1, levels):
        x = avg_pool_same(x, kernel_size, 2)
        out.append(x)
    return out


def inverse_gaussian_pyramid(levels_list: List[np.ndarray]) -> np.ndarray:
    """bfcnn/pyramid.py:289-341."""
    out = prev = None
    for lv in reversed(levels_list):
        if out is None:
            out = prev = lv
        else:
            out = upsample_bilinear_2x(out)
            out = out + (lv - upsample_bilinear_2x(prev))
            prev = lv
    return out


def laplacian_pyramid(x, levels: int, kernel_size=(5, 5)) -> List[np.ndarray]:
    """bfcnn/pyramid.py:347-398."""
    out = []
    for _ in range(levels - 1):
        down = avg_pool_same(x, kernel_size, 2)
        out.append(x - upsample_bilinear_2x(down))
        x = down
    out.append(x)
    return out


def inverse_laplacian_pyramid(levels_list: List[np.ndarray]) -> np.ndarray:
    """bfcnn/pyramid.py:404-445."""
    out = None
    for lv in reversed(levels_list):
        out = lv if out is None else upsample_bilinear_2x(out) + lv
    return out


def build_pyramid(config: Optional[Dict]):
    """bfcnn/pyramid.py:451-491: returns fn(x)->list.  type NONE builds a gaussian pyramid."""
    if config is None:
        levels, k, t = 1, (5, 5), "NONE"
    else:
        levels = config.get("levels", 1)
        k = tuple(config.get("kernel_size", (5, 5)))
        t = config.get("type", "NONE").strip().upper()
    if t in ("GAUSSIAN", "NONE"):
        return lambda x: gaussian_pyramid(x, levels, k)
    if t == "LAPLACIAN":
        return lambda x: laplacian_pyramid(x, levels, k)
    raise ValueError(f"don't know how to build pyramid type [{t}]")


def build_inverse_pyramid(config: Optional[Dict]):
    """bfcnn/pyramid.py:497-532."""
    t = "NONE" if config is None else config.get("type", "NONE").strip().upper()
    if t in ("GAUSSIAN", "NONE"):
        return inverse_gaussian_pyramid
    if t == "LAPLACIAN":
        return inverse_laplacian_pyramid
    raise ValueError(f"don't know how to build pyramid type [{t}]")


def multiscales(x: np.ndarray, no_scales: int, clip_values=True, round_values=True) -> List[np.ndarray]:
    """bfcnn/utilities.py:625-672 (multiscales_generator_fn as train_loop.py:239-247 calls it)."""
    out = [x]
    for _ in range(no_scales):
        x = avg_pool_valid_2x2(x)
        if clip_values:
            x = np.clip(x, 0.0, 255.0)
        if round_values:
            x = round_half_even(x)
        out.append(x)
    return out


# ---------------------------------------------------------------------------
# synthetic workload of SURVEY.md 8(d) (shared by tests and bench)
# ---------------------------------------------------------------------------

def synthetic_batch(batch: int, height: int, width: int, channels: int = 3, sigma: float = 20.0,
                    seed: int = 1234) -> Tuple[np.ndarray, np.ndarray]:
    """(clean_u8, noisy_u8): smooth random field (uniform u8 at 1/8 resolution, bilinear x8)
    + additive gaussian noise truncated at 2 sigma, clipped and rounded like
    bfcnn/dataset.py:209-230."""
    rng = np.random.default_rng(seed)
    h8, w8 = max(height // 8, 1), max(width // 8, 1)
    x = rng.integers(0, 256, size=(batch, h8, w8, channels)).astype(np.float64)
    for _ in range(3):
        x = upsample_bilinear_2x(x)
    x = x[:, :height, :width, :]
    if x.shape[1] < height or x.shape[2] < width:
        x = np.pad(x, ((0, 0), (0, height - x.shape[1]), (0, width - x.shape[2]), (0, 0)), mode="edge")
    clean = np.clip(np.rint(x), 0, 255)
    n = rng.standard_normal(clean.shape)
    bad = np.abs(n) > 2.0
    while bad.any():
        n[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(n) > 2.0
    noisy = np.clip(np.rint(clean + sigma * n), 0, 255)
    return clean.astype(np.uint8), noisy.astype(np.uint8)


# ---------------------------------------------------------------------------
# training-data corruption (bfcnn/dataset.py:126-239, prepare_data_fn)
# ---------------------------------------------------------------------------

def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., the counter-based generator TF's stateless ops use) on uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32
        n1 = p1 & m32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32
        n3 = p0 & m32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c0, c1, c2, c3


def truncated_standard_normal(n: int, stream: int, seed: int) -> np.ndarray:
    """tf.random.truncated_normal semantics (dataset.py:199-204, 218-223: values beyond 2 sigma are re-picked) on the
    element-indexed Philox stream the HIP kernel uses: attempt a = 0, 1, ... of element i takes Philox(counter =
    (i lo, i hi, a, stream), key = seed), Box-Muller on (r0, r1) and (r2, r3), first |z| <= 2 in the order
    z0, z1, z2, z3."""
    idx = np.arange(n, dtype=np.uint64)
    lo, hi = idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32)
    out = np.zeros(n, np.float64)
    todo = np.ones(n, bool)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for attempt in range(16):
        if not todo.any():
            break
        sel = np.nonzero(todo)[0]
        r = _philox4x32_10(lo[sel], hi[sel], np.full(sel.size, attempt, np.uint64), np.full(sel.size, stream, np.uint64), k0, k1)
        done = np.zeros(sel.size, bool)
        val = np.zeros(sel.size, np.float64)
        for h in range(2):
            u1 = ((r[2 * h] >> np.uint64(8)).astype(np.float64) + 1.0) / 16777216.0
            u2 = (r[2 * h + 1] >> np.uint64(8)).astype(np.float64) / 16777216.0
            rad = np.sqrt(-2.0 * np.log(u1))
            for z in (rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)):
                take = ~done & (np.abs(z) <= 2.0)
                val[take] = z[take]
                done |= take
        out[sel[done]] = val[done]
        todo[sel[done]] = False
    return out


def prepare_data(input_batch: np.ndarray, flip_left_right: bool, flip_up_down: bool, mult_std: float, add_std: float,
                 seed: int) -> Tuple[np.ndarray, np.ndarray]:
    """dataset.py:126-239 with the per-batch random choices made by the caller: flips (:131-159), tf.round (:234),
    x * truncated_normal(1, mult_std) (:193-208) then + truncated_normal(0, add_std) (:211-227) -- a std of 0 means the
    term is not applied -- and tf.round (:230).  Returns (input_batch, noisy_batch)."""
    x = np.asarray(input_batch, np.float64)
    if flip_left_right:
        x = x[:, :, ::-1, :]
    if flip_up_down:
        x = x[:, ::-1, :, :]
    clean = round_half_even(x)
    v = clean.copy()
    n = v.size
    if mult_std > 0:
        v = v * (1.0 + mult_std * truncated_standard_normal(n, 0, seed).reshape(v.shape))
    if add_std > 0:
        v = v + add_std * truncated_standard_normal(n, 1, seed).reshape(v.shape)
    return clean, round_half_even(v)
