"""Gradient oracle for training resnet backbones outside the 16-filter 3x3 family (the shipped bottleneck / depthwise config, the
`add_gates` channel gate): the forward of oracle/resnet_generic_oracle.py restated with torch-CPU fp64 tensor ops, so that
autograd supplies d(total loss)/d(every trainable tensor).

TEST INFRASTRUCTURE ONLY (tests/, never the product path).  The inference-mode forward here is checked against the NumPy
restatement (tests/test_resnet_generic_train_oracle.py), which pins it; what it adds is the training side:
  * bfcnn/train_loop.py:259-312 -- hydra(noisy, training=True), denoiser loss * depth weight + regularization * sum(model.losses);
  * keras BatchNormalization(center=False) under training=True (bfcnn/utilities.py:204-206): batch mean / biased variance,
    moving statistics updated with DEFAULT_BN_MOMENTUM, the Bessel-corrected variance into the moving variance (fused kernel);
  * the regularisers the builder attaches (backbone_resnet.py:128-176): `kernel_regularizer` on the base convolution,
    `block_regularizer[j]` on block convolution j (kernel_regularizer / depthwise_regularizer), "l2" on the two gate Dense
    kernels (backbone_blocks.py:146-160), the denoiser section's `kernel_regularizer` on the head, nothing on BatchNorm gammas;
    keras strings: "l1" = 0.01 sum |w|, "l2" = 0.01 sum w^2.
"""
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import bfcnn_oracle as O
from . import resnet_generic_oracle as R
from . import unet_torch as T

DT = torch.float64


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


_act = T._act                       # linear | relu | leaky_relu (0.3) / _01 / _001 | gelu (erf form)


def conv_same(x, w, groups=1):
    """x [B,H,W,Cin], keras kernel [kh,kw,Cin/groups,Cout]"""
    kh, kw = w.shape[0], w.shape[1]
    return _nhwc(F.conv2d(_nchw(x), w.permute(3, 2, 0, 1), padding=(kh // 2, kw // 2), groups=groups))


def depthwise_mult_same(x, w):
    """keras DepthwiseConv2D kernel [kh,kw,C,m], output channel c * m + j"""
    kh, kw, C, m = w.shape
    wt = w.permute(2, 3, 0, 1).reshape(C * m, 1, kh, kw)
    return _nhwc(F.conv2d(_nchw(x), wt, padding=(kh // 2, kw // 2), groups=C))


def views(spec: R.GenericResnetSpec, flat) -> Dict[str, torch.Tensor]:
    out, o = {}, 0
    for name, shape, _ in spec.tensors():
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].reshape(shape)
        o += n
    return out


def state_views(spec, flat):
    out, o = {}, 0
    for name, shape in spec.state_tensors():
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].reshape(shape)
        o += n
    return out


def _batch_norm(t, base, P, S, new_state, training):
    """keras BatchNormalization(scale=True, center=False) (backbone_resnet.py:129-135)"""
    gamma = P[base + "/gamma"]
    if not training:
        return gamma * (t - S[base + "/moving_mean"]) / torch.sqrt(S[base + "/moving_variance"] + R.BN_EPS)
    n = t.shape[0] * t.shape[1] * t.shape[2]
    mu = t.mean(dim=(0, 1, 2))
    var = ((t - mu) ** 2).mean(dim=(0, 1, 2))
    mom = O.DEFAULT_BN_MOMENTUM
    new_state[base + "/moving_mean"] = (S[base + "/moving_mean"] * mom + mu * (1 - mom)).detach()
    new_state[base + "/moving_variance"] = (S[base + "/moving_variance"] * mom + var * (n / max(n - 1, 1)) * (1 - mom)).detach()
    return gamma * (t - mu) / torch.sqrt(var + R.BN_EPS)


def avgpool_same(x, pool, stride):
    """keras AveragePooling2D(pool, strides, padding="same"): TF SAME padding, divisor = taps inside the image (resnet_generic_oracle.avgpool_same)"""
    B, H, W, C = x.shape
    OH, OW = -(-H // stride[0]), -(-W // stride[1])
    pt = max((OH - 1) * stride[0] + pool[0] - H, 0) // 2
    pl = max((OW - 1) * stride[1] + pool[1] - W, 0) // 2
    rows = []
    for oy in range(OH):
        y0, y1 = max(oy * stride[0] - pt, 0), min(oy * stride[0] - pt + pool[0], H)
        cols = []
        for ox in range(OW):
            x0, x1 = max(ox * stride[1] - pl, 0), min(ox * stride[1] - pl + pool[1], W)
            cols.append(x[:, y0:y1, x0:x1].mean(dim=(1, 2)))
        rows.append(torch.stack(cols, dim=1))
    return torch.stack(rows, dim=1)


def selector_prefilter(sel, flags, pre_w, pool):
    """resnet_generic_oracle.selector_prefilter (custom_layers_selector.py:160-185; utilities.py:566-620) in torch"""
    x = sel
    if "use_conv1x1_selector" in flags:
        x = x @ pre_w.reshape(pre_w.shape[-2], pre_w.shape[-1])
    if "use_global_normalization" in flags:
        mu = x.mean(dim=(1, 2), keepdim=True)
        x = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(dim=(1, 2), keepdim=True) + 1e-3)
    if "use_local_normalization" in flags:
        mu = avgpool_same(x, pool, (1, 1))
        x = (x - mu) / torch.sqrt(avgpool_same((x - mu) ** 2, pool, (1, 1)) + 1e-3)
    if "use_lowpass" in flags:
        x = (1.0 - torch.tanh(4.0 * x) ** 4) * x
    if "use_highpass" in flags:
        x = torch.tanh(4.0 * x) ** 4 * x
    return x


def selector_block(x1, x2, sel, w0, w1, scale_type, activation_type, pool, stride):
    """custom_layers_selector.py:81-330 as resnet_generic_oracle.selector_block restates it (no optional pre-filters)"""
    st = scale_type.lower()
    w0, w1 = w0.reshape(w0.shape[-2], w0.shape[-1]), w1.reshape(w1.shape[-2], w1.shape[-1])
    if st in ("local", "multiscale", "mixed"):
        if st == "local":
            u = avgpool_same(sel, pool, stride)
        elif st == "multiscale":
            u = torch.cat([avgpool_same(sel, (pool[0] // 2, pool[1] // 2), stride), avgpool_same(sel, pool, stride),
                           avgpool_same(sel, (pool[0] * 2, pool[1] * 2), stride)], dim=-1)
        else:
            loc = avgpool_same(sel, pool, stride)
            u = torch.cat([loc, torch.zeros_like(loc) + sel.mean(dim=(1, 2), keepdim=True)], dim=-1)
        u = u @ w0
        u = torch.where(u > 0, u, 0.3 * u)
        u = torch.relu(u @ w1)
        u = T._resize(u, u.shape[1] * stride[0], u.shape[2] * stride[1])
    elif st == "global":
        u = sel.mean(dim=(1, 2)) @ w0
        u = torch.where(u > 0, u, R.SELECTOR_GLOBAL_LEAKY * u)
        u = torch.relu(u @ w1)[:, None, None, :]
    else:
        raise ValueError(scale_type)
    p = 2.5 - u
    s = torch.clamp(0.2 * p + 0.5, 0.0, 1.0) if activation_type.lower() == "hard" else torch.sigmoid(p)
    return x1 * s + x2 * (1.0 - s)


def hydra(spec: R.GenericResnetSpec, P, S, x, training: bool, drop_scale=None):
    """returns (prediction, new state dict).  drop_scale: {block index: per-sample factor [B]} = RandomOnOff's draw
    (0 or 1 / (1 - rate); custom_layers.py:107-127), training only."""
    new_state = dict(S)
    drop_scale = drop_scale or {}
    scaled = lambda t, name: t * torch.relu(P[name] + 1.0)
    xn = (torch.clamp(x, spec.v_min, spec.v_max) - spec.v_min) / (spec.v_max - spec.v_min) - 0.5
    f = _act(conv_same(xn, P["base/kernel"]), spec.base_activation)
    if spec.add_initial_bn:
        f = _batch_norm(f, "initial_bn", P, S, new_state, training)
    for i in range(spec.no_layers):
        t = f
        for j, (dm, g, a) in enumerate(zip(spec.block_depthwise, spec.block_groups, spec.block_activation)):
            w = P[f"block{i}/conv{j}/kernel"]
            t = depthwise_mult_same(t, w) if dm != -1 else conv_same(t, w, g)
            if j >= 1 and spec.use_bn:
                base = f"block{i}/bn{j}"
                gamma = P[base + "/gamma"]
                if training:
                    n = t.shape[0] * t.shape[1] * t.shape[2]
                    mu = t.mean(dim=(0, 1, 2))
                    var = ((t - mu) ** 2).mean(dim=(0, 1, 2))
                    t = gamma * (t - mu) / torch.sqrt(var + R.BN_EPS)
                    mom = O.DEFAULT_BN_MOMENTUM
                    new_state[base + "/moving_mean"] = (S[base + "/moving_mean"] * mom + mu * (1 - mom)).detach()
                    new_state[base + "/moving_variance"] = (S[base + "/moving_variance"] * mom + var * (n / max(n - 1, 1)) * (1 - mom)).detach()
                else:
                    t = gamma * (t - S[base + "/moving_mean"]) / torch.sqrt(S[base + "/moving_variance"] + R.BN_EPS)
            t = _act(t, a)
            if j == 0:
                first = t                                                # x_1st_conv: the selector layer (backbone_blocks.py:229-231)
            if j == 1 and spec.add_gates:
                y = torch.relu(t.mean(dim=(1, 2)) @ P[f"block{i}/gate/dense0/kernel"])
                y = torch.clamp(0.2 * (y @ P[f"block{i}/gate/dense1/kernel"]) + 0.5, 0.0, 1.0)
                t = t * y[:, None, None, :]
        if spec.add_channelwise_scaling:
            t = scaled(t, f"block{i}/channelwise/w0")
        if spec.add_learnable_multiplier:
            t = scaled(t, f"block{i}/multiplier/w0")
        if training and i in drop_scale:
            t = t * drop_scale[i].reshape(-1, 1, 1, 1)
        if spec.selector:
            if spec.selector[6]:
                first = selector_prefilter(first, spec.selector[6], P.get(f"block{i}/selector/pre/kernel"), spec.selector[3])
            kind = "dense" if spec.selector[0] == "global" else "conv"
            f = selector_block(f, t, first, P[f"block{i}/selector/{kind}0/kernel"], P[f"block{i}/selector/{kind}1/kernel"],
                               spec.selector[0], spec.selector[1], spec.selector[3], spec.selector[4])
        else:
            f = t + f
    if spec.add_final_bn:
        f = _batch_norm(f, "final_bn", P, S, new_state, training)
    if spec.add_concat_input:
        f = torch.cat([f, xn], dim=-1)
    if spec.add_channelwise_scaling:
        f = scaled(f, "channelwise/w0")
    if spec.add_learnable_multiplier:
        f = scaled(f, "multiplier/w0")
    h = _act(conv_same(f, P["head/conv0/kernel"]), spec.head_activation)
    h = conv_same(h, P["head/conv1/kernel"])
    p = torch.tanh(2.0 * h) * 0.51
    return (torch.clamp(p, -0.5, 0.5) + 0.5) * (spec.v_max - spec.v_min) + spec.v_min, new_state


def regularizer_kind(spec: R.GenericResnetSpec, name: str, kind: str):
    if kind == "bn_gamma":
        return None
    if kind == "channelwise":
        return "l1_0.1"                      # L1(DEFAULT_CHANNELWISE_MULTIPLIER_L1) (backbone_resnet.py:190-195; constants.py:13)
    if kind == "multiplier":
        return "l1_1.0"                      # L1(DEFAULT_MULTIPLIER_L1) (:197-202; constants.py:12)
    if name.startswith("base/"):
        return spec.kernel_regularizer
    if name.startswith("head/"):
        return spec.head_regularizer
    if "/gate/" in name:
        return "l2"
    if "/selector/" in name:
        return spec.selector[5]
    j = int(name.split("/")[1][4:])
    return spec.block_regularizer[j] if spec.block_regularizer else spec.kernel_regularizer


def regularization(spec, P):
    total = torch.zeros((), dtype=DT)
    for name, _, kind in spec.tensors():
        rk = regularizer_kind(spec, name, kind)
        if rk == "l1":
            total = total + 0.01 * P[name].abs().sum()
        elif rk == "l2":
            total = total + 0.01 * (P[name] ** 2).sum()
        elif rk == "l1_0.1":
            total = total + 0.1 * P[name].abs().sum()
        elif rk == "l1_1.0":
            # Multiplier hands its regulariser to the non-trainable w1 (= 1.0) as well (custom_layers.py:1067-1074): a constant 1.0
            total = total + 1.0 * P[name].abs().sum() + 1.0
        elif rk not in (None, "none"):
            raise ValueError(rk)
    return total


def train_step(spec: R.GenericResnetSpec, ls: O.LossSpec, params: np.ndarray, state: np.ndarray, gt: np.ndarray, noisy: np.ndarray,
               depth_weight: float = 1.0, drop_scale=None):
    """train_step_single_gpu (bfcnn/train_loop.py:259-312): returns (total, model-loss dict, denoiser-loss dict, prediction,
    flat gradient, new flat state)."""
    flat = torch.tensor(np.asarray(params, np.float64), dtype=DT, requires_grad=True)
    P = views(spec, flat)
    S = state_views(spec, torch.tensor(np.asarray(state, np.float64), dtype=DT))
    ds = {k: torch.tensor(np.asarray(v, np.float64)) for k, v in (drop_scale or {}).items()}
    pred, new_state = hydra(spec, P, S, torch.from_numpy(noisy.astype(np.float64)), True, ds)
    dl = T.denoiser_loss(ls, torch.from_numpy(gt.astype(np.float64)), pred)
    reg = regularization(spec, P)
    total = dl["total_loss"] * depth_weight + reg * ls.regularization
    total.backward()
    st = np.concatenate([new_state[n].numpy().ravel() for n, _ in spec.state_tensors()]) if spec.state_tensors() else np.zeros(0)
    ml = {"regularization_loss": float(reg.detach()), "total_loss": float(reg.detach() * ls.regularization)}
    return float(total.detach()), ml, {k: float(v.detach()) for k, v in dl.items()}, pred.detach().numpy(), flat.grad.numpy().copy(), st


def infer(spec, params, state, x):
    P = views(spec, torch.tensor(np.asarray(params, np.float64), dtype=DT))
    S = state_views(spec, torch.tensor(np.asarray(state, np.float64), dtype=DT))
    return hydra(spec, P, S, torch.from_numpy(np.asarray(x, np.float64)), False)[0].numpy()
