"""Gradient oracle for `unet_laplacian` training: the forward of oracle/unet_oracle.py restated with torch-CPU fp64 tensor
ops, so that autograd supplies d(total loss)/d(every trainable tensor).

TEST INFRASTRUCTURE ONLY (tests/, never the product path).  The forward here is checked against the NumPy restatement
(tests/test_unet_train_oracle.py: every output to 1e-10), which is what pins it; autograd then differentiates exactly that
function.  What it adds to unet_oracle.py is the training side of the reference:

  * bfcnn/train_loop.py:259-312 -- ground truth pyramid (utilities.py:625-685 multiscales_generator_fn), one denoiser loss
    per output scale times its depth weight, plus model.losses times `regularization`;
  * bfcnn/loss.py:40-113, 190-247 -- keras-relu hinge / cutoff L1, per-image RMSE, 1 - mean tf.image.ssim;
  * the regularisers the builder attaches: kernel_regularizer "l2" on base / down / up / head convolutions and the depthwise
    kernels (backbone_unet_laplacian.py:179-250), SoftOrthonormalConstraintRegularizer(0.01, 0, 1e-4) on the ConvNext
    1x1 convolutions (custom_layers.py:951-978, when use_soft_orthonormal_regularization) and on the attention block's
    four convolutions (backbone_unet_laplacian.py:332; regularizers.py:283-338), l1(1e-6) on every
    ChannelLearnableMultiplier (custom_layers.py:266-268), nothing on LayerNorm gammas;
  * training-mode randomness as explicit inputs: StochasticDepth keeps / drops a block's branch per SAMPLE (keras Dropout
    with noise_shape [B,1,1,1], kept branches scaled by 1 / (1 - rate); custom_layers.py:174-209,
    backbone_unet_laplacian.py:351-352, 557-558) -> `depth_scale[prefix]` = [B] multipliers; the attention's dropout on the
    softmax weights (keras.layers.Attention(dropout=...), custom_layers.py:1322-1326) -> `attn_scale[prefix]` = [B,T,T].
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import bfcnn_oracle as O
from . import unet_oracle as U

DT = torch.float64
SOFTORTHONORMAL = (0.01, 0.0, 1e-4)        # constants.py:19-21: lambda, l1, l2


def _act(x, name):
    name = (name or "linear").lower().strip()
    if name == "linear":
        return x
    if name == "relu":
        return torch.relu(x)
    if name.startswith("leaky_relu"):
        alpha = {"leaky_relu": 0.3, "leaky_relu_01": 0.1, "leaky_relu_001": 0.01}[name]
        return torch.where(x > 0, x, alpha * x)
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / np.sqrt(2.0)))
    raise NotImplementedError(name)


def _conv(x, w, stride=1):
    """keras Conv2D padding same, HWIO kernel, NHWC tensor."""
    kh, kw = w.shape[:2]
    _, pt, pb = O.same_pads(x.shape[1], kh, stride)
    _, pl, pr = O.same_pads(x.shape[2], kw, stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, w.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)


def _depthwise(x, w):
    kh, kw, C, _ = w.shape
    _, pt, pb = O.same_pads(x.shape[1], kh, 1)
    _, pl, pr = O.same_pads(x.shape[2], kw, 1)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, w.permute(2, 3, 0, 1), groups=C).permute(0, 2, 3, 1)


def _layer_norm(x, gamma, eps=U.LN_EPS):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma


def _multiplier(x, w):
    return torch.tanh(torch.relu(1.0 + w)) * x


def _avg_pool_same(x, k):
    """AveragePooling2D(k, strides 1, same): TF padding (the extra tap after for even k), divisor = in-bounds taps."""
    _, pt, pb = O.same_pads(x.shape[1], k, 1)
    _, pl, pr = O.same_pads(x.shape[2], k, 1)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    ones = F.pad(torch.ones((1, 1, x.shape[1], x.shape[2]), dtype=x.dtype), (pl, pr, pt, pb))
    ssum = F.avg_pool2d(xp, k, stride=1) * (k * k)
    cnt = F.avg_pool2d(ones, k, stride=1) * (k * k)
    return (ssum / cnt).permute(0, 2, 3, 1)


def _up2(x):
    return F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)


def _up2_nearest(x):
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def _resize(x, oh, ow):
    return F.interpolate(x.permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear", align_corners=False, antialias=False).permute(0, 2, 3, 1)


def views(spec: U.UnetLaplacianSpec, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {n: flat[o:o + int(np.prod(s))].reshape(s) for n, (o, s) in spec.offsets().items()}


def check_trainable_graph(spec: U.UnetLaplacianSpec):
    """the graph family the training path is built for (configs/unet_laplacian_v5.json and its depth / width / filter
    variations)."""
    bad = []
    if spec.downsample_type not in ("strides", "conv2d", "maxpool"): bad.append(f"downsample_type {spec.downsample_type}")
    if spec.upsample_type not in ("upsample_laplacian_conv2d", "upsample_nearest_conv2d", "upsample_bilinear_conv2d", "bilinear", "nn", "nearest"):
        bad.append(f"upsample_type {spec.upsample_type}")
    if not (spec.use_laplacian or spec.use_laplacian_averaging): bad.append("no laplacian split")
    if getattr(spec, "use_concat", False) and spec.use_attention_gates: bad.append("use_concat with use_attention_gates")
    if bad:
        raise NotImplementedError("unet_laplacian training: " + ", ".join(bad))


def backbone(spec, P, xn, depth_scale=None, attn_scale=None):
    a = spec.activation
    depth_scale, attn_scale = depth_scale or {}, attn_scale or {}

    def convnext(prefix, x):
        t = _depthwise(x, P[f"{prefix}/dw/kernel"])
        if spec.use_ln:
            t = _layer_norm(t, P[f"{prefix}/ln/gamma"])
        t = _act(_conv(t, P[f"{prefix}/pw1/kernel"]), spec.mlp_activation or a)
        t = _conv(t, P[f"{prefix}/pw2/kernel"])
        if spec.use_gamma:
            t = _multiplier(t, P[f"{prefix}/gamma/w"])
        return t

    def attention(prefix, x):
        B, H, W, C = x.shape
        rh, rw = (H, W) if spec.attention_full else spec.attention_resolution
        t = x if spec.attention_full else _resize(x, rh, rw)
        if spec.use_ln:
            t = _layer_norm(t, P[f"{prefix}/ln/gamma"])
        seq = (B * rh, rw) if spec.attention_full else (B, rh * rw)   # the archive's graph: one sequence per image row
        def qkv(n):
            z = _conv(t, P[f"{prefix}/{n}/kernel"])
            z = _act(z, spec.attention_activation) if spec.attention_activation else torch.where(z > 0, z, spec.attention_alpha * z)
            return z.reshape(*seq, -1)
        # ... and its [query, key, value] hand-over where keras reads [query, value, key] (oracle/unet_oracle.py attention)
        q, v, k = (qkv("query"), qkv("key"), qkv("value")) if spec.attention_full else (qkv("query"), qkv("value"), qkv("key"))
        p = torch.softmax(q @ k.transpose(1, 2), dim=-1)
        if prefix in attn_scale:
            p = p * attn_scale[prefix]
        t = (p @ v).reshape(B, rh, rw, -1)
        if spec.attention_full:
            if spec.use_ln:
                t = _layer_norm(t, P[f"{prefix}/ln1/gamma"])
        else:
            t = _resize(t, H, W)
        t = _conv(t, P[f"{prefix}/out/kernel"])
        return _multiplier(t, P[f"{prefix}/gamma/w"])

    def branch(prefix, t):
        return t * depth_scale[prefix].reshape(-1, 1, 1, 1) if prefix in depth_scale else t

    x = _act(_conv(xn, P["base/kernel"]), a)
    nodes = {}
    for d in range(spec.depth):
        for w in range(spec.width):
            pre = f"enc{d}_{w}"
            x = x + branch(pre, attention(pre, x) if (spec.use_self_attention and d == spec.depth - 1) else convnext(pre, x))
        if spec.use_output_normalization and spec.use_ln and not spec.output_norm_at_heads:
            x = _layer_norm(x, P[f"enc{d}/out_ln/gamma"])
        if spec.level_activation:
            x = _act(x, a)
        nodes[d] = x
        if d != spec.depth - 1:
            k = spec.gaussian_kernel_size
            if spec.use_laplacian_averaging:
                smooth = _avg_pool_same(x, k)
            else:
                g = torch.from_numpy(U.gaussian_kernel_3((k, k)))
                smooth = _depthwise(x, g[:, :, None, None].repeat(1, 1, x.shape[-1], 1))
            nodes[d] = x - smooth
            if spec.downsample_type == "strides":
                x = _act(_conv(smooth[:, ::2, ::2, :], P[f"down{d}/kernel"]), a)
            elif spec.downsample_type == "maxpool":                   # MaxPooling2D(2, 2, same) + 1x1 (downsampling.py:56-68)
                mp = F.max_pool2d(smooth.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
                x = _act(_conv(mp, P[f"down{d}/kernel"]), a)
            else:                                                     # conv2d: 2 x 2, strides 2, same (downsampling.py:45-55)
                x = _act(_conv(smooth, P[f"down{d}/kernel"], stride=2), a)
    outs = {spec.depth - 1: nodes[spec.depth - 1]}
    for d in reversed(range(spec.depth - 1)):
        low = outs[d + 1]
        if spec.upsample_type == "upsample_laplacian_conv2d" and (a == "linear" or spec.upsample_linear):   # upsampling.py:80-90
            up = _up2(_conv(low, P[f"up{d}/kernel"]))
        elif spec.upsample_type in ("bilinear", "nn", "nearest"):     # UpSampling2D alone (upsampling.py:103-116)
            up = _up2(low) if spec.upsample_type == "bilinear" else _up2_nearest(low)
        else:
            up = _act(_conv(_up2_nearest(low) if spec.upsample_type == "upsample_nearest_conv2d" else _up2(low), P[f"up{d}/kernel"]), a)
        enc = nodes[d]
        if spec.use_attention_gates:                                  # AdditiveAttentionGate.call (custom_layers.py:805-832)
            yg = _conv(_layer_norm(enc, P[f"gate{d}/y_ln/gamma"]) if spec.use_ln else enc, P[f"gate{d}/y/kernel"])
            xg = _conv(_layer_norm(up, P[f"gate{d}/x_ln/gamma"]) if spec.use_ln else up, P[f"gate{d}/x/kernel"])
            z = xg + yg
            o = _multiplier(_conv(torch.where(z > 0, z, 0.1 * z), P[f"gate{d}/o/kernel"]), P[f"gate{d}/scale/w"])
            enc = enc * torch.sigmoid(4.0 * o)
        x = torch.cat([enc, up], dim=-1) if getattr(spec, "use_concat", False) else enc + up      # Concatenate / Add (:516-519)
        if spec.use_mix_project:                                      # backbone_unet_laplacian.py:521-527
            x = _act(_conv(x, P[f"mix{d}/kernel"]), a)
        for w in range(spec.width):
            pre = f"dec{d}_{w}"
            y = convnext(pre, x)
            # the Add (and the StochasticDepth in front of it) exists only when the channel counts agree: the first block behind a
            # Concatenate without mix projection maps 2 C -> C and stands alone (:557-560)
            x = x + branch(pre, y) if y.shape[-1] == x.shape[-1] else y
        if spec.use_output_normalization and spec.use_ln and not spec.output_norm_at_heads:
            x = _layer_norm(x, P[f"dec{d}/out_ln/gamma"])
        outs[d] = x
    if spec.use_output_normalization and spec.use_ln and spec.output_norm_at_heads:
        last = spec.depth - 1
        outs = {d: _layer_norm(outs[d], P[f"enc{d}/out_ln/gamma" if d == last else f"dec{d}/out_ln/gamma"]) for d in outs}
    return [outs[d] for d in range(spec.depth)]


def hydra(spec, P, x, depth_scale=None, attn_scale=None):
    xn = torch.clamp(x, spec.v_min, spec.v_max) / (spec.v_max - spec.v_min) - 0.5
    outs = []
    for i, f in enumerate(backbone(spec, P, xn, depth_scale, attn_scale)):
        h = _act(_conv(f, P[f"head{i}/conv0/kernel"]), spec.head_activation)
        h = _conv(h, P[f"head{i}/conv1/kernel"])
        p = torch.tanh(2.0 * h) * 0.51
        outs.append((torch.clamp(p, -0.5, 0.5) + 0.5) * (spec.v_max - spec.v_min) + spec.v_min)
    return outs


# ---- losses ------------------------------------------------------------------------------------------------------------

def _keras_relu(x, threshold, max_value):
    return torch.clamp(torch.where(x > threshold, x, torch.zeros_like(x)), max=max_value)


def _ssim_mean(gt, pred, max_val=255.0):
    g = torch.from_numpy(O.ssim_gauss_kernel(7, 1.5))
    C = gt.shape[-1]
    k = g[None, None].repeat(C, 1, 1, 1)
    cf = lambda z: F.conv2d(z.permute(0, 3, 1, 2), k, groups=C)
    x, y = pred, gt
    a, b, s, q = cf(x), cf(y), cf(x * y), cf(x * x + y * y)
    c1, c2 = (0.01 * max_val) ** 2, (0.03 * max_val) ** 2
    S = (2 * a * b + c1) / (a * a + b * b + c1) * (2 * s - 2 * a * b + c2) / (q - a * a - b * b + c2)
    return S.mean()


def denoiser_loss(ls: O.LossSpec, gt, pred):
    e = gt - pred
    mae = lambda h, c: _keras_relu(e.abs(), h, c).mean()
    rmse = lambda h, c: torch.sqrt((_keras_relu(e, h, c) ** 2).mean(dim=(1, 2, 3)) + O.DEFAULT_EPSILON).mean()
    mae_pl = mae(ls.hinge, ls.cutoff) if ls.mae_multiplier > 0 else torch.zeros((), dtype=DT)
    mse_pl = rmse(ls.hinge, ls.cutoff * ls.cutoff) if ls.mse_multiplier > 0 else torch.zeros((), dtype=DT)
    ssim_l = 1.0 - _ssim_mean(gt, pred) if ls.ssim_multiplier > 0 else torch.zeros((), dtype=DT)
    return {"total_loss": mae_pl * ls.mae_multiplier + mse_pl * ls.mse_multiplier + ssim_l * ls.ssim_multiplier,
            "mae_loss": mae(0.0, 255.0), "mse_loss": rmse(0.0, 255.0), "ssim_loss": ssim_l}


def soft_orthonormal(w, lam=SOFTORTHONORMAL[0], l1=SOFTORTHONORMAL[1], l2=SOFTORTHONORMAL[2]):
    """regularizers.py:283-338 on a kernel [kh,kw,cin,cout]: wt = [cout, kh*kw*cin], G = wt wt^T."""
    wt = w.permute(3, 0, 1, 2).reshape(w.shape[3], -1)
    G = wt @ wt.T
    r = lam * ((G - torch.eye(G.shape[0], dtype=w.dtype)) ** 2).sum()
    if l1 > 0:
        r = r + l1 * G.abs().sum()
    if l2 > 0:
        r = r + l2 * (G * G).sum()
    return r


def regularizer_kind(spec: U.UnetLaplacianSpec, name: str, kind: str, soft_orthonormal_convnext: bool = True) -> str:
    """which keras regulariser the builder attaches to a tensor: l2 | soft_orthonormal | l1_1e-6 | none."""
    if kind == "ln_gamma":
        return "none"
    if kind == "multiplier":
        return "l1_1e-6"
    if kind == "depthwise":
        return "l2"
    leaf = name.split("/")[1]
    if name.startswith("gate"):                # AdditiveAttentionGate convolutions (custom_layers.py:726-740)
        return "soft_orthonormal" if soft_orthonormal_convnext else "l2_1e-4"
    if leaf in ("pw1", "pw2"):
        return "soft_orthonormal" if soft_orthonormal_convnext else "l2"
    if leaf in ("key", "query", "value", "out"):
        return "soft_orthonormal"
    return "l2"


def regularization(spec, P, soft_orthonormal_convnext=True):
    total = torch.zeros((), dtype=DT)
    for name, shape, kind in spec.tensors():
        rk = regularizer_kind(spec, name, kind, soft_orthonormal_convnext)
        if rk == "l2":
            total = total + 0.01 * (P[name] ** 2).sum()
        elif rk == "l2_1e-4":
            total = total + 1e-4 * (P[name] ** 2).sum()
        elif rk == "l1_1e-6":
            total = total + 1e-6 * P[name].abs().sum()
        elif rk == "soft_orthonormal":
            total = total + soft_orthonormal(P[name])
    return total


def ground_truth_pyramid(gt: np.ndarray, scales: int) -> List[np.ndarray]:
    """multiscales_generator_fn (utilities.py:625-685): 2x2 VALID average, clip, round, per extra scale."""
    out = [gt.astype(np.float64)]
    for _ in range(scales - 1):
        out.append(np.clip(O.round_half_even(O.avg_pool_valid_2x2(out[-1])), 0.0, 255.0))
    return out


def train_step(spec: U.UnetLaplacianSpec, ls: O.LossSpec, params: np.ndarray, gt: np.ndarray, noisy: np.ndarray,
               depth_weights, depth_scale: Optional[Dict[str, np.ndarray]] = None,
               attn_scale: Optional[Dict[str, np.ndarray]] = None, soft_orthonormal_convnext: bool = True):
    """train_step_single_gpu (bfcnn/train_loop.py:259-312) for the multi-output hydra: returns (total, model_loss dict,
    [denoiser loss dict per scale], [prediction per scale], flat gradient)."""
    check_trainable_graph(spec)
    flat = torch.tensor(np.asarray(params, np.float64), dtype=DT, requires_grad=True)
    P = views(spec, flat)
    ds = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in (depth_scale or {}).items()}
    at = {k: torch.from_numpy(np.asarray(v, np.float64)) for k, v in (attn_scale or {}).items()}
    preds = hydra(spec, P, torch.from_numpy(noisy.astype(np.float64)), ds, at)
    gts = ground_truth_pyramid(gt, spec.depth)
    total_d = torch.zeros((), dtype=DT)
    dls = []
    for i, p in enumerate(preds):
        dl = denoiser_loss(ls, torch.from_numpy(gts[i]), p)
        total_d = total_d + dl["total_loss"] * float(depth_weights[i])
        dls.append({k: float(v.detach()) for k, v in dl.items()})
    reg = regularization(spec, P, soft_orthonormal_convnext)
    total = total_d + reg * ls.regularization
    total.backward()
    return (float(total.detach()), {"regularization_loss": float(reg.detach()), "total_loss": float((reg * ls.regularization).detach())}, dls,
            [p.detach().numpy() for p in preds], flat.grad.numpy().copy())
