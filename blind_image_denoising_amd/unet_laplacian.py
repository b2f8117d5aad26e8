"""`unet_laplacian` backbone + one denoiser head per scale (bfcnn/backbone_unet_laplacian.py:35-615,
bfcnn/model.py:58-162, 251-359) on the operators of csrc/unet_ops.hip.  Inference only.

The builder walks the same graph the reference builder assembles from keras layers and issues one C-ABI operator
per fused group (torch tensors are the containers of the intermediate activations, nothing is computed by torch):

    first conv 5x5 (+normalise, +activation)
    per level d:   width x [dw kxk + LayerNorm] -> [1x1 C->4C, act, 1x1 4C->C, multiplier, +skip]   (ConvNextBlock)
                   or, on the deepest level with use_self_attention, ConvolutionalSelfAttention
                   LayerNorm + activation;  Laplacian split (smooth, x - smooth, ::2 slice);  1x1 + activation
    decoder d:     skip + act(bilinear x2 (1x1 (lower)))  ->  width x ConvNextBlock(dw 1x1)  ->  LayerNorm
    head i:        1x1 C_i -> 32 + activation,  1x1 32 -> 3, tanh(2x)*0.51, denormalise
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _native as N

LN_EPSILON = 1e-3          # DEFAULT_LN_EPSILON (bfcnn/constants.py:10); keras LayerNormalization default as well
ACT_CODES = {"linear": (0, 0.0), "relu": (1, 0.0), "leaky_relu": (2, 0.3), "leakyrelu": (2, 0.3),
             "leaky_relu_01": (2, 0.1), "leakyrelu_01": (2, 0.1), "leaky_relu_001": (2, 0.01), "leakyrelu_001": (2, 0.01),
             "gelu": (3, 0.0)}      # bfcnn/utilities.py:229-267


def _act(name: Optional[str]) -> Tuple[int, float]:
    name = (name or "linear").strip().lower()
    if name not in ACT_CODES:
        raise NotImplementedError(f"activation [{name}] is outside the hot path")
    return ACT_CODES[name]


def _call(fn_name: str, *args):
    N.check(getattr(N.lib(), fn_name)(*args), None, fn_name)


# ---------------------------------------------------------------------------------------------
# operator wrappers (float32 NHWC cuda tensors in, new tensor out)
# ---------------------------------------------------------------------------------------------

def pack_pointwise(w: torch.Tensor) -> torch.Tensor:
    """[1,1,cin,cout] / [cin,cout] kernel -> matrix-core operand order."""
    cin, cout = w.shape[-2], w.shape[-1]
    out = torch.empty(cin * cout, dtype=torch.float32, device=w.device)
    _call("bf_op_pack_pointwise", N.ptr(w.contiguous()), N.ptr(out), cin, cout, N.stream_ptr(w))
    return out


def pointwise(x: torch.Tensor, wp: torch.Tensor, cout: int, act: str = "linear", mult: torch.Tensor = None,
              res: torch.Tensor = None, alpha: Optional[float] = None) -> torch.Tensor:
    cin = x.shape[-1]
    npix = x.numel() // cin
    out = torch.empty(x.shape[:-1] + (cout,), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_pointwise", N.ptr(x), N.ptr(out), N.ptr(wp), N.ptr(mult), N.ptr(res), npix, cin, cout, code,
          a if alpha is None else alpha, N.stream_ptr(x))
    return out


def pointwise_ex(x: torch.Tensor, wp: torch.Tensor, cout: int, mode: int, act: str = "linear", mult: torch.Tensor = None,
                 res: torch.Tensor = None, add: torch.Tensor = None) -> torch.Tensor:
    """mode 1: act(x . w + res);  mode 2: res * sigmoid(4 * mult * (x . w)) + add   (AdditiveAttentionGate epilogues)."""
    cin = x.shape[-1]
    out = torch.empty(x.shape[:-1] + (cout,), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_pointwise_ex", N.ptr(x), N.ptr(out), N.ptr(wp), N.ptr(mult), N.ptr(res), N.ptr(add), x.numel() // cin, cin, cout,
          code, a, mode, N.stream_ptr(x))
    return out


def convnext_mlp(x: torch.Tensor, skip: Optional[torch.Tensor], w1p: torch.Tensor, w2p: torch.Tensor,
                 mult: Optional[torch.Tensor], act: str) -> torch.Tensor:
    C = x.shape[-1]
    out = torch.empty_like(x)
    code, a = _act(act)
    _call("bf_op_convnext_mlp", N.ptr(x), N.ptr(skip), N.ptr(out), N.ptr(w1p), N.ptr(w2p), N.ptr(mult), x.numel() // C, C,
          code, a, N.stream_ptr(x))
    return out


def pack_mlp_h3(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """[1,1,C,4C] and [1,1,4C,C] fp32 kernels -> split-f16 fragments of the f16x3 MLP kernel."""
    C = int(w1.shape[-2])
    nbytes = int(N.lib().bf_op_mlp_h3_pack_bytes(C))
    if nbytes < 0:
        raise NotImplementedError(f"split-f16 MLP: C={C} (32 and 64 are built)")
    out = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
    _call("bf_op_pack_mlp_h3", N.ptr(w1.contiguous()), N.ptr(w2.contiguous()), N.ptr(out), C, N.stream_ptr(w1))
    return out


def convnext_mlp_h3(x: torch.Tensor, skip: Optional[torch.Tensor], packed: torch.Tensor, mult: Optional[torch.Tensor],
                    act: str) -> torch.Tensor:
    C = x.shape[-1]
    out = torch.empty_like(x)
    code, a = _act(act)
    _call("bf_op_convnext_mlp_h3", N.ptr(x), N.ptr(skip), N.ptr(out), N.ptr(packed), N.ptr(mult), x.numel() // C, C, code, a,
          N.stream_ptr(x))
    return out


def convnext_block1_h3(x: torch.Tensor, dw: torch.Tensor, gamma: Optional[torch.Tensor], packed: torch.Tensor,
                       mult: Optional[torch.Tensor], act: str, eps: float = LN_EPSILON) -> torch.Tensor:
    """x + ConvNextBlock(x) for a block with a 1x1 depthwise convolution (dw [C]), one kernel."""
    C = x.shape[-1]
    out = torch.empty_like(x)
    code, a = _act(act)
    _call("bf_op_convnext_block1_h3", N.ptr(x), N.ptr(out), N.ptr(dw), N.ptr(gamma), eps, N.ptr(packed), N.ptr(mult),
          x.numel() // C, C, code, a, N.stream_ptr(x))
    return out


def convnext_block1_up_h3(enc: torch.Tensor, low: torch.Tensor, dw: torch.Tensor, gamma: Optional[torch.Tensor], packed: torch.Tensor,
                          mult: Optional[torch.Tensor], act: str, act_up: str = "linear", eps: float = LN_EPSILON) -> torch.Tensor:
    """x = enc + act_up(bilinear 2x of low); x + ConvNextBlock(x) (1x1 depthwise, 32 channels): upsample_act_add and
    convnext_block1_h3 as one kernel -- x is never written."""
    B, OH, OW, C = enc.shape
    if tuple(low.shape) != (B, OH // 2, OW // 2, C) or OH % 2 or OW % 2:
        raise ValueError(f"low-resolution map {tuple(low.shape)} does not sit under {tuple(enc.shape)}")
    out = torch.empty_like(enc)
    code, a = _act(act)
    ucode, ua = _act(act_up)
    _call("bf_op_convnext_block1_up_h3", N.ptr(enc), N.ptr(low), N.ptr(out), N.ptr(dw), N.ptr(gamma), eps, N.ptr(packed), N.ptr(mult),
          B, OH, OW, C, code, a, ucode, ua, N.stream_ptr(enc))
    return out


def pack_mlp_h3_chain(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """[32,128] and [128,32] fp32 kernels -> the split-f16 operand of convnext_chain32_h3"""
    if int(w1.shape[-2]) != 32:
        raise NotImplementedError("the chain kernel is built for 32 channels")
    out = torch.empty(int(N.lib().bf_op_mlp_h3_pack_bytes(32)), dtype=torch.uint8, device=w1.device)
    _call("bf_op_pack_mlp_h3_chain", N.ptr(w1.contiguous()), N.ptr(w2.contiguous()), N.ptr(out), 32, N.stream_ptr(w1))
    return out


def convnext_chain32_h3(x: torch.Tensor, low: Optional[torch.Tensor], blocks, act: str, act_up: str = "linear",
                        eps: float = LN_EPSILON) -> torch.Tensor:
    """1..3 pixel-wise ConvNext blocks of 32 channels in one kernel; blocks = [(packed, dw [32], gamma [32] | None, mult [32] | None)];
    low: the first block's input is x + act_up(bilinear 2x of low)"""
    import ctypes as C
    B, OH, OW, Cc = x.shape
    n = len(blocks)
    if Cc != 32 or not 1 <= n <= 3:
        raise ValueError(f"chain of {n} blocks on {Cc} channels: built for 1..3 blocks of 32 channels")
    if low is not None and (tuple(low.shape) != (B, OH // 2, OW // 2, Cc) or OH % 2 or OW % 2):
        raise ValueError(f"low-resolution map {tuple(low.shape)} does not sit under {tuple(x.shape)}")
    out = torch.empty_like(x)
    code, a = _act(act)
    ucode, ua = _act(act_up)
    arr = lambda k: (C.c_void_p * n)(*[N.ptr(b[k]) for b in blocks])
    _call("bf_op_convnext_chain32_h3", N.ptr(x), N.ptr(low), N.ptr(out), n, arr(0), arr(1), arr(2), arr(3), eps, B, OH, OW, code, a,
          ucode, ua, N.stream_ptr(x))
    return out


def convnext_block_h3(x: torch.Tensor, dw: torch.Tensor, gamma: Optional[torch.Tensor], packed: torch.Tensor,
                      mult: Optional[torch.Tensor], act: str, eps: float = LN_EPSILON) -> torch.Tensor:
    """x + ConvNextBlock(x) for a 32-channel block with a k x k depthwise convolution (dw [k,k,C], k = 3 | 5), one kernel."""
    B, H, W, C = x.shape
    out = torch.empty_like(x)
    code, a = _act(act)
    _call("bf_op_convnext_block_h3", N.ptr(x), N.ptr(out), N.ptr(dw), int(dw.shape[0]), N.ptr(gamma), eps, N.ptr(packed),
          N.ptr(mult), B, H, W, C, code, a, N.stream_ptr(x))
    return out


def dwconv_ln(x: torch.Tensor, w: Optional[torch.Tensor], gamma: Optional[torch.Tensor], act: str = "linear",
              eps: float = LN_EPSILON) -> torch.Tensor:
    """depthwise k x k (w [k,k,C,1] or None) -> LayerNorm(center=False) * gamma (or None) -> activation."""
    B, H, W, C = x.shape
    k = 0 if w is None else int(w.shape[0])
    out = torch.empty_like(x)
    code, a = _act(act)
    _call("bf_op_dwconv_ln", N.ptr(x), N.ptr(out), N.ptr(w), N.ptr(gamma), B, H, W, C, k, eps, code, a, N.stream_ptr(x))
    return out


def smooth_split(x: torch.Tensor, k: int, gauss: Optional[torch.Tensor] = None, down_stride: int = 2
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(x - smooth(x), smooth(x)[:, ::2, ::2, :]) or, with down_stride 1, (x - smooth(x), smooth(x))."""
    B, H, W, C = x.shape
    lap = torch.empty_like(x)
    down = torch.empty_like(x) if down_stride == 1 else \
        torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
    _call("bf_op_smooth_split", N.ptr(x), N.ptr(lap), N.ptr(down), N.ptr(gauss), B, H, W, C, k, down_stride, N.stream_ptr(x))
    return lap, down


def pack_conv(w: torch.Tensor) -> torch.Tensor:
    """[kh,kw,cin,cout] kernel -> tap-major matrix-core operand order."""
    kh, kw, cin, cout = w.shape
    return torch.cat([pack_pointwise(w[i, j].contiguous()) for i in range(kh) for j in range(kw)])


def conv2d(x: torch.Tensor, wp: torch.Tensor, cout: int, k: int, stride: int = 1, act: str = "linear",
           res: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Conv2D k x k, strides, padding="same": res + act(conv(x) + bias)."""
    B, H, W, cin = x.shape
    out = torch.empty((B, -(-H // stride), -(-W // stride), cout), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_conv2d", N.ptr(x), N.ptr(out), N.ptr(wp), N.ptr(res), N.ptr(bias), B, H, W, cin, cout, k, k, stride, code, a,
          N.stream_ptr(x))
    return out


def dwconv_mult(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act: str = "linear") -> torch.Tensor:
    """DepthwiseConv2D k x k with depth_multiplier m (w [k,k,C,m]) + bias + activation."""
    B, H, W, C = x.shape
    k, m = int(w.shape[0]), int(w.shape[-1])
    out = torch.empty((B, H, W, C * m), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_dwconv_mult", N.ptr(x), N.ptr(out), N.ptr(w), N.ptr(bias), B, H, W, C, m, k, code, a, N.stream_ptr(x))
    return out


def dwmult_pointwise(x: torch.Tensor, wd: torch.Tensor, bias1: Optional[torch.Tensor], act1: str, wp: torch.Tensor, cout: int,
                     bias2: Optional[torch.Tensor], act2: str, res: Optional[torch.Tensor]) -> torch.Tensor:
    """res + act2(act1(depthwise_kxk_xm(x) + bias1) . w + bias2), one kernel (wd [k,k,C,m])."""
    B, H, W, C = x.shape
    k, m = int(wd.shape[0]), int(wd.shape[-1])
    out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    (c1, a1), (c2, a2) = _act(act1), _act(act2)
    _call("bf_op_dwmult_pointwise", N.ptr(x), N.ptr(out), N.ptr(wd), N.ptr(bias1), c1, a1, N.ptr(wp), N.ptr(bias2), c2, a2,
          N.ptr(res), B, H, W, C, m, k, cout, N.stream_ptr(x))
    return out


def pack_bneck_h3(w0: torch.Tensor, wd: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """operand of bneck_block_h3: w0 [32,32], wd [3,3,32,4], w2 [128,32] (fp32, BatchNorm scales folded in)"""
    if tuple(w0.shape) != (32, 32) or tuple(wd.shape) != (3, 3, 32, 4) or tuple(w2.shape) != (128, 32):
        raise ValueError(f"bottleneck operands {tuple(w0.shape)}, {tuple(wd.shape)}, {tuple(w2.shape)}: built for 32 -> 32 -> 3x3 x4 -> 32")
    packed = torch.empty(int(N.lib().bf_op_bneck_h3_pack_bytes()), dtype=torch.uint8, device=w0.device)
    _call("bf_op_pack_bneck_h3", N.ptr(w0.contiguous()), N.ptr(wd.contiguous()), N.ptr(w2.contiguous()), N.ptr(packed), N.stream_ptr(w0))
    return packed


def bneck_block_h3(x: torch.Tensor, packed: torch.Tensor, shift0: Optional[torch.Tensor], act0: str, shift1: Optional[torch.Tensor],
                   act1: str, shift2: Optional[torch.Tensor], act2: str, add_res: bool = True) -> torch.Tensor:
    """[x +] act2(act1(depthwise3x3_x4(act0(x . w0 + shift0)) + shift1) . w2 + shift2), one kernel, split-f16 GEMMs"""
    B, H, W, C = x.shape
    out = torch.empty_like(x)
    (c0, a0), (c1, a1), (c2, a2) = _act(act0), _act(act1), _act(act2)
    _call("bf_op_bneck_block_h3", N.ptr(x), N.ptr(out), N.ptr(packed), N.ptr(shift0), c0, a0, N.ptr(shift1), c1, a1, N.ptr(shift2), c2, a2,
          int(bool(add_res)), B, H, W, N.stream_ptr(x))
    return out


def maxpool2(x: torch.Tensor) -> torch.Tensor:
    B, H, W, C = x.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
    _call("bf_op_maxpool2", N.ptr(x), N.ptr(out), B, H, W, C, N.stream_ptr(x))
    return out


def norm_smooth_split(x: torch.Tensor, gamma: Optional[torch.Tensor], act: str, k: int, gauss: Optional[torch.Tensor] = None,
                      eps: float = LN_EPSILON) -> Tuple[torch.Tensor, torch.Tensor]:
    """y = act(LayerNorm(x) * gamma) -> (y - smooth(y), smooth(y)[:, ::2, ::2, :]) in one kernel."""
    B, H, W, C = x.shape
    lap = torch.empty_like(x)
    down = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_norm_smooth_split", N.ptr(x), N.ptr(gamma), eps, code, a, N.ptr(gauss), N.ptr(lap), N.ptr(down), B, H, W, C, k,
          N.stream_ptr(x))
    return lap, down


def upsample_act_add(x: torch.Tensor, other: Optional[torch.Tensor], act: str = "linear") -> torch.Tensor:
    B, H, W, C = x.shape
    out = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_upsample_act_add", N.ptr(x), N.ptr(other), N.ptr(out), B, H, W, C, code, a, N.stream_ptr(x))
    return out


def conv2d_transpose(x: torch.Tensor, w: torch.Tensor, stride: int = 2, act: str = "linear") -> torch.Tensor:
    """Conv2DTranspose(k, strides s, padding "same", use_bias=False): upsample_type "conv2d_transpose"
    (bfcnn/upsampling.py:37-48); w [k,k,cout,cin] as keras stores it."""
    B, H, W, cin = x.shape
    k, cout = int(w.shape[0]), int(w.shape[2])
    if w.shape[1] != k or w.shape[3] != cin:
        raise ValueError(f"kernel {tuple(w.shape)} does not fit {cin} input channels")
    out = torch.empty((B, H * stride, W * stride, cout), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_conv2d_transpose", N.ptr(x), N.ptr(w), N.ptr(out), B, H, W, cin, cout, k, int(stride), code, a, N.stream_ptr(x))
    return out


def resize_bilinear(x: torch.Tensor, oh: int, ow: int) -> torch.Tensor:
    B, H, W, C = x.shape
    out = torch.empty((B, oh, ow, C), dtype=torch.float32, device=x.device)
    _call("bf_op_resize_bilinear", N.ptr(x), N.ptr(out), B, H, W, C, oh, ow, N.stream_ptr(x))
    return out


def attention(q: torch.Tensor, v: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    B, T, A = q.shape
    out = torch.empty_like(q)
    _call("bf_op_attention", N.ptr(q), N.ptr(v), N.ptr(k), N.ptr(out), B, T, A, N.stream_ptr(q))
    return out


def attention_interleaved(qvk: torch.Tensor, A: int) -> torch.Tensor:
    """qvk [B,T,3A] = query | value | key side by side (one 1x1 convolution wrote all three) -> softmax(q k^T) v, [B,T,A]."""
    B, T, ld = qvk.shape
    out = torch.empty((B, T, A), dtype=torch.float32, device=qvk.device)
    if not qvk.is_contiguous():
        raise ValueError("tensor must be contiguous")
    at = lambda floats: N.C.c_void_p(qvk.data_ptr() + 4 * floats)
    _call("bf_op_attention_ld", at(0), at(A), at(2 * A), N.ptr(out), B, T, A, ld, N.stream_ptr(qvk))
    return out


def first_conv(x: torch.Tensor, w: torch.Tensor, H: int, W: int, act: str, normalize: bool, v_min: float, v_max: float,
               arith: int = 0) -> torch.Tensor:
    """x [B,Hs,Ws,cin] uint8 / float32 (0..255 scale), zero-padded to [H,W] before normalisation.
    arith 1: split-f16 matrix-core kernel for the k x k 3 -> 32 shapes, k = 3, 5, 7 (csrc/unet_h3_first.hip); 0: exact fp32."""
    B, Hs, Ws, cin = x.shape
    k, cout = int(w.shape[0]), int(w.shape[-1])
    out = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    code, a = _act(act)
    if arith == 1 and k in (3, 5, 7) and (cin, cout) == (3, 32):
        _call("bf_op_first_conv_h3k", N.ptr(x), int(x.dtype == torch.uint8), N.ptr(out), N.ptr(w), B, Hs, Ws, H, W, k, int(normalize),
              v_min, v_max, code, a, N.stream_ptr(x))
        return out
    _call("bf_op_first_conv", N.ptr(x), int(x.dtype == torch.uint8), N.ptr(out), N.ptr(w), B, Hs, Ws, H, W, cin, cout, k,
          int(normalize), v_min, v_max, code, a, N.stream_ptr(x))
    return out


def head_out(x: torch.Tensor, w: torch.Tensor, Ho: int, Wo: int, as_uint8: bool, denormalize: bool, v_min: float,
             v_max: float, status: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, H, W, hf = x.shape
    cout = int(w.shape[-1])
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.uint8 if as_uint8 else torch.float32, device=x.device)
    _call("bf_op_head_out", N.ptr(x), N.ptr(w), N.ptr(out), int(as_uint8), B, H, W, Ho, Wo, hf, cout, int(denormalize),
          v_min, v_max, N.ptr(status), N.stream_ptr(x))
    return out


def head_fused(x: torch.Tensor, gamma: Optional[torch.Tensor], w0p: torch.Tensor, act: str, w1: torch.Tensor, Ho: int, Wo: int,
               as_uint8: bool, denormalize: bool, v_min: float, v_max: float, hf: int = 32, eps: float = LN_EPSILON,
               status: Optional[torch.Tensor] = None, arith: int = 0) -> torch.Tensor:
    """[LayerNorm * gamma] -> 1x1 C->hf + act -> 1x1 hf->cout -> tanh(2x)*0.51 -> denormalise [-> uint8], one kernel.
    arith 1: the first 1x1 with split-f16 operands on the f16 matrix cores for C = 32 / 64 (bf_op_head_fused_h3); 0: exact fp32."""
    B, H, W, C = x.shape
    cout = int(w1.shape[-1])
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.uint8 if as_uint8 else torch.float32, device=x.device)
    code, a = _act(act)
    _call("bf_op_head_fused_h3" if arith == 1 and C in (32, 64) and hf == 32 else "bf_op_head_fused", N.ptr(x), N.ptr(gamma), eps,
          N.ptr(w0p), code, a, N.ptr(w1), N.ptr(out), int(as_uint8), B, H, W, Ho, Wo, C, hf, cout, int(denormalize), v_min, v_max,
          N.ptr(status), N.stream_ptr(x))
    return out


def channel_multiplier(w: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(w)
    _call("bf_op_channel_multiplier", N.ptr(w), N.ptr(out), w.numel(), N.stream_ptr(w))
    return out


def gaussian_kernel(kernel_size: Tuple[int, int]) -> np.ndarray:
    """GaussianFilter's fixed kernel (custom_layers.py:146-158; utilities.py:272-321): nsig = (k-1)/2, sigma 1."""
    ax = [np.linspace(-(k - 1) / 2, (k - 1) / 2, k, dtype=np.float64) for k in kernel_size]
    gx, gy = np.meshgrid(ax[0], ax[1])
    g = np.exp(-(gx * gx + gy * gy) / 2.0)
    return (g / g.sum()).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# the model
# ---------------------------------------------------------------------------------------------

class UnetLaplacianHydra:
    """hydra(x) for backbone type "unet_laplacian": returns one denormalised output per scale, full resolution first
    (model.py:117-142).  Trainable tensors live in one flat float32 vector (`params`) in graph-construction order;
    `trainable_variables` lists (name, shape, kind, offset)."""

    multi_output = True           # DenoiserModule keeps output 0
    # True: when host arrays are handed back and the status word reports a non-finite value in front of a head's tanh (an
    # activation left the f16 range inside a split-f16 operator), switch to the exact-fp32 operators and repeat the call
    auto_exact_fallback = True

    class _Desc:
        def __init__(self, cin, cout):
            self.in_channels, self.out_channels = cin, cout

    def _status(self) -> torch.Tensor:
        """int32 status word on the device, cleared (bf_op_fill32) at the start of a forward, OR-ed by the head kernels."""
        if getattr(self, "_status_word", None) is None:
            self._status_word = torch.empty(1, dtype=torch.int32, device=self.device)
        _call("bf_op_fill32", N.ptr(self._status_word), 0, 1, N.stream_ptr(self._status_word))
        return self._status_word

    def status_tensor(self) -> Optional[torch.Tensor]:
        return getattr(self, "_status_word", None)

    def check_status(self, raise_on_overflow: bool = True) -> bool:
        """synchronises and reads the status word of the last forward (see HydraModel.check_status)."""
        st = self.status_tensor()
        if st is None or not (int(st.item()) & N.BF_STATUS_F16_RANGE):
            return True
        if raise_on_overflow:
            raise FloatingPointError("an activation left the f16 range inside the split-f16 operators; "
                                     "call set_option('arith', 0) to run the exact-fp32 operators")
        return False

    def set_option(self, key: str, value: int):
        if key not in ("arith", "fuse_up_block", "fuse_chain") or int(value) not in (0, 1):
            raise ValueError(f"unknown option {key}={value}")
        setattr(self, key, int(value))
        self.version = getattr(self, "version", 0) + 1

    def __init__(self, config: Dict, device=None, seed: Optional[int] = None):
        bb, dn = config["backbone"], config["denoiser"]
        self.config = config
        for key, want in dict(use_bn=False, use_bias=False,
                              use_complex_base=False, use_value_compressor=False, use_global_pool_information=False,
                              multiple_scale_outputs=True).items():
            default = True if key in ("multiple_scale_outputs",) else False
            if bb.get(key, default) != want:
                raise NotImplementedError(f"unet_laplacian: {key}={bb.get(key, default)} is outside the built graph")
        # Concatenate([encoder feature, upsampled]) instead of Add in the decoder nodes: the reference builder's default
        # (backbone_unet_laplacian.py:52, 516-517); every shipped configuration sets it to false
        self.use_concat = bool(bb.get("use_concat", True))
        self.downsample_type = bb.get("downsample_type", "strides").strip().lower()
        if self.downsample_type not in ("strides", "conv2d", "maxpool"):
            raise ValueError(f"don't know how to handle [{self.downsample_type}]")          # downsampling.py:73-75
        if dn.get("use_bn", False) or dn.get("use_ln", False) or dn.get("use_bias", False):
            raise NotImplementedError("denoiser head: use_bn / use_ln / use_bias are outside the built graph")
        self.depth = int(bb.get("depth", 5))
        self.width = int(bb.get("width", 1) or 1)
        if self.width <= 0:
            self.width = 1
        if self.depth <= 0:
            raise ValueError("depth and width must be > 0")                        # backbone_unet_laplacian.py:125-126
        rate = bb.get("convolutional_self_attention_dropout_rate", 0.0)
        if rate < 0 or rate > 1:
            raise ValueError("convolutional_self_attention_dropout_rate must be >= 0 and <= 1")
        if bb.get("use_soft_orthonormal_regularization", False) and bb.get("use_soft_orthogonal_regularization", False):
            raise ValueError("only one use_soft_orthonormal_regularization or use_soft_orthogonal_regularization "
                             "must be turned on")
        self.filters = int(bb.get("filters", 32))
        self.max_filters = int(bb.get("max_filters", -1))
        self.multiplier = float(bb.get("filters_level_multiplier", 2.0))
        self.in_channels = int(bb["input_shape"][-1])
        self.enc_k = int(bb.get("encoder_kernel_size", 5))
        self.dec_k = int(bb.get("decoder_kernel_size", 3))
        self.gauss_k = int(bb.get("gaussian_kernel_size", 3))
        self.activation = (bb.get("activation", "leaky_relu_01") or "leaky_relu_01").strip().lower()
        _act(self.activation)
        self.upsample_type = bb.get("upsample_type", "bilinear").strip().lower()
        if self.upsample_type == "conv2d_transpose":
            # the builder hands Conv2DTranspose the level's base parameters, strides (1, 1) (backbone_unet_laplacian.py:179-188,
            # 246-250; upsampling.py:52-59): nothing is upsampled and the decoder's Add cannot be built in the reference either
            raise NotImplementedError("unet_laplacian: upsample_type conv2d_transpose (stride 1 in the reference builder: the "
                                      "decoder Add has mismatched shapes there) is not built")
        if self.upsample_type not in ("upsample_laplacian_conv2d", "upsample_bilinear_conv2d", "upsample_nearest_conv2d",
                                      "bilinear", "nn", "nearest"):
            raise ValueError(f"don't know how to handle [{self.upsample_type}]")             # upsampling.py:118-120
        self.use_ln = bool(bb.get("use_ln", True))
        self.use_gamma = bool(bb.get("use_gamma", True))
        self.use_laplacian = bool(bb.get("use_laplacian", True))
        self.use_laplacian_averaging = bool(bb.get("use_laplacian_averaging", True))
        if not (self.use_laplacian or self.use_laplacian_averaging):
            raise NotImplementedError("unet_laplacian without the Laplacian split")
        self.use_mix_project = bool(bb.get("use_mix_project", True))
        self.use_self_attention = bool(bb.get("use_self_attention", False))
        self.use_attention_gates = bool(bb.get("use_attention_gates", False))
        if self.use_concat and self.use_attention_gates:
            raise NotImplementedError("unet_laplacian: use_concat together with use_attention_gates is outside the built graph")
        self.use_output_normalization = bool(bb.get("use_output_normalization", False))
        # keras 2.13 cannot resolve the string "leaky_relu" ConvolutionalSelfAttention gives its Conv2D layers
        # (custom_layers.py:1272-1282); later keras resolve it to negative_slope 0.2, which is what is built here
        self.attention_alpha = float(bb.get("attention_alpha", 0.2))
        self.attention_resolution = (16, 16)
        # graph revision of the reference's trained archive (pretrained/unet_laplacian_v5.6, older code than the snapshot
        # builder; keys of this package, set by keras_import.config_from_archive_graph -- oracle/unet_oracle.py explains them)
        self.mlp_activation = (bb.get("convnext_activation") or self.activation).strip().lower()
        _act(self.mlp_activation)
        self.level_activation = bool(bb.get("encoder_level_activation", True))
        self.output_norm_at_heads = bool(bb.get("output_normalization_at_heads", False))
        self.attention_rows = bool(bb.get("attention_full_resolution", False))
        self.attention_activation = (bb.get("attention_activation") or "").strip().lower()
        if self.attention_activation in ("leaky_relu", "leakyrelu"):
            self.attention_activation = ""                       # LeakyReLU(attention_alpha), see above
        if self.attention_activation:
            _act(self.attention_activation)
        self.upsample_linear = bool(bb.get("upsample_linear", False))
        vr = bb.get("value_range", [0, 255])
        self.v_min, self.v_max = float(vr[0]), float(vr[1])
        self.head_filters = int(dn.get("filters", 32))
        self.head_activation = (dn.get("activation", "linear") or "linear").strip().lower()
        _act(self.head_activation)
        self.out_channels = int(dn.get("output_channels", 3))
        for d in range(self.depth):
            C = self.level_filters(d)
            # 256 channels: the deepest level only -- as the self-attention bottleneck, or as ConvNext blocks whose MLP runs in
            # 128-channel slices of the hidden layer over the existing 1x1 operators (_convnext)
            ok = C in (32, 64, 128) or (C == 256 and d == self.depth - 1)
            if not ok:
                raise NotImplementedError(f"unet_laplacian: level {d} has {C} channels (ConvNext levels: 32/64/128, and 256 at the "
                                          f"deepest level)")
        if self.head_filters not in (32, 64, 128):
            raise NotImplementedError("denoiser head filters must be 32, 64 or 128")
        if self.use_concat and not self.use_mix_project and 2 * self.level_filters(max(self.depth - 2, 0)) > 128 and self.depth > 1:
            raise NotImplementedError("unet_laplacian: use_concat without use_mix_project runs the first decoder block of a level on 2 C "
                                      "channels; the depthwise + LayerNorm operator takes up to 128")

        self.desc = self._Desc(self.in_channels, self.out_channels)
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        # 1: ConvNext MLPs with 32 / 64 channels on the f16 matrix cores with split-f16 operands (csrc/unet_h3.hip);
        # 0: every GEMM in exact fp32 (csrc/unet_ops.hip).  Same tests, same bars.
        self.arith = 1
        # 1: a level's first decoder block forms its input enc + act(up(low)) while loading it (32 channels, 1x1 depthwise:
        # bf_op_convnext_block1_up_h3); 0: upsample_act_add writes the node and the block reads it back.  Same bits either way.
        self.fuse_up_block = 1
        # 1: the pixel-wise decoder blocks of a 32-channel level (decoder_kernel_size 1) run up to three per launch, the level's node formed
        # on load (bf_op_convnext_chain32_h3); 0: one launch per block
        self.fuse_chain = 1
        self._inventory = self._build_inventory()
        self.n_params = sum(int(np.prod(s)) for _, s, _ in self._inventory)
        self.params = torch.from_numpy(self._initial_values(seed)).to(self.device)
        self._packed = None

    @classmethod
    def from_keras_archive(cls, path: str, device=None) -> "UnetLaplacianHydra":
        """hydra with the graph and the trained tensors of a `.keras` archive written by `model.save`
        (bfcnn/export_model.py:106-110), e.g. the reference's pretrained/unet_laplacian_v5.6/model_hydra.keras."""
        from . import keras_import
        config, h5 = keras_import.read_archive(path)
        model = cls(config, device=device, seed=0)
        model.set_weights(keras_import.params_from_archive(config, h5, model._inventory))
        return model

    # -- inventory ---------------------------------------------------------------------------
    def level_filters(self, d: int) -> int:
        f = int(round(self.filters * max(1, self.multiplier ** d)))              # backbone_unet_laplacian.py:198-201
        return min(self.max_filters, f) if self.max_filters > 0 else f

    def _is_attention(self, d: int) -> bool:
        return self.use_self_attention and d == self.depth - 1                    # :324

    def _build_inventory(self) -> List[Tuple[str, Tuple[int, ...], str]]:
        out = [("base/kernel", (5, 5, self.in_channels, self.filters), "conv")]
        A = self.filters

        def block(prefix, C, k, attn, cin=None):
            cin = C if cin is None else cin              # first decoder block behind a Concatenate: 2 C channels in, C out
            if attn:
                if self.use_ln:
                    out.append((f"{prefix}/ln/gamma", (C,), "ln_gamma"))
                for n in ("key", "query", "value"):
                    out.append((f"{prefix}/{n}/kernel", (1, 1, C, A), "conv"))
                if self.attention_rows and self.use_ln:
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
                block(f"enc{d}_{w}", C, self.enc_k, self._is_attention(d))
            if self.use_output_normalization and self.use_ln and (d == self.depth - 1 or not self.output_norm_at_heads):
                out.append((f"enc{d}/out_ln/gamma", (C,), "ln_gamma"))
            if d != self.depth - 1:
                kd = 2 if self.downsample_type == "conv2d" else 1
                out.append((f"down{d}/kernel", (kd, kd, C, self.level_filters(d + 1)), "conv"))
        for d in reversed(range(self.depth - 1)):
            C = self.level_filters(d)
            if self.upsample_type == "upsample_laplacian_conv2d":
                out.append((f"up{d}/kernel", (1, 1, self.level_filters(d + 1), C), "conv"))
            elif self.upsample_type in ("upsample_bilinear_conv2d", "upsample_nearest_conv2d"):
                out.append((f"up{d}/kernel", (3, 3, self.level_filters(d + 1), C), "conv"))
            if self.use_attention_gates:                    # AdditiveAttentionGate.build order (custom_layers.py:749-790)
                out.append((f"gate{d}/x/kernel", (1, 1, C, C), "conv"))
                if self.use_ln:
                    out.append((f"gate{d}/x_ln/gamma", (C,), "ln_gamma"))
                out.append((f"gate{d}/y/kernel", (1, 1, C, C), "conv"))
                if self.use_ln:
                    out.append((f"gate{d}/y_ln/gamma", (C,), "ln_gamma"))
                out.append((f"gate{d}/o/kernel", (1, 1, C, C), "conv"))
                out.append((f"gate{d}/scale/w", (C,), "multiplier"))
            cat = 2 * C if self.use_concat else C
            if self.use_mix_project:
                out.append((f"mix{d}/kernel", (1, 1, cat, C), "conv"))
                cat = C
            for w in range(self.width):
                block(f"dec{d}_{w}", C, self.dec_k, False, cin=cat if w == 0 else C)
            if self.use_output_normalization and self.use_ln:
                out.append((f"dec{d}/out_ln/gamma", (C,), "ln_gamma"))
        for i in range(self.depth):
            out.append((f"head{i}/conv0/kernel", (1, 1, self.level_filters(i), self.head_filters), "conv"))
            out.append((f"head{i}/conv1/kernel", (1, 1, self.head_filters, self.out_channels), "conv"))
        return out

    @property
    def trainable_variables(self):
        o, res = 0, []
        for name, shape, kind in self._inventory:
            res.append((name, shape, kind, o))
            o += int(np.prod(shape))
        return res

    def count_params(self) -> int:
        return self.n_params

    def _initial_values(self, seed) -> np.ndarray:
        from .model import glorot_normal
        rng = np.random.default_rng(seed)
        parts = []
        for _, shape, kind in self._inventory:
            if kind in ("conv", "depthwise"):
                a = glorot_normal(shape, rng)
            elif kind == "ln_gamma":
                a = np.ones(shape)
            else:                                                # truncated_normal(0, 0.01) (custom_layers.py:271)
                a = rng.normal(0.0, 0.01, shape)
                bad = np.abs(a) > 0.02
                while bad.any():
                    a[bad] = rng.normal(0.0, 0.01, int(bad.sum()))
                    bad = np.abs(a) > 0.02
            parts.append(np.asarray(a, np.float32).ravel())
        return np.concatenate(parts)

    def mark_dirty(self):
        """the flat parameter vector changed in place (optimizer step): drop the packed operands"""
        self._packed = None
        self.version = getattr(self, "version", 0) + 1

    def get_weights(self) -> np.ndarray:
        return self.params.detach().cpu().numpy()

    def set_weights(self, params: np.ndarray):
        params = np.ascontiguousarray(params, np.float32).ravel()
        if params.size != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {params.size}")
        self.params.copy_(torch.from_numpy(params))
        self.mark_dirty()

    # -- packing -----------------------------------------------------------------------------
    def _pack(self) -> Dict[str, torch.Tensor]:
        """device views of the tensors; 1x1 kernels additionally in matrix-core order, multipliers as tanh(relu(1+w))."""
        if self._packed is not None:
            return self._packed
        P: Dict[str, torch.Tensor] = {}
        for name, shape, kind, off in self.trainable_variables:
            n = int(np.prod(shape))
            t = self.params[off:off + n]
            if off % 4:                                          # operators want 16-byte aligned buffers
                t = t.clone()
            t = t.view(shape)
            if kind == "conv" and shape[0] == 1 and shape[2] * shape[3] >= 256 * 1024:
                P[name] = t.contiguous()                         # (256-channel MLP: packed in slices below)
            elif kind == "conv" and shape[0] == 1 and shape[2] % 16 == 0 and shape[3] % 16 == 0:
                P[name] = pack_pointwise(t)
            elif kind == "conv" and shape[0] > 1 and shape[2] % 16 == 0 and shape[3] % 16 == 0:
                P[name] = pack_conv(t)
            elif kind == "multiplier":
                P[name] = channel_multiplier(t)
            elif kind == "depthwise":
                P[name] = t.reshape(shape[0], shape[1], shape[2]).contiguous()
            else:
                P[name] = t.contiguous()
        for name, shape, kind, off in self.trainable_variables:
            # 256-channel ConvNext MLP (256 -> 1024 -> 256): the hidden layer in eight slices of 128 channels, each a pair of 1x1
            # operators the library has (256 -> 128, 128 -> 256); the full kernels are not packed (no operator takes them)
            if name.endswith("/pw1/kernel") and shape[2] == 256 and shape[3] == 1024:
                prefix = name[:-len("/pw1/kernel")]
                w1 = self.params[off:off + 256 * 1024].view(256, 1024)
                o2 = dict((v[0], v[3]) for v in self.trainable_variables)[f"{prefix}/pw2/kernel"]
                w2 = self.params[o2:o2 + 1024 * 256].view(1024, 256)
                P[f"{prefix}/mlp256"] = [(pack_pointwise(w1[:, 128 * j:128 * (j + 1)].contiguous().view(1, 1, 256, 128)),
                                          pack_pointwise(w2[128 * j:128 * (j + 1)].contiguous().view(1, 1, 128, 256))) for j in range(8)]
        if self.use_concat and self.use_mix_project:
            # the 1x1 behind a Concatenate([enc, up]) = enc . W[:C] + up . W[C:]: the two halves as operands of their own, the
            # concatenated map is never formed
            for name, shape, kind, off in self.trainable_variables:
                if name.startswith("mix") and name.endswith("/kernel"):
                    C = shape[3]
                    w = self.params[off:off + 2 * C * C].view(2 * C, C)
                    P[name[:-len("/kernel")] + "/a"] = pack_pointwise(w[:C].contiguous().view(1, 1, C, C))
                    P[name[:-len("/kernel")] + "/b"] = pack_pointwise(w[C:].contiguous().view(1, 1, C, C))
        for name, shape, kind, off in self.trainable_variables:
            if name.endswith("/pw1/kernel") and shape[2] in (32, 64) and shape[3] == 4 * shape[2]:
                prefix = name[:-len("/pw1/kernel")]
                n1, n2 = int(np.prod(shape)), int(np.prod(shape))
                o2 = dict((v[0], v[3]) for v in self.trainable_variables)[f"{prefix}/pw2/kernel"]
                P[f"{prefix}/mlp_h3"] = pack_mlp_h3(self.params[off:off + n1].clone().view(shape[2], shape[3]),
                                                   self.params[o2:o2 + n2].clone().view(shape[3], shape[2]))
                if shape[2] == 32 and prefix.startswith("dec") and self.dec_k == 1:      # pixel-wise decoder blocks: the chain kernel's operand
                    P[f"{prefix}/mlp_h3c"] = pack_mlp_h3_chain(self.params[off:off + n1].clone().view(shape[2], shape[3]),
                                                               self.params[o2:o2 + n2].clone().view(shape[3], shape[2]))
        if self.filters == 32:
            # the three projections of an attention block as ONE 1x1 convolution 128 -> 96: query | "value" operand | "key" operand side
            # by side = query_conv | key_conv | value_conv in the archive's wiring (attention_rows), query_conv | value_conv | key_conv in
            # the snapshot builder's (custom_layers.py:1353-1360) -- the order `_attention` hands them to the attention operator in
            tv = {v[0]: v for v in self.trainable_variables}
            order = ("query", "key", "value") if self.attention_rows else ("query", "value", "key")
            for name, (_, shape, _, off) in list(tv.items()):
                if name.endswith("/query/kernel") and shape[2] == 128:
                    prefix = name[:-len("/query/kernel")]
                    mats = []
                    for nm in order:
                        o = tv[f"{prefix}/{nm}/kernel"][3]
                        mats.append(self.params[o:o + shape[2] * shape[3]].view(shape[2], shape[3]))
                    P[f"{prefix}/qvk"] = pack_pointwise(torch.cat(mats, dim=1).contiguous().view(1, 1, shape[2], 3 * shape[3]))
        if not self.use_laplacian_averaging:
            P["gauss"] = torch.from_numpy(gaussian_kernel((self.gauss_k, self.gauss_k))).to(self.device)
        self._packed = P
        return P

    # -- forward -----------------------------------------------------------------------------
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("unet_laplacian inference needs the GPU: there is no CPU execution path")

    def _fused_up_block(self, P, d: int, C: int, skip, low: torch.Tensor, up_act: str) -> bool:
        """the node enc + act(up(low)) can be formed inside the level's first decoder block (bf_op_convnext_block1_up_h3): 32 channels,
        split-f16 MLP, 1x1 depthwise, nothing between the Add and the block (no gate, no Concatenate, no mix projection)"""
        prefix = f"dec{d}_0"
        return bool(self.fuse_up_block and skip is not None and C == 32 and self.arith == 1 and self.width >= 1
                    and not self.use_mix_project and f"{prefix}/mlp_h3" in P and P[f"{prefix}/dw/kernel"].shape[0] == 1
                    and _act(up_act)[0] in (0, 1, 2) and low.shape[1] * 2 == skip.shape[1] and low.shape[2] * 2 == skip.shape[2])

    def _chain_len(self, P, d: int, w: int, C: int) -> int:
        """how many of the level's decoder blocks from block w on run in one launch of the chain kernel (0: none)"""
        if not (self.fuse_chain and self.arith == 1 and C == 32 and self.dec_k == 1) or _act(self.mlp_activation)[0] == 0:
            return 0
        n = 0
        while n < 3 and w + n < self.width and f"dec{d}_{w + n}/mlp_h3c" in P and tuple(P[f"dec{d}_{w + n}/dw/kernel"].shape) == (1, 1, 32):
            n += 1
        return n

    def _chain_block(self, P, prefix: str):
        return (P[f"{prefix}/mlp_h3c"], P[f"{prefix}/dw/kernel"].view(-1), P.get(f"{prefix}/ln/gamma") if self.use_ln else None,
                P.get(f"{prefix}/gamma/w") if self.use_gamma else None)

    def _convnext_up(self, P, prefix: str, enc: torch.Tensor, low: torch.Tensor, up_act: str) -> torch.Tensor:
        mult = P.get(f"{prefix}/gamma/w") if self.use_gamma else None
        gamma = P.get(f"{prefix}/ln/gamma") if self.use_ln else None
        return convnext_block1_up_h3(enc, low, P[f"{prefix}/dw/kernel"].view(-1), gamma, P[f"{prefix}/mlp_h3"], mult, self.mlp_activation,
                                     up_act)

    def _convnext(self, P, prefix: str, x: torch.Tensor) -> torch.Tensor:
        mult = P.get(f"{prefix}/gamma/w") if self.use_gamma else None
        gamma = P.get(f"{prefix}/ln/gamma") if self.use_ln else None
        dw = P[f"{prefix}/dw/kernel"]
        if self.arith == 1 and f"{prefix}/mlp_h3" in P and dw.shape[0] == 1:
            return convnext_block1_h3(x, dw.view(-1), gamma, P[f"{prefix}/mlp_h3"], mult, self.mlp_activation)
        if self.arith == 1 and f"{prefix}/mlp_h3" in P and dw.shape[0] in (3, 5) and x.shape[-1] == 32:
            return convnext_block_h3(x, dw, gamma, P[f"{prefix}/mlp_h3"], mult, self.mlp_activation)
        t = dwconv_ln(x, dw, gamma)
        if f"{prefix}/mlp256" in P:
            # 256 channels: out = x + m * sum_j act(t . W1[:, j]) . W2[j, :] over eight 128-channel slices of the hidden layer
            acc = None
            for w1p, w2p in P[f"{prefix}/mlp256"]:
                hdn = pointwise(t, w1p, 128, self.mlp_activation)
                acc = pointwise(hdn, w2p, 256, "linear", res=acc)
            out = torch.empty_like(x)
            B = x.shape[0]
            _call("bf_op_scale_add", N.ptr(x), N.ptr(acc), N.ptr(mult), None, N.ptr(out), B, x.numel() // (B * 256), 256, N.stream_ptr(x))
            return out
        if self.arith == 1 and f"{prefix}/mlp_h3" in P:
            return convnext_mlp_h3(t, x, P[f"{prefix}/mlp_h3"], mult, self.mlp_activation)
        return convnext_mlp(t, x, P[f"{prefix}/pw1/kernel"], P[f"{prefix}/pw2/kernel"],
                            P.get(f"{prefix}/gamma/w") if self.use_gamma else None, self.mlp_activation)

    def _attention(self, P, prefix: str, x: torch.Tensor) -> torch.Tensor:
        B, H, W, C = x.shape
        A = self.filters
        qkv_act = dict(act=self.attention_activation) if self.attention_activation else \
            dict(act="leaky_relu", alpha=self.attention_alpha)
        if self.attention_rows:
            # archive revision: no resize, one sequence per image row, operands in the order the archive's graph wires them
            # (scores = query_conv . value_conv^T, output = softmax . key_conv), LayerNorm on the product
            t = dwconv_ln(x, None, P[f"{prefix}/ln/gamma"]) if self.use_ln else x
            if f"{prefix}/qvk" in P:                           # the three projections in one pass over t
                t = attention_interleaved(pointwise(t, P[f"{prefix}/qvk"], 3 * A, **qkv_act).view(B * H, W, 3 * A), A).view(B, H, W, A)
            else:
                q, v, k = (pointwise(t, P[f"{prefix}/{n}/kernel"], A, **qkv_act).view(B * H, W, A) for n in ("query", "key", "value"))
                t = attention(q, v, k).view(B, H, W, A)
            if self.use_ln:
                t = dwconv_ln(t, None, P[f"{prefix}/ln1/gamma"])
            return pointwise(t, P[f"{prefix}/out/kernel"], C, "linear", mult=P[f"{prefix}/gamma/w"], res=x)
        rh, rw = self.attention_resolution
        t = resize_bilinear(x, rh, rw)
        if self.use_ln:
            t = dwconv_ln(t, None, P[f"{prefix}/ln/gamma"])
        if f"{prefix}/qvk" in P:                               # the three projections in one pass over t
            t = attention_interleaved(pointwise(t, P[f"{prefix}/qvk"], 3 * A, **qkv_act).view(B, rh * rw, 3 * A), A).view(B, rh, rw, A)
        else:
            q, v, k = (pointwise(t, P[f"{prefix}/{n}/kernel"], A, **qkv_act).view(B, rh * rw, A) for n in ("query", "value", "key"))
            t = attention(q, v, k).view(B, rh, rw, A)
        t = resize_bilinear(t, H, W)
        return pointwise(t, P[f"{prefix}/out/kernel"], C, "linear", mult=P[f"{prefix}/gamma/w"], res=x)

    def backbone(self, x: torch.Tensor, H: int, W: int, defer_output_norm: bool = False) -> List[torch.Tensor]:
        """x: [B,Hs,Ws,cin] image on the value_range scale (zero-padded to [H,W]); returns the per-scale feature maps.
        defer_output_norm: level 0 is returned BEFORE its output LayerNorm (nothing else in the graph reads that
        normalised map -- the deeper ones feed the decoder above them -- so the fused head kernel applies it)."""
        P = self._pack()
        step = 2 ** (self.depth - 1)
        if H % step or W % step:
            raise ValueError(f"height and width must be multiples of {step} (the decoder adds x2-upsampled maps to the "
                             f"::2-sliced ones; got {H}x{W})")
        a = self.activation
        f = first_conv(x, P["base/kernel"], H, W, a, True, self.v_min, self.v_max, arith=self.arith)
        nodes = {}
        for d in range(self.depth):
            for w in range(self.width):
                f = self._attention(P, f"enc{d}_{w}", f) if self._is_attention(d) else self._convnext(P, f"enc{d}_{w}", f)
            inline_norm = self.use_output_normalization and self.use_ln and not self.output_norm_at_heads
            gamma = P[f"enc{d}/out_ln/gamma"] if inline_norm else None
            la = a if self.level_activation else "linear"
            if d != self.depth - 1:
                gauss = None if self.use_laplacian_averaging else P["gauss"]
                Cn = self.level_filters(d + 1)
                if self.downsample_type == "strides":                     # downsampling.py:60-72
                    if self.gauss_k in (3, 5):   # output LayerNorm + activation + Laplacian split in one kernel
                        lap, down = norm_smooth_split(f, gamma, la, self.gauss_k, gauss)
                    else:
                        lap, down = smooth_split(self._level_out(f, gamma, la), self.gauss_k, gauss)
                    f = pointwise(down, P[f"down{d}/kernel"], Cn, a)
                else:
                    lap, smooth = smooth_split(self._level_out(f, gamma, la), self.gauss_k, gauss, down_stride=1)
                    if self.downsample_type == "conv2d":                   # 2x2 stride 2 (:45-55)
                        f = conv2d(smooth, P[f"down{d}/kernel"], Cn, 2, 2, a)
                    else:                                                   # maxpool + 1x1 (:56-68)
                        f = pointwise(maxpool2(smooth), P[f"down{d}/kernel"], Cn, a)
                nodes[d] = lap
            else:
                f = self._level_out(f, gamma, la)
                nodes[d] = f
        outs = {self.depth - 1: nodes[self.depth - 1]}
        for d in reversed(range(self.depth - 1)):
            low = outs[d + 1]
            C = self.level_filters(d)
            # gated: the Add happens in the gate kernel; Concatenate: the up-sampled map stays on its own
            skip = None if (self.use_attention_gates or self.use_concat) else nodes[d]
            first_done = 0                                                 # 1: block dec{d}_0 already ran, fused with the up-sampling
            if self.upsample_type == "upsample_laplacian_conv2d":
                # 1x1 and the bilinear resize are both linear: the 1x1 runs on the low-resolution map (1/4 of the work;
                # upsampling.py:80-90 makes the same exchange itself when the activation is linear)
                low = pointwise(low, P[f"up{d}/kernel"], C, "linear")
                up_act = "linear" if self.upsample_linear else a
                if self._fused_up_block(P, d, C, skip, low, up_act) and self._chain_len(P, d, 0, C):
                    first_done = self._chain_len(P, d, 0, C)                # the node and the level's first blocks in one launch
                    f = convnext_chain32_h3(skip, low, [self._chain_block(P, f"dec{d}_{j}") for j in range(first_done)],
                                            self.mlp_activation, up_act)
                elif self._fused_up_block(P, d, C, skip, low, up_act):
                    f, first_done = self._convnext_up(P, f"dec{d}_0", skip, low, up_act), 1
                else:
                    f = upsample_act_add(low, skip, up_act)
            elif self.upsample_type in ("upsample_bilinear_conv2d", "upsample_nearest_conv2d"):
                from .pyramid import upsample_2x                           # UpSampling2D, then Conv2D 3x3 + activation
                up = upsample_2x(low, bilinear=self.upsample_type == "upsample_bilinear_conv2d")
                f = conv2d(up, P[f"up{d}/kernel"], C, 3, 1, a, res=skip)
            else:
                if low.shape[-1] != nodes[d].shape[-1]:
                    raise ValueError(f"Add of [{nodes[d].shape[-1]}] and [{low.shape[-1]}] channels: upsample_type "
                                     f"[{self.upsample_type}] needs equal filters on both levels")
                if self.upsample_type == "bilinear" and self._fused_up_block(P, d, C, skip, low, "linear") and self._chain_len(P, d, 0, C):
                    first_done = self._chain_len(P, d, 0, C)
                    f = convnext_chain32_h3(skip, low, [self._chain_block(P, f"dec{d}_{j}") for j in range(first_done)],
                                            self.mlp_activation, "linear")
                elif self.upsample_type == "bilinear" and self._fused_up_block(P, d, C, skip, low, "linear"):
                    f, first_done = self._convnext_up(P, f"dec{d}_0", skip, low, "linear"), 1
                elif self.upsample_type == "bilinear":
                    f = upsample_act_add(low, skip, "linear")
                else:
                    from .pyramid import upsample_2x
                    f = upsample_2x(low, other=skip, bilinear=False)
            if self.use_attention_gates:
                # AdditiveAttentionGate([encoder feature, upsampled]) (backbone_unet_laplacian.py:497-509; custom_layers.py:805-832)
                # then Add([gated encoder feature, upsampled]): f = enc * sigmoid(4 * scale * conv_o(lrelu(conv_x(LN up) + conv_y(LN enc)))) + up
                enc, up = nodes[d], f
                y = pointwise(dwconv_ln(enc, None, P.get(f"gate{d}/y_ln/gamma")) if self.use_ln else enc, P[f"gate{d}/y/kernel"], C)
                o = pointwise_ex(dwconv_ln(up, None, P.get(f"gate{d}/x_ln/gamma")) if self.use_ln else up, P[f"gate{d}/x/kernel"], C,
                                 1, "leaky_relu_01", res=y)
                f = pointwise_ex(o, P[f"gate{d}/o/kernel"], C, 2, mult=P[f"gate{d}/scale/w"], res=enc, add=up)
            first = first_done
            if self.use_concat and self.use_mix_project:
                # act(Concatenate([enc, up]) . W) = act(enc . W[:C] + up . W[C:])
                t = pointwise(nodes[d], P[f"mix{d}/a"], C, "linear")
                f = pointwise_ex(f, P[f"mix{d}/b"], C, 1, a, res=t)
            elif self.use_concat:
                # first decoder block on the 2 C channels of Concatenate([enc, up]): depthwise + LayerNorm over 2 C, 1x1 2C -> 4C,
                # 1x1 4C -> C, multiplier, NO skip (the channel counts differ: backbone_unet_laplacian.py:557-560)
                cat = torch.empty(f.shape[:-1] + (2 * C,), dtype=torch.float32, device=f.device)
                _call("bf_op_concat_channels", N.ptr(nodes[d]), N.ptr(f), None, N.ptr(cat), f.numel() // C, C, C, 0, N.stream_ptr(f))
                pre = f"dec{d}_0"
                t = dwconv_ln(cat, P[f"{pre}/dw/kernel"], P.get(f"{pre}/ln/gamma") if self.use_ln else None)
                hdn = pointwise(t, P[f"{pre}/pw1/kernel"], 4 * C, self.mlp_activation)
                f = pointwise(hdn, P[f"{pre}/pw2/kernel"], C, "linear", mult=P.get(f"{pre}/gamma/w") if self.use_gamma else None)
                first = 1
            elif self.use_mix_project:
                f = pointwise(f, P[f"mix{d}/kernel"], self.level_filters(d), a)
            w = first
            while w < self.width:
                nb = self._chain_len(P, d, w, C)
                if nb:                                                      # pixel-wise blocks of 32 channels: up to three per launch
                    f = convnext_chain32_h3(f, None, [self._chain_block(P, f"dec{d}_{w + j}") for j in range(nb)], self.mlp_activation)
                    w += nb
                else:
                    f = self._convnext(P, f"dec{d}_{w}", f)
                    w += 1
            if self.use_output_normalization and self.use_ln and not self.output_norm_at_heads \
                    and not (defer_output_norm and d == 0):
                f = dwconv_ln(f, None, P[f"dec{d}/out_ln/gamma"])
            outs[d] = f
        if self.use_output_normalization and self.use_ln and self.output_norm_at_heads and not defer_output_norm:
            outs = {d: dwconv_ln(f, None, P[self._out_ln_name(d)]) for d, f in outs.items()}
        return [outs[d] for d in range(self.depth)]

    def _out_ln_name(self, d: int) -> str:
        return f"enc{d}/out_ln/gamma" if d == self.depth - 1 else f"dec{d}/out_ln/gamma"

    @staticmethod
    def _level_out(f: torch.Tensor, gamma: Optional[torch.Tensor], act: str) -> torch.Tensor:
        return f if (gamma is None and act == "linear") else dwconv_ln(f, None, gamma, act)

    def _head(self, P, i: int, f: torch.Tensor, Ho: int, Wo: int, as_uint8: bool, deferred_norm: bool = False,
              status: Optional[torch.Tensor] = None) -> torch.Tensor:
        gamma = None
        if deferred_norm and self.use_output_normalization and self.use_ln:
            if self.output_norm_at_heads:                      # every scale arrives un-normalised
                gamma = P[self._out_ln_name(i)]
            elif i == 0 and self.depth > 1:
                gamma = P[f"dec{i}/out_ln/gamma"]
        if self.head_filters == 32:
            return head_fused(f, gamma, P[f"head{i}/conv0/kernel"], self.head_activation, P[f"head{i}/conv1/kernel"], Ho, Wo,
                              as_uint8, True, self.v_min, self.v_max, status=status, arith=self.arith)
        if gamma is not None:
            f = dwconv_ln(f, None, gamma)
        h = pointwise(f, P[f"head{i}/conv0/kernel"], self.head_filters, self.head_activation)
        return head_out(h, P[f"head{i}/conv1/kernel"], Ho, Wo, as_uint8, True, self.v_min, self.v_max, status=status)

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
        """float32 (or uint8) [B,H,W,3] on the 0..255 scale -> list of float32 outputs, full resolution first."""
        if training:
            raise NotImplementedError("unet_laplacian: hydra(x, training=True) on its own is not built; the training step "
                                      "(forward + losses + gradients) is build_train_functions(...).train_step_single_gpu")
        self._require_gpu()
        x, was_numpy = self._as_device(x)
        B, H, W, _ = x.shape
        P = self._pack()
        outs = []
        status = self._status()
        for i, f in enumerate(self.backbone(x, H, W, defer_output_norm=True)):
            outs.append(self._head(P, i, f, f.shape[1], f.shape[2], False, deferred_norm=True, status=status))
        if was_numpy:
            torch.cuda.synchronize(self.device)
            if not self.check_status(raise_on_overflow=not (self.auto_exact_fallback and self.arith != 0)):
                self.set_option("arith", 0)
                return self(x.cpu().numpy())
            return [o.cpu().numpy() for o in outs]
        return outs

    def predict(self, x):
        return self(x)

    def infer_u8(self, image: torch.Tensor, cast_to_uint8: bool = True) -> torch.Tensor:
        """DenoiserModule.__call__ for this model (module_denoiser.py:46-75): pad to a power of two, hydra, first output,
        crop, round half to even, cast.  Only the full-resolution head is evaluated."""
        from .utilities import next_power_of_2
        self._require_gpu()
        B, Hs, Ws, _ = image.shape
        H, W = next_power_of_2(Hs), next_power_of_2(Ws)
        P = self._pack()
        status = self._status()
        f = self.backbone(image, H, W, defer_output_norm=True)[0]
        return self._head(P, 0, f, Hs, Ws, cast_to_uint8, deferred_norm=True, status=status)
