"""The three layers the reference exports at package level (bfcnn/__init__.py:25-28; custom_layers.py:107-127, 1028-1160) as callables
on device tensors, over the same C-ABI operators the resnet builder's flags use:
`Multiplier` / `ChannelwiseMultiplier`: x * activation(w0 + w1) with one scalar / one value per channel, w0 trainable (zeros at
creation), w1 the constant `multiplier`; `RandomOnOff`: Dropout of a whole sample's tensor (training only)."""
from typing import Optional

import numpy as np
import torch

from . import _native as N


def _check_gpu(x):
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dim() < 2:
        raise RuntimeError("the layer runs on MI355X device tensors [B, ..., C]: there is no CPU execution path")
    return x.to(torch.float32).contiguous()


class _ScaleLayer:
    _per_channel = False

    def __init__(self, multiplier: float = 1.0, regularizer=None, trainable: bool = True, activation: str = "linear", name=None, **kwargs):
        act = (activation or "linear").strip().lower()
        if act not in ("linear", "relu"):
            raise NotImplementedError(f"{type(self).__name__}: activation [{activation}] (linear and relu are built)")
        self._multiplier, self._activation, self.regularizer, self.trainable, self.name = float(multiplier), act, regularizer, trainable, name
        self.w0: Optional[torch.Tensor] = None

    def build(self, channels: int, device):
        self.w0 = torch.zeros(channels if self._per_channel else 1, dtype=torch.float32, device=device)   # zeros at creation

    def __call__(self, inputs):
        x = _check_gpu(inputs)
        C = int(x.shape[-1])
        if self.w0 is None:
            self.build(C, x.device)
        L = N.lib()
        m = torch.empty(C, dtype=torch.float32, device=x.device)
        fn = L.bf_op_relu_shift if self._activation == "relu" else L.bf_op_linear_shift
        N.check(fn(N.ptr(self.w0), int(self.w0.numel()), self._multiplier, N.ptr(m), C, N.stream_ptr(m)), None, "bf_op_relu_shift")
        B = int(x.shape[0])
        out = torch.empty_like(x)
        N.check(L.bf_op_scale_add(None, N.ptr(x), N.ptr(m), None, N.ptr(out), B, x.numel() // (B * C), C, N.stream_ptr(x)), None,
                "bf_op_scale_add")
        return out

    @property
    def weights(self):
        """[w0 (trainable), w1 (constant)] as host tensors, in keras' order (custom_layers.py:1053-1074)"""
        if self.w0 is None:
            return []
        return [self.w0.detach().cpu(), torch.full((1,), self._multiplier, dtype=torch.float32)]

    def get_config(self):
        return {"w0": None if self.w0 is None else self.w0.cpu().numpy(), "w1": np.array([self._multiplier], np.float32),
                "regularizer": self.regularizer, "activation": self._activation}


class Multiplier(_ScaleLayer):
    """custom_layers.py:1028-1091"""
    _per_channel = False


class ChannelwiseMultiplier(_ScaleLayer):
    """custom_layers.py:1097-1160"""
    _per_channel = True


class RandomOnOff:
    """custom_layers.py:107-127: keras Dropout(rate, noise_shape = [B, 1, ..., 1]) -- a whole sample is kept (scaled by 1 / (1 - rate))
    or zeroed; the identity unless training=True"""

    def __init__(self, rate: float = 0.5, seed: Optional[int] = None, **kwargs):
        if not 0.0 <= rate < 1.0:
            raise ValueError("rate must be in [0, 1)")
        self._rate = float(rate)
        self._rng = np.random.default_rng(seed)

    def __call__(self, inputs, training=None):
        x = _check_gpu(inputs)
        if not training or self._rate == 0.0:
            return x
        B, C = int(x.shape[0]), int(x.shape[-1])
        keep = (self._rng.uniform(size=B) >= self._rate).astype(np.float32) / np.float32(1.0 - self._rate)
        s = torch.from_numpy(keep).to(x.device)
        out = torch.empty_like(x)
        N.check(N.lib().bf_op_scale_add(None, N.ptr(x), None, N.ptr(s), N.ptr(out), B, x.numel() // (B * C), C, N.stream_ptr(x)), None,
                "bf_op_scale_add")
        return out
