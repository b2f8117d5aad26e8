"""Gaussian / Laplacian pyramids and their inverses (bfcnn/pyramid.py:238-532) on the HBM-bound
resampling kernels of csrc/pyramid.hip.  Tensors are float32 NHWC on the GPU."""
from enum import Enum
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from . import _native as N
from .constants import TYPE_STR
from .custom_logger import logger

DEFAULT_KERNEL_SIZE = (5, 5)      # bfcnn/pyramid.py:20


class PyramidType(Enum):
    """bfcnn/pyramid.py:214-233."""
    NONE = 1
    GAUSSIAN = 2
    LAPLACIAN = 3

    @staticmethod
    def from_string(type_str: str) -> "PyramidType":
        if type_str is None:
            raise ValueError("type_str must not be null")
        if not isinstance(type_str, str):
            raise ValueError("type_str must be string")
        if len(type_str.strip()) <= 0:
            raise ValueError("stripped type_str must not be empty")
        return PyramidType[type_str.strip().upper()]

    def to_string(self) -> str:
        return self.name


def _dev(x):
    was_numpy = isinstance(x, np.ndarray)
    if was_numpy:
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    if x.dim() != 4:
        raise ValueError("expected a rank-4 NHWC tensor")
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("pyramid kernels need the GPU: there is no CPU execution path")
        x = x.cuda()
    return x.to(torch.float32).contiguous(), was_numpy


def _same_out(n: int) -> int:
    return (n + 1) // 2


def avg_pool_s2_same(x: torch.Tensor, kernel_size=DEFAULT_KERNEL_SIZE) -> torch.Tensor:
    """keras AveragePooling2D(pool_size=kernel_size, strides=(2,2), padding="same") (pyramid.py:266-270)."""
    B, H, W, C = x.shape
    out = torch.empty((B, _same_out(H), _same_out(W), C), dtype=torch.float32, device=x.device)
    N.check(N.lib().bf_avgpool_s2_same(N.ptr(x), N.ptr(out), B, H, W, C, int(kernel_size[0]), int(kernel_size[1]),
                                       N.stream_ptr(x)), None, "bf_avgpool_s2_same")
    return out


def upsample_2x(x: torch.Tensor, other: torch.Tensor = None, bilinear: bool = True, alpha: float = 1.0,
                beta: float = 1.0) -> torch.Tensor:
    """alpha * UpSampling2D(2, bilinear|nearest)(x) + beta * other (pyramid.py:319-325, 380-385, 434-437)."""
    B, H, W, C = x.shape
    out = torch.empty((B, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    if other is not None and tuple(other.shape) != tuple(out.shape):
        raise ValueError(f"shape mismatch: up({tuple(x.shape)}) vs {tuple(other.shape)}")
    N.check(N.lib().bf_upsample2x(N.ptr(x), N.ptr(other), N.ptr(out), B, H, W, C, 1 if bilinear else 0,
                                  float(alpha), float(beta), N.stream_ptr(x)), None, "bf_upsample2x")
    return out


def laplacian_split(x: torch.Tensor, kernel_size=DEFAULT_KERNEL_SIZE):
    """one Laplacian level (pyramid.py:374-385): (down, x - up(down)) with down = avg_pool_s2_same(x).  One fused kernel where the
    shape allows it (x read once: bf_laplacian_split), else the two operators; the results are bitwise the same either way."""
    B, H, W, C = x.shape
    if H % 2 == 0 and W % 2 == 0:
        down = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        lap = torch.empty_like(x)
        rc = N.lib().bf_laplacian_split(N.ptr(x), N.ptr(down), N.ptr(lap), B, H, W, C, int(kernel_size[0]), int(kernel_size[1]),
                                        N.stream_ptr(x))
        if rc == N.BF_OK:
            return down, lap
        if rc != N.BF_EUNSUPPORTED:
            N.check(rc, None, "bf_laplacian_split")
    down = avg_pool_s2_same(x, kernel_size)
    return down, upsample_2x(down, x, True, -1.0, 1.0)               # x - up(down)


def strided_slice_2(x: torch.Tensor) -> torch.Tensor:
    """x[:, ::2, ::2, :] (bfcnn/downsampling.py:61)."""
    B, H, W, C = x.shape
    out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
    N.check(N.lib().bf_strided_slice2(N.ptr(x), N.ptr(out), B, H, W, C, N.stream_ptr(x)), None, "bf_strided_slice2")
    return out


def avg_pool2_valid(x: torch.Tensor, clip_values: bool = False, round_values: bool = False) -> torch.Tensor:
    """tf.nn.avg_pool2d(2x2, stride 2, VALID) [+clip][+round] (bfcnn/utilities.py:655-663)."""
    B, H, W, C = x.shape
    out = torch.empty((B, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
    N.check(N.lib().bf_avgpool2_valid(N.ptr(x), N.ptr(out), B, H, W, C, int(clip_values), int(round_values),
                                      N.stream_ptr(x)), None, "bf_avgpool2_valid")
    return out


def multiscales_generator_fn(shape=None, no_scales: int = 1, clip_values: bool = False, round_values: bool = False, **kwargs):
    """bfcnn/utilities.py:625-685: ground-truth pyramid for deep supervision."""
    def multiscale_fn(n):
        n, was_numpy = _dev(n)
        scales = [n]
        for _ in range(no_scales):
            n = avg_pool2_valid(n, clip_values, round_values)
            scales.append(n)
        return [s.cpu().numpy() for s in scales] if was_numpy else scales
    return multiscale_fn


class _PyramidModel:
    def __init__(self, name, levels, fn):
        self.name, self.levels, self._fn = name, levels, fn

    def __call__(self, x, training=False):
        return self._fn(x)

    def predict(self, x):
        return self._fn(x)


def build_gaussian_pyramid_model(input_dims, levels: int, kernel_size=DEFAULT_KERNEL_SIZE, trainable=False,
                                 name="gaussian_pyramid"):
    """bfcnn/pyramid.py:238-283."""
    def fn(x):
        x, was_numpy = _dev(x)
        out = [x]
        for _ in range(1, levels):
            x = avg_pool_s2_same(x, kernel_size)
            out.append(x)
        return [o.cpu().numpy() for o in out] if was_numpy else out
    return _PyramidModel(name, levels, fn)


def build_inverse_gaussian_pyramid_model(input_dims, levels: int, trainable=False, name="inverse_gaussian_pyramid"):
    """bfcnn/pyramid.py:289-341."""
    def fn(xs):
        if not isinstance(xs, (list, tuple)):
            xs = [xs]
        if len(xs) != levels:
            raise ValueError(f"expected {levels} levels, got {len(xs)}")
        conv = [_dev(x) for x in xs]
        was_numpy = conv[0][1]
        output = previous = None
        for level_x, _ in reversed(conv):
            if output is None:
                output = previous = level_x
            else:
                diff = upsample_2x(previous, level_x, True, -1.0, 1.0)      # level_x - up(previous)
                output = upsample_2x(output, diff, True, 1.0, 1.0)         # up(output) + diff
                previous = level_x
        return output.cpu().numpy() if was_numpy else output
    return _PyramidModel(name, levels, fn)


def build_laplacian_pyramid_model(input_dims, levels: int, kernel_size=DEFAULT_KERNEL_SIZE, trainable=False,
                                  name="laplacian_pyramid"):
    """bfcnn/pyramid.py:347-398."""
    logger.info(f"building laplacian pyramid model with: {levels} levels")

    def fn(x):
        x, was_numpy = _dev(x)
        out = []
        for _ in range(levels - 1):
            x, lap = laplacian_split(x, kernel_size)
            out.append(lap)
        out.append(x)
        return [o.cpu().numpy() for o in out] if was_numpy else out
    return _PyramidModel(name, levels, fn)


def build_inverse_laplacian_pyramid_model(input_dims, levels: int, trainable=False, name="inverse_laplacian_pyramid"):
    """bfcnn/pyramid.py:404-445."""
    logger.info(f"building inverse laplacian pyramid model with: {levels} levels")

    def fn(xs):
        if not isinstance(xs, (list, tuple)):
            xs = [xs]
        if len(xs) != levels:
            raise ValueError(f"expected {levels} levels, got {len(xs)}")
        conv = [_dev(x) for x in xs]
        was_numpy = conv[0][1]
        output = None
        for level_x, _ in reversed(conv):
            output = level_x if output is None else upsample_2x(output, level_x, True, 1.0, 1.0)
        return output.cpu().numpy() if was_numpy else output
    return _PyramidModel(name, levels, fn)


def build_pyramid_model(input_dims: Union[Tuple, List], config: Dict):
    """bfcnn/pyramid.py:451-491 (type NONE builds a gaussian pyramid, as the reference does)."""
    if config is None:
        no_levels, kernel_size, pyramid_type = 1, DEFAULT_KERNEL_SIZE, PyramidType.from_string("NONE")
    else:
        no_levels = config.get("levels", 1)
        kernel_size = tuple(config.get("kernel_size", DEFAULT_KERNEL_SIZE))
        pyramid_type = PyramidType.from_string(config.get(TYPE_STR, "NONE"))
    if pyramid_type in (PyramidType.GAUSSIAN, PyramidType.NONE):
        return build_gaussian_pyramid_model(input_dims=input_dims, levels=no_levels, kernel_size=kernel_size)
    if pyramid_type == PyramidType.LAPLACIAN:
        return build_laplacian_pyramid_model(input_dims=input_dims, levels=no_levels, kernel_size=kernel_size)
    raise ValueError("don't know how to build pyramid type [{0}]".format(pyramid_type))


def build_inverse_pyramid_model(input_dims: Union[Tuple, List], config: Dict):
    """bfcnn/pyramid.py:497-532."""
    if config is None:
        no_levels, pyramid_type = 1, PyramidType.from_string("NONE")
    else:
        no_levels = config.get("levels", 1)
        pyramid_type = PyramidType.from_string(config.get(TYPE_STR, "NONE"))
    if pyramid_type in (PyramidType.GAUSSIAN, PyramidType.NONE):
        return build_inverse_gaussian_pyramid_model(input_dims=input_dims, levels=no_levels)
    if pyramid_type == PyramidType.LAPLACIAN:
        return build_inverse_laplacian_pyramid_model(input_dims=input_dims, levels=no_levels)
    raise ValueError("don't know how to build pyramid type [{0}]".format(pyramid_type))
