"""`load_image` (bfcnn/file_operations.py:101-159): read an image file, decode it, optionally fit it into `image_size` the way
`tf.image.resize_with_pad` does, add a batch axis, normalise.  Host-side I/O in front of the engine (decoding through Pillow; nothing
here is on the hot path, and nothing here touches the GPU): returns a NumPy array where the reference returns a tf.Tensor."""
from pathlib import Path
from typing import Any, Optional, Tuple

import numpy as np


def _resize_bilinear(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """tf.image.resize(method=BILINEAR, antialias=False) of [H,W,C]: half-pixel centres, edge-clamped taps"""
    H, W, _ = img.shape

    def taps(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(src)
        f = src - i0
        return np.clip(i0, 0, n_in - 1).astype(np.int64), np.clip(i0 + 1, 0, n_in - 1).astype(np.int64), f
    y0, y1, fy = taps(H, oh)
    x0, x1, fx = taps(W, ow)
    a = img.astype(np.float64)
    rows = a[y0] * (1.0 - fy)[:, None, None] + a[y1] * fy[:, None, None]
    return (rows[:, x0] * (1.0 - fx)[None, :, None] + rows[:, x1] * fx[None, :, None]).astype(np.float32)


def resize_with_pad(img: np.ndarray, target_height: int, target_width: int) -> np.ndarray:
    """tf.image.resize_with_pad(image, th, tw, method=BILINEAR, antialias=False): the aspect ratio is kept (sizes floored), the rest
    is zero padding, centred (the odd pixel goes to the bottom / right); float32 result as tf.image.resize returns it"""
    H, W, C = img.shape
    ratio = max(W / float(target_width), H / float(target_height))
    rh_f, rw_f = H / ratio, W / ratio
    rh, rw = max(int(np.floor(rh_f)), 1), max(int(np.floor(rw_f)), 1)
    ph, pw = max(0, int(np.floor((target_height - rh_f) / 2.0))), max(0, int(np.floor((target_width - rw_f) / 2.0)))
    out = np.zeros((target_height, target_width, C), np.float32)
    out[ph:ph + rh, pw:pw + rw] = _resize_bilinear(img, rh, rw)
    return out


def load_image(path: Any, image_size: Optional[Tuple[int, int]] = None, num_channels: int = 3, dtype=np.uint8, interpolation: str = "bilinear",
               expand_dims: bool = False, normalize: bool = False) -> np.ndarray:
    from PIL import Image
    if isinstance(path, Path):
        path = str(path)
    if str(interpolation).lower() not in ("bilinear", "resizemethod.bilinear"):
        raise NotImplementedError(f"load_image: interpolation [{interpolation}] (bilinear, the default, is built)")
    if num_channels not in (1, 3, 4):
        raise ValueError("num_channels must be 1, 3 or 4")                          # tf.image.decode_image: 0, 1, 3 or 4
    with Image.open(path) as im:
        img = np.asarray(im.convert({1: "L", 3: "RGB", 4: "RGBA"}[num_channels]), dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    if image_size is not None:
        img = resize_with_pad(img, int(image_size[0]), int(image_size[1]))
    if expand_dims:
        img = img[None]
    if normalize:                                                                  # layer_normalize(img, 0, 255) (utilities.py:449-461)
        return (np.clip(img.astype(np.float32), 0.0, 255.0) / np.float32(255.0) - np.float32(0.5)).astype(np.float32)
    return img.astype(dtype)                                                       # tf.cast: float -> integer truncates
