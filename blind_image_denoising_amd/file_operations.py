"""`load_image` (bfcnn/file_operations.py:101-159): read an image file, decode it, optionally fit it into `image_size` the way
`tf.image.resize_with_pad` does, add a batch axis, normalise.  Host-side I/O in front of the engine (decoding through Pillow; nothing
here is on the hot path, and nothing here touches the GPU): returns a NumPy array where the reference returns a tf.Tensor."""
import glob
import itertools
import os
from pathlib import Path
from typing import Any, Generator, List, Optional, Tuple, Union

import numpy as np

from .custom_logger import logger

SUPPORTED_IMAGE_LIST_FORMATS = (".bmp", ".gif", ".jpeg", ".jpg", ".png")            # file_operations.py:18


def merge_iterators(*iterators):
    """file_operations.py:23-35: round-robin over the iterators until all are exhausted"""
    empty = {}
    for values in itertools.zip_longest(*iterators, fillvalue=empty):
        for value in values:
            if value is not empty:
                yield value


def index_directory_gen(directory: str, formats: Tuple = SUPPORTED_IMAGE_LIST_FORMATS) -> Generator[str, None, None]:
    """file_operations.py:87-96: every image file below `directory`"""
    for filename in glob.iglob(os.path.join(directory, "**/*"), recursive=True):
        if filename.lower().endswith(formats):
            yield filename


def image_filenames_generator(directory: Union[str, List[str]], verbose: bool = True):
    """file_operations.py:40-83: a function returning a generator of the image filenames of one or several directories (interleaved)"""
    if isinstance(directory, str):
        directory = [directory]
    if not isinstance(directory, list):
        raise ValueError(f"don't know what to do with [{directory}]")

    def gen_fn():
        return merge_iterators(*[index_directory_gen(directory=d) for d in directory])
    if verbose:
        total = 0
        for d in directory:
            n = sum(1 for _ in index_directory_gen(directory=d))
            total += n
            logger.info(f"directory [{d}]: [{n}] samples")
        logger.info(f"total number of samples: [{total}]")
    return gen_fn


def random_crops(image: np.ndarray, no_crops_per_image: int = 16, crop_size: Tuple[int, int] = (64, 64), rng=None,
                 extrapolation_value: float = 0.0) -> np.ndarray:
    """utilities.random_crops (utilities.py:467-561) of ONE image [H,W,C] (the loader hands it one image at a time): boxes of
    crop_size / image_size of the image at uniformly random positions, sampled with tf.image.crop_and_resize's bilinear rule
    (output row i reads y1 (H-1) + i (y2 - y1)(H-1) / (crop_h - 1): the crop is resampled, not sliced), cast back to the image's dtype"""
    rng = rng or np.random.default_rng()
    H, W, C = image.shape
    ch, cw = int(crop_size[0]), int(crop_size[1])
    if H <= 0 or W <= 0:
        return np.zeros((no_crops_per_image, ch, cw, C), image.dtype)
    ry, rx = ch / float(H), cw / float(W)
    out = np.empty((no_crops_per_image, ch, cw, C), np.float32)
    img = image.astype(np.float32)
    for k in range(no_crops_per_image):
        y1 = max(rng.uniform(0.0, max(1.0 - ry, 0.0)), 0.0)
        x1 = max(rng.uniform(0.0, max(1.0 - rx, 0.0)), 0.0)
        y2, x2 = min(y1 + ry, 1.0), min(x1 + rx, 1.0)
        ys = y1 * (H - 1) + np.arange(ch) * ((y2 - y1) * (H - 1) / (ch - 1) if ch > 1 else 0.0) if ch > 1 else np.array([0.5 * (y1 + y2) * (H - 1)])
        xs = x1 * (W - 1) + np.arange(cw) * ((x2 - x1) * (W - 1) / (cw - 1) if cw > 1 else 0.0) if cw > 1 else np.array([0.5 * (x1 + x2) * (W - 1)])
        y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
        fy, fx = (ys - y0).astype(np.float32), (xs - x0).astype(np.float32)
        oky, okx = (ys >= 0) & (ys <= H - 1), (xs >= 0) & (xs <= W - 1)
        y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
        x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
        rows = img[y0c] * (1.0 - fy)[:, None, None] + img[y1c] * fy[:, None, None]
        crop = rows[:, x0c] * (1.0 - fx)[None, :, None] + rows[:, x1c] * fx[None, :, None]
        crop[~oky] = extrapolation_value
        crop[:, ~okx] = extrapolation_value
        out[k] = crop
    return out.astype(image.dtype)


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
