"""
The corruption half of the reference's input pipeline on the device: `prepare_data_fn` of
bfcnn/dataset.py:126-239 (whole-batch flips, multiplicative and additive truncated-normal noise, rounding) as ONE
HIP kernel (`bf_noise_augment`), so that a training step is not fed by a host pipeline.  The tf.data part of
`dataset_builder` (file listing, decoding, cropping, shuffling, batching: dataset.py:241-297) is host-side I/O in front of it
(`_DirectoryDataset`, file_operations.py); `dataset_builder` also takes an iterable of clean batches directly and yields the
(input_batch, noisy_batch) pairs `train_loop` consumes.

Random numbers: the per-BATCH choices of the reference (two flips, whether each noise is applied, the two standard
deviations ~ U[min, max]) come from a seeded host generator; the per-element noise is Philox4x32-10 indexed by the
element, keyed by a per-batch seed from the same generator (TensorFlow's own stream cannot be reproduced outside
TensorFlow; the distribution and the order of operations are the reference's).
"""
from collections import namedtuple
from typing import Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .custom_logger import logger


def noise_augment(input_batch: torch.Tensor, flip_left_right: bool = False, flip_up_down: bool = False,
                  mult_std: float = 0.0, add_std: float = 0.0, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """one bf_noise_augment call: (round(flip(x)), round(round(flip(x)) * tn(1, mult_std) + tn(0, add_std)))."""
    if input_batch.dim() != 4:
        raise ValueError(f"expected a [B,H,W,C] batch, got {tuple(input_batch.shape)}")
    if not input_batch.is_cuda:
        raise RuntimeError("noise_augment runs on the MI355X: the engine has no CPU execution path")
    x = input_batch.to(torch.float32).contiguous()
    B, H, W, C = x.shape
    clean, noisy = torch.empty_like(x), torch.empty_like(x)
    flip = (1 if flip_left_right else 0) | (2 if flip_up_down else 0)
    N.check(N.lib().bf_noise_augment(N.ptr(x), N.ptr(clean), N.ptr(noisy), B, H, W, C, flip, float(mult_std), float(add_std),
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, N.stream_ptr(x)), None, "bf_noise_augment")
    return clean, noisy


class PrepareData:
    """`prepare_data_fn` (bfcnn/dataset.py:126-239) built from the same `dataset` config section
    (dataset.py:92-121): random_left_right, random_up_down, additional_noise, multiplicative_noise."""

    def __init__(self, config: Dict, seed: Optional[int] = None):
        self.use_left_right = bool(config.get("random_left_right", False))
        self.use_up_down = bool(config.get("random_up_down", False))
        additional = list(config.get("additional_noise", []))
        multiplicative = list(config.get("multiplicative_noise", []))
        self.use_additive = len(additional) > 0                         # dataset.py:99-104
        self.use_multiplicative = len(multiplicative) > 0
        self.additive = (min(additional), max(additional)) if additional else (1.0, 1.0)
        self.multiplicative = (min(multiplicative), max(multiplicative)) if multiplicative else (1.0, 1.0)
        # random_blur, inpaint_drop_rate, random_rotate, quantization, use_jpeg_noise: dataset_builder reads them (dataset.py:84-105) and
        # prepare_data_fn (:123-239) never uses them -- the reference's own shipped configs set random_blur / random_rotate /
        # inpaint_drop_rate and train without any of the three.  Accepted and without effect here as well, so that those config files
        # run unchanged; the names that were set are kept for the log.
        self.ignored = [k for k in ("random_blur", "use_jpeg_noise") if config.get(k, False)] + \
            [k for k in ("random_rotate", "inpaint_drop_rate") if float(config.get(k, 0.0)) > 0.0] + \
            (["quantization"] if int(config.get("quantization", -1)) > 1 else [])
        if self.ignored:
            logger.info(f"dataset options without effect in prepare_data_fn (as in the reference): {self.ignored}")
        self.rng = np.random.default_rng(seed)

    def draw(self) -> Dict:
        """the per-batch scalars, in the order the reference draws them (dataset.py:141-142, 170-187)."""
        r = self.rng
        flip_lr, flip_ud = r.uniform() > 0.5, r.uniform() > 0.5
        opt_add, opt_mult = r.uniform() > 0.5, r.uniform() > 0.5
        add_std = r.uniform(self.additive[0], self.additive[1])
        mult_std = r.uniform(self.multiplicative[0], self.multiplicative[1])
        return {"flip_left_right": bool(flip_lr and self.use_left_right), "flip_up_down": bool(flip_ud and self.use_up_down),
                "add_std": float(add_std) if (opt_add and self.use_additive) else 0.0,
                "mult_std": float(mult_std) if (opt_mult and self.use_multiplicative) else 0.0,
                "seed": int(r.integers(0, 2 ** 63 - 1))}

    def __call__(self, input_batch: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return noise_augment(input_batch, **self.draw())


DatasetResults = namedtuple("DatasetResults", ["config", "batch_size", "input_shape", "training", "testing"])      # dataset.py:27-35


class _DirectoryDataset:
    """the tf.data pipeline of bfcnn/dataset.py:241-297 as a re-iterable: every iteration is one pass over the image files --
    shuffle the filenames (buffer 1024) -> load_image -> `no_crops_per_image` random crops of `input_shape` -> prepare_data_fn on
    that image's crops (ONE draw of the flip / noise scalars per image, on the device: bf_noise_augment) -> unbatch -> shuffle
    (buffer batch_size * 128) -> batches of `batch_size`, remainder dropped.  Decoding and cropping are host-side I/O; everything
    from the clean crops on lives on the GPU."""

    def __init__(self, config: Dict, directories, device=None, seed: Optional[int] = None):
        from .file_operations import image_filenames_generator
        self.config, self.device = config, device
        self.batch_size = int(config["batch_size"])
        self.input_shape = list(config["input_shape"])
        color_mode = config.get("color_mode", "rgb").strip().lower()
        if color_mode not in ("rgb", "rgba", "grayscale"):
            raise ValueError('`color_mode` must be one of {"rgb", "rgba", "grayscale"}. ' f"Received: color_mode={color_mode}")
        self.num_channels = {"rgb": 3, "rgba": 4, "grayscale": 1}[color_mode]
        self.no_crops = int(config.get("no_crops_per_image", 1))
        self.prepare = PrepareData(config, seed)
        self.rng = np.random.default_rng(seed)
        self.filenames_fn = image_filenames_generator(directory=directories)

    @staticmethod
    def _shuffled(it, buffer_size, rng):
        """tf.data shuffle(buffer_size): a buffer that is filled, then one random element out for every element in"""
        buf = []
        for item in it:
            buf.append(item)
            if len(buf) >= buffer_size:
                yield buf.pop(int(rng.integers(len(buf))))
        while buf:
            yield buf.pop(int(rng.integers(len(buf))))

    def _samples(self):
        from .file_operations import load_image, random_crops
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        for path in self._shuffled(self.filenames_fn(), 1024, self.rng):
            try:
                img = load_image(path=path, image_size=None, num_channels=self.num_channels, expand_dims=False, normalize=False)
            except Exception as e:                       # an unreadable file must not end an epoch
                logger.warning(f"skipping [{path}]: {e}")
                continue
            crops = random_crops(img, no_crops_per_image=self.no_crops, crop_size=(self.input_shape[0], self.input_shape[1]), rng=self.rng)
            clean, noisy = self.prepare(torch.from_numpy(crops).to(dev))
            for k in range(clean.shape[0]):
                yield clean[k], noisy[k]

    def __iter__(self):
        batch = []
        for sample in self._shuffled(self._samples(), self.batch_size * 128, self.rng):
            batch.append(sample)
            if len(batch) == self.batch_size:            # drop_remainder=True
                yield torch.stack([c for c, _ in batch]), torch.stack([n for _, n in batch])
                batch = []


def dataset_builder(config: Dict, clean_batches: Iterable = None, device=None, seed: Optional[int] = None):
    """bfcnn/dataset.py:40-305.  With `config["inputs"]` (image directories) and no `clean_batches`: the reference's pipeline, returned
    as `DatasetResults(config, batch_size, input_shape, training, testing=None)` whose `training` yields (input_image_batch,
    noisy_image_batch) device tensors, one pass over the files per iteration.  With `clean_batches` (any iterable of clean [B,H,W,C]
    batches in value range, numpy or torch): the corruption stage alone, as a generator of the same pairs."""
    logger.info(f"creating dataset_builder with configuration [{config}]")
    if clean_batches is None:
        inputs = config.get("inputs") if isinstance(config, dict) else None
        if inputs is None:
            raise ValueError("the dataset configuration names no `inputs` directories and no clean_batches were given")
        if isinstance(inputs, list):
            directories = [i.get("directory", None) for i in inputs]
        elif isinstance(inputs, dict):
            directories = [config.get("directory", None)]                     # dataset.py:68-69 (as written there)
        else:
            raise ValueError("dont know how to handle anything else than list and dict")
        directories = [d for d in directories if d]
        if not directories:
            raise ValueError("don't know how to handle non directory datasets")  # dataset.py:254-255
        ds = _DirectoryDataset(config, directories, device=device, seed=seed)
        return DatasetResults(config=config, batch_size=ds.batch_size, input_shape=ds.input_shape, training=ds, testing=None)
    prepare = PrepareData(config, seed)

    def gen():
        for batch in clean_batches:
            t = torch.as_tensor(np.asarray(batch) if not isinstance(batch, torch.Tensor) else batch)
            dev = device if device is not None else (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
            yield prepare(t.to(dev))
    return gen()
