"""`load_image` (bfcnn/file_operations.py:101-159) as the reference's tests/bfcnn/test_file_operations.py exercises it on lena.jpg (the
KITTI / Megadepth files of that test are not in this repository's fixtures), plus the resize_with_pad geometry against torch."""
import pathlib

import numpy as np
import pytest

import blind_image_denoising_amd as bf
from blind_image_denoising_amd.file_operations import resize_with_pad

LENA_IMAGE_PATH = pathlib.Path(__file__).parent / "golden" / "lena.jpg"


@pytest.mark.parametrize("num_channels", [1, 3])
@pytest.mark.parametrize("target_size", [(32, 32), (64, 64), (128, 128), (256, 256), (512, 512)])
def test_grayscale_load_image(num_channels, target_size):
    x = bf.load_image(path=LENA_IMAGE_PATH, num_channels=num_channels, image_size=target_size, expand_dims=True, normalize=True)
    assert x.shape[0] == 1
    assert x.shape[1:3] == target_size
    assert x.shape[3] == num_channels
    assert x.dtype == np.float32 and -0.5 <= x.min() < x.max() <= 0.5


def test_load_image_defaults_and_padding_geometry():
    import torch
    import torch.nn.functional as F
    raw = bf.load_image(str(LENA_IMAGE_PATH))
    assert raw.shape == (512, 512, 3) and raw.dtype == np.uint8
    assert np.array_equal(bf.load_image(LENA_IMAGE_PATH, image_size=(512, 512)), raw)          # same size: the identity
    # a wide target: the image keeps its aspect ratio, sits in the middle, the bands left and right are zero
    wide = bf.load_image(LENA_IMAGE_PATH, image_size=(64, 200), dtype=np.float32)
    assert wide.shape == (64, 200, 3) and not wide[:, :68].any() and not wide[:, 132:].any() and wide[:, 68:132].min() > 0
    # the resampling itself: tf.image.resize bilinear without antialiasing = half-pixel centres, as torch's align_corners=False
    a = np.random.default_rng(0).uniform(0, 255, (37, 53, 3)).astype(np.float32)
    got = resize_with_pad(a, 20, 71)
    rw = int(np.floor(53 / (37 / 20.0)))
    ref = F.interpolate(torch.from_numpy(a.astype(np.float64)).permute(2, 0, 1)[None], size=(20, rw), mode="bilinear", align_corners=False)[0]
    pw = int(np.floor((71 - 53 / (37 / 20.0)) / 2))
    assert np.abs(got[:, pw:pw + rw] - ref.permute(1, 2, 0).numpy()).max() < 1e-3 and not got[:, :pw].any() and not got[:, pw + rw:].any()
    with pytest.raises(NotImplementedError):
        bf.load_image(LENA_IMAGE_PATH, interpolation="nearest")
