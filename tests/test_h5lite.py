"""The minimal HDF5 reader (blind_image_denoising_amd/h5lite.py) against the one Keras weight file the reference ships.
Runs where /root/reference exists (this container); skipped on the GPU box, where the reference is absent."""
import os

import numpy as np
import pytest

from blind_image_denoising_amd import h5lite

ARCHIVE = "/root/reference/bfcnn/pretrained/unet_laplacian_v5.6/model_hydra.keras"
pytestmark = pytest.mark.skipif(not os.path.exists(ARCHIVE), reason="reference archive not present")


def test_reads_every_tensor_of_the_trained_v56_archive():
    d = h5lite.read_keras_archive(ARCHIVE)
    assert len(d) == 95 and sum(v.size for v in d.values()) == 334976       # == count_params of the v5 graph family
    assert all(v.dtype == np.float32 and np.isfinite(v).all() for v in d.values())
    base = [v for k, v in d.items() if k.endswith("functional/_layer_checkpoint_dependencies/conv2d/vars/0")][0]
    assert base.shape == (5, 5, 3, 32) and 0.05 < np.abs(base).mean() < 0.3
    dw = [v for k, v in d.items() if k.endswith("conv_next_block/conv_1/vars/0")][0]
    assert dw.shape == (5, 5, 32, 1)
    heads = sorted(v.shape for k, v in d.items() if "functional_3/" in k or "functional_5/" in k or "functional_7/" in k)
    assert heads == [(1, 1, 32, 3)] * 3 + [(1, 1, 32, 32), (1, 1, 64, 32), (1, 1, 128, 32)]
    # the trained graph is older than the snapshot builder: its attention blocks carry a second
    # LayerNorm (ln_1, 32 channels) that backbone_unet_laplacian.py no longer builds
    assert sum(k.endswith("ln_1/vars/0") for k in d) == 3


def test_rejects_what_is_not_hdf5():
    with pytest.raises(ValueError):
        h5lite.H5File(b"not an hdf5 file at all")
