"""blind_image_denoising_amd.regularizers on the GPU: the shape tests of the reference's tests/bfcnn/test_regularizer.py (same
parametrisation) plus the VALUES against the formulas of bfcnn/regularizers.py:159-338 in NumPy fp64."""
import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf

pytestmark = pytest.mark.gpu

SHAPES_4D = [(5, 5, 3, 2), (1, 1, 1, 4), (2, 2, 2, 8), (4, 4, 4, 16), (8, 8, 8, 32)]
SHAPES_2D = [(3, 2), (1, 4), (2, 8), (4, 16), (8, 32)]


def _rand(shape, seed=0):
    x = np.random.default_rng(seed).uniform(-1, 1, shape).astype(np.float32)
    return x, torch.from_numpy(x).cuda()


def _wt(x):
    return x.T if x.ndim == 2 else x.transpose(3, 0, 1, 2).reshape(x.shape[3], -1)


@pytest.mark.parametrize("shape", SHAPES_4D)
def test_4d_reshape_to_2d(shape):
    x, xd = _rand(shape)
    x_reshaped = bf.regularizers.reshape_to_2d(xd)
    assert x_reshaped.shape[0] == shape[3]
    assert x_reshaped.shape[1] == (shape[0] * shape[1] * shape[2])
    assert np.array_equal(x_reshaped.cpu().numpy(), _wt(x))


@pytest.mark.parametrize("shape", SHAPES_2D)
def test_2d_reshape_to_2d(shape):
    x, xd = _rand(shape)
    x_reshaped = bf.regularizers.reshape_to_2d(xd)
    assert x_reshaped.shape[0] == shape[1]
    assert x_reshaped.shape[1] == shape[0]
    assert np.array_equal(x_reshaped.cpu().numpy(), x.T)


@pytest.mark.parametrize("shape", SHAPES_4D + SHAPES_2D)
def test_wt_x_w(shape):
    x, xd = _rand(shape)
    wt_w = bf.regularizers.wt_x_w(xd)
    assert wt_w.shape[0] == shape[-1]
    assert wt_w.shape[1] == shape[-1]
    ref = _wt(x.astype(np.float64)) @ _wt(x.astype(np.float64)).T
    assert np.abs(wt_w.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", SHAPES_4D + SHAPES_2D)
def test_create_soft_orthogonal_constraint(shape):
    x, xd = _rand(shape, seed=1)
    G = _wt(x.astype(np.float64)) @ _wt(x.astype(np.float64)).T
    n = G.shape[0]
    result = bf.regularizers.SoftOrthonormalConstraintRegularizer(1.0)(xd)
    assert result >= 0
    want = ((G - np.eye(n)) ** 2).sum() + 0.001 * np.abs(G).sum()                      # defaults: lambda 1, l1 0.001, l2 0
    assert abs(result.item() - want) <= 2e-5 * max(1.0, want)
    Gm = G * (1.0 - np.eye(n))
    for lam, l1, l2 in ((1.0, 0.01, 0.0), (0.5, 0.0, 0.25)):
        got = bf.regularizers.SoftOrthogonalConstraintRegularizer(lam, l1, l2)(xd)
        want = lam * (Gm ** 2).sum() + l1 * np.abs(Gm).sum() + l2 * (Gm ** 2).sum()
        assert got >= 0 and abs(got.item() - want) <= 2e-5 * max(1.0, want)


@pytest.mark.parametrize("config", [["l1"], ["l1l2"], ["l1", "l2"], ["soft_orthogonal"], ["soft_orthogonal", "l1"],
                                    [{"type": "soft_orthonormal", "config": {"lambda_coefficient": 0.01, "l1_coefficient": 0.0, "l2_coefficient": 1e-4}}],
                                    "l2"])
def test_builder(config):
    prune_fns = bf.regularizers.builder(config=config)
    assert prune_fns is not None
    x, xd = _rand((3, 3, 4, 8), seed=2)
    x64 = x.astype(np.float64)
    G = _wt(x64) @ _wt(x64).T
    Gm = G * (1.0 - np.eye(8))
    terms = {"l1": 0.01 * np.abs(x64).sum(), "l2": 0.01 * (x64 ** 2).sum(), "l1l2": 0.0,
             "soft_orthogonal": (Gm ** 2).sum() + 0.01 * np.abs(Gm).sum()}
    if isinstance(config, list) and isinstance(config[0], dict):
        want = 0.01 * ((G - np.eye(8)) ** 2).sum() + 1e-4 * (G ** 2).sum()             # constants.py:19-21: the ConvNext layers' setting
    else:
        want = sum(terms[c] for c in (config if isinstance(config, list) else [config]))
    assert abs(prune_fns(xd).item() - want) <= 2e-5 * max(1.0, want)
    with pytest.raises(RuntimeError, match="MI355X"):
        prune_fns(torch.from_numpy(x))
    with pytest.raises(KeyError):
        bf.regularizers.builder("l3")
