"""The gradient oracle of unet_laplacian training (oracle/unet_torch.py, torch-CPU fp64 autograd) against the NumPy
restatement it mirrors (oracle/unet_oracle.py): same outputs, same loss terms; and its own consistency (finite differences)."""
import numpy as np
import pytest

from oracle import bfcnn_oracle as O
from oracle import unet_oracle as U
from oracle import unet_torch as T


def _setup(depth=3, width=2, filters=32, seed=3, B=2, S=64):
    cfg = U.canonical_config(depth=depth, width=width, filters=filters)
    spec = U.UnetLaplacianSpec.from_config(cfg["model"])
    params = U.init_params(spec, seed=seed)
    clean, noisy = O.synthetic_batch(B, S, S, seed=seed + 1)
    return spec, params, clean.astype(np.float64), noisy.astype(np.float64)


def test_forward_equals_the_numpy_restatement():
    spec, params, clean, noisy = _setup()
    ls = O.LossSpec(hinge=3.5, cutoff=255.0, mae_multiplier=1.0, ssim_multiplier=1.0, mse_multiplier=0.5, regularization=0.01)
    total, ml, dls, preds, grads = T.train_step(spec, ls, params, clean, noisy, [1.0, 0.5, 0.25])
    ref = U.hydra_forward(spec, params, noisy)
    for p, r in zip(preds, ref):
        assert p.shape == r.shape and np.abs(p - r).max() < 1e-9
    gts = T.ground_truth_pyramid(clean, spec.depth)
    for i in range(spec.depth):
        want = O.denoiser_loss(ls, gts[i], ref[i])
        for k in ("total_loss", "mae_loss", "mse_loss", "ssim_loss"):
            assert abs(dls[i][k] - want[k]) < 1e-9 * max(1.0, abs(want[k])), (i, k)
    assert np.isfinite(grads).all() and grads.shape == params.shape


def test_gradient_against_finite_differences():
    spec, params, clean, noisy = _setup(depth=2, width=1, filters=16, S=32)
    ls = O.LossSpec(hinge=0.0, cutoff=255.0, mae_multiplier=1.0, ssim_multiplier=1.0, mse_multiplier=0.5, regularization=0.01)
    params = params.astype(np.float64)
    dw = [1.0, 0.7]
    total, _, _, _, grads = T.train_step(spec, ls, params, clean, noisy, dw)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(params.size)
    d /= np.linalg.norm(d)
    eps = 1e-5
    lp = T.train_step(spec, ls, params + eps * d, clean, noisy, dw)[0]
    lm = T.train_step(spec, ls, params - eps * d, clean, noisy, dw)[0]
    assert abs((lp - lm) / (2 * eps) - grads @ d) <= 2e-4 * max(1.0, abs(grads @ d))


def test_soft_orthonormal_matches_its_definition():
    """regularizers.py:283-338: lambda * ||W^T W - I||_F^2 + l2 * sum (W^T W)^2 on the [cout, cin] reshape."""
    import torch
    rng = np.random.default_rng(1)
    w = rng.standard_normal((1, 1, 8, 24))
    wt = w.transpose(3, 0, 1, 2).reshape(24, -1)
    G = wt @ wt.T
    want = 0.01 * ((G - np.eye(24)) ** 2).sum() + 1e-4 * (G * G).sum()
    assert abs(float(T.soft_orthonormal(torch.from_numpy(w))) - want) < 1e-10 * want


@pytest.mark.parametrize("options", [{"decoder_kernel_size": 5, "downsample_type": "conv2d", "gaussian_kernel_size": 2,
                                      "upsample_type": "upsample_nearest_conv2d", "use_laplacian_averaging": True},
                                     {"use_attention_gates": True, "upsample_type": "upsample_nearest_conv2d"},
                                     {"use_mix_project": True, "downsample_type": "maxpool"},
                                     {"upsample_type": "bilinear", "filters_level_multiplier": 1.0},
                                     {"upsample_type": "nn", "filters_level_multiplier": 1.0, "activation": "relu"},
                                     {"use_concat": True, "use_mix_project": True},
                                     {"use_concat": True, "use_mix_project": False, "decoder_kernel_size": 3}],
                         ids=["v6", "v3", "mix-maxpool", "plain-bilinear", "plain-nearest", "concat-mix (builder defaults)", "concat-no-mix"])
def test_forward_of_the_other_shipped_graphs_equals_the_numpy_restatement(options):
    """the options of configs/unet_laplacian_v6.json (2 x 2 averaging, conv2d down-sampling, nearest + 3 x 3 up-sampling, 5 x 5
    decoder depthwise) and v3 / v4 (AdditiveAttentionGate) in the torch gradient oracle against unet_oracle.py"""
    import torch
    cfg = U.canonical_config(depth=3, width=1, filters=32)
    cfg["model"]["backbone"].update(options)
    spec = U.UnetLaplacianSpec.from_config(cfg["model"])
    params = U.init_params(spec, seed=2)
    x = np.random.default_rng(0).uniform(0, 255, (2, 32, 32, 3))
    ref = U.hydra_forward(spec, params, x)
    got = T.hydra(spec, T.views(spec, torch.tensor(params.astype(np.float64))), torch.from_numpy(x))
    for g, r in zip(got, ref):
        assert np.abs(g.numpy() - r).max() < 1e-9



def test_forward_of_the_trained_archive_graph_equals_the_numpy_restatement():
    """the graph revision of the reference's trained unet_laplacian_v5.6 archive (GELU in the convnext MLP and on query / key /
    value, row-wise full-resolution attention with its second LayerNorm, no level activation, linear up-sampling, the output
    LayerNorms in front of the heads) on the archive's own weights: the torch gradient oracle against unet_oracle.py, which
    tests/test_unet_pretrained.py pins on the archive's operator list and the reference's acceptance test."""
    import sys, os, torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import unet_v56 as V
    z, cfg = V.load()
    spec = U.UnetLaplacianSpec.from_config(cfg)
    T.check_trainable_graph(spec)
    params = np.asarray(z["params"], np.float64)
    x = V.corrupt(z["kitti"][:1, 32:96, 16:112], 20.0, seed=3).astype(np.float64)
    ref = U.hydra_forward(spec, params, x)
    got = T.hydra(spec, T.views(spec, torch.tensor(params)), torch.from_numpy(x))
    for g, r in zip(got, ref):
        assert np.abs(g.numpy() - r).max() < 1e-8
    ls = O.LossSpec(hinge=0.0, cutoff=255.0, mae_multiplier=1.0, ssim_multiplier=1.0, mse_multiplier=0.5, regularization=0.01)
    clean = z["kitti"][:1, 32:96, 16:112].astype(np.float64)
    total, _, _, _, grads = T.train_step(spec, ls, params, clean, x, [1.0, 0.5, 0.25])
    d = np.random.default_rng(0).standard_normal(params.size)
    d /= np.linalg.norm(d)
    eps = 1e-5
    lp = T.train_step(spec, ls, params + eps * d, clean, x, [1.0, 0.5, 0.25])[0]
    lm = T.train_step(spec, ls, params - eps * d, clean, x, [1.0, 0.5, 0.25])[0]
    assert abs((lp - lm) / (2 * eps) - grads @ d) <= 2e-4 * max(1.0, abs(grads @ d))
