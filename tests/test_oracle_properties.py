"""CPU tests of the oracle: golden fixtures (freeze the restatement), known-answer tests that
need no TensorFlow, and the reference's own pyramid round-trip identity on its own fixture."""
import pathlib

import numpy as np
import pytest

from oracle import bfcnn_oracle as O

G = pathlib.Path(__file__).resolve().parent / "golden"


def test_oracle_matches_golden_conv():
    z = np.load(G / "conv3x3_c16.npz")
    for s in range(5):
        y = O.conv2d_same(z[f"x{s}"].astype(np.float64), z[f"w{s}"].astype(np.float64))
        assert np.abs(y - z[f"y{s}"]).max() < 1e-13


def test_oracle_matches_golden_net():
    z = np.load(G / "net_2blocks.npz")
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=2)["model"])
    assert spec.param_count() == z["params"].size and spec.state_count() == z["state"].size
    y = O.hydra_forward(spec, z["params"], z["state"], z["noisy"].astype(np.float64))
    assert np.abs(y - z["hydra_f32"]).max() < 1e-10
    assert np.array_equal(O.denoiser_module_call(spec, z["params"], z["state"], z["noisy"]), z["out_u8"])
    assert np.array_equal(O.denoiser_module_call(spec, z["params"], z["state"], z["ragged"]), z["ragged_u8"])


def test_oracle_matches_golden_train_step():
    z = np.load(G / "train_step.npz")
    n = np.load(G / "net_2blocks.npz")
    cfg = O.canonical_config(no_layers=2)
    spec, ls = O.ResnetSpec.from_config(cfg["model"]), O.LossSpec.from_config(cfg["loss"])
    total, ml, dl, pred, grads, new_state = O.train_step_single_gpu(
        spec, ls, n["params"], n["state"], z["clean"].astype(np.float64), z["noisy"].astype(np.float64))
    assert abs(total - z["total"]) < 1e-12 and np.abs(grads - z["grads"]).max() < 1e-12
    assert np.abs(new_state - z["new_state"]).max() < 1e-12


def test_oracle_matches_golden_train_step_with_all_loss_terms():
    z, t, n = np.load(G / "train_step_full_loss.npz"), np.load(G / "train_step.npz"), np.load(G / "net_2blocks.npz")
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=2)["model"])
    ls = O.LossSpec.from_config({"hinge": 3.5, "cutoff": 255.0, "mae_multiplier": 1.0, "mse_multiplier": 0.5,
                                 "ssim_multiplier": 1.0, "regularization": 0.01})
    total, ml, dl, pred, grads, _ = O.train_step_single_gpu(spec, ls, n["params"], n["state"], t["clean"].astype(np.float64),
                                                             t["noisy"].astype(np.float64), depth_weight=0.8)
    assert abs(total - z["total"]) < 1e-12 and abs(dl[0]["ssim_loss"] - z["ssim"]) < 1e-14
    assert np.abs(grads - z["grads"]).max() < 1e-12


def test_param_counts_match_survey():
    """1x6 -> 28,784 trainable parameters, 1x18 -> 84,272."""
    for n, cnt in ((6, 28784), (18, 84272)):
        assert O.ResnetSpec.from_config(O.canonical_config(no_layers=n)["model"]).param_count() == cnt


def test_kat_mid_grey_maps_to_mid_grey():
    """f32 input 127.5 normalises to 0; with moving_mean = 0 the bias-free net outputs exactly 0,
    tanh(0) = 0, denormalised 127.5 everywhere (SURVEY.md 8c KAT iii)."""
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=3)["model"])
    params, state = O.init_params(spec, seed=1, nontrivial_bn=False)
    y = O.hydra_forward(spec, params, state, np.full((1, 16, 16, 3), 127.5))
    assert np.array_equal(y, np.full_like(y, 127.5))
    assert O.round_half_even(np.array([127.5]))[0] == 128.0


def test_kat_backbone_is_degree_one_homogeneous():
    """bias-free property (README.md:31-46): features(alpha*x) = alpha*features(x) for alpha > 0 when
    moving_mean = 0 (only the head's tanh and non-zero BN means break it)."""
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=2)["model"])
    params, state = O.init_params(spec, seed=2, nontrivial_bn=False)
    rng = np.random.default_rng(0)
    x = 127.5 + rng.uniform(-20, 20, (1, 12, 12, 3))
    _, c1 = O.hydra_forward(spec, params, state, x, want_cache=True)
    _, c2 = O.hydra_forward(spec, params, state, 127.5 + 3.0 * (x - 127.5), want_cache=True)
    assert np.abs(c2["feat"] - 3.0 * c1["feat"]).max() < 1e-12 * np.abs(c1["feat"]).max() * 10


def test_rounding_is_half_to_even():
    z = np.load(G / "rounding.npz")
    assert list(z["r"]) == [0, 2, 2, 126, 128, 254, 254, 0, 255, 0, 255]


def test_pow2_float_formula_equals_integer_next_pow2():
    """the reference computes the padded size in float32 (utilities.py:736-751); the engine uses the
    exact integer next power of two: identical for every n <= 4096."""
    for n in range(1, 4097):
        assert O.pow2_target(n) == 1 << (n - 1).bit_length()


@pytest.mark.parametrize("size", [32, 64, 128, 256])
def test_shape_contract_u8_in_u8_out(size):
    """tests/bfcnn/test_model_denoiser.py:61-70: DenoiserModule keeps shape and dtype."""
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=1)["model"])
    params, state = O.init_params(spec, seed=0)
    x = np.random.default_rng(size).integers(0, 256, (1, size, size, 3)).astype(np.uint8)
    y = O.denoiser_module_call(spec, params, state, x)
    assert y.shape == x.shape and y.dtype == np.uint8


def _lena(size, channels):
    from PIL import Image
    im = Image.open(G / "lena.jpg").convert("L" if channels == 1 else "RGB").resize((size, size), Image.BILINEAR)
    x = np.asarray(im, dtype=np.float64).reshape(1, size, size, channels)
    return x / 255.0 - 0.5          # load_image(..., normalize=True)


@pytest.mark.parametrize("ptype", [None, "laplacian", "gaussian"])
@pytest.mark.parametrize("levels", [1, 3])
@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("size", [64, 256])
def test_pyramid_round_trip_reference_identity(ptype, levels, channels, size):
    """the reference's only numeric pin on the path (tests/bfcnn/test_pyramid.py:22-409):
    len(pyramid(x)) == levels and mean(|inverse(pyramid(x)) - x|) < 1e-7, kernel_size (3,3)."""
    if ptype is None:
        cfg, levels = None, 1
    else:
        cfg = {"levels": levels, "type": ptype, "kernel_size": (3, 3), "xy_max": (1.0, 1.0)}
    x = _lena(size, channels)
    pyr = O.build_pyramid(cfg)(x)
    assert len(pyr) == levels
    rec = O.build_inverse_pyramid(cfg)(pyr)
    assert np.mean(np.abs(rec - x)) < 1e-7


def test_pyramid_golden():
    z = np.load(G / "pyramid.npz")
    x = z["x"]
    assert np.abs(O.upsample_bilinear_2x(x) - z["up_bilinear"]).max() < 1e-14
    for k in (2, 3, 5):
        assert np.abs(O.avg_pool_same(x, (k, k), 2) - z[f"pool{k}"]).max() < 1e-14


def test_mae_is_keras_relu_threshold_not_soft_hinge():
    """loss.py:53-57: |e| if |e| > hinge else 0 (not |e| - hinge), then min(., cutoff)."""
    e = np.array([[[[0.2, 0.5, 0.6, 300.0]]]])
    assert abs(O.mae_diff(e, hinge=0.5, cutoff=255.0) - (0.6 + 255.0) / 4) < 1e-12
