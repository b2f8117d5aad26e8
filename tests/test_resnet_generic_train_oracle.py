"""Pins oracle/resnet_generic_torch.py (the autograd gradient oracle of the generic resnet family): its inference forward equals
the NumPy restatement, its training-mode BatchNorm equals bfcnn_oracle.bn_train, and autograd agrees with central differences."""
import numpy as np
import pytest

from oracle import bfcnn_oracle as O
from oracle import resnet_generic_oracle as R
from oracle import resnet_generic_torch as T


def _spec(gates, layers=2, **over):
    cfg = R.shipped_config()
    cfg["backbone"].update({"no_layers": layers, "add_gates": gates, **over})
    return R.GenericResnetSpec.from_config(cfg)


@pytest.mark.parametrize("gates", [False, True])
def test_inference_forward_equals_the_numpy_restatement(gates):
    spec = _spec(gates)
    params, state = R.init_params(spec, seed=3)
    x = np.random.default_rng(0).uniform(0, 255, (2, 16, 24, 3))
    a, b = T.infer(spec, params, state, x), R.hydra_forward(spec, params, state, x)
    assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())


def test_gate_tensors_follow_the_second_convolution():
    names = [n for n, _, _ in _spec(True, layers=1).tensors()]
    assert names.index("block0/gate/dense0/kernel") == names.index("block0/bn1/gamma") + 1
    assert names.index("block0/conv2/kernel") == names.index("block0/gate/dense1/kernel") + 1
    assert dict((n, s) for n, s, _ in _spec(True, layers=1).tensors())["block0/gate/dense0/kernel"] == (128, 16)


def test_training_step_state_update_and_central_differences():
    spec = _spec(True, layers=1)
    ls = O.LossSpec.from_config({"hinge": 0.5, "cutoff": 255.0, "mae_multiplier": 1.0, "regularization": 0.01})
    params, state = R.init_params(spec, seed=5)
    params = params.astype(np.float64)
    clean, noisy = O.synthetic_batch(2, 12, 12, seed=1)
    total, ml, dl, pred, grads, new_state = T.train_step(spec, ls, params, state, clean, noisy)
    assert np.isfinite(grads).all() and grads.shape == params.shape and new_state.shape == state.shape
    assert abs(total - (dl["total_loss"] + ml["total_loss"])) < 1e-12
    assert not np.allclose(new_state, state)                          # moving statistics moved
    rng = np.random.default_rng(2)
    idx = rng.choice(params.size, 6, replace=False)
    for i in idx:
        e = 1e-5 * max(1.0, abs(params[i]))
        p1, p2 = params.copy(), params.copy()
        p1[i] += e
        p2[i] -= e
        num = (T.train_step(spec, ls, p1, state, clean, noisy)[0] - T.train_step(spec, ls, p2, state, clean, noisy)[0]) / (2 * e)
        # (a piecewise-linear loss through ReLUs: a central difference crosses a few kinks; 3e-3 is what that leaves)
        assert abs(num - grads[i]) <= 3e-3 * max(abs(grads[i]), 1e-3), (i, num, grads[i])


def test_batchnorm_training_semantics_match_bfcnn_oracle():
    import torch
    spec = _spec(False, layers=1)
    params, state = R.init_params(spec, seed=7)
    x = np.random.default_rng(3).uniform(0, 255, (3, 8, 8, 3))
    P = T.views(spec, torch.tensor(params.astype(np.float64)))
    S = T.state_views(spec, torch.tensor(state.astype(np.float64)))
    _, new = T.hydra(spec, P, S, torch.from_numpy(x), True)
    # reproduce conv0 -> relu -> depthwise of block 0 in NumPy and push it through bn_train
    Pn = {k: v.numpy() for k, v in P.items()}
    f = O.conv2d_same(O.layer_normalize(x, 0.0, 255.0), Pn["base/kernel"])
    t = np.maximum(O.conv2d_same(f, Pn["block0/conv0/kernel"]), 0)
    t = R.depthwise_mult_same(t, Pn["block0/conv1/kernel"])
    _, _, nm, nv = O.bn_train(t, Pn["block0/bn1/gamma"], S["block0/bn1/moving_mean"].numpy(), S["block0/bn1/moving_variance"].numpy())
    assert np.abs(new["block0/bn1/moving_mean"].numpy() - nm).max() < 1e-12
    assert np.abs(new["block0/bn1/moving_variance"].numpy() - nv).max() < 1e-12
