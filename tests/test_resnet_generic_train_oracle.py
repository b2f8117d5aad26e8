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


FLAGS = dict(add_initial_bn=True, add_final_bn=True, add_channelwise_scaling=True, add_learnable_multiplier=True, dropout_rate=0.5)


def test_builder_flags_forward_and_regularisers():
    """add_initial_bn / add_final_bn, ChannelwiseMultiplier / Multiplier closing every block and the backbone, RandomOnOff: the torch
    restatement equals the NumPy one at inference; the regularisers are L1(0.1) on a channelwise w0, L1(1.0) on a multiplier w0 plus the
    constant L1(1.0)|w1 = 1| the Multiplier layer adds for its non-trainable weight (custom_layers.py:1067-1074)"""
    import torch
    spec = _spec(False, **FLAGS)
    names = [n for n, _, _ in spec.tensors()]
    assert names.index("initial_bn/gamma") == 1 and names[-5:-2] == ["final_bn/gamma", "channelwise/w0", "multiplier/w0"]
    assert names.index("block0/multiplier/w0") == names.index("block0/channelwise/w0") + 1 == names.index("block1/conv0/kernel") - 1
    assert [n for n, _ in spec.state_tensors()][:2] == ["initial_bn/moving_mean", "initial_bn/moving_variance"]
    params, state = R.init_params(spec, seed=4)
    x = np.random.default_rng(0).uniform(0, 255, (2, 16, 24, 3))
    a, b = T.infer(spec, params, state, x), R.hydra_forward(spec, params, state, x)
    assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    plain = _spec(False)
    P = T.views(spec, torch.tensor(params.astype(np.float64)))
    extra = float(T.regularization(spec, P)) - float(T.regularization(plain, {n: P[n] for n, _, _ in plain.tensors()}))
    w = {n: P[n].numpy() for n, _, k in spec.tensors() if k in ("channelwise", "multiplier")}
    want = sum(0.1 * np.abs(v).sum() if "channelwise" in n else 1.0 * np.abs(v).sum() + 1.0 for n, v in w.items())
    assert abs(extra - want) < 1e-9


def test_builder_flags_training_step_central_differences():
    spec = _spec(True, layers=1, **FLAGS)
    ls = O.LossSpec.from_config({"hinge": 0.5, "cutoff": 255.0, "mae_multiplier": 1.0, "regularization": 0.01})
    params, state = R.init_params(spec, seed=6)
    params = params.astype(np.float64)
    clean, noisy = O.synthetic_batch(2, 12, 12, seed=2)
    drop = {0: np.array([2.0, 0.0])}
    total, ml, dl, pred, grads, new_state = T.train_step(spec, ls, params, state, clean, noisy, drop_scale=drop)
    off = {}
    o = 0
    for n, s, k in spec.tensors():
        off[n] = o
        o += int(np.prod(s))
    for name in ("initial_bn/gamma", "final_bn/gamma", "block0/channelwise/w0", "block0/multiplier/w0", "channelwise/w0", "multiplier/w0"):
        i = off[name]
        e = 1e-5
        p1, p2 = params.copy(), params.copy()
        p1[i] += e
        p2[i] -= e
        num = (T.train_step(spec, ls, p1, state, clean, noisy, drop_scale=drop)[0] -
               T.train_step(spec, ls, p2, state, clean, noisy, drop_scale=drop)[0]) / (2 * e)
        assert abs(num - grads[i]) <= 3e-3 * max(abs(grads[i]), 1e-3), (name, num, grads[i])
    # a dropped sample's branch contributes nothing: its prediction does not depend on block 0's weights
    p3 = params.copy()
    p3[off["block0/conv0/kernel"]:off["block0/conv0/kernel"] + 64] += 0.1
    pred3 = T.train_step(spec, ls, p3, state, clean, noisy, drop_scale={0: np.array([0.0, 0.0])})[3]
    pred0 = T.train_step(spec, ls, params, state, clean, noisy, drop_scale={0: np.array([0.0, 0.0])})[3]
    assert np.abs(pred3 - pred0).max() < 1e-12


@pytest.mark.parametrize("sp", [dict(scale_type="local", pool_size=(8, 8)), dict(scale_type="multiscale", pool_size=(8, 8), activation_type="soft"),
                                dict(scale_type="mixed", pool_size=(16, 16)), dict(scale_type="global")],
                         ids=["local", "multiscale-soft", "mixed", "global"])
def test_selector_block_forward_and_gradient(sp):
    """selector_block (custom_layers_selector.py:81-330) in the torch restatement: equal to the NumPy one at inference, its kernels
    regularised with the block's `kernel_regularizer` (default "l1", :88), autograd against central differences on them"""
    spec = _spec(False, selector_params=sp)
    params, state = R.init_params(spec, seed=9)
    x = np.random.default_rng(1).uniform(0, 255, (2, 16, 24, 3))
    a, b = T.infer(spec, params, state, x), R.hydra_forward(spec, params, state, x)
    assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
    ls = O.LossSpec.from_config({"hinge": 0.0, "cutoff": 255.0, "mae_multiplier": 1.0, "regularization": 0.01})
    params = params.astype(np.float64)
    clean, noisy = O.synthetic_batch(2, 16, 24, seed=2)
    total, ml, dl, pred, grads, _ = T.train_step(spec, ls, params, state, clean, noisy)
    off, o = {}, 0
    for n, s, k in spec.tensors():
        off[n] = (o, int(np.prod(s)))
        o += int(np.prod(s))
    kind = "dense" if sp["scale_type"] == "global" else "conv"
    rng = np.random.default_rng(3)
    for name in (f"block0/selector/{kind}0/kernel", f"block1/selector/{kind}1/kernel"):
        o0, n0 = off[name]
        assert np.abs(grads[o0:o0 + n0]).max() > 0
        d = np.zeros(params.size)
        d[o0:o0 + n0] = rng.standard_normal(n0)
        d /= np.linalg.norm(d)
        e = 1e-5
        num = (T.train_step(spec, ls, params + e * d, state, clean, noisy)[0] - T.train_step(spec, ls, params - e * d, state, clean, noisy)[0]) / (2 * e)
        assert abs(num - grads @ d) <= 5e-3 * max(abs(grads @ d), 1e-3), (name, num, grads @ d)
