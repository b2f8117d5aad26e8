"""
Pins the NumPy oracle against an independent second implementation (PyTorch-CPU
float64: F.conv2d / F.batch_norm / F.interpolate / F.avg_pool2d + autograd).
The reference itself (TensorFlow) is not runnable here, see oracle/bfcnn_oracle.py
header.  Agreement <= 1e-10 is required before golden fixtures are frozen.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bfcnn_oracle as O

torch.set_num_threads(4)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).double()


def _nchw(x):
    return _t(x).permute(0, 3, 1, 2).contiguous()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def _w(w):  # HWIO -> OIHW
    return _t(w).permute(3, 2, 0, 1).contiguous()


def torch_hydra(spec, P, S, x, training):
    """Independent torch restatement of the hydra graph (fp64)."""
    new_state = {}
    xn = (x.clamp(spec.v_min, spec.v_max) - spec.v_min) / (spec.v_max - spec.v_min) - 0.5
    k = spec.kernel_size
    f = F.conv2d(xn, P["base/kernel"], padding=k // 2)
    for i in range(spec.no_layers):
        t = f
        nb = len(spec.block_kernels)
        for j in range(nb):
            kk = spec.block_kernels[j]
            t = F.conv2d(t, P[f"block{i}/conv{j}/kernel"], padding=kk // 2)
            if j >= 1 and spec.use_bn:
                g = P[f"block{i}/bn{j}/gamma"]
                mm = S[f"block{i}/bn{j}/moving_mean"].clone()
                mv = S[f"block{i}/bn{j}/moving_variance"].clone()
                # torch momentum = 1 - keras momentum; torch also feeds the unbiased
                # variance into the running update (same as keras' fused kernel)
                t = F.batch_norm(t, mm, mv, weight=g, bias=None, training=training,
                                 momentum=1.0 - spec.bn_momentum, eps=spec.bn_eps)
                new_state[f"block{i}/bn{j}/moving_mean"] = mm
                new_state[f"block{i}/bn{j}/moving_variance"] = mv
            if j < nb - 1:
                t = F.relu(t)
        f = f + t
    h0 = F.conv2d(f, P["head/conv0/kernel"])
    h1 = F.conv2d(h0, P["head/conv1/kernel"])
    p = torch.tanh(2.0 * h1) * 0.51
    y = (p.clamp(-0.5, 0.5) + 0.5) * (spec.v_max - spec.v_min) + spec.v_min
    return y, new_state


def _torch_params(spec, params, state, requires_grad=False):
    P = {}
    for n, (o, s) in spec.offsets().items():
        a = params[o:o + int(np.prod(s))].reshape(s).astype(np.float64)
        P[n] = (_w(a) if len(s) == 4 else _t(a)).requires_grad_(requires_grad)
    S = {n: _t(state[o:o + int(np.prod(s))].reshape(s).astype(np.float64))
         for n, (o, s) in spec.state_offsets().items()}
    return P, S


@pytest.mark.parametrize("k", [1, 3, 5, 7])
def test_conv_same_matches_torch(k):
    rng = np.random.default_rng(k)
    x = rng.standard_normal((2, 11, 13, 5))
    w = rng.standard_normal((k, k, 5, 7))
    ref = _nhwc(F.conv2d(_nchw(x), _w(w), padding=k // 2))
    assert np.abs(O.conv2d_same(x, w) - ref).max() < 1e-11


def test_conv_grads_match_autograd():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 9, 10, 4))
    w = rng.standard_normal((3, 3, 4, 6))
    dy = rng.standard_normal((2, 9, 10, 6))
    xt, wt = _nchw(x).requires_grad_(True), _w(w).requires_grad_(True)
    (F.conv2d(xt, wt, padding=1) * _nchw(dy)).sum().backward()
    assert np.abs(O.conv2d_same_grad_input(dy, w) - _nhwc(xt.grad)).max() < 1e-11
    dw = wt.grad.permute(2, 3, 1, 0).numpy()
    assert np.abs(O.conv2d_same_grad_kernel(x, dy, 3, 3) - dw).max() < 1e-10


@pytest.mark.parametrize("training", [False, True])
def test_hydra_forward_matches_torch(training):
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=3)["model"])
    params, state = O.init_params(spec, seed=3)
    _, noisy = O.synthetic_batch(2, 24, 40, seed=5)
    x = noisy.astype(np.float64)
    P, S = _torch_params(spec, params, state)
    yt, ns = torch_hydra(spec, P, S, _nchw(x), training)
    if training:
        y, new_state = O.hydra_forward(spec, params, state, x, training=True)
        so = spec.state_offsets()
        for n, (o, s) in so.items():
            assert np.abs(new_state[o:o + s[0]] - ns[n].numpy()).max() < 1e-12, n
    else:
        y = O.hydra_forward(spec, params, state, x)
    assert np.abs(y - _nhwc(yt)).max() < 1e-9


def test_train_step_grads_match_autograd():
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=11)
    clean, noisy = O.synthetic_batch(2, 16, 20, seed=7)
    gt, x = clean.astype(np.float64), noisy.astype(np.float64)
    total, ml, dl, pred, grads, new_state = O.train_step_single_gpu(spec, ls, params, state, gt, x)

    P, S = _torch_params(spec, params, state, requires_grad=True)
    yt, _ = torch_hydra(spec, P, S, _nchw(x), True)
    err = _nchw(gt) - yt
    a = err.abs()
    d = torch.where(a > ls.hinge, a, torch.zeros_like(a)).clamp(max=ls.cutoff)
    loss = d.mean() * ls.mae_multiplier
    reg = 0.0
    for n, _, kind, r in spec.tensors():
        if kind == "conv":
            reg = reg + (0.01 * P[n].abs().sum() if r == "l1" else 0.01 * (P[n] ** 2).sum())
    tot = loss + reg * ls.regularization
    tot.backward()
    assert abs(float(tot.detach()) - total) < 1e-10
    assert abs(float(reg.detach()) - ml["regularization_loss"]) < 1e-10
    for n, (o, s) in spec.offsets().items():
        g = P[n].grad
        g = g.permute(2, 3, 1, 0).numpy() if len(s) == 4 else g.numpy()
        mine = grads[o:o + int(np.prod(s))].reshape(s)
        assert np.abs(mine - g).max() < 1e-10 * max(1.0, np.abs(g).max()), n


def _torch_ssim_mean(y, x):
    """tf.image.ssim(filter_size=7, max_val=255) restated with F.conv2d (NCHW fp64), for autograd."""
    g = torch.from_numpy(O.ssim_gauss_kernel(7, 1.5))[None, None].repeat(x.shape[1], 1, 1, 1)
    red = lambda t: F.conv2d(t, g, groups=x.shape[1])
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    m0, m1 = red(y), red(x)
    lum = (2.0 * m0 * m1 + c1) / (m0 * m0 + m1 * m1 + c1)
    cs = (2.0 * red(x * y) - 2.0 * m0 * m1 + c2) / (red(x * x + y * y) - m0 * m0 - m1 * m1 + c2)
    return (lum * cs).mean(dim=(2, 3)).mean(dim=1).mean()


def test_ssim_window_and_gradient_match_autograd():
    g = O.ssim_gauss_kernel(7, 1.5)
    assert abs(g.sum() - 1.0) < 1e-15 and np.allclose(g, g.T) and g[3, 3] == g.max()
    rng = np.random.default_rng(0)
    gt = rng.uniform(0, 255, (2, 13, 17, 3))
    pred = np.clip(gt + rng.normal(0, 25, gt.shape), 0, 255)
    val, grad = O.ssim_mean_and_grad(gt, pred)
    x = _nchw(pred).requires_grad_(True)
    v = _torch_ssim_mean(_nchw(gt), x)
    v.backward()
    assert abs(val - float(v.detach())) < 1e-14
    assert np.abs(grad - _nhwc(x.grad)).max() < 1e-16 + 1e-10 * np.abs(grad).max()
    same, _ = O.ssim_mean_and_grad(gt, gt)
    assert abs(same - 1.0) < 1e-12


@pytest.mark.parametrize("over", [{"ssim_multiplier": 1.0}, {"mse_multiplier": 0.5},
                                  {"ssim_multiplier": 1.0, "mse_multiplier": 0.5, "hinge": 3.5}])
def test_train_step_grads_with_rmse_and_ssim_terms_match_autograd(over):
    """loss.py:190-247 with all three terms: mae * m1 + rmse(hinge, cutoff^2) * m2 + (1 - mean ssim) * m3."""
    cfg = O.canonical_config(no_layers=1)
    cfg["loss"].update(over)
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=5)
    clean, noisy = O.synthetic_batch(2, 12, 14, seed=3)
    gt, x = clean.astype(np.float64), noisy.astype(np.float64)
    total, ml, dl, pred, grads, _ = O.train_step_single_gpu(spec, ls, params, state, gt, x, depth_weight=0.6)
    P, S = _torch_params(spec, params, state, requires_grad=True)
    yt, _ = torch_hydra(spec, P, S, _nchw(x), True)
    err = _nchw(gt) - yt
    a = err.abs()
    loss = torch.where(a > ls.hinge, a, torch.zeros_like(a)).clamp(max=ls.cutoff).mean() * ls.mae_multiplier
    if ls.mse_multiplier > 0:
        d = torch.where(err > ls.hinge, err, torch.zeros_like(err)).clamp(max=ls.cutoff * ls.cutoff) ** 2
        loss = loss + torch.sqrt(d.mean(dim=(1, 2, 3)) + O.DEFAULT_EPSILON).mean() * ls.mse_multiplier
    if ls.ssim_multiplier > 0:
        loss = loss + (1.0 - _torch_ssim_mean(_nchw(gt), yt)) * ls.ssim_multiplier
    reg = 0.0
    for n, _, kind, r in spec.tensors():
        if kind == "conv":
            reg = reg + (0.01 * P[n].abs().sum() if r == "l1" else 0.01 * (P[n] ** 2).sum())
    tot = loss * 0.6 + reg * ls.regularization
    tot.backward()
    assert abs(float(tot.detach()) - total) < 1e-10
    assert abs(float(loss.detach()) - dl[0]["total_loss"]) < 1e-10
    for n, (o, s) in spec.offsets().items():
        g = P[n].grad
        g = g.permute(2, 3, 1, 0).numpy() if len(s) == 4 else g.numpy()
        mine = grads[o:o + int(np.prod(s))].reshape(s)
        assert np.abs(mine - g).max() < 1e-10 * max(1.0, np.abs(g).max()), n


def test_adam_matches_torch_formula():
    """keras-2.13 Adam differs from torch.optim.Adam only in where epsilon sits
    (alpha*m/(sqrt(v)+eps) vs bias-corrected sqrt(v)); check against a literal
    restatement and check clipping."""
    rng = np.random.default_rng(1)
    p = rng.standard_normal(100)
    m = np.zeros(100)
    v = np.zeros(100)
    for it in range(3):
        g = rng.standard_normal(100) * 3
        norm = np.linalg.norm(g)
        gc = g * (1.0 / max(norm, 1.0))
        t = it + 1
        alpha = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        m_ref = 0.9 * m + 0.1 * gc
        v_ref = 0.999 * v + 0.001 * gc * gc
        p_ref = p - alpha * m_ref / (np.sqrt(v_ref) + 1e-7)
        p, m, v = O.adam_step(p, g, m, v, it, 1e-3, global_clipnorm=1.0)
        assert np.abs(p - p_ref).max() < 1e-14
        assert np.abs(m - m_ref).max() < 1e-15 and np.abs(v - v_ref).max() < 1e-15


@pytest.mark.parametrize("k", [(2, 2), (3, 3), (5, 5)])
@pytest.mark.parametrize("hw", [(16, 16), (15, 22)])
def test_avg_pool_same_matches_torch(k, hw):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, hw[0], hw[1], 3))
    oh, pt, pb = O.same_pads(hw[0], k[0], 2)
    ow, pl, pr = O.same_pads(hw[1], k[1], 2)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    ones = F.pad(torch.ones(1, 1, hw[0], hw[1], dtype=torch.float64), (pl, pr, pt, pb))
    ref = F.avg_pool2d(xp, k, 2, divisor_override=1) / F.avg_pool2d(ones, k, 2, divisor_override=1)
    assert np.abs(O.avg_pool_same(x, k, 2) - _nhwc(ref)).max() < 1e-12


def test_upsample_matches_torch():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 7, 9, 3))
    ref = _nhwc(F.interpolate(_nchw(x), scale_factor=2, mode="bilinear", align_corners=False))
    assert np.abs(O.upsample_bilinear_2x(x) - ref).max() < 1e-12
    ref = _nhwc(F.interpolate(_nchw(x), scale_factor=2, mode="nearest"))
    assert np.abs(O.upsample_nearest_2x(x) - ref).max() == 0


def test_bench_torch_cpu_training_baseline_computes_the_oracles_step():
    """bench.py --mode train's CPU baseline (torch-CPU fp32 autograd, all cores) is the SAME graph and loss as the oracle's training
    step: its first loss equals the oracle's total loss on the same two images."""
    import importlib.util
    import pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    sp = importlib.util.spec_from_file_location("_bench", root / "bench.py")
    bench = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(bench)
    cfg = O.canonical_config(no_layers=3)
    spec, ls = O.ResnetSpec.from_config(cfg["model"]), O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
    rec = bench.cpu_train_baseline_torch(spec, ls, params, state, 32, budget_s=0.2, nthreads=2)
    c1, n1 = O.synthetic_batch(2, 32, 32, sigma=20.0, seed=99)
    want = O.train_step_single_gpu(spec, ls, params, state, c1.astype(np.float64), n1.astype(np.float64))[0]
    assert abs(rec["first_loss"] - want) <= 2e-4 * abs(want), (rec["first_loss"], want)
    assert rec["value"] > 0 and rec["cores"] == 2
