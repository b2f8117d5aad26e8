"""GPU parity of the training step (train_step_single_gpu, apply_grads) against the fp64 oracle.
Tolerances (fp32 kernels vs fp64 oracle): losses 1e-5 relative; gradients 2e-4 of the tensor's
max |g| (sums over up to 1e5 pixels of fp32 products); Adam update 1e-6 absolute."""
import pathlib

import os

import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf
from blind_image_denoising_amd import _native as N
from oracle import bfcnn_oracle as O
from helpers import compare_or_explain_by_ties

pytestmark = pytest.mark.gpu
G = pathlib.Path(__file__).resolve().parent / "golden"


def _setup(no_layers, seed=42, loss_over=None, head_scale=1.0):
    cfg = O.canonical_config(no_layers=no_layers)
    if loss_over:
        cfg["loss"].update(loss_over)
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=seed, nontrivial_bn=True)
    if head_scale != 1.0:
        for name, (o, s) in spec.offsets().items():
            if name.startswith("head"):
                params[o:o + int(np.prod(s))] *= head_scale
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(params, state)
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    return cfg, spec, ls, params, state, m, fns


def _cmp_grads(spec, got, ref, rel=2e-4):
    for name, (o, s) in spec.offsets().items():
        n = int(np.prod(s))
        g, r = got[o:o + n], ref[o:o + n]
        scale = max(np.abs(r).max(), 1e-6)
        err = np.abs(g - r).max()
        assert err <= rel * scale, f"{name}: grad err {err:.3e} vs scale {scale:.3e}"


# both arithmetics of the training convolutions (split-f16 default, exact fp32): same oracle, same bars
@pytest.mark.parametrize("train_arith", [1, 0], ids=["f16x3", "f32"])
@pytest.mark.parametrize("no_layers,shape", [(1, (2, 16, 32)), (2, (2, 24, 32)), (3, (3, 33, 47))])
def test_train_step_matches_oracle(no_layers, shape, train_arith):
    cfg, spec, ls, params, state, m, fns = _setup(no_layers)
    m.set_option("train_arith", train_arith)
    clean, noisy = O.synthetic_batch(*shape, seed=7)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    assert abs(total.item() - r_total) <= 1e-5 * abs(r_total)
    assert abs(ml["regularization_loss"].item() - r_ml["regularization_loss"]) <= 1e-5 * r_ml["regularization_loss"]
    assert abs(ml["total_loss"].item() - r_ml["total_loss"]) <= 1e-5 * r_ml["total_loss"]
    for k in ("total_loss", "mae_loss", "mse_loss"):
        assert abs(dl[0][k].item() - r_dl[0][k]) <= 1e-5 * abs(r_dl[0][k]), k
    assert np.abs(pred.cpu().numpy() - r_pred).max() < 0.02
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads)
    assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5          # BN moving statistics updated


@pytest.mark.parametrize("no_layers,shape", [(1, (2, 16, 32)), (3, (3, 33, 47)), (2, (5, 70, 150))])
def test_two_convolution_backward_kernel_matches_oracle_and_the_two_kernel_path(no_layers, shape):
    """train_fused_bwd2 = 1 (bwd2_h3_kernel: both convolutions' backward of a block in one launch, dT never leaves the CU):
    the same oracle bars as the default path, and the two paths agree with each other far inside those bars."""
    cfg, spec, ls, params, state, m, fns = _setup(no_layers)
    clean, noisy = O.synthetic_batch(*shape, seed=23)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    got = {}
    for v in (0, 1):
        m.set_option("train_fused_bwd2", v)
        total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
        got[v] = (total.item(), grads.cpu().numpy().astype(np.float64))
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    assert abs(got[1][0] - r_total) <= 1e-5 * abs(r_total)
    _cmp_grads(spec, got[1][1], r_grads)
    assert got[0][0] == got[1][0]                                        # same forward
    _cmp_grads(spec, got[1][1], got[0][1], rel=5e-5)


@pytest.mark.parametrize("no_layers,shape", [(1, (2, 16, 32)), (3, (3, 33, 47)), (2, (5, 70, 150)), (4, (2, 40, 256))])
def test_block_forward_kernel_matches_oracle_and_the_two_kernel_path(no_layers, shape):
    """train_fwd_block = 2 (fwd_block_h3t_kernel wherever it can run: Add + BatchNorm of the block in front on load, conv_0 + ReLU,
    conv_1 + batch statistics in ONE row-streaming launch; bf_train_step picks it by itself from 4 096 image rows per step on):
    the same oracle bars as the two-kernel forward, the two paths agree far inside them, and the library says which one ran."""
    cfg, spec, ls, params, state, m, fns = _setup(no_layers)
    clean, noisy = O.synthetic_batch(*shape, seed=23)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    got = {}
    for v in (0, 2):
        m.set_weights(params, state)
        m.set_option("train_fwd_block", v)
        total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
        got[v] = (total.item(), grads.cpu().numpy().astype(np.float64), m.state.cpu().numpy().copy())
        kernels = N.lib().bf_get_train_kernels(m._h).decode()
        assert ("fwd_block_h3t_kernel" in kernels) == (v == 2), kernels
    gt64, x64 = gt.astype(np.float64), x.astype(np.float64)
    for v in (0, 2):
        # (a ReLU input within rounding of its kink on one pixel moves a gradient tensor past the bar: checked, not forgiven -- the
        # oracle must name such elements and agree with them on the other side; helpers.compare_or_explain_by_ties)
        def compare(ref, v=v):
            r_total, r_ml, r_dl, r_pred, r_grads, r_state = ref
            assert abs(got[v][0] - r_total) <= 1e-5 * abs(r_total)
            _cmp_grads(spec, got[v][1], r_grads)
            assert np.abs(got[v][2] - r_state).max() < 1e-5
        compare_or_explain_by_ties(compare, lambda flips: O.train_step_single_gpu(spec, ls, params, state, gt64, x64, flips=flips),
                                   lambda: O.training_step_ties(spec, ls, params, state, gt64, x64))
    assert abs(got[2][0] - got[0][0]) <= 1e-6 * abs(got[0][0])
    _cmp_grads(spec, got[2][1], got[0][1], rel=6e-4)


@pytest.mark.parametrize("no_layers,shape", [(3, (3, 33, 47)), (5, (2, 40, 256)), (2, (1, 24, 300))])
def test_batchnorm_finalisation_inside_the_block_kernels(no_layers, shape):
    """train_fold_finalize: bn_finalize / bn_bwd_finalize computed in the prologue of the next block kernel (every workgroup sums the
    partials itself, workgroup 0 writes scale / shift / moving statistics / d gamma) against the same step with the two finalisation
    kernels between the blocks: same loss, gradients and moving statistics to rounding (24 stripes of rows instead of 32), twice the same bits"""
    cfg, spec, ls, params, state, m, fns = _setup(no_layers)
    clean, noisy = O.synthetic_batch(*shape, seed=31)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    got = {}
    for v in (0, 1, 11):
        m.set_weights(params, state)
        m.set_option("train_fwd_block", 2)
        m.set_option("train_bwd_block", 2)
        m.set_option("train_fold_finalize", v % 10)
        total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
        got[v] = (total.item(), grads.cpu().numpy().astype(np.float64), m.state.cpu().numpy().copy())
    assert got[1][0] == got[11][0] and np.array_equal(got[1][1], got[11][1]) and np.array_equal(got[1][2], got[11][2])
    assert abs(got[1][0] - got[0][0]) <= 1e-6 * abs(got[0][0])
    _cmp_grads(spec, got[1][1], got[0][1], rel=2e-5)
    assert np.abs(got[1][2] - got[0][2]).max() < 1e-6


@pytest.mark.parametrize("no_layers,shape", [(1, (2, 16, 32)), (3, (3, 33, 47)), (2, (5, 70, 150)), (4, (2, 40, 256)), (2, (1, 24, 300))])
@pytest.mark.parametrize("fwd_block", [0, 2])
def test_block_backward_kernel_matches_oracle_and_the_per_convolution_path(no_layers, shape, fwd_block):
    """train_bwd_block = 2 (bwd_block_h3t_kernel wherever it can run: BatchNorm backward on load, T RECOMPUTED from the block input,
    both weight gradients, both data gradients, the skip's gradient and the next BatchNorm's sums in ONE row-streaming launch per
    block; bf_train_step picks it by itself from 8 192 strip rows per step on), with the two-kernel forward (which still writes T) and
    with the one-kernel forward (which then does not): the same oracle bars, and the library says which kernels ran."""
    cfg, spec, ls, params, state, m, fns = _setup(no_layers)
    clean, noisy = O.synthetic_batch(*shape, seed=23)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    got = {}
    for v in (0, 2):
        m.set_weights(params, state)
        m.set_option("train_fwd_block", fwd_block if v else 0)
        m.set_option("train_bwd_block", v)
        total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
        got[v] = (total.item(), grads.cpu().numpy().astype(np.float64), m.state.cpu().numpy().copy())
        kernels = N.lib().bf_get_train_kernels(m._h).decode()
        assert ("bwd_block_h3t_kernel" in kernels) == (v == 2), kernels
        assert ("fwd_block_h3t_kernel" in kernels) == (v == 2 and fwd_block == 2 and shape[2] <= 256), kernels
    gt64, x64 = gt.astype(np.float64), x.astype(np.float64)

    def compare(ref):
        r_total, r_ml, r_dl, r_pred, r_grads, r_state = ref
        assert abs(got[2][0] - r_total) <= 1e-5 * abs(r_total)
        _cmp_grads(spec, got[2][1], r_grads)
        assert np.abs(got[2][2] - r_state).max() < 1e-5
    compare_or_explain_by_ties(compare, lambda flips: O.train_step_single_gpu(spec, ls, params, state, gt64, x64, flips=flips),
                               lambda: O.training_step_ties(spec, ls, params, state, gt64, x64))
    assert abs(got[2][0] - got[0][0]) <= 1e-6 * abs(got[0][0])
    _cmp_grads(spec, got[2][1], got[0][1], rel=6e-4)


def test_train_step_many_tiles_two_stage_reductions():
    """more than 512 tile partials per BatchNorm (the two-stage fp64 reductions) and several tiles per persistent
    weight-gradient workgroup: loss, BN state and gradients against the oracle at a larger shape."""
    cfg, spec, ls, params, state, m, fns = _setup(1)
    clean, noisy = O.synthetic_batch(17, 128, 160, seed=11)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    assert abs(total.item() - r_total) <= 1e-5 * abs(r_total)
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads)
    assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5


def test_golden_train_step_and_adam():
    z, n = np.load(G / "train_step.npz"), np.load(G / "net_2blocks.npz")
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(n["params"], n["state"])
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    opt, _ = bf.optimizer_builder(cfg["train"]["optimizer"])
    total, ml, dl, pred, grads = fns.train_step_single_gpu(
        torch.from_numpy(z["clean"].astype(np.float32)), torch.from_numpy(z["noisy"].astype(np.float32)))
    assert abs(total.item() - float(z["total"])) <= 1e-5 * float(z["total"])
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), z["grads"])
    fns.apply_grads(opt, grads, None)
    assert opt.iterations == 1
    assert np.abs(m.params.cpu().numpy() - z["params_after"]).max() < 2e-6
    assert np.abs(opt.m.cpu().numpy() - z["m_after"]).max() < 1e-6
    assert np.abs(opt.v.cpu().numpy() - z["v_after"]).max() < 1e-7


def test_golden_train_step_with_all_loss_terms():
    z, t, n = np.load(G / "train_step_full_loss.npz"), np.load(G / "train_step.npz"), np.load(G / "net_2blocks.npz")
    cfg = O.canonical_config(no_layers=2)
    cfg["loss"] = {"hinge": 3.5, "cutoff": 255.0, "mae_multiplier": 1.0, "mse_multiplier": 0.5, "ssim_multiplier": 1.0,
                   "regularization": 0.01}
    spec = O.ResnetSpec.from_config(cfg["model"])
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(n["params"], n["state"])
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    total, ml, dl, pred, grads = fns.train_step_single_gpu(
        torch.from_numpy(t["clean"].astype(np.float32)), torch.from_numpy(t["noisy"].astype(np.float32)), (0.8,), 0.0, None)
    assert abs(total.item() - float(z["total"])) <= 1e-5 * float(z["total"])
    for k, key in (("ssim_loss", "ssim"), ("mae_loss", "mae"), ("mse_loss", "mse"), ("total_loss", "denoiser_total")):
        assert abs(dl[0][k].item() - float(z[key])) <= 1e-5 * abs(float(z[key])), k
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), z["grads"], rel=3e-4)


def test_adam_exact_on_given_gradient():
    """isolates bf_adam_step: feed the oracle's own gradient; with/without clipping; 3 steps."""
    cfg, spec, ls, params, state, m, fns = _setup(1)
    rng = np.random.default_rng(0)
    opt, sched = bf.optimizer_builder(cfg["train"]["optimizer"])
    p, mm, vv = params.astype(np.float64), np.zeros(params.size), np.zeros(params.size)
    for it in range(3):
        g = (rng.standard_normal(params.size) * (5.0 if it == 0 else 0.001)).astype(np.float32)   # step 0 clips
        opt.apply_gradients(torch.from_numpy(g).cuda(), m)
        p, mm, vv = O.adam_step(p, g.astype(np.float64), mm, vv, it, sched(it), global_clipnorm=1.0)
        assert np.abs(m.params.cpu().numpy() - p).max() < 2e-6, it


@pytest.mark.parametrize("mode", ["clipnorm", "clipvalue", "clipnorm_wins_over_clipvalue"])
def test_adam_per_tensor_clipping_modes(mode):
    """keras' other clipping modes (bfcnn/optimizer.py:165-169, the shipped unet configs use gradient_clipping_by_norm_local):
    clipnorm = tf.clip_by_norm on every gradient tensor, clipvalue = clip_by_value; precedence as keras 2.13."""
    cfg, spec, ls, params, state, m, fns = _setup(2)
    ocfg = {"type": "ADAM", "schedule": cfg["train"]["optimizer"]["schedule"]}
    kw = {}
    if mode != "clipvalue":
        ocfg["gradient_clipping_by_norm_local"] = kw["clipnorm"] = 0.05
    if mode != "clipnorm":
        ocfg["gradient_clipping_by_value"] = kw["clipvalue"] = 0.002
    opt, sched = bf.optimizer_builder(ocfg)
    offs = sorted(o for o, _ in spec.offsets().values()) + [params.size]
    rng = np.random.default_rng(1)
    p, mm, vv = params.astype(np.float64), np.zeros(params.size), np.zeros(params.size)
    for it in range(3):
        g = (rng.standard_normal(params.size) * (0.01 if it != 1 else 1e-4)).astype(np.float32)     # step 1 stays under the norm
        opt.apply_gradients(torch.from_numpy(g).cuda(), m)
        p, mm, vv = O.adam_step(p, g.astype(np.float64), mm, vv, it, sched(it), tensor_offsets=offs, **kw)
        assert np.abs(m.params.cpu().numpy() - p).max() < 2e-6, it
    with pytest.raises(ValueError):
        bf.optimizer_builder(dict(ocfg, gradient_clipping_by_norm=1.0, gradient_clipping_by_norm_local=1.0))


def test_loss_curve_tracks_oracle_over_steps():
    """10 optimiser steps on a tiny problem: the fp32 engine stays on the fp64 oracle's curve."""
    cfg, spec, ls, params, state, m, fns = _setup(2, seed=3)
    opt, sched = bf.optimizer_builder(cfg["train"]["optimizer"])
    clean, noisy = O.synthetic_batch(2, 16, 16, seed=5)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    p, s = params.astype(np.float64), state.astype(np.float64)
    mm, vv = np.zeros(p.size), np.zeros(p.size)
    got, ref = [], []
    for it in range(10):
        total, _, _, _, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x))
        fns.apply_grads(opt, grads, None)
        got.append(total.item())
        r_total, _, _, _, r_grads, s = O.train_step_single_gpu(spec, ls, p, s, gt.astype(np.float64), x.astype(np.float64))
        p, mm, vv = O.adam_step(p, r_grads, mm, vv, it, sched(it), global_clipnorm=1.0)
        ref.append(r_total)
    assert np.allclose(got, ref, rtol=2e-3), (got, ref)
    assert got[-1] < got[0]


def test_train_step_is_bitwise_reproducible():
    cfg, spec, ls, params, state, m, fns = _setup(2)
    clean, noisy = O.synthetic_batch(4, 40, 56, seed=9)
    gt, x = torch.from_numpy(clean.astype(np.float32)), torch.from_numpy(noisy.astype(np.float32))
    _, _, _, _, g1 = fns.train_step_single_gpu(gt, x)
    g1 = g1.clone()
    m.set_weights(params, state)
    _, _, _, _, g2 = fns.train_step_single_gpu(gt, x)
    assert torch.equal(g1, g2)


def test_training_forward_updates_moving_stats_and_matches_oracle():
    cfg, spec, ls, params, state, m, fns = _setup(2)
    _, noisy = O.synthetic_batch(2, 24, 24, seed=2)
    x = noisy.astype(np.float32)
    y = fns.train_step([torch.from_numpy(x)])
    r_y, r_state = O.hydra_forward(spec, params, state, x.astype(np.float64), training=True)
    assert np.abs(y.cpu().numpy() - r_y).max() < 0.02
    assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5
    t = fns.test_step([torch.from_numpy(x)])                   # inference mode uses the NEW moving stats
    assert np.abs(t.cpu().numpy() - O.hydra_forward(spec, params, r_state.astype(np.float32), x.astype(np.float64))).max() < 0.02


@pytest.mark.parametrize("loss_over", [{"ssim_multiplier": 1.0}, {"mse_multiplier": 0.5},
                                       {"ssim_multiplier": 1.0, "mse_multiplier": 0.5, "hinge": 3.5},      # the shipped unet configs
                                       {"ssim_multiplier": 2.0, "mae_multiplier": 0.0, "mse_multiplier": -1.0}],
                         ids=["ssim", "rmse", "shipped", "ssim_only"])
@pytest.mark.parametrize("train_arith", [1, 0], ids=["f16x3", "f32"])
def test_train_step_with_ssim_and_rmse_terms_matches_oracle(loss_over, train_arith):
    """total = mae * m1 + rmse(hinge, cutoff^2) * m2 + (1 - mean tf.image.ssim(7x7)) * m3 (bfcnn/loss.py:190-247): the two
    extra terms of csrc/loss_terms.hip, values and every gradient against the fp64 oracle (whose SSIM gradient is checked
    against torch autograd in tests/test_oracle_vs_torch.py)."""
    cfg, spec, ls, params, state, m, fns = _setup(2, loss_over=loss_over)
    m.set_option("train_arith", train_arith)
    clean, noisy = O.synthetic_batch(3, 24, 40, seed=11)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (0.7,), 0.0, None)
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64), depth_weight=0.7)
    assert abs(total.item() - r_total) <= 1e-5 * abs(r_total)
    for k in ("total_loss", "mae_loss", "mse_loss", "ssim_loss"):
        assert abs(dl[0][k].item() - r_dl[0][k]) <= 1e-5 * max(abs(r_dl[0][k]), 1e-3), k
    assert np.abs(pred.cpu().numpy() - r_pred).max() < 0.02
    # SSIM alone gives a heavy-tailed dL/dprediction (flat windows weigh 1 / (sigma^2 + c2)); through the 22-bit operands
    # of the split-f16 data-gradient convolutions the first block's kernel gradient lands at 2.4e-4 of its maximum
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads, rel=5e-4 if train_arith == 1 else 2e-4)


def test_loss_term_argument_errors():
    cfg, spec, ls, params, state, m, fns = _setup(1, loss_over={"ssim_multiplier": 1.0})
    with pytest.raises(ValueError):                                     # tf.image.ssim needs a 7x7 window
        fns.train_step_single_gpu(torch.zeros((1, 6, 16, 3)), torch.zeros((1, 6, 16, 3)))
    with pytest.raises(ValueError):
        fns.train_step_single_gpu(torch.zeros((1, 16, 16, 3)), torch.zeros((1, 8, 16, 3)))


def test_train_loop_runs_and_saves(tmp_path):
    cfg = O.canonical_config(no_layers=1)
    cfg["train"].update({"epochs": 1, "gpu_batches_per_step": 2})
    clean, noisy = O.synthetic_batch(2, 16, 16, seed=1)
    data = [(torch.from_numpy(clean.astype(np.float32)), torch.from_numpy(noisy.astype(np.float32)))] * 4
    model, hist = bf.train_loop(cfg, str(tmp_path), dataset=data)
    assert len(hist) == 2 and all(np.isfinite(hist))
    mod = bf.load_model(str(tmp_path / "final"))
    assert mod(noisy).shape == noisy.shape


def test_export_model_from_a_training_run(tmp_path):
    """bfcnn/export_model.py:19-190: config + checkpoint directory of a training run -> model directory (pipeline.json + weights) ->
    load_model returns the same uint8 denoiser as the trained model; argument checks as in the reference"""
    cfg = O.canonical_config(no_layers=2)
    cfg["train"].update({"epochs": 1, "gpu_batches_per_step": 1})
    clean, noisy = O.synthetic_batch(2, 16, 16, seed=2)
    data = [(torch.from_numpy(clean.astype(np.float32)), torch.from_numpy(noisy.astype(np.float32)))] * 3
    run, out = tmp_path / "run", tmp_path / "exported" / "model"
    model, _ = bf.train_loop(cfg, str(run), dataset=data)
    module = bf.export_model(cfg, str(run), out, to_tflite=True, test_model=True)
    assert (out / "pipeline.json").exists()
    want = bf.DenoiserModule(model)(noisy)
    assert np.array_equal(module(noisy), want) and np.array_equal(bf.load_model(str(out))(noisy), want)
    with pytest.raises(ValueError, match="Checkpoint directory"):
        bf.export_model(cfg, str(tmp_path / "nowhere"), out)
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(ValueError, match="no checkpoint"):
        bf.export_model(cfg, str(empty), out)


def test_package_level_layers():
    """bfcnn/__init__.py:25-28 exports RandomOnOff, Multiplier, ChannelwiseMultiplier (custom_layers.py:107-127, 1028-1160)"""
    r = np.random.default_rng(0)
    x = r.normal(size=(3, 5, 7, 32)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    for cls, n in ((bf.Multiplier, 1), (bf.ChannelwiseMultiplier, 32)):
        for act in ("linear", "relu"):
            layer = cls(multiplier=1.0, activation=act)
            assert np.array_equal(layer(xd).cpu().numpy(), x)                          # w0 = 0 at creation: x * (0 + 1)
            w0 = r.uniform(-2.0, 1.0, n).astype(np.float32)
            layer.w0.copy_(torch.from_numpy(w0))
            f = w0 + 1.0
            f = np.maximum(f, 0.0) if act == "relu" else f
            assert np.abs(layer(xd).cpu().numpy() - x * f).max() <= 1e-6
            assert layer.get_config()["w1"][0] == 1.0
    # tests/bfcnn/test_custom_layers.py: behind a Dense layer ([B, units]) and behind a convolution; two weights, w0 [1] / [units]
    for units in (8, 16, 32, 64):
        d2 = torch.from_numpy(r.normal(size=(10, units)).astype(np.float32)).cuda()
        for cls, n in ((bf.Multiplier, 1), (bf.ChannelwiseMultiplier, units)):
            layer = cls(multiplier=1.0, regularizer=None, trainable=True, activation="linear")
            assert layer(d2).shape == (10, units)
            assert len(layer.weights) == 2 and layer.weights[0].numpy().shape == (n,)
        d4 = torch.from_numpy(r.normal(size=(10, 32, 32, units)).astype(np.float32)).cuda()
        assert bf.ChannelwiseMultiplier(multiplier=1.0, activation="linear")(d4).shape == (10, 32, 32, units)
    drop = bf.RandomOnOff(rate=0.5, seed=3)
    assert torch.equal(drop(xd), xd) and torch.equal(drop(xd, training=False), xd)
    seen = set()
    for _ in range(20):
        y = drop(xd, training=True).cpu().numpy()
        for b in range(3):
            assert np.array_equal(y[b], 0 * x[b]) or np.allclose(y[b], 2.0 * x[b])     # a whole sample: dropped or scaled by 1 / (1 - rate)
            seen.add(bool(y[b].any()))
    assert seen == {True, False}
    with pytest.raises(RuntimeError, match="MI355X"):
        bf.Multiplier()(torch.from_numpy(x))
    assert sorted(bf.CONFIGS_DICT) == sorted(os.path.splitext(n)[0] for n, _ in bf.configs)


# ---- BASELINE configs[3]: resnet 1x18 training step, 32 images of 256x256 per GPU ------------------------------------

@pytest.mark.parametrize("train_arith", [1, 0], ids=["f16x3", "f32"])
def test_config4_network_on_a_reduced_crop_matches_oracle(train_arith):
    """the 18-block network of configs[3] itself (not a 1-3 block stand-in) on a crop the fp64 oracle finishes in seconds
    (2 x 48 x 48): loss, every gradient tensor of all 18 blocks, BN moving statistics.

    The head kernels are scaled by 0.1: a freshly initialised 18-block network drives tanh(2x) * 0.51 past the +-0.5 clip of
    the denormaliser on 75 % of the pixels, the clip's derivative is discontinuous there, and a gradient comparison would
    then measure on which side of a discontinuity fp32 rounding lands (fp64 oracle alone: weights perturbed by 1e-6
    relative move the gradients by up to 3e-2 with the clip active, 3e-4 without; tools/exp/train_depth_err.py)."""
    cfg, spec, ls, params, state, m, fns = _setup(18, head_scale=0.1)
    m.set_option("train_arith", train_arith)
    clean, noisy = O.synthetic_batch(2, 48, 48, seed=21)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    assert abs(total.item() - r_total) <= 1e-5 * abs(r_total)
    assert abs(dl[0]["mae_loss"].item() - r_dl[0]["mae_loss"]) <= 1e-5 * abs(r_dl[0]["mae_loss"])
    # 37 layers deep: the bar of the shallow tests (2e-4 of a tensor's largest gradient) with the depth's headroom
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads, rel=6e-4)
    assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5


def test_config4_network_with_its_real_initial_head_matches_oracle():
    """configs[3]'s ACTUAL initial state: a freshly initialised 18-block network with the head as glorot makes it drives
    tanh(2x) * 0.51 past the denormaliser's +-0.5 clip on most pixels.  The clip's derivative is discontinuous, so a pixel whose
    |p| lies within the forward error of 0.5 may take either side -- and only such a pixel may.  The engine's side is read off
    its own returned prediction (clipped <=> exactly v_min / v_max; hinge: |gt - prediction| in fp32) and handed to the oracle
    (train_step_single_gpu(flips=...)); every differing gate must sit within 1e-3 (clip, on the +-0.5 scale) / 4e-3 (hinge, 0..255
    scale) of its threshold in the ORACLE's own forward, the clipped fraction must agree to those few pixels, and with the same
    gates the loss and every gradient tensor of all 18 blocks must meet the usual bar."""
    cfg, spec, ls, params, state, m, fns = _setup(18, head_scale=1.0)
    clean, noisy = O.synthetic_batch(2, 48, 48, seed=21)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    gt64, x64 = gt.astype(np.float64), x.astype(np.float64)
    r_pred, _, C = O.hydra_forward(spec, params, state, x64, training=True, want_cache=True)
    p = C["p"]
    pg = pred.cpu().numpy()
    assert np.abs(pg - r_pred).max() < 0.05
    live_gpu = (pg > spec.v_min) & (pg < spec.v_max)                   # strictly inside the clip (the clip's own value: derivative 1 on
    live_ref = (p >= -0.5) & (p <= 0.5)                                # both sides at exactly +-0.5, where fp32 cannot tell anyway)
    diff = np.flatnonzero(live_gpu.ravel() != live_ref.ravel())
    clipped_ref, clipped_gpu = 1.0 - live_ref.mean(), 1.0 - live_gpu.mean()
    assert clipped_ref > 0.3, clipped_ref                             # the state this test is about
    assert abs(clipped_gpu - clipped_ref) <= max(diff.size, 1) / live_ref.size + 1e-12
    assert diff.size <= 0.01 * live_ref.size, diff.size
    flips = []
    for idx in diff:
        margin = abs(abs(p.ravel()[idx]) - 0.5)
        assert margin < 1e-3, (idx, margin)                           # a gate that differs away from the threshold is a fault
        flips.append(("clip", int(idx), float(margin)))
    a_ref = np.abs(gt64 - r_pred)
    a_gpu = np.abs(gt - pg)                                           # fp32, as the head kernel forms it
    hd = np.flatnonzero((a_gpu > np.float32(ls.hinge)).ravel() != (a_ref > ls.hinge).ravel())
    for idx in hd:
        margin = abs(a_ref.ravel()[idx] - ls.hinge)
        assert margin < 4e-3, (idx, margin)
        flips.append(("hinge", int(idx), float(margin)))
    r_total, r_ml, r_dl, _, r_grads, r_state = O.train_step_single_gpu(spec, ls, params, state, gt64, x64, flips=flips)
    assert abs(total.item() - r_total) <= 2e-5 * abs(r_total)
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads, rel=6e-4)
    assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5


def test_config4_full_shape_properties():
    """configs[3] at its real per-GPU shape (1x18, 32 x 256 x 256): properties that need no oracle at this size.
    (i) bitwise reproducible; (ii) finite; (iii) a batch that repeats 2 images 16 times has the batch statistics, the
    loss and the gradients of those 2 images alone (mean losses, BatchNorm over N,H,W) -- the B = 2 run of the same
    kernels is compared, itself pinned by the oracle tests above; (iv) the exact-fp32 arithmetic agrees with the
    split-f16 one at this depth and size; (v) one Adam step moves every tensor and keeps it finite.
    (Head kernels scaled by 0.1 as in the crop test: no pixel sits on the denormaliser's clip.)"""
    cfg, spec, ls, params, state, m, fns = _setup(18, head_scale=0.1)
    clean2, noisy2 = O.synthetic_batch(2, 256, 256, seed=31)
    gt2, x2 = torch.from_numpy(clean2.astype(np.float32)), torch.from_numpy(noisy2.astype(np.float32))
    gt32, x32 = gt2.repeat(16, 1, 1, 1), x2.repeat(16, 1, 1, 1)

    def run(gt, x, arith=1):
        m.set_weights(params, state)
        m.set_option("train_arith", arith)
        total, ml, dl, pred, grads = fns.train_step_single_gpu(gt, x, (1.0,), 0.0, None)
        return float(total.item()), grads.cpu().numpy().copy(), m.state.cpu().numpy().copy(), pred

    t_a, g_a, s_a, pred = run(gt32, x32)
    t_b, g_b, s_b, _ = run(gt32, x32)
    assert t_a == t_b and np.array_equal(g_a, g_b) and np.array_equal(s_a, s_b)                 # (i)
    assert np.isfinite(g_a).all() and np.isfinite(s_a).all() and tuple(pred.shape) == (32, 256, 256, 3)   # (ii)
    t_2, g_2, s_2, _ = run(gt2, x2)                                                               # (iii)
    assert abs(t_a - t_2) <= 1e-5 * abs(t_2)
    _cmp_grads(spec, g_a.astype(np.float64), g_2.astype(np.float64), rel=2e-4)
    # moving variance: Bessel factor n / (n - 1) differs between n = 2*65536 and n = 32*65536 by 7e-6 relative
    assert np.abs(s_a - s_2).max() < 1e-4
    t_x, g_x, s_x, _ = run(gt32, x32, arith=0)                                                    # (iv)
    assert abs(t_a - t_x) <= 1e-5 * abs(t_x)
    _cmp_grads(spec, g_a.astype(np.float64), g_x.astype(np.float64), rel=6e-4)
    m.set_weights(params, state)                                                                   # (v)
    m.set_option("train_arith", 1)
    opt, _ = bf.optimizer_builder(cfg["train"]["optimizer"])
    total, ml, dl, pred, grads = fns.train_step_single_gpu(gt32, x32, (1.0,), 0.0, None)
    fns.apply_grads(opt, grads, None)
    after = m.params.cpu().numpy()
    assert np.isfinite(after).all()
    for name, (o, s) in spec.offsets().items():
        n = int(np.prod(s))
        assert np.abs(after[o:o + n] - params[o:o + n]).max() > 0, f"{name} did not move"


def test_native_collective_single_rank():
    """bf_comm_* / bf_allreduce_grads (RCCL bound inside the library) with a world of one: the rendezvous, the
    communicator and the in-place all-reduce run on this GPU and leave the buffer as it was."""
    import ctypes as C
    L = N.lib()
    uid = C.create_string_buffer(128)
    N.check(L.bf_comm_unique_id(uid), None, "bf_comm_unique_id")
    comm = C.c_void_p()
    rc = L.bf_comm_init_rank(C.byref(comm), 1, 0, uid.raw)
    assert rc == N.BF_OK, L.bf_comm_last_error()
    g = torch.arange(84272, dtype=torch.float32, device="cuda") * 0.25
    ref = g.clone()
    rc = L.bf_allreduce_grads(None, N.ptr(g), g.numel(), comm, N.stream_ptr(g))
    assert rc == N.BF_OK, L.bf_comm_last_error()
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    assert L.bf_comm_destroy(comm) == N.BF_OK
    assert L.bf_allreduce_grads(None, N.ptr(g), g.numel(), None, N.stream_ptr(g)) == N.BF_EINVAL


def _dp_rank(rank, world, port, native, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=3 + rank, nontrivial_bn=True)      # different on purpose: broadcast must fix it
    model = bf.model_builder(cfg["model"], device=f"cuda:{rank}").hydra
    model.set_weights(params, state)
    opt, _ = bf.optimizer_builder(cfg["train"]["optimizer"])
    trainer = bf.DataParallelTrainer(model, bf.loss_function_builder(cfg["loss"]), opt, native_collective=native)
    trainer.broadcast_parameters()
    clean, noisy = O.synthetic_batch(4, 32, 32, seed=9)
    gt = bf.shard_batch(torch.from_numpy(clean.astype(np.float32)), rank, world).cuda()
    x = bf.shard_batch(torch.from_numpy(noisy.astype(np.float32)), rank, world).cuda()
    hook_ran = []
    for _ in range(2):
        trainer.step(gt, x, overlap=lambda: hook_ran.append(1))
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{int(native)}_{rank}.npy"), model.params.cpu().numpy())
    assert len(hook_ran) == 2
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("native", [False, True], ids=["torch_distributed", "c_abi_rccl"])
def test_data_parallel_trainer_two_gpus(native, tmp_path):
    """DataParallelTrainer with world 2 on two GPUs over RCCL: replicas identical after two steps and equal to the
    one-process emulation (both shards on one GPU, gradients summed, grad_scale 1/2)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_dp_rank, args=(2, port, native, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = (np.load(tmp_path / f"params_{int(native)}_{r}.npy") for r in (0, 1))
    assert np.array_equal(p0, p1)
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=3, nontrivial_bn=True)
    clean, noisy = O.synthetic_batch(4, 32, 32, seed=9)
    reps = []
    for r in range(2):
        mdl = bf.model_builder(cfg["model"], device="cuda").hydra
        mdl.set_weights(params, state)
        reps.append((mdl, bf.build_train_functions(mdl, bf.loss_function_builder(cfg["loss"])), bf.optimizer_builder(cfg["train"]["optimizer"])[0]))
    for _ in range(2):
        gs = []
        for r, (mdl, fns, opt) in enumerate(reps):
            gt = torch.from_numpy(clean[2 * r:2 * r + 2].astype(np.float32))
            x = torch.from_numpy(noisy[2 * r:2 * r + 2].astype(np.float32))
            gs.append(fns.train_step_single_gpu(gt, x)[4].clone())
        summed = gs[0] + gs[1]
        for mdl, fns, opt in reps:
            fns.apply_grads(opt, summed.clone(), None, grad_scale=0.5)
    assert np.abs(reps[0][0].params.cpu().numpy() - p0).max() <= 1e-7


# ---- blocks of one and three convolutions (backbone_resnet.py:111-113, backbone_blocks.py:174-246) -------------------

@pytest.mark.parametrize("train_arith", [1, 0], ids=["f16x3", "f32"])
@pytest.mark.parametrize("use_bn", [True, False], ids=["bn", "nobn"])
@pytest.mark.parametrize("nb,no_layers,shape", [(1, 2, (2, 24, 32)), (3, 1, (2, 16, 32)), (3, 3, (3, 33, 47)), (2, 2, (2, 24, 32))])
def test_train_step_block_variants_match_oracle(nb, no_layers, shape, use_bn, train_arith):
    """block_kernels of length 1, 2 and 3: first convolution without BatchNorm, BatchNorm on the second and third, the
    activation on every convolution but the last, skip Add -- training forward (batch statistics, moving statistics of
    every BatchNorm), loss and every gradient against the oracle."""
    cfg = O.canonical_config(no_layers=no_layers)
    cfg["model"]["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb, use_bn=use_bn)
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=nb * 10 + no_layers, nontrivial_bn=True)
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(params, state)
    m.set_option("train_arith", train_arith)
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    clean, noisy = O.synthetic_batch(*shape, seed=7 + nb)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    r_total, r_ml, r_dl, r_pred, r_grads, r_state = O.train_step_single_gpu(
        spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
    assert abs(total.item() - r_total) <= 1e-5 * abs(r_total)
    assert abs(ml["regularization_loss"].item() - r_ml["regularization_loss"]) <= 1e-5 * r_ml["regularization_loss"]
    assert np.abs(pred.cpu().numpy() - r_pred).max() < 0.02
    _cmp_grads(spec, grads.cpu().numpy().astype(np.float64), r_grads)
    if state.size:
        assert np.abs(m.state.cpu().numpy() - r_state).max() < 1e-5
    # and one optimizer step on top (per-tensor layout of the flat vector differs per variant)
    opt, _ = bf.optimizer_builder(cfg["train"]["optimizer"])
    fns.apply_grads(opt, grads, None)
    p1, _, _ = O.adam_step(params.astype(np.float64), r_grads, np.zeros(params.size), np.zeros(params.size), 0,
                           opt.lr() if opt.iterations == 0 else 1e-3, global_clipnorm=1.0)
    assert np.abs(m.params.cpu().numpy() - p1).max() < 5e-6


def test_monitoring_losses_match_oracle_through_the_c_abi():
    """loss_function_builder(...)["denoiser"](gt, prediction) (bfcnn/loss.py:190-247) on arbitrary GPU batches: the HIP loss kernels."""
    rng = np.random.default_rng(0)
    gt, pr = rng.uniform(0, 255, (2, 24, 20, 3)), rng.uniform(0, 255, (2, 24, 20, 3))
    for cfg in (O.canonical_config()["loss"], {"mse_multiplier": 0.5, "hinge": 3.5}):        # ssim_multiplier defaults to 1.0
        got = bf.loss_function_builder(cfg)["denoiser"](torch.from_numpy(gt.astype(np.float32)).cuda(), torch.from_numpy(pr.astype(np.float32)).cuda())
        ref = O.denoiser_loss(O.LossSpec.from_config(cfg), gt, pr)
        for k in ("total_loss", "mae_loss", "mse_loss", "ssim_loss"):
            assert abs(float(got[k]) - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), (k, float(got[k]), ref[k])


def test_model_losses_and_model_loss_match_oracle():
    """keras model.losses / loss_fn_map["model"](model) (bfcnn/loss.py:181-187): the regulariser kernel, not torch arithmetic."""
    cfg, spec, ls, params, state, m, fns = _setup(2)
    got = bf.loss_function_builder(cfg["loss"])["model"](m)
    ref = O.model_loss(spec, ls, params, np.float64)
    assert abs(float(got["regularization_loss"]) - ref["regularization_loss"]) <= 1e-5 * ref["regularization_loss"]
    assert abs(float(got["total_loss"]) - ref["total_loss"]) <= 1e-5 * ref["total_loss"]
    assert len(m.losses) == 1 + 2 * 2 + 2                          # base, two kernels per block, two head kernels


def test_the_engine_learns_to_denoise(tmp_path):
    """end to end through the public API, as a user of the reference would run it: an image directory -> dataset_builder (random crops,
    flips, additive noise on the device) -> train_loop (canonical resnet 1x6 from its glorot initialisation, L1 loss, Adam, checkpoints)
    -> load_model -> uint8 denoiser, judged like the reference's test_pretrained.py on a part of the image the training never saw:
    PSNR and MAE better than the noisy input's (measured: 22.3 -> 30.5 dB; tools/exp/learn_to_denoise.py is the same run as a script)."""
    import pathlib
    from PIL import Image
    lena = Image.open(pathlib.Path(__file__).parent / "golden" / "lena.jpg")
    (tmp_path / "img").mkdir()
    lena.crop((0, 0, 512, 384)).save(tmp_path / "img" / "train.png")
    held = np.asarray(lena.convert("RGB"))[384:512, 0:512][None]
    cfg = O.canonical_config(no_layers=6)
    cfg["train"].update({"epochs": 30, "gpu_batches_per_step": 1, "seed": 7})      # 1 920 steps of 16 crops: about 6 s
    cfg["train"]["optimizer"]["schedule"]["config"]["learning_rate"] = 2e-3
    cfg["loss"] = {"hinge": 0.0, "cutoff": 255.0, "mae_multiplier": 1.0, "ssim_multiplier": 0.0, "regularization": 0.01}
    cfg["dataset"] = {"batch_size": 16, "color_mode": "rgb", "no_crops_per_image": 64 * 16, "value_range": [0, 255], "clip_value": True,
                      "round_values": True, "random_up_down": True, "random_left_right": True, "input_shape": [64, 64, 3],
                      "additional_noise": [20, 20.0001], "inputs": [{"directory": str(tmp_path / "img")}]}
    model, hist = bf.train_loop(cfg, str(tmp_path / "run"))
    assert len(hist) == 1920 and np.isfinite(hist).all() and np.mean(hist[-20:]) < 0.1 * hist[0]
    noisy = np.clip(np.round(held + np.random.default_rng(0).normal(0, 20.0, held.shape)), 0, 255).astype(np.uint8)
    den = bf.load_model(str(tmp_path / "run" / "final"))(noisy)
    psnr = lambda a, b: 10 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    mae = lambda a, b: np.abs(a.astype(np.float64) - b.astype(np.float64)).mean()
    # (unseeded runs of tools/exp/learn_to_denoise.py end between +3.9 and +8.2 dB; the seed in the train section pins this one)
    assert psnr(held, den) > psnr(held, noisy) + 3.0, (psnr(held, noisy), psnr(held, den))
    assert mae(held, den) < 0.75 * mae(held, noisy)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BF_SWEEP_N", 24))))     # BF_SWEEP_N=300: a longer hunt
def test_random_training_configurations_and_options_match_oracle(seed):
    """a seeded sweep over the engine's training path: depth, base kernel size, 1 / 2 / 3 convolutions per block, BatchNorm on / off, the
    three loss terms and the hinge, ragged shapes, and every set_option switch of the training step (arithmetic of the convolutions,
    fused forward / backward kernels, double-buffered backward, tile order): loss, prediction, every gradient tensor and the moving
    statistics against the oracle"""
    rng = np.random.default_rng(11000 + seed)
    nb = int(rng.choice([1, 2, 2, 2, 3]))
    cfg = O.canonical_config(no_layers=int(rng.integers(1, 5)), kernel_size=int(rng.choice([1, 3, 5, 7])))
    cfg["model"]["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb, use_bn=bool(rng.random() < 0.8))
    cfg["loss"].update({"hinge": float(rng.choice([0.0, 0.5, 3.5])), "mse_multiplier": float(rng.choice([0.0, 0.5])),
                        "ssim_multiplier": float(rng.choice([0.0, 1.0])), "regularization": 0.01})
    spec = O.ResnetSpec.from_config(cfg["model"])
    ls = O.LossSpec.from_config(cfg["loss"])
    params, state = O.init_params(spec, seed=seed, nontrivial_bn=True)
    for name, (o, s) in spec.offsets().items():              # keep tanh(2x) * 0.51 inside the denormaliser's clip (see the config-4 test)
        if name.startswith("head"):
            params[o:o + int(np.prod(s))] *= 0.3
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(params, state)
    try:
        fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    except NotImplementedError as e:
        pytest.skip(f"refused: {e}")
    opts = {"train_arith": int(rng.random() < 0.7), "train_fused_fwd": int(rng.integers(2)), "train_fused_bwd": int(rng.integers(2)), "train_fused_bwd2": int(rng.integers(2)),
            "train_fwd_block": int(rng.integers(3)), "train_bwd_block": int(rng.integers(3)), "train_fold_finalize": int(rng.integers(2)), "train_bwd_dbuf": int(rng.random() < 0.3), "train_zigzag": int(rng.integers(2))}
    for k, v in opts.items():
        m.set_option(k, v)
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(8, 60)), int(rng.integers(8, 70))
    # a ReLU / hinge input within rounding of its kink on one pixel moves a gradient tensor past the bar (about 1 % of the
    # configurations here).  That is CHECKED, not forgiven: on a mismatch the oracle must name the elements that sit on a kink, and
    # the result must equal the oracle with those gates on the other side (helpers.compare_or_explain_by_ties)
    clean, noisy = O.synthetic_batch(B, H, W, seed=seed)
    gt, x = clean.astype(np.float32), noisy.astype(np.float32)
    m.set_weights(params, state)
    try:
        total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    except NotImplementedError as e:
        pytest.skip(f"refused: {e}")
    got_grads, got_pred, got_state = grads.cpu().numpy().astype(np.float64), pred.cpu().numpy(), m.state.cpu().numpy()
    gt64, x64 = gt.astype(np.float64), x.astype(np.float64)

    def compare(ref):
        r_total, r_ml, r_dl, r_pred, r_grads, r_state = ref
        assert abs(total.item() - r_total) <= 2e-5 * abs(r_total), (opts, total.item(), r_total)
        assert np.abs(got_pred - r_pred).max() < 0.02
        _cmp_grads(spec, got_grads, r_grads, rel=6e-4)
        if state.size:
            assert np.abs(got_state - r_state).max() < 1e-5

    compare_or_explain_by_ties(compare, lambda flips: O.train_step_single_gpu(spec, ls, params, state, gt64, x64, flips=flips),
                               lambda: O.training_step_ties(spec, ls, params, state, gt64, x64))
