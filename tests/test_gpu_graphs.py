"""The C-ABI entry points make no allocation, synchronisation or host round trip, so a whole DenoiserModule call can be
captured into a HIP graph and replayed (include/bfcnn_hip.h: "graph-capturable"): small-batch inference is launch-bound
(20 launches for resnet 1x18, ~60 for unet_laplacian v5) and one graph launch replaces them."""
import numpy as np
import pytest
import torch

import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
from oracle import unet_oracle as U

pytestmark = pytest.mark.gpu


def _capture_and_check(module, a, b):
    static_in = torch.from_numpy(a).cuda()
    ref_a, ref_b = module(a), module(b)                 # direct calls (also warm-up: packing, workspace, kernel attributes)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_out = module(static_in)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(static_out.cpu().numpy(), ref_a)
    static_in.copy_(torch.from_numpy(b).cuda())
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(static_out.cpu().numpy(), ref_b)


def test_resnet_denoiser_module_in_a_hip_graph():
    cfg = O.canonical_config(no_layers=6)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=3)
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(params, state)
    _, a = O.synthetic_batch(1, 64, 64, seed=1)
    _, b = O.synthetic_batch(1, 64, 64, seed=2)
    _capture_and_check(bf.DenoiserModule(m), a, b)


def test_unet_laplacian_denoiser_module_in_a_hip_graph():
    cfg = U.canonical_config(depth=3, width=1)
    spec = U.UnetLaplacianSpec.from_config(cfg["model"])
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(U.init_params(spec, seed=4))
    _, a = O.synthetic_batch(1, 64, 64, seed=1)
    _, b = O.synthetic_batch(1, 64, 64, seed=2)
    _capture_and_check(bf.DenoiserModule(m), a, b)


def test_trained_unet_from_the_registry_in_a_hip_graph():
    """the packaged trained network (row attention, GELU MLPs, Gaussian split): same capture, real frames."""
    import unet_v56 as V
    z, _ = V.load()
    a = V.corrupt(z["kitti"][:1, :64, :128], 20.0, seed=1)
    b = V.corrupt(z["kitti"][1:2, 64:128, :128], 20.0, seed=2)
    _capture_and_check(bf.load_denoiser_model("unet_laplacian_v5.6"), a, b)


def test_train_step_with_all_loss_terms_in_a_hip_graph():
    """bf_train_step with the RMSE and SSIM terms on (two head passes + csrc/loss_terms.hip) captures and replays bitwise."""
    cfg = O.canonical_config(no_layers=2)
    cfg["loss"].update({"hinge": 3.5, "mse_multiplier": 0.5, "ssim_multiplier": 1.0})
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=6)
    m = bf.model_builder(cfg["model"], device="cuda").hydra
    m.set_weights(params, state)
    fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
    clean, noisy = O.synthetic_batch(2, 32, 32, seed=3)
    gt, x = torch.from_numpy(clean.astype(np.float32)).cuda(), torch.from_numpy(noisy.astype(np.float32)).cuda()
    ref = fns.train_step_single_gpu(gt, x)
    m.set_weights(params, state)                       # the step updated the BN moving statistics
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fns.train_step_single_gpu(gt, x)
    m.set_weights(params, state)
    g.replay()
    torch.cuda.synchronize()
    assert out[0].item() == ref[0].item()
    assert torch.equal(out[4], ref[4])


@pytest.mark.parametrize("family", ["resnet", "unet_laplacian"])
def test_graphed_denoiser_module_replays_per_shape(family):
    """GraphedDenoiserModule: one captured graph per input shape, results identical to the direct module on every call (device and
    host inputs, shapes revisited after others grew the engine's workspace, least-recently-used eviction)."""
    if family == "resnet":
        cfg = O.canonical_config(no_layers=6)
        spec = O.ResnetSpec.from_config(cfg["model"])
        params, state = O.init_params(spec, seed=3)
        m = bf.model_builder(cfg["model"], device="cuda").hydra
        m.set_weights(params, state)
    else:
        cfg = U.canonical_config(depth=3, width=1)
        spec = U.UnetLaplacianSpec.from_config(cfg["model"])
        m = bf.model_builder(cfg["model"], device="cuda").hydra
        m.set_weights(U.init_params(spec, seed=4))
    direct = bf.DenoiserModule(m)
    graphed = bf.GraphedDenoiserModule(bf.DenoiserModule(m), max_shapes=2)
    shapes = [(1, 32, 32), (2, 64, 48), (1, 32, 32), (3, 96, 128), (1, 32, 32), (2, 64, 48)]
    for k, (B, H, W) in enumerate(shapes):
        _, img = O.synthetic_batch(B, H, W, seed=20 + k)
        ref = direct(img)
        got = graphed(img)                                               # host array in, host array out
        assert isinstance(got, np.ndarray) and np.array_equal(got, ref), (k, B, H, W)
        dev = graphed(torch.from_numpy(img).cuda())                      # device tensor in, device tensor out
        assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), ref)
        assert graphed.check_status()
        assert len(graphed.captured_shapes()) <= 2
    assert graphed.captured_shapes()[-1] == (2, 64, 48, 3)
    graphed.invalidate()
    assert graphed.captured_shapes() == []
    with pytest.raises(ValueError):
        graphed(np.zeros((1, 8, 8, 3), np.float32))
    assert graphed(np.zeros((0, 8, 8, 3), np.uint8)).shape == (0, 8, 8, 3)


@pytest.mark.parametrize("family", ["resnet", "unet_laplacian"])
def test_graphed_denoiser_module_follows_new_weights_and_options(family):
    """A captured graph replays kernels that read the PACKED weights of the moment of capture: after set_weights / mark_dirty (a
    training or optimizer step) / set_option the graphed module must return what the direct module returns (the graph is keyed on
    the model's version and re-captured), never the old weights' output and never freed operands (round-3 advisor finding)."""
    if family == "resnet":
        cfg = O.canonical_config(no_layers=4)
        spec = O.ResnetSpec.from_config(cfg["model"])
        m = bf.model_builder(cfg["model"], device="cuda").hydra
        weights = [O.init_params(spec, seed=s) for s in (3, 11)]
        set_w = lambda k: m.set_weights(*weights[k])
    else:
        cfg = U.canonical_config(depth=3, width=1)
        spec = U.UnetLaplacianSpec.from_config(cfg["model"])
        m = bf.model_builder(cfg["model"], device="cuda").hydra
        weights = [U.init_params(spec, seed=s) for s in (4, 12)]
        set_w = lambda k: m.set_weights(weights[k])
    set_w(0)
    direct = bf.DenoiserModule(m)
    graphed = bf.GraphedDenoiserModule(bf.DenoiserModule(m))
    _, img = O.synthetic_batch(2, 64, 48, seed=5)
    dev = torch.from_numpy(img).cuda()
    first = graphed(dev).cpu().numpy()
    assert np.array_equal(first, direct(img))
    set_w(1)                                                             # new weights after the capture
    ref = direct(img)
    assert not np.array_equal(ref, first)
    assert np.array_equal(graphed(dev).cpu().numpy(), ref)
    assert np.array_equal(graphed(img), ref)
    with torch.no_grad():                                                # in-place change + mark_dirty, as an optimizer step does
        m.params.mul_(0.5)
    m.mark_dirty()
    ref2 = direct(img)
    assert not np.array_equal(ref2, ref)
    assert np.array_equal(graphed(dev).cpu().numpy(), ref2)
    m.set_option("arith", 0)                                             # another arithmetic: the old graph holds the other kernels
    v = m.version
    assert np.array_equal(graphed(dev).cpu().numpy(), direct(img))
    assert graphed._graphs[tuple(dev.shape)][3][0] == v
    assert len(graphed.captured_shapes()) == 1
