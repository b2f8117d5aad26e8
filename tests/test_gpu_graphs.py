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
