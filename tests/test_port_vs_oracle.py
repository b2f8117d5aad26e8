"""The C port that bench.py times as `cpu_baseline` must compute the same function as the oracle."""
import numpy as np
import pytest

from oracle import bfcnn_oracle as O
from oracle import port


@pytest.mark.parametrize("no_layers,hw,k", [(2, (32, 32), 3), (3, (20, 45), 3), (1, (16, 16), 5)])
def test_port_matches_oracle(no_layers, hw, k):
    spec = O.ResnetSpec.from_config(O.canonical_config(no_layers=no_layers, kernel_size=k)["model"])
    params, state = O.init_params(spec, seed=no_layers)
    _, noisy = O.synthetic_batch(2, hw[0], hw[1], seed=3)
    got = port.forward_u8(spec, params, state, noisy, port.lib(rebuild=True))
    ref = O.denoiser_module_call(spec, params, state, noisy)
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01
