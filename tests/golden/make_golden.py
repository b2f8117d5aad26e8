"""
Generates the committed golden fixtures under tests/golden/ from the fp64 NumPy oracle
(oracle/bfcnn_oracle.py), after the oracle has been cross-checked against PyTorch-CPU fp64
(tests/test_oracle_vs_torch.py must pass first; this script re-asserts the key agreements).

The reference itself cannot be run in the build image (TensorFlow is absent), so these are
oracle-generated vectors, not reference outputs: conv / BN / head / loss / Adam parity stays
"unpinned by the reference" (see oracle header); the fixtures freeze the restatement so that a
later edit of the oracle cannot silently move the target.

lena.jpg is the reference's own test fixture (images/test/etc/lena.jpg), copied as data for the
pyramid round-trip test (tests/bfcnn/test_pyramid.py).

Run:  python tests/golden/make_golden.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import bfcnn_oracle as O   # noqa: E402

OUT = pathlib.Path(__file__).resolve().parent


FULL_LOSS = {"hinge": 3.5, "cutoff": 255.0, "mae_multiplier": 1.0, "mse_multiplier": 0.5, "ssim_multiplier": 1.0,
             "regularization": 0.01}


def full_loss_step(spec, params, state, clean_t, noisy_t):
    ls = O.LossSpec.from_config(FULL_LOSS)
    total, ml, dl, pred, grads, _ = O.train_step_single_gpu(spec, ls, params, state, clean_t.astype(np.float64),
                                                             noisy_t.astype(np.float64), depth_weight=0.8)
    np.savez_compressed(OUT / "train_step_full_loss.npz", total=total, denoiser_total=dl[0]["total_loss"],
                        ssim=dl[0]["ssim_loss"], mae=dl[0]["mae_loss"], mse=dl[0]["mse_loss"], grads=grads)


def main():
    # (1) single 3x3 C16 SAME convolution, 5 seeds
    d = {}
    for seed in range(5):
        rng = np.random.default_rng(100 + seed)
        x = rng.standard_normal((1, 16, 24, 16)).astype(np.float32)
        w = (rng.standard_normal((3, 3, 16, 16)) * 0.1).astype(np.float32)
        d[f"x{seed}"], d[f"w{seed}"] = x, w
        d[f"y{seed}"] = O.conv2d_same(x.astype(np.float64), w.astype(np.float64))
    np.savez_compressed(OUT / "conv3x3_c16.npz", **d)

    # (2)+(3) 2-block net at 64x64 (and a ragged 40x50 that exercises pad_to_power_of_2)
    cfg = O.canonical_config(no_layers=2)
    spec = O.ResnetSpec.from_config(cfg["model"])
    params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
    clean, noisy = O.synthetic_batch(2, 64, 64, seed=1234)
    _, ragged = O.synthetic_batch(1, 40, 50, seed=99)
    hydra_f32 = O.hydra_forward(spec, params, state, noisy.astype(np.float64))
    np.savez_compressed(
        OUT / "net_2blocks.npz", params=params, state=state, clean=clean, noisy=noisy, ragged=ragged,
        hydra_f32=hydra_f32, out_u8=O.denoiser_module_call(spec, params, state, noisy),
        ragged_u8=O.denoiser_module_call(spec, params, state, ragged))

    # (4) one training step + one Adam step on a 2-block net, 2x24x32
    ls = O.LossSpec.from_config(cfg["loss"])
    clean_t, noisy_t = O.synthetic_batch(2, 24, 32, seed=7)
    total, ml, dl, pred, grads, new_state = O.train_step_single_gpu(
        spec, ls, params, state, clean_t.astype(np.float64), noisy_t.astype(np.float64))
    p1, m1, v1 = O.adam_step(params.astype(np.float64), grads, np.zeros_like(grads), np.zeros_like(grads), 0,
                             O.exponential_decay(1e-3, 40000, 0.9, 0), global_clipnorm=1.0)
    np.savez_compressed(
        OUT / "train_step.npz", clean=clean_t, noisy=noisy_t, total=total, reg=ml["regularization_loss"],
        mae=dl[0]["mae_loss"], mse=dl[0]["mse_loss"], denoiser_total=dl[0]["total_loss"], pred=pred, grads=grads,
        new_state=new_state, params_after=p1, m_after=m1, v_after=v1)

    # (4b) the same step with all three loss terms on, weights as the reference's shipped configs (loss.py:190-247)
    full_loss_step(spec, params, state, clean_t, noisy_t)

    # (5) pyramid / resampling
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 18, 22, 3))
    d = {"x": x, "up_bilinear": O.upsample_bilinear_2x(x), "up_nearest": O.upsample_nearest_2x(x),
         "slice2": O.strided_slice_2x(x), "pool_valid": O.avg_pool_valid_2x2(x)}
    for k in (2, 3, 5):
        d[f"pool{k}"] = O.avg_pool_same(x, (k, k), 2)
    np.savez_compressed(OUT / "pyramid.npz", **d)

    # (6) rounding edge cases of the uint8 store (tf.round = half-to-even)
    v = np.array([0.5, 1.5, 2.5, 126.5, 127.5, 253.5, 254.5, 0.49999997, 254.50002, -0.2, 255.4], np.float64)
    np.savez_compressed(OUT / "rounding.npz", v=v, r=np.clip(O.round_half_even(v), 0, 255).astype(np.uint8))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
