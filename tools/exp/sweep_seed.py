"""Replays one seed of tests/test_gpu_training.py::test_random_training_configurations_and_options_match_oracle and prints, per gradient
tensor, the error against the oracle for train_fused_bwd2 = 0 / 1 (and of the two against each other)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(11000 + seed)
nb = int(rng.choice([1, 2, 2, 2, 3]))
cfg = O.canonical_config(no_layers=int(rng.integers(1, 5)), kernel_size=int(rng.choice([1, 3, 5, 7])))
cfg["model"]["backbone"].update(block_kernels=[3] * nb, block_filters=[16] * nb, use_bn=bool(rng.random() < 0.8))
cfg["loss"].update({"hinge": float(rng.choice([0.0, 0.5, 3.5])), "mse_multiplier": float(rng.choice([0.0, 0.5])),
                    "ssim_multiplier": float(rng.choice([0.0, 1.0])), "regularization": 0.01})
spec = O.ResnetSpec.from_config(cfg["model"]); ls = O.LossSpec.from_config(cfg["loss"])
params, state = O.init_params(spec, seed=seed, nontrivial_bn=True)
for name, (o, s) in spec.offsets().items():
    if name.startswith("head"):
        params[o:o + int(np.prod(s))] *= 0.3
m = bf.model_builder(cfg["model"], device="cuda").hydra
m.set_weights(params, state)
fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
opts = {"train_arith": int(rng.random() < 0.7), "train_fused_fwd": int(rng.integers(2)), "train_fused_bwd": int(rng.integers(2)), "train_fused_bwd2": int(rng.integers(2)),
        "train_bwd_dbuf": int(rng.random() < 0.3), "train_zigzag": int(rng.integers(2))}
B, H, W = int(rng.integers(1, 4)), int(rng.integers(8, 60)), int(rng.integers(8, 70))
print("nb", nb, "cfg", cfg["model"]["backbone"], cfg["loss"], "opts", opts, "shape", (B, H, W))
clean, noisy = O.synthetic_batch(B, H, W, seed=seed)
gt, x = clean.astype(np.float32), noisy.astype(np.float32)
ref = O.train_step_single_gpu(spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
ties = O.training_step_ties(spec, ls, params, state, gt.astype(np.float64), x.astype(np.float64))
print("ties:", len(ties), ties[:20])
got = {}
for v in (0, 1):
    for k, val in opts.items():
        m.set_option(k, val)
    m.set_option("train_fused_bwd2", v)
    m.set_weights(params, state)
    total, ml, dl, pred, grads = fns.train_step_single_gpu(torch.from_numpy(gt), torch.from_numpy(x), (1.0,), 0.0, None)
    got[v] = grads.cpu().numpy().astype(np.float64)
    print(f"bwd2={v}: total {total.item():.9g} oracle {ref[0]:.9g}")
for name, (o, s) in spec.offsets().items():
    n = int(np.prod(s)); r = ref[4][o:o + n]; sc = max(np.abs(r).max(), 1e-6)
    e0, e1, e01 = np.abs(got[0][o:o + n] - r).max() / sc, np.abs(got[1][o:o + n] - r).max() / sc, np.abs(got[0][o:o + n] - got[1][o:o + n]).max() / sc
    print(f"{name:28s} rel err bwd2=0 {e0:.2e}  bwd2=1 {e1:.2e}  0 vs 1 {e01:.2e}")
