"""bf_train_step on a batch that repeats 2 images R times vs the 2 images alone (mathematically identical loss / gradients)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O

L = int(os.environ.get("LAYERS", 18))
cfg = O.canonical_config(no_layers=L)
spec = O.ResnetSpec.from_config(cfg["model"])
params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
for name, (o, s) in spec.offsets().items():
    if name.startswith("head"):
        params[o:o + int(np.prod(s))] *= 0.1
for S in (64, 256):
    clean2, noisy2 = O.synthetic_batch(2, S, S, seed=31)
    gt2, x2 = torch.from_numpy(clean2.astype(np.float32)), torch.from_numpy(noisy2.astype(np.float32))
    for arith in (1, 0):
        m = bf.model_builder(cfg["model"], device="cuda").hydra
        fns = bf.build_train_functions(m, bf.loss_function_builder(cfg["loss"]))
        base = None
        for R in (1, 2, 4, 8, 16):
            m.set_weights(params, state)
            m.set_option("train_arith", arith)
            total, ml, dl, pred, grads = fns.train_step_single_gpu(gt2.repeat(R, 1, 1, 1), x2.repeat(R, 1, 1, 1), (1.0,), 0.0, None)
            g = grads.cpu().numpy().astype(np.float64)
            st = m.state.cpu().numpy().astype(np.float64)
            if base is None:
                base = (float(total.item()), g, st, pred[:2].cpu().numpy())
                continue
            worst = max((np.abs(g[o:o + int(np.prod(s))] - base[1][o:o + int(np.prod(s))]).max() / max(np.abs(base[1][o:o + int(np.prod(s))]).max(), 1e-6), n)
                        for n, (o, s) in spec.offsets().items())
            print(f"S={S} arith={arith} B={2 * R:2d}: loss rel {abs(total.item() - base[0]) / abs(base[0]):.1e} worst grad {worst[0]:.1e} ({worst[1]}) "
                  f"state diff {np.abs(st - base[2]).max():.1e} pred diff {np.abs(pred[:2].cpu().numpy() - base[3]).max():.1e}", flush=True)
