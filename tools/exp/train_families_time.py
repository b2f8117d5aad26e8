"""Step times of the two operator-walk training graphs (parity rows, not benchmark configurations): unet_laplacian v5 and the
shipped bottleneck resnet, one GPU, synthetic batches."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
from oracle import resnet_generic_oracle as R
from oracle import unet_oracle as U

LOSS = {"hinge": 3.5, "cutoff": 255.0, "mae_multiplier": 1.0, "mse_multiplier": 0.5, "ssim_multiplier": 1.0, "regularization": 0.01}


def run(name, cfg, B, S, depth_weights=None):
    model = bf.model_builder(cfg, device="cuda").hydra
    fns = bf.build_train_functions(model, bf.loss_function_builder(LOSS))
    opt, _ = bf.optimizer_builder({"type": "Adam", "gradient_clipping_by_norm_local": 1.0,
                                   "schedule": {"type": "exponential_decay", "config": {"decay_rate": 0.9, "decay_steps": 1000, "learning_rate": 1e-4}}})
    clean, noisy = O.synthetic_batch(min(B, 4), S, S, seed=3)
    reps = (B + clean.shape[0] - 1) // clean.shape[0]
    gt = torch.from_numpy(np.concatenate([clean] * reps)[:B].astype(np.float32)).cuda()
    x = torch.from_numpy(np.concatenate([noisy] * reps)[:B].astype(np.float32)).cuda()
    step = lambda: fns.apply_grads(opt, fns.train_step_single_gpu(gt, x, depth_weights)[4], None)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: batch {B} {S}x{S}: {dt * 1e3:.1f} ms per step = {B / dt:.0f} images/s")


ucfg = U.canonical_config()
ucfg = ucfg["model"] if "model" in ucfg else ucfg
run("unet_laplacian v5 (depth 3, width 3, 32/64/128)", ucfg, 8, 256, [1.0, 0.5, 0.25])
run("resnet bottleneck 1x6 (shipped config)", R.shipped_config(), 16, 128)
