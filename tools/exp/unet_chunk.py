"""Does running the unet_laplacian graph on batch slices (intermediates stay in the 256 MB Infinity Cache between
producer and consumer kernels) beat one pass over the whole batch?  Usage: python tools/exp/unet_chunk.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O, unet_oracle as U

cfg = U.canonical_config()
spec = U.UnetLaplacianSpec.from_config(cfg["model"])
model = bf.model_builder(cfg["model"], device="cuda").hydra
model.set_weights(U.init_params(spec, seed=42))
module = bf.DenoiserModule(model)
_, base = O.synthetic_batch(4, 512, 512, seed=1)
noisy = torch.from_numpy(np.concatenate([base] * 8)).cuda()
for chunk in (32, 16, 8, 4, 2, 1):
    def step():
        return [module(noisy[i:i + chunk]) for i in range(0, 32, chunk)]
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"chunk {chunk:3d}: {dt * 1e3:7.2f} ms per 32 images = {32 / dt:7.0f} images/s", flush=True)
