"""Throughput of the reference's trained unet_laplacian_v5.6 network (tests/golden/unet_v56.npz) on the HIP path next to
the snapshot-builder graph of the same size: 32 frames of 512x512, DenoiserModule uint8 -> uint8.
Usage: python tools/exp/unet_v56_bench.py [steps]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O, unet_oracle as U

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
z = np.load(os.path.join(ROOT, "tests", "golden", "unet_v56.npz"))
cfg56 = json.loads(bytes(z["config"]).decode())
m56 = bf.model_builder(cfg56, device="cuda").hydra
m56.set_weights(z["params"])
cfg5 = U.canonical_config()["model"]
m5 = bf.model_builder(cfg5, device="cuda").hydra
m5.set_weights(U.init_params(U.UnetLaplacianSpec.from_config(cfg5), seed=42))
_, base = O.synthetic_batch(4, 512, 512, seed=1)
noisy = torch.from_numpy(np.concatenate([base] * 8)).cuda()
for name, m in (("v5.6 archive graph", m56), ("v5 snapshot graph", m5)):
    module = bf.DenoiserModule(m)
    for _ in range(2):
        module(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        module(noisy)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name:20s}: {dt * 1e3:7.2f} ms per 32 frames = {32 / dt:7.0f} images/s", flush=True)
