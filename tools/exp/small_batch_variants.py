"""Small-batch latency of the resnet 1x18 module per tile-kernel variant (1: 16 x 32 tiles, 2: 16 x 16 tiles, two workgroups per CU)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
cfg = O.canonical_config(no_layers=18)
spec = O.ResnetSpec.from_config(cfg["model"])
params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
m = bf.model_builder(cfg["model"], device="cuda").hydra
m.set_weights(params, state)
mod = bf.DenoiserModule(m)
SHAPES = [(1, 256, 256)] if os.environ.get("ONLY_ONE") else [(1, 128, 128), (1, 256, 256), (2, 256, 256), (3, 256, 256), (4, 256, 256), (6, 256, 256), (8, 256, 256), (1, 512, 512), (1, 1024, 1024), (4, 512, 512)]
for (B, H, W) in SHAPES:
    _, img = O.synthetic_batch(B, H, W, seed=1)
    x = torch.from_numpy(img).cuda()
    res = {}
    for v in (-1, 1, 2):
        m.set_option("h3_variant", v)
        for _ in range(5): mod(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): mod(x)
        torch.cuda.synchronize(); res[v] = (time.perf_counter() - t0) / 50 * 1e6
    print(f"{B} x {H} x {W}: default {res[-1]:8.1f} us  variant 1 {res[1]:8.1f}  variant 2 {res[2]:8.1f}   tiles(16x32) {B * ((H + 15) // 16) * ((W + 31) // 32)}", flush=True)
