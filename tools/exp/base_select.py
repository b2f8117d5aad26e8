"""resnet 1x6 DenoiserModule time per call with the row-streaming base convolution (base_rows 1) and the vector kernel (0)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
cfg = O.canonical_config(no_layers=6)
spec = O.ResnetSpec.from_config(cfg["model"])
params, state = O.init_params(spec, seed=42, nontrivial_bn=True)
m = bf.model_builder(cfg["model"], device="cuda").hydra
m.set_weights(params, state)
mod = bf.DenoiserModule(m)
def t(x, n=40):
    for _ in range(5): mod(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): mod(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (B, H, W) in [(1, 128, 128), (1, 256, 256), (2, 256, 256), (4, 256, 256), (8, 256, 256), (16, 256, 256), (32, 256, 256), (64, 256, 256), (1, 512, 512), (1, 1024, 1024), (4, 1024, 1024), (64, 64, 64)]:
    x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    t(x, 5)
    r = {}
    for v in (1, 0, 1, 0):
        m.set_option("base_rows", v)
        r.setdefault(v, []).append(t(x))
    m.set_option("base_rows", 1)
    print(f"{B:3d} x {H} x {W}: rows {min(r[1]):8.1f} us   vector {min(r[0]):8.1f} us   rows of chunks {B * H * ((W + 255) // 256)}", flush=True)
