"""resnet 1x18 DenoiserModule time per call over batch / size regimes: library default against the tile kernel (h3_pair 0 + h3_variant 1 or 2)
and against two blocks per launch forced wherever it can run (h3_pair 2)."""
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
if len(sys.argv) > 1 and sys.argv[1] == "odd":
    shapes_override = [(512, 8, 256), (256, 16, 256), (128, 32, 256), (64, 32, 512), (4, 2048, 128), (1, 4096, 256), (1, 16, 40000), (2000, 4, 64), (300, 20, 100), (8, 300, 300), (5, 700, 130)]
else:
    shapes_override = None
shapes = [(2, 256, 256), (4, 256, 256), (6, 256, 256), (8, 256, 256), (10, 256, 256), (12, 256, 256), (16, 256, 256), (24, 256, 256),
          (1, 512, 512), (2, 512, 512), (3, 512, 512), (4, 512, 512), (8, 512, 512), (1, 1024, 1024), (2, 1024, 1024), (1, 1080, 1920),
          (4, 128, 128), (16, 128, 128), (32, 128, 128), (64, 128, 128), (1, 2048, 2048)]
if shapes_override: shapes = shapes_override
def t(x, n=30):
    for _ in range(4): mod(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): mod(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (B, H, W) in shapes:
    img = np.random.default_rng(1).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    res = {}
    t(x, 5)
    for name, opts in (("default", {"h3_pair": 1, "h3_variant": -1}), ("tiles32", {"h3_pair": 0, "h3_variant": 1}), ("tiles16", {"h3_pair": 0, "h3_variant": 2}),
                       ("pairs", {"h3_pair": 2, "h3_variant": -1}), ("stream1", {"h3_pair": 0, "h3_variant": 4})):
        for k, v in opts.items(): m.set_option(k, v)
        res[name] = t(x)
        if name == "default": kern = m.block_kernel()
    best = min(res, key=res.get)
    print(f"{B:3d} x {H} x {W}: " + "  ".join(f"{k} {v:8.1f}" for k, v in res.items()) + f"   default = {kern[0].replace('fused_block', '')} x{kern[1]}  best = {best}" + ("" if res['default'] <= 1.03 * res[best] else "   <<<"), flush=True)
