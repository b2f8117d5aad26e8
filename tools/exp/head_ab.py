"""A/B of the head folded into the last block (option fused_head) on the bench workload."""
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
_, base = O.synthetic_batch(16, 256, 256, seed=1)
x = torch.from_numpy(np.concatenate([base] * 8)).cuda()
outs = {}
for rep in range(2):
    for v in (1, 0):
        m.set_option("fused_head", v)
        for _ in range(5):
            y = mod(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            y = mod(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        outs[v] = y.cpu().numpy()
        print(f"fused_head={v}: {dt * 1e3:.3f} ms per batch of 128 = {128 / dt:.0f} images/s", flush=True)
d = np.abs(outs[0].astype(int) - outs[1].astype(int))
print("max LSB difference between the two paths:", d.max(), "fraction differing:", (d > 0).mean())
