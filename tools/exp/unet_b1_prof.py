"""unet_laplacian v5, one 512 x 512 image through DenoiserModule, 60 calls (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import blind_image_denoising_amd as bf
from oracle import unet_oracle as U
B = int(os.environ.get("B", 1))
ucfg = U.canonical_config()
um = bf.model_builder(ucfg["model"], device="cuda").hydra
um.set_weights(U.init_params(U.UnetLaplacianSpec.from_config(ucfg["model"]), seed=42))
x = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (B, 512, 512, 3), dtype=np.uint8)).cuda()
mod = bf.DenoiserModule(um)
for _ in range(60):
    mod(x)
torch.cuda.synchronize()
