"""How far inside the parity bars the HIP path sits on the reference's trained network and a real frame:
normalised MAE of the three hydra outputs and the share of uint8 pixels off by one, both arithmetics.
Usage: python tools/exp/v56_margins.py"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import blind_image_denoising_amd as bf
from oracle import unet_oracle as U
import unet_v56 as V
z, cfg = V.load()
spec = U.UnetLaplacianSpec.from_config(cfg)
m = bf.model_builder(cfg, device="cuda").hydra
m.set_weights(z["params"])
params = np.asarray(z["params"])
clean = z["kitti"][:1, 32:160, 16:208]
noisy = V.corrupt(clean, 20.0, seed=3)
ref = U.hydra_forward(spec, params, noisy.astype(np.float64))
want = U.denoiser_module_call(spec, params, noisy)
for arith in (1, 0):
    m.set_option("arith", arith)
    got = m(noisy.astype(np.float32))
    maes = [float(np.abs(np.asarray(g, np.float64) - r).mean() / 255.0) for g, r in zip(got, ref)]
    den = bf.DenoiserModule(m)(noisy)
    d = np.abs(den.astype(np.int32) - want.astype(np.int32))
    print("arith", arith, "normalised MAE per scale", ["%.2e" % v for v in maes], "u8 max", d.max(), "frac off", float((d > 0).mean()))
