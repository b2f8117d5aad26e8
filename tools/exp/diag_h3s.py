import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import blind_image_denoising_amd as bf
from oracle import bfcnn_oracle as O
from helpers import fused_block_h3_gpu
m = bf.model_builder(O.canonical_config(no_layers=0)["model"], device="cuda").hydra
m.set_option("h3_variant", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
rng = np.random.default_rng(15)
B, H, W = 1, 64, 96
x = rng.standard_normal((B, H, W, 16)).astype(np.float32)
w1 = (rng.standard_normal((3, 3, 16, 16)) * 0.1).astype(np.float32)
w2 = (rng.standard_normal((3, 3, 16, 16)) * 0.1).astype(np.float32)
sc, sh = np.ones(16, np.float32), np.zeros(16, np.float32)
t = np.maximum(O.conv2d_same(x.astype(np.float64), w1.astype(np.float64)), 0)
ref = x + O.conv2d_same(t, w2.astype(np.float64))
got = fused_block_h3_gpu(x, w1, w2, sc, sh, 1)
err = np.abs(got - ref)
bad = err > 1e-3
print("bad elements:", bad.sum(), "of", bad.size)
ys, xs, cs = np.where(bad[0])
print("rows:", sorted(set(ys.tolist())))
print("cols:", sorted(set(xs.tolist())))
print("channels:", sorted(set(cs.tolist())))
