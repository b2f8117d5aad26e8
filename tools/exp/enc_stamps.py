"""Per-phase cycles of uh_enc32u_kernel from a UH_ENC_STAMP=1 build (BFCNN_HIP_LIB=lib/variants/libbfcnn_hip_UH_ENC_STAMP1.so):
producer waves 0-3: [0] row walk of a batch, [1] step barrier; consumer waves 4-7: [0] skip request of the next batch,
[1] staging read + split, [2] the two GEMMs, [3] output arithmetic + stores, [4] step barrier.  Read the SHARES: stamps add fences."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from blind_image_denoising_amd import unet_laplacian as UL, _native as N
B, S, Cc, k = 32, 512, 32, int(os.environ.get("ENC_K", 5))
x = torch.randn((B, S, S, Cc), device="cuda")
dw = torch.randn((k, k, Cc), device="cuda") * 0.2
g = torch.rand(Cc, device="cuda") + 0.5
w1, w2 = torch.randn((Cc, 4 * Cc), device="cuda") / Cc ** 0.5, torch.randn((4 * Cc, Cc), device="cuda") / (4 * Cc) ** 0.5
pk = UL.pack_mlp_h3(w1, w2)
mult = torch.rand(Cc, device="cuda")
f = lambda: UL.convnext_block_h3(x, dw, g, pk, mult, "leaky_relu_01")
for _ in range(3):
    f()
torch.cuda.synchronize()
L = N.lib()
L.bf_debug_enc_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.bf_debug_enc_stamps(None, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
buf = np.zeros(256 * 12 * 8, dtype=np.uint64)
assert L.bf_debug_enc_stamps(buf.ctypes.data_as(C.c_void_p), 0) == 0
d = buf.reshape(256, 12, 8).astype(np.float64)
us = e0.elapsed_time(e1) * 1e3
print(f"launch {us:.0f} us; s_memtime ticks per wave (100 MHz ticks x clock ratio): producers {d[:, :4].sum(axis=2).mean():.0f}, consumers {d[:, 4:8].sum(axis=2).mean():.0f}")
nsteps = 2 * (B * (S // 16) * (S // 32)) / 256
for name, waves, phases in (("producer", slice(0, 4), ["row walk", "barrier"]),
                            ("consumer", slice(4, 8), ["skip request", "staging read + split", "GEMMs", "scale + add + store", "barrier"])):
    dd = d[:, waves].mean(axis=(0, 1))
    tot = dd.sum()
    print(f"{name}: " + ", ".join(f"{n} {dd[i] / nsteps:.0f} ({100 * dd[i] / tot:.0f} %)" for i, n in enumerate(phases)) + f"; per step {tot / nsteps:.0f}")
