"""Runs the full-row streaming block of an H3V_ABLATE=32 build (BFCNN_HIP_LIB=...) on the bench shape and prints the
average cycles a wave spends per step in each phase (s_memtime stamps; read the SHARES, not the total: stamps add fences).
Role A (conv1, waves 0-3) and B (conv2, waves 4-7): compute | - | barrier.  Role C (memory, waves 8-11): issue of the
stores and DMA pieces | wait for the DMA of row s+1 | barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blind_image_denoising_amd import _native as N

B, H, W = int(os.environ.get("B", 128)), int(os.environ.get("H", 256)), 256
L = N.lib()
L.bf_debug_set_fused_dbg.argtypes = [C.c_void_p]
L.bf_debug_set_h3_variant(4)
x = torch.randn((B, H, W, 16), device="cuda")
out = torch.empty_like(x)
w1 = torch.randn((3, 3, 16, 16), device="cuda") * 0.1
w2 = torch.randn((3, 3, 16, 16), device="cuda") * 0.1
sc, sh = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
scratch = torch.zeros(int(L.bf_debug_fused_block_h3_scratch_floats(B, H, W)), device="cuda")
NWG, NW = 256, 12
dbg = torch.zeros(NWG * NW * 8, dtype=torch.int64, device="cuda")
L.bf_debug_set_fused_dbg(C.c_void_p(dbg.data_ptr()))
call = lambda: L.bf_debug_fused_block_h3(N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(scratch), B, H, W, 1, None)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
dbg.zero_()
assert call() == 0
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NWG, NW, 8).astype(np.float64)      # [block][wave][phase]
cyc = d[:, :, :3].sum(axis=2)
ticks = d[:, :, 3]
live = ticks > 0
print(f"workgroups with work: {(cyc.sum(axis=1) > 0).sum()} of {NWG}; cycles per wave: mean {cyc[live].mean():.0f} max {cyc.max():.0f}; "
      f"in-kernel clock {np.median(cyc[live] / ticks[live]) * 100:.0f} MHz (s_memtime / s_memrealtime)")
names = ["compute / memory issue", "wait DMA of row s+1", "barrier"]
for role, waves in (("A conv1", slice(0, 4)), ("B conv2", slice(4, 8)), ("C loaders", slice(8, 10)), ("C storers", slice(10, 12))):
    dd = d[:, waves, :3]
    tot = dd.sum()
    print(f"role {role}: cycles per wave {dd.sum(axis=2).mean():.0f}")
    for k in range(3):
        per_wave = dd[:, :, k].mean(axis=0)
        print(f"  {names[k]:32s} {dd[:, :, k].mean():10.0f}  ({dd[:, :, k].sum() / tot * 100:5.1f} %)   by wave: " + " ".join(f"{v:9.0f}" for v in per_wave))
