"""Runs the split-f16 row-streaming block of an H3_ABLATE=32 build on the bench shape and prints the average cycles a
wave spends per tile in each phase (s_memtime stamps; read the SHARES, not the total: stamps add fences)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blind_image_denoising_amd import _native as N

B, H, W = int(os.environ.get("B", 128)), 256, 256
L = N.lib()
L.bf_debug_set_fused_dbg.argtypes = [C.c_void_p]
x = torch.randn((B, H, W, 16), device="cuda")
out = torch.empty_like(x)
w1 = torch.randn((3, 3, 16, 16), device="cuda") * 0.1
w2 = torch.randn((3, 3, 16, 16), device="cuda") * 0.1
sc, sh = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
scratch = torch.zeros(int(L.bf_debug_fused_block_h3_scratch_floats(B, H, W)), device="cuda")
NWG, NW = 256, 8
dbg = torch.zeros(NWG * NW * 8, dtype=torch.int64, device="cuda")
L.bf_debug_set_fused_dbg(C.c_void_p(dbg.data_ptr()))
for _ in range(3):
    rc = L.bf_debug_fused_block_h3(N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(scratch), B, H, W, 1, None)
    assert rc == 0
torch.cuda.synchronize()
# wall time of the whole debug call (pack + 2 conversions + block) and of the conversions alone -> shader clock estimate
def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_all = timed(lambda: L.bf_debug_fused_block_h3(N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(scratch), B, H, W, 1, None))
dbg.zero_()
rc = L.bf_debug_fused_block_h3(N.ptr(x), N.ptr(w1), N.ptr(w2), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(scratch), B, H, W, 1, None)
torch.cuda.synchronize()
print(f"debug call (pack + fp32->split + block + split->fp32): {t_all:.1f} us")
d = dbg.cpu().numpy().reshape(NWG, NW, 8).astype(np.float64)      # [block][wave][phase]
tiles = B * 16 * 8 / NWG
names = ["next-tile index math", "conv1 (+ DMA issue)", "barrier A", "conv2 + stores", "vmcnt (next tile's DMA)", "barrier B", "tile index math", "-"]
tot = d.sum(axis=2).mean()
print(f"cycles per tile per wave (avg over {NWG} workgroups x {NW} waves), total {tot / tiles:.0f}")
for k in (6, 0, 1, 2, 3, 4, 5):
    per_wave = d[:, :, k].mean(axis=0) / tiles
    print(f"  {names[k]:26s} {d[:, :, k].mean() / tiles:9.0f}  ({d[:, :, k].sum() / d.sum() * 100:5.1f} %)   by wave: " + " ".join(f"{v:7.0f}" for v in per_wave))
