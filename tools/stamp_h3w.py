"""Times fused_block2_h3w_kernel (two residual blocks per launch, fused_h3w.hip) on the bench shape through
bf_debug_fused_block2_h3 and, for an H3V_ABLATE=32 build (BFCNN_HIP_LIB=...), prints the average cycles a wave spends per step in
each phase (s_memtime stamps; read the SHARES: stamps add fences).  Roles: A1 (conv1a, waves 0-2, stores), B1 (conv2a, 3-5),
A2 (conv1b, 6-8, DMA), B2 (conv2b, 9-11).  Phases: compute + memory issue | wait for the DMA of row s+1 (A2 only) | barrier."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blind_image_denoising_amd import _native as N

B, H, W = int(os.environ.get("B", 128)), int(os.environ.get("H", 256)), int(os.environ.get("W", 256))
L = N.lib()
L.bf_debug_set_fused_dbg.argtypes = [C.c_void_p]
x = torch.randn((B, H, W, 16), device="cuda")
out = torch.empty_like(x)
w4 = torch.randn((4, 3, 3, 16, 16), device="cuda") * 0.1
sc, sh = torch.ones((2, 16), device="cuda"), torch.zeros((2, 16), device="cuda")
scratch = torch.zeros(int(L.bf_debug_fused_block2_h3_scratch_floats(B, H, W)), device="cuda")
NWG, NW = 256, 12
dbg = torch.zeros(NWG * NW * 8, dtype=torch.int64, device="cuda")
L.bf_debug_set_fused_dbg(C.c_void_p(dbg.data_ptr()))
call = lambda: L.bf_debug_fused_block2_h3(N.ptr(x), N.ptr(w4), N.ptr(sc), N.ptr(sh), N.ptr(out), N.ptr(scratch), B, H, W, 1, 0, None)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()

# kernel time alone: the debug entry also converts fp32 <-> split-planar, so time the whole call and the conversions apart
import time
def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
t_all = timed(call)
print(f"{os.environ.get('BFCNN_HIP_LIB', 'default lib')}: debug call (convert in + kernel + convert out) {t_all:.1f} us")

dbg.zero_()
assert call() == 0
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(NWG, NW, 8).astype(np.float64)      # [block][wave][phase]
if d.sum() == 0:
    sys.exit(0)
cyc = d[:, :, :3].sum(axis=2)
ticks = d[:, :, 3]
live = ticks > 0
mhz = np.median(cyc[live] / ticks[live]) * 100
print(f"workgroups with work: {(cyc.sum(axis=1) > 0).sum()} of {NWG}; cycles per wave: mean {cyc[live].mean():.0f} max {cyc.max():.0f}; "
      f"in-kernel clock {mhz:.0f} MHz (s_memtime / s_memrealtime) -> {cyc[live].mean() / mhz:.1f} us")
names = ["compute / memory issue", "wait DMA of row s+1", "barrier"]
nsteps = (H + 12 + 5) // 6 * 6
for role, waves in (("A1 conv1a + stores", slice(0, 3)), ("B1 conv2a", slice(3, 6)), ("A2 conv1b + DMA", slice(6, 9)), ("B2 conv2b", slice(9, 12))):
    dd = d[:, waves, :3]
    tot = dd.sum()
    print(f"role {role}: cycles per wave {dd.sum(axis=2).mean():.0f} = {dd.sum(axis=2).mean() / nsteps:.0f} per step")
    for k in range(3):
        per_wave = dd[:, :, k].mean(axis=0)
        print(f"  {names[k]:32s} {dd[:, :, k].mean():10.0f}  ({dd[:, :, k].sum() / tot * 100:5.1f} %)   by wave: " + " ".join(f"{v:9.0f}" for v in per_wave))
