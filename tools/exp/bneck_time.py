"""time bf_op_bneck_block_h3 alone (batch 64 x 256 x 256) with and without the skip: how much of a launch is the second read of x"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from blind_image_denoising_amd import _native as N, unet_laplacian as UL

B, S = 64, 256
x = torch.randn((B, S, S, 32), device="cuda")
y = torch.empty_like(x)
pk = UL.pack_bneck_h3(torch.randn((32, 32), device="cuda") * 0.2, torch.randn((3, 3, 32, 4), device="cuda") * 0.1, torch.randn((128, 32), device="cuda") * 0.1)
b1, b2 = torch.randn(128, device="cuda") * 0.1, torch.randn(32, device="cuda") * 0.1
L = N.lib()
for add in (1, 0, 1, 0):
    f = lambda: N.check(L.bf_op_bneck_block_h3(N.ptr(x), N.ptr(y), N.ptr(pk), None, 1, 0.0, N.ptr(b1), 1, 0.0, N.ptr(b2), 1, 0.0, add, B, S, S, N.stream_ptr(x)), None, "bneck")
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"add_res={add}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch")
