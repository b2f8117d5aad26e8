"""time of bwd_block_h3t_kernel / fwd_block_h3t_kernel through their single-kernel entries at the configs[3] shape (32 x 256 x 256), with
whatever library BFCNN_HIP_LIB names (tools/ablate_unit.sh train_bwd_h3t H3U_ABLATE 1 2 ... builds the timing variants).
The entries also pack the weights and reduce the partials (a few small launches): the same constant in every variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from blind_image_denoising_amd import _native as N

L = N.lib()
B, H, W = 32, 256, 256
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
a, gr, c, bnc = r(B, H, W, 16), r(B, H, W, 16) * 0.1, r(B, H, W, 16), r(B, H, W, 16)
coef = torch.cat([torch.ones(16), torch.full((16,), 0.1), torch.full((16,), 0.01)]).cuda()
w0, w1 = r(3, 3, 16, 16) * 0.1, r(3, 3, 16, 16) * 0.1
out, dw1, dw0, st = torch.empty_like(a), torch.empty(2304, device="cuda"), torch.empty(2304, device="cuda"), torch.empty(32, device="cuda")
scr = torch.empty(int(L.bf_debug_bwd_block_h3t_scratch_floats(B, H, W)), device="cuda")
calls = [0]

def bwd():
    N.check(L.bf_debug_bwd_block_h3t(N.ptr(a), N.ptr(gr), N.ptr(c), N.ptr(coef), N.ptr(w0), N.ptr(w1), N.ptr(bnc), N.ptr(out), N.ptr(dw1),
                                     N.ptr(dw0), N.ptr(st), N.ptr(scr), B, H, W, 1, calls[0] & 1, N.stream_ptr(a)), None, "bwd_block")
    calls[0] += 1

ao, co, st2 = torch.empty_like(a), torch.empty_like(a), torch.empty(32, device="cuda")
sc, sh = torch.ones(16, device="cuda"), torch.zeros(16, device="cuda")
scr2 = torch.empty(int(L.bf_debug_fwd_block_h3t_scratch_floats(B, H, W)), device="cuda")

def fwd():
    N.check(L.bf_debug_fwd_block_h3t(N.ptr(a), N.ptr(c), N.ptr(sc), N.ptr(sh), N.ptr(w0), N.ptr(w1), N.ptr(ao), None, N.ptr(co), N.ptr(st2),
                                     N.ptr(scr2), B, H, W, 1, calls[0] & 1, N.stream_ptr(a)), None, "fwd_block")
    calls[0] += 1

def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

which = sys.argv[1:] or ["bwd", "fwd"]
print(os.path.basename(os.environ.get("BFCNN_HIP_LIB", "default")), " ".join(f"{k} {timed(dict(bwd=bwd, fwd=fwd)[k]):.1f} us" for k in which), flush=True)
