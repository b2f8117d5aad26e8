"""Times the level-0 encoder ConvNext block kernel of whatever library BFCNN_HIP_LIB points to (ablation variants)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from blind_image_denoising_amd import unet_laplacian as UL
B, S, C, k = 32, 512, 32, int(os.environ.get("ENC_K", 5))
x = torch.randn((B, S, S, C), device="cuda")
dw = torch.randn((k, k, C), device="cuda") * 0.2
g = torch.rand(C, device="cuda") + 0.5
w1, w2 = torch.randn((C, 4 * C), device="cuda") / C ** 0.5, torch.randn((4 * C, C), device="cuda") / (4 * C) ** 0.5
pk = UL.pack_mlp_h3(w1, w2)
mult = torch.rand(C, device="cuda")
f = lambda: UL.convnext_block_h3(x, dw, g, pk, mult, "leaky_relu_01")
if os.environ.get("ENC_VARIANT"):
    from blind_image_denoising_amd import _native as N
    N.lib().bf_op_set_variant(b"enc32", int(os.environ["ENC_VARIANT"]))
for _ in range(3):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('BFCNN_HIP_LIB', 'default'):70s} {e0.elapsed_time(e1) * 100:8.1f} us per launch")
