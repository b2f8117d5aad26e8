"""Times the split-f16 ConvNext MLP kernels of whatever library BFCNN_HIP_LIB points to: decoder block (1x1 depthwise + LayerNorm +
MLP + Add) at C = 32 / 64 and the bare MLP + skip at C = 64, on the level sizes of the unet bench (32 x 512^2, 32 x 256^2)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from blind_image_denoising_amd import unet_laplacian as UL
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
out = []
for C, S in ((32, 512), (64, 256)):
    x = torch.randn((32, S, S, C), device="cuda")
    dw = torch.randn(C, device="cuda"); g = torch.rand(C, device="cuda") + 0.5
    w1, w2 = torch.randn((C, 4 * C), device="cuda") / C ** 0.5, torch.randn((4 * C, C), device="cuda") / (4 * C) ** 0.5
    pk = UL.pack_mlp_h3(w1, w2); mult = torch.rand(C, device="cuda")
    out.append(f"block1 C={C}: {t(lambda: UL.convnext_block1_h3(x, dw, g, pk, mult, 'leaky_relu_01')):7.1f} us")
    out.append(f"mlp+skip C={C}: {t(lambda: UL.convnext_mlp_h3(x, x, pk, mult, 'leaky_relu_01')):7.1f} us")
print(os.path.basename(os.environ.get("BFCNN_HIP_LIB", "default")), " | ".join(out))
