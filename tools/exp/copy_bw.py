"""HBM roof for this path's access pattern: device-to-device copy of one activation tensor (read 537 MB + write 537 MB,
the algorithmic traffic of one fused block launch at B=128, 256x256, 16 channels, 4 B)."""
import torch
n = 128 * 256 * 256 * 16
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
for name, fn in [("copy_ (read+write)", lambda: y.copy_(x)), ("x.sum() (read only)", lambda: x.sum()), ("y.zero_() (write only)", lambda: y.zero_()),
                 ("y = x*2+1 (read+write)", lambda: torch.add(x, 1.0, alpha=2.0, out=y))]:
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = n * 4 * (2 if "read+write" in name else 1)
    print(f"{name:28s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s")
