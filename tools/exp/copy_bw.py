"""Achievable HBM bandwidth of plain streaming kernels on this chip (reference points for the roofline fractions): device copy
(read + write), fill (write only), sum (read only), 537 MB tensors = one activation of the bench batch."""
import torch
n = 128 * 256 * 256 * 16
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
def t(f, reps=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
b = n * 4
us = t(lambda: y.copy_(x)); print(f"copy  {us:7.1f} us  {2 * b / us / 1e6:6.2f} TB/s (read + write)")
us = t(lambda: y.fill_(1.0)); print(f"fill  {us:7.1f} us  {b / us / 1e6:6.2f} TB/s (write)")
us = t(lambda: x.sum()); print(f"sum   {us:7.1f} us  {b / us / 1e6:6.2f} TB/s (read)")
z = torch.empty_like(x)
us = t(lambda: torch.add(x, y, out=z)); print(f"add   {us:7.1f} us  {3 * b / us / 1e6:6.2f} TB/s (2 reads + write)")
