"""Memory-system ceilings on the box, with torch's own kernels (fill = write only, sum = read
only, copy = both), at a cache-resident and an HBM-sized footprint."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (115, 460, 2048):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev, dtype=torch.float32); y = torch.empty_like(x)
    x.normal_()
    tw = t(lambda: x.fill_(1.0)); tr = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
    print("%5d MB  write %.2f TB/s  read %.2f TB/s  copy %.2f TB/s (r+w)" % (
        mb, mb * 1.048576e6 / tw / 1e12, mb * 1.048576e6 / tr / 1e12, 2 * mb * 1.048576e6 / tc / 1e12))
