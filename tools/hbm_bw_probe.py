import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (32, 128, 402, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    us = t(lambda: b.copy_(a))
    print("copy  %5d MB: %8.1f us  %.2f TB/s (read+write)" % (mb, us, 2 * n * 4 / us / 1e6))
    us = t(lambda: a.sum())
    print("sum   %5d MB: %8.1f us  %.2f TB/s (read)" % (mb, us, n * 4 / us / 1e6))
    us = t(lambda: b.fill_(1.0))
    print("fill  %5d MB: %8.1f us  %.2f TB/s (write)" % (mb, us, n * 4 / us / 1e6))
