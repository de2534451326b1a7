#!/usr/bin/env python3
"""Device streaming bandwidth seen by one kernel at the sizes of this path's layers: d2d copy (read + write) and fill (write only)."""
import torch
for mb in (8, 26, 44, 64, 128, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    reps = 20
    for _ in range(reps):
        y.copy_(x)
    e.record(); torch.cuda.synchronize()
    tc = a.elapsed_time(e) / reps * 1e-3
    a.record()
    for _ in range(reps):
        y.fill_(1.0)
    e.record(); torch.cuda.synchronize()
    tf = a.elapsed_time(e) / reps * 1e-3
    print("%5d MB: copy %.1f us = %.2f TB/s (r+w)   fill %.1f us = %.2f TB/s" % (mb, tc * 1e6, 2 * mb * 1.048576e6 / tc / 1e12, tf * 1e6, mb * 1.048576e6 / tf / 1e12), flush=True)
