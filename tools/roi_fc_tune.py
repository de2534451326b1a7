#!/usr/bin/env python3
"""Tile / split-K sweep for the second-stage GEMM of the engine: [320 rois] x [8192] -> 128 (pre-composed DSA mix + fc1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, orehip
dev = torch.device("cuda"); L = orehip.lib()
M, K, N = 320, 8192, 128
x = torch.randn(1, 1, M, K, device=dev)
w = orehip.pack_conv_weight(torch.randn(N, K, 1, 1) / K ** 0.5).to(dev)
b = torch.randn(N, device=dev)
out = torch.empty(1, 1, M, N, device=dev)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
ref = None
print("auto plan: %.1f us" % timeit(lambda: orehip.conv2d(x, w, N, 1, 1, shift=b, relu_cout=N, out=out)))
ref = out.clone()
for t in ((16, 32, 1, 1, 4), (32, 32, 1, 1, 4), (32, 64, 1, 1, 4), (32, 128, 1, 1, 4), (32, 64, 2, 1, 2), (64, 64, 2, 1, 2), (64, 128, 2, 1, 2), (64, 128, 2, 2, 1), (16, 64, 1, 1, 4)):
    for S in (1, 2, 4, 8, 16):
        L.ore_conv_set_plan_override(*t)
        try:
            us = timeit(lambda: orehip.conv2d(x, w, N, 1, 1, shift=b, relu_cout=N, out=out, splitk=S))
            err = float((out - ref).abs().max())
            print("%-20s S=%2d  %7.1f us  maxdiff %.1e" % (t, S, us, err), flush=True)
        except orehip.OreError as ex:
            print("%-20s S=%2d  n/a (%s)" % (t, S, str(ex)[:50]))
L.ore_conv_set_plan_override(0, 0, 0, 0, 0)
