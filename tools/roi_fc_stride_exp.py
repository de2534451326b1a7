#!/usr/bin/env python3
"""Does the 32 KB row stride of the second-stage GEMM's A matrix ([320, 8192] fp32) hot-spot memory channels?  Times the same GEMM
(k_conv_kw, production plan and a few tiles) with the activation rows padded by 0 / 16 / 64 / 80 floats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

dev = torch.device("cuda")
L = orehip.lib()
M, K, N = 320, 8192, 128
w = orehip.pack_conv_weight(torch.randn(N, K, 1, 1) / K ** 0.5).to(dev)
out = torch.empty(1, 1, M, N, device=dev)
flops = 2.0 * M * K * N


def t(x, reps=50):
    for _ in range(5):
        orehip.conv2d(x, w, N, 1, 1, Cin=K, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        orehip.conv2d(x, w, N, 1, 1, Cin=K, out=out)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


base = torch.randn(1, 1, M, K, device=dev)
for padf in (0, 16, 32, 64, 80, 272):
    x = torch.zeros(1, 1, M, K + padf, device=dev)
    x[..., :K] = base
    line = "row stride %5d floats:" % (K + padf)
    for tile in (None, (16, 48, 2, 4), (32, 64, 2, 8), (32, 64, 2, 4), (16, 48, 2, 8), (32, 32, 2, 8)):
        if tile is None:
            L.ore_conv_set_plan_override(-3, 0, 0, 0, 0)
        else:
            L.ore_conv_set_plan_override(-3, *tile)
        try:
            us = t(x)
            line += "  %s %.1f us" % ("plan" if tile is None else "%dx%d S%d" % (tile[0], tile[1], tile[3]), us)
        except orehip.OreError:
            pass
    print(line, flush=True)
L.ore_conv_set_plan_override(-3, 0, 0, 0, 0)
