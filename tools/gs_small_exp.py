#!/usr/bin/env python3
"""k_conv_gs against k_conv_kw on the mid-size 1x1 layers the plan sends to k_conv_kw (lat3 6400 x 256 -> 128, conv3 8400 x 256 -> 128,
lat4 1600 x 384 -> 128, s4cat 1600 x 544 -> 384).  usage: python tools/gs_small_exp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, orehip
dev = torch.device("cuda")
L = orehip.lib()
g = torch.Generator().manual_seed(0)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for name, M, K, N in (("lat3", 6400, 256, 128), ("conv3", 8400, 256, 128), ("lat4", 1600, 384, 128), ("s4cat", 1600, 544, 384), ("s3cat", 6400, 352, 256)):
    x = torch.randn(1, 1, M, K, generator=g).to(dev)
    w = orehip.pack_conv_weight(torch.randn(N, K, 1, 1, generator=g) * 0.05).to(dev)
    sh = torch.randn(N, generator=g).to(dev)
    out = torch.empty(1, 1, M, N, device=dev)
    fn = lambda: orehip.conv2d(x, w, N, 1, 1, shift=sh, relu_cout=N, out=out)
    row = ["auto %6.2f us" % t(fn)]
    ref = out.clone()
    for bm, bn in ((64, 64), (64, 128), (32, 128)):
        L.ore_conv_set_plan_override(-4, bm, bn, 0, 0)
        try:
            row.append("gs %dx%d %6.2f us" % (bm, bn, t(fn)))
            assert float((out - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        except Exception as e:
            row.append("gs %dx%d n/a" % (bm, bn))
        L.ore_conv_set_plan_override(-4, 0, 0, 0, 0)
    print("%-6s M=%5d K=%3d N=%3d  %s" % (name, M, K, N, "   ".join(row)), flush=True)
