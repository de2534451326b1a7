#!/usr/bin/env python3
"""Sweep tile / ring depth / split-K of k_conv_kw for one layer shape: python tools/conv_kw_sweep.py H W Cin Cout k [stride]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

H, W, Cin, Cout, k = (int(v) for v in sys.argv[1:6])
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda")
L = orehip.lib()
x = torch.randn(1, H, W, Cin, device=dev)
w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(dev)
Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
out = torch.empty(1, Ho, Wo, Cout, device=dev)
flops = 2.0 * Ho * Wo * Cout * Cin * k * k


def t(reps=30):
    for _ in range(3):
        orehip.conv2d(x, w, Cout, k, stride, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        orehip.conv2d(x, w, Cout, k, stride, out=out)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


L.ore_conv_set_plan_override(-2, 0, 0, 0, 0)
base = t()
ref = out.clone()
print("r01 plan: %.2f us (%.1f TF/s)" % (base, flops / base / 1e6))
L.ore_conv_set_plan_override(-2, 2, 0, 0, 0)
res = []
for bm in (16, 32):
    for bn in (16, 32, 48, 64, 80):
        if bn > (Cout + 15) // 16 * 16:
            continue
        for nw in (4, 8, 16):                          # waves per block (in-block K split): 8 / 16 exist for the small fp32 tiles
            if nw > 4 and (bm != 16 or bn > 48 or (nw == 16 and bn > 32)):
                continue
            L.ore_conv_set_plan_override(-6, nw, 0, 0, 0)
            for ns in ((2, 3, 4) if nw == 4 else (2,)):
                for S in (1, 2, 4, 8):
                    L.ore_conv_set_plan_override(-3, bm, bn, ns, S)
                    try:
                        us = t(20)
                    except orehip.OreError:
                        continue
                    err = float((out - ref).abs().max() / ref.abs().max())
                    res.append((us, bm, bn, ns, S, err, nw))
L.ore_conv_set_plan_override(-6, 0, 0, 0, 0)
L.ore_conv_set_plan_override(-3, 0, 0, 0, 0)
L.ore_conv_set_plan_override(-2, 1, 0, 0, 0)
res.sort()
for us, bm, bn, ns, S, err, nw in res[:10]:
    print("%6.2f us  %5.1f TF/s  tile %2dx%2d ns %d S %d nw %2d  relerr %.1e" % (us, flops / us / 1e6, bm, bn, ns, S, nw, err))
