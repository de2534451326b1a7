#!/usr/bin/env python3
"""k_conv_gs tile experiment on the two large 1x1 concat convs (s2cat 160x160 320->112, s3cat 80x80 352->256): python tools/gs_tile_exp.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

L = orehip.lib()
dev = torch.device("cuda")


def t(x, w, Cout, reps=50):
    out = torch.empty(1, x.shape[1], x.shape[2], Cout, device=dev)
    for _ in range(5):
        orehip.conv2d(x, w, Cout, 1, 1, out=out)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        orehip.conv2d(x, w, Cout, 1, 1, out=out)
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


for (H, W, Cin, Cout) in ((160, 160, 320, 112), (80, 80, 352, 256)):
    x = torch.randn(1, H, W, Cin, device=dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5).to(dev)
    L.ore_conv_set_plan_override(-4, 0, 0, 0, 0)
    line = "%dx%d %d->%d: plan %.2f us" % (H, W, Cin, Cout, t(x, w, Cout))
    for (bm, bn) in ((64, 112), (64, 64), (32, 64), (128, 112), (128, 64), (64, 128), (32, 128), (64, 80)):
        for ns in (3, 4):
            L.ore_conv_set_plan_override(-4, bm, bn, ns, 0)
            try:
                line += "  %dx%d/%d %.2f" % (bm, bn, ns, t(x, w, Cout))
            except orehip.OreError:
                pass
    L.ore_conv_set_plan_override(-4, 0, 0, 0, 0)
    print(line, flush=True)
