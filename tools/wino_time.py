#!/usr/bin/env python3
"""Time the 3x3 large-M layers on the Winograd kernel (and the direct kernels beside it) with HIP events, back-to-back launches.
usage: wino_time.py [mode ...]   mode = ore_conv_set_plan_override(-7, mode): 0 direct, 1 pipelined Winograd, 3 phase-serial Winograd.
ORE_WINO_DBG=<bits> skips phases of the serial build (timing experiments; results are then wrong)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, orehip
LAYERS = [("stem2", 1, 320, 320, 64, 64), ("s2l0", 1, 160, 160, 128, 64), ("s2l1", 1, 160, 160, 64, 64), ("out3", 1, 80, 80, 128, 128),
          ("s3l0", 1, 80, 80, 112, 80), ("s3l1", 1, 80, 80, 80, 80), ("out4", 1, 40, 40, 128, 128), ("s4l1", 1, 40, 40, 96, 96), ("out5", 1, 20, 20, 128, 128), ("s5l1", 1, 20, 20, 112, 112),
          ("s3l1_480", 1, 60, 80, 80, 80), ("s4l1_800", 1, 38, 50, 96, 96), ("s4l1x16", 16, 40, 40, 96, 96), ("s5l1x16", 16, 20, 20, 112, 112),
          ("stem2x16", 16, 320, 320, 64, 64), ("s2l0x16", 16, 160, 160, 128, 64), ("s3l1x16", 16, 80, 80, 80, 80)]
modes = [int(a) for a in sys.argv[1:]] or [0, 1, 3]
dev = torch.device("cuda")
L = orehip.lib()
for name, B, H, W, Cin, Cout in LAYERS:
    x = torch.randn(B, H, W, Cin, device=dev)
    wt = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    w = orehip.pack_conv_weight(wt).to(dev)
    U = orehip.winograd_weight(w, Cout, Cin)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    out = torch.empty(B, H, W, Cout, device=dev)
    fl = 2.0 * B * H * W * Cin * Cout * 9
    row = []
    for m in modes:
        L.ore_conv_set_plan_override(-7, m, 0, 0, 0)
        for _ in range(5):
            orehip.conv2d(x, w, Cout, 3, 1, scale=sc, shift=sh, relu_cout=Cout, out=out, w_wino=U)
        torch.cuda.synchronize()
        n = 50 if B == 1 else 10
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            orehip.conv2d(x, w, Cout, 3, 1, scale=sc, shift=sh, relu_cout=Cout, out=out, w_wino=U)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / n
        row.append("mode %d: %8.2f us %6.1f TF/s" % (m, us, fl / us / 1e6))
    print("%-9s %s" % (name, "   ".join(row)), flush=True)
L.ore_conv_set_plan_override(-7, 1, 0, 0, 0)
