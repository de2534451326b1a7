#!/usr/bin/env python3
"""How does a 3x3 conv kernel's time scale with the number of blocks per CU?  (k_conv3x3_patch<4>: one block = a 4x16-pixel tile.)
Times the stage-2 layer-0 shape family (Cin -> 64, 3x3) at grids of 128..1024 blocks, fp32 and bf16 operands, so that the MFMA phase
(fp32 - bf16) and the non-MFMA floor (bf16) can be read per resident-block count.  usage: conv_occupancy_exp.py [Cin]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
print("# Cin", Cin)
Cout, k = 64, 3
dev = torch.device("cuda")
w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
print("blocks  H x W      fp32_us  bf16_us  mfma_us_at_peak(one block per CU share)")
for ty, tx in ((8, 16), (16, 16), (24, 16), (40, 10), (32, 16), (48, 16), (64, 16), (128, 16)):
    H, W = 4 * ty, 16 * tx
    x = torch.randn(1, H, W, Cin, device=dev)
    out = torch.empty(1, H, W, Cout, device=dev)
    res = {}
    for mode in ("fp32", "bf16"):
        prev = orehip.set_conv_precision(mode)
        orehip.lib().ore_conv_set_plan_override(-1, 4, 0, 0, 0)          # force k_conv3x3_patch<4>
        for _ in range(5):
            orehip.conv2d(x, w, Cout, k, 1, out=out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            orehip.conv2d(x, w, Cout, k, 1, out=out)
        b.record()
        torch.cuda.synchronize()
        res[mode] = a.elapsed_time(b) / 50 * 1e3
        orehip.set_conv_precision(prev)
        orehip.lib().ore_conv_set_plan_override(-1, -1, 0, 0, 0)
    fl = 2.0 * H * W * Cout * Cin * 9
    print("%5d  %3d x %3d  %8.2f %8.2f   %8.2f" % (ty * tx, H, W, res["fp32"], res["bf16"], fl / 157.3e12 * 1e6))
