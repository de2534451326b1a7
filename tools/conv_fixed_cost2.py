#!/usr/bin/env python3
"""Fixed cost of one conv launch vs block count: K=16 (one K step) at M=6400, N=80 under forced tiles (GPU box only)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import orehip  # noqa: E402
from conv_fixed_cost import timeit  # noqa: E402

dev = torch.device("cuda")
L = orehip.lib()
for (H, W, Cout) in ((80, 80, 80), (160, 160, 64)):
    for Cin in (16, 64):
        x = torch.randn(1, H, W, Cin, device=dev)
        w = orehip.pack_conv_weight(torch.randn(Cout, Cin, 1, 1)).to(dev)
        out = torch.empty(1, H, W, Cout, device=dev)
        ws = torch.zeros(L.ore_conv_workspace_floats(), device=dev)
        for tile in ((128, 0, 4, 1, 1), (64, 0, 4, 1, 1), (32, 0, 1, 1, 4), (32, 32, 1, 1, 4), (16, 32, 1, 1, 4), (16, 16, 1, 1, 4)):
            bn = tile[1] or (Cout + 15) // 16 * 16
            L.ore_conv_set_plan_override(tile[0], bn, tile[2], tile[3], tile[4])
            try:
                us = timeit(lambda: orehip.conv2d(x, w, Cout, 1, 1, out=out, workspace=ws, splitk=1))
                blocks = -(-H * W // tile[0]) * -(-((Cout + 15) // 16 * 16) // bn)
                print("M=%d N=%d K=%d tile %dx%d waves %dx%dx%d blocks=%d: %.2f us" % (H * W, Cout, Cin, tile[0], bn, tile[2], tile[3], tile[4], blocks, us), flush=True)
            except orehip.OreError as e:
                print("tile", tile, "unavailable")
        L.ore_conv_set_plan_override(0, 0, 0, 0, 0)
