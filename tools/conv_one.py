#!/usr/bin/env python3
"""Run one conv layer shape N times (for rocprofv3 --pmc runs).  usage: conv_one.py <layer> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, orehip
from conv_tune import LAYERS
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
_, H, W, Cin, Cout, k, stride = [l for l in LAYERS if l[0] == name][0]
dev = torch.device("cuda")
x = torch.randn(1, H, W, Cin, device=dev)
w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
out = torch.empty(1, Ho, Wo, Cout, device=dev)
for _ in range(reps):
    orehip.conv2d(x, w, Cout, k, stride, out=out)
torch.cuda.synchronize()
