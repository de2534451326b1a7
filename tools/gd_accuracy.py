#!/usr/bin/env python3
"""k_conv_gd against the kernels it replaces, both against an fp64 convolution of the same fp32 inputs (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, torch.nn.functional as F
import orehip as ore
L = ore.lib()
g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, k, s) in ((80, 80, 352, 256, 1, 1), (160, 160, 320, 112, 1, 1), (320, 320, 64, 128, 3, 2)):
    x = torch.relu(torch.randn(1, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, s, k // 2)
    xn, wp = x.permute(0, 2, 3, 1).contiguous().cuda(), ore.pack_conv_weight(w).cuda()
    out = {}
    for name, mode in (("plan (gd)", 1), ("gs / igemm", 0)):
        L.ore_conv_set_plan_override(-14, mode, 0, 0, 0)
        y = ore.conv2d(xn, wp, Cout, k, s).permute(0, 3, 1, 2).cpu().double()
        e = (y - ref).abs()
        out[name] = y
        print("%dx%d %d->%d k%d s%d  %-11s max|err| / max|ref| %.3e   rms err / rms ref %.3e" % (H, W, Cin, Cout, k, s, name, e.max() / ref.abs().max(), (e ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()))
    L.ore_conv_set_plan_override(-14, 1, 0, 0, 0)
    print("      gd vs gs max diff / max|ref| %.3e" % float((out["plan (gd)"] - out["gs / igemm"]).abs().max() / ref.abs().max()))
