#!/usr/bin/env python3
"""Is a kernel slower when other kernels ran since its last launch (cold instruction cache)?  k_stem1 launched back to back (its code
stays resident) against k_stem1 alternated with a different kernel family in between; rocprofv3 --kernel-trace gives the durations:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ic -o ic -- python3 tools/icache_exp.py"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch
import orehip as ore
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 255, (1, 3, 640, 640), dtype=torch.uint8, generator=g).cuda()
w = torch.randn(64, 3, 3, 3, generator=g).cuda()
sc, sh = torch.ones(64).cuda(), torch.zeros(64).cuda()
x = torch.randn(1, 40, 40, 96, generator=g).cuda()
wp = ore.pack_conv_weight(torch.randn(96, 96, 3, 3, generator=g) / 30).cuda()
x2 = torch.randn(1, 80, 80, 256, generator=g).cuda()
wp2 = ore.pack_conv_weight(torch.randn(128, 256, 1, 1, generator=g) / 16).cuda()
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
torch.cuda.synchronize()
if mode in ("both", "warm"):
    for _ in range(60):
        ore.stem1(img, 640, 640, (100., 110., 120.), (50., 55., 60.), w, sc, sh)
    torch.cuda.synchronize()
if mode in ("both", "cold"):
    for _ in range(60):
        ore.stem1(img, 640, 640, (100., 110., 120.), (50., 55., 60.), w, sc, sh)
        ore.conv2d(x, wp, 96, 3, 1)
        ore.conv2d(x2, wp2, 128, 1, 1)
        ore.maxpool3x3s2(x2)
    torch.cuda.synchronize()
print("done", mode)
