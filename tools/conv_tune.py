#!/usr/bin/env python3
"""Sweep tile / split-K configurations of the implicit-GEMM conv for every layer shape of the 640x640 bs=1 eval path
and print the measured time per configuration (GPU box only).  Feeds the plan table in csrc/ore_conv.hip."""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

LAYERS = [  # name, H, W, Cin, Cout, k, stride
    ("stem2", 320, 320, 64, 64, 3, 1), ("stem3", 320, 320, 64, 128, 3, 2),
    ("s2l0", 160, 160, 128, 64, 3, 1), ("s2l1", 160, 160, 64, 64, 3, 1), ("s2cat", 160, 160, 320, 112, 1, 1),
    ("s3l0", 80, 80, 112, 80, 3, 1), ("s3l1", 80, 80, 80, 80, 3, 1), ("s3cat", 80, 80, 352, 256, 1, 1),
    ("s4l0", 40, 40, 256, 96, 3, 1), ("s4l1", 40, 40, 96, 96, 3, 1), ("s4cat", 40, 40, 544, 384, 1, 1),
    ("s5l0", 20, 20, 384, 112, 3, 1), ("s5l1", 20, 20, 112, 112, 3, 1), ("s5cat", 20, 20, 720, 512, 1, 1),
    ("lat5", 20, 20, 512, 128, 1, 1), ("out5", 20, 20, 128, 128, 3, 1), ("lat4", 40, 40, 384, 128, 1, 1),
    ("out4", 40, 40, 128, 128, 3, 1), ("lat3", 80, 80, 256, 128, 1, 1), ("out3", 80, 80, 128, 128, 3, 1),
    ("conv3", 84, 100, 256, 128, 1, 1), ("tower", 84, 100, 128, 128, 3, 1), ("pred", 84, 100, 128, 5, 3, 1),
]
TILES = [(128, 0, 4, 1, 1), (128, 128, 2, 2, 1), (64, 0, 4, 1, 1), (64, 128, 2, 2, 1), (128, 64, 2, 1, 2), (128, 128, 2, 1, 2),
         (64, 128, 2, 1, 2), (64, 112, 2, 1, 2), (64, 96, 2, 1, 2), (64, 80, 2, 1, 2), (64, 64, 2, 1, 2), (64, 48, 2, 1, 2),
         (64, 32, 2, 1, 2), (32, 128, 2, 1, 2), (32, 64, 2, 1, 2), (32, 128, 1, 1, 4), (32, 112, 1, 1, 4), (32, 96, 1, 1, 4),
         (32, 80, 1, 1, 4), (32, 64, 1, 1, 4), (32, 48, 1, 1, 4), (32, 32, 1, 1, 4), (16, 64, 1, 1, 4), (16, 32, 1, 1, 4)]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]


def main():
    only = sys.argv[1:] or None
    dev = torch.device("cuda")
    L = orehip.lib()
    for name, H, W, Cin, Cout, k, stride in LAYERS:
        if only and name not in only:
            continue
        x = torch.randn(1, H, W, Cin, device=dev)
        w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
        C16 = (Cout + 15) // 16 * 16
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        M = Ho * Wo
        out = torch.empty(1, Ho, Wo, Cout, device=dev)
        flops = 2.0 * M * Cout * Cin * k * k
        res = []
        for (bm, bn, wgm, wgn, wgk) in TILES:
            bn_eff = min(bn, C16) if bn else (C16 if C16 <= 128 else 128)
            if bn == 0 and bn_eff == 128:
                continue  # whole-N tiles of 128 use the 2x2 layout
            if bn and bn > C16 and bn_eff in [t[1] for t in TILES if t[0] == bm and t[2:] == (wgm, wgn, wgk) and t[1] and t[1] <= C16]:
                continue
            tiles = -(-M // bm) * -(-C16 // bn_eff)
            for S in SPLITS:
                nsteps = -(-(k * k * Cin // 16) // wgk)
                if S > 1 and (S > nsteps // 2 or tiles * S > 4096 or tiles > 1024 or S * bm * bn_eff * 4 > 256 * 1024):
                    continue
                L.ore_conv_set_plan_override(bm, bn_eff, wgm, wgn, wgk)
                try:
                    for _ in range(3):
                        orehip.conv2d(x, w, Cout, k, stride, out=out, splitk=S)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    reps = 30
                    e0.record()
                    for _ in range(reps):
                        orehip.conv2d(x, w, Cout, k, stride, out=out, splitk=S)
                    e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / reps
                    res.append((us, bm, bn_eff, wgm, wgn, wgk, S, tiles * S))
                except orehip.OreError as ex:
                    pass
        L.ore_conv_set_plan_override(0, 0, 0, 0, 0)
        if k == 3 and stride == 1 and C16 % 64 == 0:          # 3x3 "patch" kernel, tile height 8 / 4
            for mode in (102, 16, 8, 4):
                L.ore_conv_set_plan_override(-1, mode, 0, 0, 0)
                for _ in range(3):
                    orehip.conv2d(x, w, Cout, k, stride, out=out, splitk=1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    orehip.conv2d(x, w, Cout, k, stride, out=out, splitk=1)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 30
                print(f"     patch TH={mode}: {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s")
            L.ore_conv_set_plan_override(-1, 0, 0, 0, 0)       # generic kernels only for the sweep below / 'auto' = generic
        for _ in range(3):
            orehip.conv2d(x, w, Cout, k, stride, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            orehip.conv2d(x, w, Cout, k, stride, out=out)
        e1.record()
        torch.cuda.synchronize()
        auto_us = e0.elapsed_time(e1) * 1e3 / 30
        L.ore_conv_set_plan_override(-1, -1, 0, 0, 0)
        res.sort()
        print(f"== {name}: M={M} N={Cout} K={k * k * Cin}  {flops / 1e9:.2f} GF  ideal@155TF {flops / 155e6:.1f} us   auto={auto_us:.1f} us")
        for us, bm, bn, wgm, wgn, wgk, S, blocks in res[:6]:
            print(f"     {us:8.1f} us  {flops / us / 1e6:6.1f} TF/s  tile {bm}x{bn} waves {wgm}x{wgn}x{wgk} S={S} blocks={blocks}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
