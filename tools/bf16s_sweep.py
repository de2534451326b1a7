#!/usr/bin/env python3
"""Tile sweep of the bf16-STORAGE conv kernels (k_conv_gs / k_conv_kw, SB builds) over the 28 conv launches of a 640x640 image.
usage: bf16s_sweep.py [layer-name-substring]   -> one line per (layer, kernel, tile) with the time of back-to-back launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, orehip
from conv_layers_table import LAYERS, EXTRA
L = orehip.lib()
dev = torch.device("cuda")
shapes = [(n, 1, h, w, ci, co, k, s) for n, h, w, ci, co, k, s in LAYERS]
shapes += [("conv3", 1, 8400, 1, 256, 128, 1, 1), ("tower", 1, 8400, 1, 128, 128, 3, 1)]   # level-major rows as one tall image (1x1: exact; 3x3: same work)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
GS = [(64, 64), (128, 64), (128, 128), (64, 128), (32, 64), (32, 128), (64, 112), (128, 112), (64, 80)]
KW = [(16, 16), (16, 32), (16, 48), (16, 64), (16, 80), (32, 16), (32, 32), (32, 48), (32, 64), (32, 80)]
KD = [(16, 16, 4, 4), (16, 16, 4, 8), (16, 16, 8, 4), (16, 16, 16, 2), (16, 32, 4, 4), (16, 32, 8, 2), (32, 32, 4, 4), (32, 32, 8, 2), (16, 48, 4, 4), (16, 48, 8, 2),
      (16, 64, 4, 2), (16, 80, 4, 2), (32, 64, 4, 2), (32, 80, 4, 2), (64, 64, 4, 2), (32, 48, 4, 2), (32, 16, 4, 4)]


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for name, B, H, W, Cin, Cout, k, stride in shapes:
    if flt and flt not in name:
        continue
    x = torch.zeros(B * H * W + 2, Cin, dtype=torch.bfloat16, device=dev)
    x[:-2] = torch.randn(B * H * W, Cin, device=dev).to(torch.bfloat16)
    xd = x[:-2].view(B, H, W, Cin)
    w = orehip.pack_conv_weight_bf16(torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    fn = lambda: orehip.conv2d(xd, w, Cout, k, stride, out=out)      # noqa: E731
    res = [("auto", timed(fn))]
    for bm, bn in GS:
        if bn > (Cout + 15) // 16 * 16 or (Cout + 15) // 16 * 16 % 16:
            continue
        for ns in (3, 4):
            L.ore_conv_set_plan_override(-4, bm, bn, ns, 0)
            try:
                res.append((f"gs {bm}x{bn} ns{ns}", timed(fn)))
            except Exception:
                pass
    L.ore_conv_set_plan_override(-4, 0, 0, 3, 0)
    for bm, bn in KW:
        if bn > (Cout + 15) // 16 * 16:
            continue
        for sk in (1, 2, 4):
            L.ore_conv_set_plan_override(-3, bm, bn, 2, sk)
            try:
                res.append((f"kw {bm}x{bn} S{sk}", timed(fn)))
            except Exception:
                pass
    L.ore_conv_set_plan_override(-3, 0, 0, 0, 0)
    for bm, bn, nw, sb in KD:                                    # k_conv_kd's bf16-storage builds (round 4)
        if bn > (Cout + 15) // 16 * 16:
            continue
        L.ore_conv_set_plan_override(-13, bm, bn, nw, sb)
        try:
            res.append((f"kd {bm}x{bn} w{nw} b{sb}", timed(fn)))
        except Exception:
            pass
    L.ore_conv_set_plan_override(-13, 0, 0, 0, 0)
    res.sort(key=lambda t: t[1])
    print("%-7s M=%6d Cin=%4d Cout=%4d k=%d s=%d | auto %.1f | best: %s" % (name, B * Ho * Wo, Cin, Cout, k, stride, dict(res)["auto"],
          "  ".join("%s %.1f" % r for r in res[:5])), flush=True)
