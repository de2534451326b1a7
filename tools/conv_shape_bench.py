#!/usr/bin/env python3
"""Time single conv / weight-gradient calls at given shapes (the plan the library picks), 20 repetitions between two stream events.

    python tools/conv_shape_bench.py [--precision fp32|bf16] conv:B,H,W,Cin,Cout,k[,stride] wgrad:B,H,W,Cin,Cout,k ...
`conv` = orehip.conv2d on a packed random weight (Winograd form passed for 3x3 stride 1 where a build exists), `wgrad` =
orehip.conv2d_wgrad.  Prints us per call and algorithmic TFLOP/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))

import torch  # noqa: E402


def main():
    import orehip
    args = sys.argv[1:]
    if args and args[0] == "--precision":
        orehip.set_conv_precision(args[1])
        args = args[2:]
    print("# ore_version %d" % orehip.lib().ore_version())
    for spec in args:
        kind, rest = spec.split(":")
        v = [int(t) for t in rest.split(",")]
        B, H, W, Cin, Cout, k = v[:6]
        stride = v[6] if len(v) > 6 else 1
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, H, W, Cin, generator=g).cuda()
        if kind == "conv":
            w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
            wp = orehip.pack_conv_weight(w).cuda() if Cin % 16 == 0 else orehip.pack_conv_weight(torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 16 - Cin % 16))).cuda()
            ww = orehip.winograd_weight(wp, Cout, Cin) if (k == 3 and stride == 1 and orehip.winograd_covers(Cout, Cin)) else None
            fn = lambda: orehip.conv2d(x, wp, Cout, k, stride, k // 2, w_wino=ww)
            fl = 2.0 * B * (H // stride) * (W // stride) * Cout * Cin * k * k
        else:
            dz = torch.randn(B, H, W, Cout, generator=g).cuda()
            fn = lambda: orehip.conv2d_wgrad(x, dz, k)
            fl = 2.0 * B * H * W * Cout * Cin * k * k
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("%-44s %9.1f us %8.1f TFLOP/s" % (spec, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
