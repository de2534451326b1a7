#!/usr/bin/env python3
"""A/B per layer shape of the 640x640 bs=1 path: k_conv_igemm / patch kernels (plan of round 1) vs k_conv_kw (csrc/ore_conv_kw.hip),
interleaved rounds in one process.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

LAYERS = [  # name, H, W, Cin, Cout, k, stride, colsum
    ("stem3", 320, 320, 64, 128, 3, 2, 0), ("s2l0", 160, 160, 128, 64, 3, 1, 0), ("s2l1", 160, 160, 64, 64, 3, 1, 0),
    ("s2cat", 160, 160, 320, 112, 1, 1, 1),
    ("s3l0", 80, 80, 112, 80, 3, 1, 0), ("s3l1", 80, 80, 80, 80, 3, 1, 0), ("s3cat", 80, 80, 352, 256, 1, 1, 1),
    ("s4l0", 40, 40, 256, 96, 3, 1, 0), ("s4l1", 40, 40, 96, 96, 3, 1, 0), ("s4cat", 40, 40, 544, 384, 1, 1, 1),
    ("s5l0", 20, 20, 384, 112, 3, 1, 0), ("s5l1", 20, 20, 112, 112, 3, 1, 0), ("s5cat", 20, 20, 720, 512, 1, 1, 1),
    ("lat5*", 20, 20, 512, 128, 1, 1, 0), ("out5", 20, 20, 128, 128, 3, 1, 0), ("lat4*", 40, 40, 384, 128, 1, 1, 0),
    ("out4", 40, 40, 128, 128, 3, 1, 0), ("lat3*", 80, 80, 256, 128, 1, 1, 0), ("out3", 80, 80, 128, 128, 3, 1, 0),
    ("conv3", 84, 100, 256, 128, 1, 1, 0), ("tower", 84, 100, 128, 128, 3, 1, 0), ("roi_fc", 1, 320, 8192, 128, 1, 1, 0),
]


def main():
    only = sys.argv[1:] or None
    dev = torch.device("cuda")
    L = orehip.lib()
    tot = [0.0, 0.0, 0.0]
    print("%-8s %9s %9s %9s   %s" % ("layer", "r01 us", "kw auto", "kw forced", "GFLOP  TF/s(best)"))
    for name, H, W, Cin, Cout, k, stride, cs in LAYERS:
        if only and name.rstrip("*") not in only:
            continue
        x = torch.randn(1, H, W, Cin, device=dev)
        w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5).to(dev)
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        out = torch.empty(1, Ho, Wo, Cout, device=dev)
        flops = 2.0 * Ho * Wo * Cout * Cin * k * k

        def run():
            if cs:
                orehip.conv2d(x, w, Cout, k, stride, scale=sc, shift=sh, relu_cout=Cout, out=out, want_colsum=True)
            else:
                orehip.conv2d(x, w, Cout, k, stride, scale=sc, shift=sh, relu_cout=Cout, out=out)
        res, outs = [], []
        best = [1e9, 1e9, 1e9]
        for rnd in range(3):
            for mi, mode in enumerate((0, 1, 2)):
                L.ore_conv_set_plan_override(-2, mode, 0, 0, 0)
                for _ in range(3):
                    run()
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(40):
                    run()
                e.record()
                torch.cuda.synchronize()
                best[mi] = min(best[mi], a.elapsed_time(e) / 40 * 1e3)
                if rnd == 0:
                    outs.append(out.clone())
        err = max(float((outs[i] - outs[0]).abs().max() / outs[0].abs().max()) for i in (1, 2))
        for i in range(3):
            tot[i] += best[i]
        print("%-8s %9.2f %9.2f %9.2f   %.3f  %.1f   relerr %.1e" % (name, best[0], best[1], best[2], flops / 1e9, flops / min(best) / 1e6, err), flush=True)
    L.ore_conv_set_plan_override(-2, 1, 0, 0, 0)
    print("%-8s %9.2f %9.2f %9.2f" % ("sum", *tot))


if __name__ == "__main__":
    main()
