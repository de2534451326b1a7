#!/usr/bin/env python3
"""Fixed cost vs per-K-step cost of the implicit-GEMM conv on small (latency-bound) layers: time 1x1 and 3x3 convs over a sweep
of Cin at fixed M, N inside a hipGraph of 20 back-to-back launches (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402


def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


def main():
    dev = torch.device("cuda")
    for (H, W, Cout, k) in ((80, 80, 80, 1), (80, 80, 80, 3), (20, 20, 112, 1), (20, 20, 112, 3), (40, 40, 96, 3)):
        line = []
        for Cin in (16, 64, 128, 256, 512):
            x = torch.randn(1, H, W, Cin, device=dev)
            w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
            out = torch.empty(1, H, W, Cout, device=dev)
            ws = torch.zeros(orehip.lib().ore_conv_workspace_floats(), device=dev)
            us = timeit(lambda: orehip.conv2d(x, w, Cout, k, 1, out=out, workspace=ws))
            line.append("Cin=%d K=%d: %.1f us" % (Cin, Cin * k * k, us))
        print("M=%d N=%d k=%d | " % (H * W, Cout, k) + " | ".join(line), flush=True)
    # an empty-ish kernel for the launch floor inside a graph
    a = torch.zeros(64, device=dev)
    print("graph floor (tiny torch add): %.2f us" % timeit(lambda: a.add_(1.0)))


if __name__ == "__main__":
    main()
