#!/usr/bin/env python3
"""How much of a GEMM-shaped layer's time is tile imbalance?  The stage-3 concat shape (K = 352, N = 256) and the stage-2 concat shape
(K = 320, N = 112) on k_conv_gd at row counts that give exactly 1, ~1.56 (the real layer) and 2 tiles per CU; 20 launches captured in a
hipGraph, replay time / 20.  GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip  # noqa: E402

dev = torch.device("cuda")


def timed(M, Cin, Cout, reps=20):
    x = torch.randn(1, 1, M, Cin, device=dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5).to(dev)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    out = torch.empty(1, 1, M, Cout, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            orehip.conv2d(x, w, Cout, 1, 1, scale=sc, shift=sh, relu_cout=Cout, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                orehip.conv2d(x, w, Cout, 1, 1, scale=sc, shift=sh, relu_cout=Cout, out=out)
    for _ in range(3):
        g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 10 / reps * 1e3


for name, Cin, Cout, tile_rows, ntile_n, Ms in (("s3cat 352->256 (64x64 tiles, 4 column tiles)", 352, 256, 64, 4, (4096, 6400, 8192)),
                                               ("s2cat 320->112 (128x64 tiles, 2 column tiles)", 320, 112, 128, 2, (16384, 25600, 32768)),
                                               ("lat3 / conv3 256->128", 256, 128, 64, 2, (4096, 6400, 8192, 8400))):
    for M in Ms:
        tiles = -(-M // tile_rows) * ntile_n
        us = timed(M, Cin, Cout)
        peak = 2.0 * M * Cin * Cout / 157.3e12 * 1e6
        print("%-48s M %6d  tiles %4d (%.2f per CU)  %7.2f us   MFMA at peak %5.2f us  -> %.0f %%" % (name, M, tiles, tiles / 256, us, peak, 100 * peak / us))
