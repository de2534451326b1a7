#!/usr/bin/env python3
"""stem_1 alone at the three sizes of the path: one eval image, the 16 query images and the 384 support crops of a bs-16 training step."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "faster-orefsdet_amd"))
import torch  # noqa: E402
import orehip as oh  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


w = torch.randn(64, 3, 3, 3).cuda() * 0.1
sc, sh = torch.rand(64).cuda() + 0.5, torch.randn(64).cuda() * 0.1
print("# ore_version %d" % oh.lib().ore_version())
for name, x in (("eval 1x640x640 u8", torch.randint(0, 256, (1, 3, 640, 640), dtype=torch.uint8).cuda()),
                ("query 16x640x640 u8", torch.randint(0, 256, (16, 3, 640, 640), dtype=torch.uint8).cuda()),
                ("support 384x240x240 f32", torch.rand(384, 3, 240, 240).cuda() * 255)):
    Hp = (x.shape[-2] + 31) // 32 * 32
    for std in ((1.0, 1.0, 1.0), (57.375, 57.12, 58.395)):
        us = t(lambda: oh.stem1(x, Hp, Hp, (103.53, 116.28, 123.675), std, w, sc, sh))
        print("%-26s std %-24s %8.1f us" % (name, std, us))
