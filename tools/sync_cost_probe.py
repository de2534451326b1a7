#!/usr/bin/env python3
"""What an idle torch.cuda.synchronize() costs the host, before and after the detector (its engine, streams, events) exists --
the protocol's per-image sync (d2z:evaluation/evaluator.py:151-161) is on the critical path of the headline number."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402


def idle_sync_us(n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


x = torch.zeros(8, device="cuda")
print("bare process, idle device:            %.2f us per torch.cuda.synchronize()" % idle_sync_us())
s = [torch.cuda.Stream() for _ in range(4)]
print("+ 4 more streams (unused):            %.2f us" % idle_sync_us())
for q in s:
    with torch.cuda.stream(q):
        x.add_(1)
print("+ those streams used once:            %.2f us" % idle_sync_us())
import bench  # noqa: E402
model = bench.build_model("cuda")[0]
model.eval()
img = bench.synth_image(7).cuda()
inp = [{"image": img, "height": 640, "width": 640}]
with torch.no_grad():
    for _ in range(5):
        model(inp)
print("+ the detector after 5 eval calls:    %.2f us" % idle_sync_us())
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(300):
        model(inp)
        torch.cuda.synchronize()
print("protocol loop: %.1f us per image" % ((time.perf_counter() - t0) / 300 * 1e6))
