#!/usr/bin/env python3
"""A/B aid: engine replay time per image (back to back, no per-image sync) with an alternative build of the library:
ORE_LIB=libore_hip_x1.so python tools/lib_ab.py [kernel-name-substring ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import orehip
if os.environ.get("ORE_LIB"):
    orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", os.environ["ORE_LIB"])
import torch, bench
model, cfg = bench.build_model(torch.device("cuda", 0))
imgs = [bench.synth_image(i).cuda() for i in range(4)]
for i in range(30):
    model([{"image": imgs[i % 4], "height": 640, "width": 640}]); torch.cuda.synchronize()
eng = model._engine
best = 1e9
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300):
        eng.eval_forward(imgs[i % 4])
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 300)
print("%-24s %.1f us per image back to back" % (os.environ.get("ORE_LIB", "libore_hip.so"), best * 1e6))
