#!/usr/bin/env python3
"""Where the host time of the reference FPS protocol goes: cProfile of model([{image,height,width}]) + synchronize on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model, cfg = bench.build_model(dev)
    img = bench.synth_image(0).to(dev)
    req = [{"image": img, "height": 640, "width": 640}]
    for _ in range(20):
        model(req)
        torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        model(req)
        torch.cuda.synchronize()
    print("ms per call", (time.perf_counter() - t0) / n * 1e3)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        model(req)
        torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
