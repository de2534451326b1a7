#!/usr/bin/env python3
"""Is the in-flight eval loop GPU-bound or host-bound?  Times the enqueue loop alone (no sync) and the loop + sync, for 1/2/4/8
engines; also the same with one host thread per engine (ctypes releases the GIL inside the HIP calls)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
if len(sys.argv) > 1:
    model.conv_operands = sys.argv[1]
imgs = [bench.synth_image(i).to(dev) for i in range(4)]
K = 800
for n in (1, 2, 4, 8):
    engines = [model.make_engine() for _ in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]

    def step(i):
        with torch.cuda.stream(streams[i % n]):
            engines[i % n].eval_forward(imgs[i % 4], use_graph=True)
    for i in range(4 * n):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()

    def worker(j):
        with torch.cuda.stream(streams[j]):
            for i in range(K // n):
                engines[j].eval_forward(imgs[i % 4], use_graph=True)
    th = [threading.Thread(target=worker, args=(j,)) for j in range(n)]
    t3 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("engines %d: single host thread: enqueue %.0f us/img, total %.0f us/img (%.0f img/s) | one thread per engine: enqueue %.0f, total %.0f us/img (%.0f img/s)" % (
        n, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, K / (t2 - t0), (t4 - t3) / K * 1e6, (t5 - t3) / K * 1e6, K / (t5 - t3)), flush=True)
    for e in engines:
        e.close()
