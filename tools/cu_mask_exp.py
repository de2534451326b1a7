#!/usr/bin/env python3
"""Experiment: 4 bs=1 engines on HIP streams restricted to disjoint CU sets (hipExtStreamCreateWithCUMask) vs the plain 4 streams.
usage: cu_mask_exp.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
imgs = [bench.synth_image(i).to(dev) for i in range(4)]
K = 600


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(((1 if b in bits else 0) << (b - 32 * w)) for b in range(32 * w, 32 * w + 32)) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(streams, label):
    n = len(streams)
    engines = [model.make_engine() for _ in range(n)]

    def step(i):
        with torch.cuda.stream(streams[i % n]):
            engines[i % n].eval_forward(imgs[i % 4], use_graph=True)
    for i in range(4 * n):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    torch.cuda.synchronize()
    print("%-40s %8.1f img/s" % (label, K / (time.perf_counter() - t0)), flush=True)
    for e in engines:
        e.close()


run([torch.cuda.Stream(dev) for _ in range(4)], "4 plain streams")
allb = list(range(256))
if len(sys.argv) > 1 and sys.argv[1] == "fine":
    run([masked_stream(set(b for b in allb if b % 8 == i)) for i in range(8)], "8 eighths (bit mod 8), 8 streams")
    run([masked_stream(set(b for b in allb if b % 8 == i // 2 * 1 + 0 * i)) for i in range(8)], "8 streams on 4 eighths pairs")
    run([masked_stream(set(b for b in allb if b % 4 == i % 4)) for i in range(8)], "4 quarters (bit mod 4), 2 streams each")
    run([masked_stream(set(b for b in allb if b % 16 == i)) for i in range(16)], "16 sixteenths (bit mod 16), 16 streams")
    run([masked_stream(set(b for b in allb if b % 6 == i)) for i in range(6)], "6 sixths (bit mod 6), 6 streams")
    sys.exit(0)
run([masked_stream(set(allb[i * 128:(i + 1) * 128])) for i in range(2)] * 2, "2 halves (bit ranges), 2 streams each")
run([masked_stream(set(b for b in allb if b % 2 == i)) for i in range(2)] * 2, "2 halves (even/odd bits), 2 streams each")
run([masked_stream(set(allb[i * 64:(i + 1) * 64])) for i in range(4)], "4 quarters (bit ranges)")
run([masked_stream(set(b for b in allb if b % 4 == i)) for i in range(4)], "4 quarters (bit mod 4)")
run([masked_stream(set(b for b in allb if (b // 32) % 2 == i)) for i in range(2)] * 2, "2 halves (alternating 32-bit words)")
