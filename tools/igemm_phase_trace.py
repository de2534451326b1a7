#!/usr/bin/env python3
"""Phase timeline of k_conv_igemm (trace build: make -C faster-orefsdet_amd/csrc trace).  Per K step (thread 0 of every block, median over
blocks): issue of the prefetch loads -> [LDS fragment reads + MFMAs] -> [park next tile in LDS] -> [barrier].
usage: igemm_phase_trace.py H W Cin Cout k stride [fp32|bf16]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orehip  # noqa: E402

orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", "libore_hip_trace.so")
H, W, Cin, Cout, k, stride = (int(v) for v in sys.argv[1:7])
mode = sys.argv[7] if len(sys.argv) > 7 else "fp32"
dev = torch.device("cuda")
x = torch.randn(1, H, W, Cin, device=dev)
w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
orehip.set_conv_precision(mode)
orehip.lib().ore_conv_set_plan_override(-1, 0, 0, 0, 0)          # no patch kernels: the generic implicit-GEMM kernel
for _ in range(3):
    y = orehip.conv2d(x, w, Cout, k, stride)
nb = 8192
buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
orehip.lib().ore_debug_set_trace(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
y = orehip.conv2d(x, w, Cout, k, stride)
torch.cuda.synchronize()
orehip.lib().ore_debug_set_trace(C.c_void_p(0))
t = buf.cpu().numpy().reshape(nb, 64).astype(np.int64)
t = t[t[:, 62] > 0]
med = lambda a: int(np.median(a))  # noqa: E731
print("# %dx%d Cin %d Cout %d k %d stride %d, %s: %d blocks (blockIdx.x) stamped; shader clocks" % (H, W, Cin, Cout, k, stride, mode, len(t)))
print("block duration median %d; prologue (index setup .. first tile in LDS) %d; epilogue %d" % (
    med(t[:, 62] - t[:, 0]), med(t[:, 1] - t[:, 0]), med(t[:, 62] - t[:, 61])))
print("step  loads-issue  frag-reads+MFMA  park-next-tile  barrier   total")
prev = t[:, 1]
tot = [0, 0, 0, 0]
n = 0
for st in range(14):
    b = 2 + 4 * st
    if b + 3 >= 61 or not (t[:, b] > 0).all():
        break
    d = [med(t[:, b] - prev), med(t[:, b + 1] - t[:, b]), med(t[:, b + 2] - t[:, b + 1]), med(t[:, b + 3] - t[:, b + 2])]
    print("%4d %11d %16d %15d %8d %7d" % (st, d[0], d[1], d[2], d[3], sum(d)))
    tot = [a + c for a, c in zip(tot, d)]
    prev = t[:, b + 3]
    n += 1
print("mean over %d steps: loads-issue %d  frag+MFMA %d  park %d  barrier %d" % (n, tot[0] // n, tot[1] // n, tot[2] // n, tot[3] // n))
