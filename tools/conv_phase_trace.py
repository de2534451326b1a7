#!/usr/bin/env python3
"""Phase timeline of k_conv3x3_patch<4> (make -C faster-orefsdet_amd/csrc trace -> lib/libore_hip_trace.so): s_memtime stamps of
thread 0 of every block at the phase boundaries -> per-phase cycles (median over blocks).  usage: conv_phase_trace.py [blocks_y tiles_x Cin]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orehip  # noqa: E402

orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", "libore_hip_trace.so")
ty, tx, Cin = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 16, 128)
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
Cout, k = 64, 3
dev = torch.device("cuda")
H, W = 4 * ty, 16 * tx
nb = ty * tx
x = torch.randn(1, H, W, Cin, device=dev)
w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k)).to(dev)
out = torch.empty(1, H, W, Cout, device=dev)
orehip.set_conv_precision(mode)
orehip.lib().ore_conv_set_plan_override(-1, 4, 0, 0, 0)
for _ in range(3):
    orehip.conv2d(x, w, Cout, k, 1, out=out)
buf = torch.zeros(nb * 64, dtype=torch.int64, device=dev)
orehip.lib().ore_debug_set_trace(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
orehip.conv2d(x, w, Cout, k, 1, out=out)
torch.cuda.synchronize()
orehip.lib().ore_debug_set_trace(C.c_void_p(0))
t = buf.cpu().numpy().reshape(nb, 64).astype(np.int64)
t0 = t[:, 0].min()
ns = Cin // 16
print("# %d blocks (%dx%d), Cin %d, %s; cycles are s_memtime ticks" % (nb, H, W, Cin, mode))
print("kernel span (first start .. last end): %d ticks; block start spread %d; block duration median %d (min %d max %d)" % (
    t[:, 62].max() - t0, t[:, 0].max() - t0, np.median(t[:, 62] - t[:, 0]), (t[:, 62] - t[:, 0]).min(), (t[:, 62] - t[:, 0]).max()))
med = lambda a: int(np.median(a))  # noqa: E731
print("prologue (start -> first gload issued): %d" % med(t[:, 1] - t[:, 0]))
rows = []
for sl in range(ns):
    b = 2 + sl * 4
    prev = t[:, 1] if sl == 0 else t[:, b - 1]
    rows.append((sl, med(t[:, b] - prev), med(t[:, b + 1] - t[:, b]), med(t[:, b + 2] - t[:, b + 1]), med(t[:, b + 3] - t[:, b + 2])))
print("slab  lstore(+wait for the slab's global loads)  barrier1  gload-issue+9 taps (LDS reads + MFMA)  barrier2")
for r in rows:
    print("%4d %12d %28d %22d %22d" % r)
print("epilogue (last barrier -> stores issued): %d" % med(t[:, 62] - t[:, 2 + (ns - 1) * 4 + 3]))
tot = [sum(r[i] for r in rows) for i in range(1, 5)]
print("sum over slabs: lstore %d  barrier1 %d  taps %d  barrier2 %d" % tuple(tot))
