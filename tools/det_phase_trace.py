#!/usr/bin/env python3
"""Phase stamps (s_memtime) of the first-stage detection tail at the bench shape: k_level_select (block 0) and the consumer wave of
k_nms_scan_t, from the trace build of the library (make -C faster-orefsdet_amd/csrc trace).  GPU."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orehip  # noqa: E402

orehip.LIB_PATH = os.path.join(ROOT, "faster-orefsdet_amd", "lib", "libore_hip_trace.so")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model, cfg = bench.build_model(dev)
e = model.engine()
img = bench.synth_image(1).to(dev)
L = orehip.lib()
for _ in range(3):
    e.eval_forward(img, use_graph=False)
torch.cuda.synchronize()
buf = torch.zeros(512, dtype=torch.int64, device=dev)
rbuf = torch.zeros(64, dtype=torch.int64, device=dev)
L.ore_debug_set_trace_roi(C.c_void_p(rbuf.data_ptr()))
L.ore_debug_set_trace_det(C.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
e.eval_forward(img, use_graph=False)
torch.cuda.synchronize()
L.ore_debug_set_trace_det(C.c_void_p(0))
L.ore_debug_set_trace_roi(C.c_void_p(0))
rt = rbuf.cpu().numpy().astype(np.int64)
print("k_roi_tail, clocks since entry:", {n: int(rt[i] - rt[0]) for i, n in enumerate(
    ["entry", "count read", "compaction", "rank sort", "IoU bits", "greedy", "detections", "postprocess + record"]) if rt[i]})
t = buf.cpu().numpy().astype(np.int64)
c = e.buffer("counts").cpu().numpy().ravel()
print("counts (n_pre, n_keep):", c[:2])
ls = t[:9]
print("k_level_select block 0, clocks since entry:", {n: int(ls[i] - ls[0]) for i, n in enumerate(
    ["entry", "sigmoid+count", "scan", "radix0", "radix1", "radix2", "own counts", "scan2", "emit"]) if ls[i]})
s0 = t[16]
print("k_nms_scan_t: init barrier %d, scores %d, end %d (clocks since entry)" % (t[17] - s0, t[18] - s0, t[19] - s0))
nb = (int(c[0]) + 63) // 64
rows = []
for w in range(nb):
    a, b_, d = t[20 + 3 * w], t[21 + 3 * w], t[22 + 3 * w]
    if a == 0:
        break
    prev = t[22 + 3 * (w - 1)] if w else t[18]
    rows.append((w, int(a - prev), int(b_ - a), int(d - b_), int(a - s0)))
print("block: wait-for-column / AND loop / fixpoint+bookkeeping (clocks); at")
for r in rows:
    print("  %2d: %6d %6d %6d   @%7d" % r)
tot = np.array([[r[1], r[2], r[3]] for r in rows]).sum(0)
print("sums: wait %d, AND %d, rest %d clocks over %d blocks" % (tot[0], tot[1], tot[2], len(rows)))
print("producers: column: issue start @, landed @ (clocks since kernel entry), latency")
for c in range(nb):
    a, b_ = t[128 + 2 * c], t[129 + 2 * c]
    if a:
        print("  col %2d (wave %2d): %7d %7d  %6d" % (c, 1 + c % 15, a - s0, b_ - s0, b_ - a))
print("fixpoint iterations per block:", [int(x) for x in t[320:320 + len(rows)]])
