#!/usr/bin/env python3
"""Do two parallel branches of a hipGraph overlap on this ROCm?  A lateral-like 1x1 conv (15 us, many blocks) beside a chain of small
3x3 convs (stage-4/5-like, ~10 us each at 25-100 blocks), captured once as one serial stream and once as a fork / join of two streams.
usage: python tools/graph_branch_exp.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch, orehip
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def mk(Cin, Cout, k, H, W):
    x = torch.randn(1, H, W, Cin, generator=g).to(dev)
    w = orehip.pack_conv_weight(torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dev)
    out = torch.empty(1, H, W, Cout, device=dev)
    return lambda: orehip.conv2d(x, w, Cout, k, 1, out=out)
lat = mk(256, 128, 1, 80, 80)
chain = [mk(96, 96, 3, 40, 40) for _ in range(4)] + [mk(112, 112, 3, 20, 20) for _ in range(4)]
def serial():
    lat()
    for f in chain: f()
s2 = torch.cuda.Stream()
def forked():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        lat()
    for f in chain: f()
    cur.wait_stream(s2)
def only_chain():
    for f in chain: f()
res = {}
for name, fn in (("chain only", only_chain), ("lateral only", lat), ("serial", serial), ("fork / join", forked)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    a.record()
    for _ in range(n): gr.replay()
    b.record(); torch.cuda.synchronize()
    res[name] = a.elapsed_time(b) * 1e3 / n
    print("%-14s %8.1f us per replay" % (name, res[name]), flush=True)
