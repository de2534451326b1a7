#!/usr/bin/env python3
"""Every conv / linear / weight-gradient call of ONE eager training step (bs 16 by default), timed one by one with stream events:
shape, algorithmic GFLOP, microseconds, TFLOP/s -- the per-layer table of the training step (the eval path has
tools/conv_layers_table.py).  The calls are intercepted at the Python entry points orehip.conv2d / orehip.conv2d_wgrad; the
event pair adds a few microseconds per call and serialises nothing that was not already serial (one stream).

    python tools/train_conv_table.py [--batch 16] [--precision fp32] > gpurun_out/train_conv_table.txt"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))

import torch  # noqa: E402


def build_items(bench, batch):
    from detectron2.structures import Boxes, Instances
    g = torch.Generator().manual_seed(1)
    items = []
    for b in range(batch):
        wh = torch.rand(17, 2, generator=g) * 120 + 30
        ctr = torch.rand(17, 2, generator=g) * (640 - wh) + wh / 2
        inst = Instances((640, 640))
        inst.gt_boxes = Boxes(torch.cat([ctr - wh / 2, ctr + wh / 2], 1).cuda())
        inst.gt_classes = torch.zeros(17, dtype=torch.int64, device="cuda")
        sup = torch.stack([bench.synth_image(100 + 50 * b + i, 240, 240) for i in range(24)]).cuda()
        side = torch.rand(24, 2, generator=g) * 120 + 80
        c = torch.rand(24, 2, generator=g) * (240 - side) + side / 2
        items.append({"image": bench.synth_image(7 + b, 640, 640).cuda(), "instances": inst, "support_images": sup,
                      "support_bboxes": torch.cat([c - side / 2, c + side / 2], 1).numpy()})
    return items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"))
    a = ap.parse_args()
    import bench
    import orehip
    from fewx.solver import build_lr_scheduler, build_optimizer
    orehip.set_conv_precision(a.precision)
    model, cfg = bench.build_model("cuda")
    model.train()
    model.train_graph = False
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    items = build_items(bench, a.batch)

    def step():
        losses = model(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rows = []
    real_conv, real_wgrad = orehip.conv2d, orehip.conv2d_wgrad

    def conv2d(x, w_packed, Cout, k, stride=1, pad=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = real_conv(x, w_packed, Cout, k, stride, pad, **kw)
        e1.record()
        B, H, W, ld = x.shape
        Cin = kw.get("Cin") or ld - kw.get("in_coff", 0)
        Ho, Wo = y.shape[1], y.shape[2]
        rows.append(("conv", (B, H, W), Cin, Cout, k, stride, 2.0 * B * Ho * Wo * Cout * Cin * k * k, e0, e1,
                     "wino" if kw.get("w_wino") is not None else ""))
        return y

    def conv2d_wgrad(x, dz, k, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_wgrad(x, dz, k, **kw)
        e1.record()
        B, H, W, ld = x.shape
        Cin = kw.get("Cin") or ld - kw.get("x_coff", 0)
        Cout = kw.get("Cout") or dz.shape[-1] - kw.get("dz_coff", 0)
        rows.append(("wgrad", (B, H, W), Cin, Cout, k, 1, 2.0 * B * H * W * Cout * Cin * k * k, e0, e1, ""))
        return r

    orehip.conv2d, orehip.conv2d_wgrad = conv2d, conv2d_wgrad
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    step()
    s1.record()
    torch.cuda.synchronize()
    orehip.conv2d, orehip.conv2d_wgrad = real_conv, real_wgrad
    tot_us = sum(r[7].elapsed_time(r[8]) * 1e3 for r in rows)
    tot_gf = sum(r[6] for r in rows) / 1e9
    print("# conv / linear / weight-gradient calls of one eager training step, batch %d, %s, ore_version %d" % (a.batch, a.precision, orehip.lib().ore_version()))
    print("# %d calls, %.1f GFLOP, %.2f ms in these calls of %.2f ms for the step (eager, events per call)" % (len(rows), tot_gf, tot_us / 1e3, s0.elapsed_time(s1)))
    print("%-6s %-18s %5s %5s %2s %2s %9s %9s %8s %s" % ("kind", "B,H,W", "Cin", "Cout", "k", "s", "GFLOP", "us", "TFLOP/s", ""))
    agg = collections.OrderedDict()
    for kind, shp, Cin, Cout, k, s, fl, e0, e1, note in rows:
        us = e0.elapsed_time(e1) * 1e3
        key = (kind, shp, Cin, Cout, k, s, note)
        v = agg.setdefault(key, [0, 0.0, 0.0])
        v[0] += 1; v[1] += fl; v[2] += us
    for (kind, shp, Cin, Cout, k, s, note), (n, fl, us) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        print("%-6s %-18s %5d %5d %2d %2d %9.2f %9.1f %8.1f %s x%d" % (kind, "%d,%d,%d" % shp, Cin, Cout, k, s, fl / 1e9, us, fl / us / 1e6, note, n))


if __name__ == "__main__":
    main()
