#!/usr/bin/env python3
"""Which torch-native (aten) operators one EAGER training step of the bs-16 configuration dispatches, by call site: the small
launches between the library's own kernels.  TorchDispatchMode sees every operator that reaches the dispatcher (forward, autograd's
backward and the optimizer); the first frame inside faster-orefsdet_amd/ (or bench/tools) names the call site, "<backward>" when
the operator is issued by the autograd engine.

    python tools/train_glue_probe.py [--batch 16] > gpurun_out/train_glue.txt"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

PKG = os.path.join(ROOT, "faster-orefsdet_amd")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.numel = collections.Counter()
        self.shapes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in ("aten.view", "aten._unsafe_view", "aten.expand", "aten.permute", "aten.t.", "aten.transpose", "aten.slice",
                                   "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.reshape",
                                   "aten.split", "aten.unbind", "aten.empty", "aten.sym_", "aten.is_", "aten.stride", "aten.size", "aten.narrow",
                                   "aten.lift_fresh", "aten._local_scalar", "aten.result_type", "aten.new_empty")):
            return out
        site = "<backward / engine>"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if fr.filename.startswith(PKG) or fr.filename.endswith("bench.py"):
                site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                break
        n = out.numel() if isinstance(out, torch.Tensor) else -1
        self.rows[(site, name)] += 1
        self.numel[(site, name)] = max(self.numel[(site, name)], n)
        if site.startswith("<backward") and "aten.add" in name and isinstance(out, torch.Tensor):
            self.shapes[(name, tuple(out.shape))] += 1
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    import bench
    import orehip
    from fewx.solver import build_lr_scheduler, build_optimizer
    model, cfg = bench.build_model("cuda")
    model.train()
    model.train_graph = False
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    from detectron2.structures import Boxes, Instances
    g = torch.Generator().manual_seed(1)
    items = []
    for b in range(a.batch):
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

    def step():
        losses = model(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with Count() as c:
        step()
    torch.cuda.synchronize()
    tot = sum(c.rows.values())
    print("# aten operators of one eager training step (batch %d), views excluded: %d, ore_version %d" % (a.batch, tot, orehip.lib().ore_version()))
    by_site = collections.Counter()
    for (site, name), n in c.rows.items():
        by_site[site] += n
    print("## by call site")
    for site, n in by_site.most_common():
        print("%5d  %s" % (n, site))
    print("## the autograd engine's adds by shape (sums of gradients of tensors with several consumers; in-place adds into .grad)")
    for (name, shape), n in sorted(c.shapes.items(), key=lambda kv: -kv[1]):
        print("%5d  %-22s %s" % (n, name, shape))
    print("## by (call site, operator)   count  max numel")
    for (site, name), n in sorted(c.rows.items(), key=lambda kv: -kv[1]):
        print("%5d %10d  %-70s %s" % (n, c.numel[(site, name)], site, name))


if __name__ == "__main__":
    main()
