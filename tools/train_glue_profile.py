#!/usr/bin/env python3
"""Where do the ATen element-wise / copy / fill kernels of a bs = 16 training step come from?  torch.profiler with stacks, grouped by
the innermost frame inside this repository (or, where the profiler has no Python frames, by operand shapes).  usage: python tools/train_glue_profile.py [batch]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
# reuse bench.train_leg's setup by monkeypatching its timing away: build the step closure ourselves
from detectron2.structures import Boxes, Instances
from fewx.solver import build_lr_scheduler, build_optimizer
model, cfg = bench.build_model(dev)
model.train(); model.train_graph = False
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for n, p in model.named_parameters():
        if n.startswith("roi_heads.") and p.dim() > 1:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).to(dev) * (p[0].numel() ** -0.5))
opt = build_optimizer(cfg, model); sched = build_lr_scheduler(cfg, opt)
items = []
for b in range(batch):
    wh = torch.rand(17, 2, generator=g) * 120 + 30
    ctr = torch.rand(17, 2, generator=g) * (640 - wh) + wh / 2
    inst = Instances((640, 640)); inst.gt_boxes = Boxes(torch.cat([ctr - wh / 2, ctr + wh / 2], 1).to(dev)); inst.gt_classes = torch.zeros(17, dtype=torch.int64, device=dev)
    sup = torch.stack([bench.synth_image(100 + 50 * b + i, 240, 240) for i in range(24)]).to(dev)
    side = torch.rand(24, 2, generator=g) * 120 + 80
    c = torch.rand(24, 2, generator=g) * (240 - side) + side / 2
    items.append({"image": bench.synth_image(7 + b, 640, 640).to(dev), "instances": inst, "support_images": sup, "support_bboxes": torch.cat([c - side / 2, c + side / 2], 1).numpy()})

def step():
    losses = model(items); opt.zero_grad(); sum(losses.values()).backward(); opt.step(); sched.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    t = getattr(ev, "self_device_time_total", getattr(ev, "self_cuda_time_total", 0))
    if t <= 0:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if ("faster-orefsdet_amd" in fr or "/bench.py" in fr) and "orehip/__init__" not in fr:
            site = fr.split("faster-orefsdet_amd/")[-1][:90]
            break
    if site == "?":                                       # no Python frames (this torch build): the operand shapes tell the call site
        site = str([tuple(x) for x in (ev.input_shapes or []) if x])[:110]
    k = (ev.name, site)
    agg[k][0] += t; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("ATen ops with device time: %.2f ms per step" % (tot / 1e3))
for (name, site), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.3f ms %5d x  %-28s %s" % (t / 1e3, n, name, site))
