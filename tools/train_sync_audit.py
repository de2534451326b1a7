#!/usr/bin/env python3
"""Which calls of one training step synchronise the host with the GPU?  Runs two warm steps, then one step under
torch.cuda.set_sync_debug_mode("warn") and prints every warning with the innermost frames of this repository.
    python tools/train_sync_audit.py [--batch 2]"""
import argparse
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=320)
    a = ap.parse_args()
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.structures import Boxes, Instances
    from fewx.solver import build_lr_scheduler, build_optimizer
    from oracle import ref_model as R
    from oracle import ref_train as T
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "faster-orefsdet_amd", "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", 4])
    cfg.freeze()
    m = build_model(cfg)
    sd = R.synth_roi_state(R.synth_state_dict(0), 0)
    sd["roi_heads.box_head.0.fc1.weight"] *= 0.02
    m.load_state_dict(sd, strict=False)
    m.train()
    opt = build_optimizer(cfg, m)
    sched = build_lr_scheduler(cfg, opt)
    items = []
    for b in range(a.batch):
        img, gt, sup, sbox = T.synth_train_inputs(b, (a.size, a.size), n_gt=9, shots=4, support_hw=112)
        inst = Instances((a.size, a.size))
        inst.gt_boxes, inst.gt_classes = Boxes(gt.cuda()), torch.zeros(len(gt), dtype=torch.int64).cuda()
        items.append({"image": img.cuda(), "instances": inst, "support_images": sup.cuda(), "support_bboxes": sbox.numpy()})

    def step():
        losses = m(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    seen = {}

    def show(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if ROOT in f.filename and "train_sync_audit" not in f.filename]
        key = tuple((os.path.relpath(f.filename, ROOT), f.lineno) for f in st[-3:])
        seen[key] = seen.get(key, 0) + 1
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    step()
    torch.cuda.set_sync_debug_mode("default")
    print(f"{sum(seen.values())} synchronising calls in one step (batch {a.batch}):")
    for key, n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(f"  x{n:3d}  " + "  <-  ".join(f"{f}:{l}" for f, l in reversed(key)))


if __name__ == "__main__":
    main()
