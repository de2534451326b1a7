#!/usr/bin/env python3
"""Train-step benchmark (SURVEY 8d metric ii): forward + backward + gradient exchange + clip/SGD, images/s at 640x640 with
24 support crops of 240x240 per query (finetune_vovnet.yaml), batch 1 per GPU like the reference.  Synthetic data and weights.

    python tools/bench_train.py --steps 10 --warmup 3               # 1 GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --gpus N ...
Prints one JSON line (rank 0).  Not the headline metric of BASELINE.json (bench.py measures that); recorded in DESIGN.md."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--shots", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1, help="query images per GPU per step (BASELINE configs[2]: 16)")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the shape-static dense part (fwd + bwd) into hipGraphs")
    ap.add_argument("--no-gd-large", action="store_true", help="A/B aid: k_conv_gd only up to M = 32768 as in round 4 (plan override -14 3)")
    ap.add_argument("--no-kd", action="store_true", help="A/B aid: keep the small-M layers on k_conv_kw (plan override -12 0)")
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"), help="bf16 = BASELINE configs[4]: frozen stages in bf16 storage, "
                    "bf16 MFMA operands in the trainable convs' forward / data / weight gradients, everything else fp32")
    ap.add_argument("--graph-step", action="store_true", help="the WHOLE iteration (forward, losses, backward, clip + SGD) as one replayed hipGraph "
                    "(fewx.solver.GraphedTrainStep; single process)")
    ap.add_argument("--roi-bwd", default=None, choices=("tiled", "fixed", "atomic"), help="A/B aid: the ROIAlign backward form (default: the library's)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl")
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.structures import Boxes, Instances
    from fewx.solver import FlatDataParallel, build_lr_scheduler, build_optimizer
    from oracle import ref_model as R                      # synthetic weights / inputs only (not the thing measured)
    from oracle import ref_train as T
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "faster-orefsdet_amd", "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", a.shots])
    cfg.freeze()
    torch.manual_seed(0)
    m = build_model(cfg)
    sd = R.synth_roi_state(R.synth_state_dict(0), 0)
    sd["roi_heads.box_head.0.fc1.weight"] *= 0.02
    m.load_state_dict(sd, strict=False)
    m.train()
    m.train_graph = bool(a.graph)
    import orehip
    orehip.set_conv_precision(a.precision)
    if a.roi_bwd:
        orehip.ROI_BWD_DETERMINISTIC, orehip.ROI_BWD_MODE = a.roi_bwd != "atomic", "tiled" if a.roi_bwd == "tiled" else "fixed"
    if os.environ.get("ORE_GD_MODE"):
        orehip.lib().ore_conv_set_plan_override(-14, int(os.environ["ORE_GD_MODE"]), 0, 0, 0)
    if a.no_gd_large:
        orehip.lib().ore_conv_set_plan_override(-14, 3, 0, 0, 0)
    if a.no_kd:
        orehip.lib().ore_conv_set_plan_override(-12, 0, 0, 0, 0)
        orehip.lib().ore_conv_set_plan_override(-10, 0, 0, 0, 0)
    model = FlatDataParallel(m, cfg, overlap=not a.no_overlap) if world > 1 else m
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    items = []
    for b in range(a.batch):
        img, gt, sup, sbox = T.synth_train_inputs(rank * a.batch + b, (a.size, a.size), n_gt=17, shots=a.shots, support_hw=240)
        inst = Instances((a.size, a.size))
        inst.gt_boxes, inst.gt_classes = Boxes(gt.cuda()), torch.zeros(len(gt), dtype=torch.int64).cuda()
        items.append({"image": img.cuda(), "instances": inst, "support_images": sup.cuda(), "support_bboxes": sbox.numpy()})

    stepper = None
    if a.graph_step:
        assert world == 1, "--graph-step is single-process"
        from fewx.solver import GraphedTrainStep
        stepper = GraphedTrainStep(m, opt)

    def step():
        if stepper is not None:
            losses = stepper(items)
            sched.step()
            return losses
        losses = model(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()
        return losses

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "train_images_per_second", "value": world * a.batch * a.steps / el, "unit": "img/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * el / a.steps, "higher_is_better": True,
                          "scaling": "weak", "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
                          "config": {"workload": "finetune_vovnet.yaml train step, %d x (1 query %dx%d + %d support 240x240) per GPU" % (a.batch, a.size, a.size, a.shots),
                                     "batch_per_gpu": a.batch,
                                     "bucket_bytes": 4 * opt.bucket.size},
                          "train_graph": bool(a.graph), "train_graph_error": m.__dict__.get("_ore_train_graph_error"),
                          "whole_step_graph": None if stepper is None else {"replays": stepper.replays, "eager_steps": stepper.eager_steps, "error": stepper.error},
                          "losses": {k: float(v.detach()) for k, v in losses.items()}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
