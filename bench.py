#!/usr/bin/env python3
"""Faster-OreFSDet on MI355X -- headline benchmark (BASELINE.json: images/s at 640x640, 25-shot).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload at N=1 = BASELINE.json configs[1]: finetune_vovnet.yaml, 25-shot eval-only, bs=1, 640x640 synthetic image,
cached support prototypes, random-init weights (no dataset / checkpoint exists offline).  One "step" = one complete eval
forward of the detector (fused preprocess + VoVNet-19-slim-eSE + FPN -> query<->support correlation ->
CenterNet head -> sigmoid/top-k/decode/NMS proposals, then 8f row 1: ROIAlign -> support-guided mix + fc1 -> cls/box -> NMS ->
top-100 detections) over one image that is already resident in HBM, replayed as ONE hipGraph.  By default 4 such bs=1 forwards
are kept in flight per GPU (one engine + HIP stream each: bs=1 leaves most CUs idle in the small pyramid layers); the strictly
one-image-at-a-time rate is measured in the same run and printed as "sequential".
N>1: pure data parallel, every rank runs the same per-GPU work on its own images, no data-path collective (weak scaling);
RCCL (backend "nccl") is used only for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernel = the fp32 MFMA implicit-GEMM conv (k_conv_igemm): algorithmic FLOPs of all its launches in
               one image / their summed duration, measured live with HIP events on the launch stream (eager passes after
               the timed region), against the 157.3 TFLOP/s fp32 matrix peak of gfx950.
  cpu_baseline the CPU oracle (oracle/: plain-PyTorch fp32 + C decode/NMS restatement of the reference) timed on this
               box's host cores on a bounded sample of the same workload (kind "port").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "faster-orefsdet_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (same table); only used with --conv-operands bf16


def synth_image(seed, h=640, w=640):
    """uint8 BGR CHW ore-like texture (low-passed noise in [40,200]); content does not change the work."""
    g = torch.Generator().manual_seed(2000 + seed)
    x = torch.rand(1, 3, h // 4 + 2, w // 4 + 2, generator=g)
    x = torch.nn.functional.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
    x = x + 0.15 * (torch.rand(1, 3, h, w, generator=g) - 0.5)
    return (x.clamp(0, 1) * 160 + 40).round().to(torch.uint8)[0]


def build_model(device):
    """finetune_vovnet.yaml through the fewx registry surface, reference initialisers + non-trivial FrozenBN statistics."""
    from fewx.config import get_cfg
    from detectron2.modeling import build_model as _build
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "faster-orefsdet_amd", "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "INPUT.MAX_SIZE_TEST", 640])  # init on the host, then move
    cfg.freeze()
    torch.manual_seed(0)
    model = _build(cfg)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for name, mod in model.named_modules():
            if type(mod).__name__ == "FrozenBatchNorm2d":
                n = mod.num_features
                mod.weight.copy_(torch.rand(n, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(n, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
                mod.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            elif isinstance(mod, torch.nn.Conv2d) and "bottom_up" in name and mod.bias is None:
                torch.nn.init.kaiming_normal_(mod.weight, generator=g)
    model.to(device).eval()
    C = cfg.MODEL.FPN.OUT_CHANNELS
    support = {f"p{l}": {0: torch.randn(1, C, s, s, generator=g) * 0.1} for l, s in ((3, 32), (4, 16), (5, 8))}
    support["rcnn_8"] = {0: torch.randn(24, C, 8, 8, generator=g) * 0.1}
    support["rcnn_4"] = {0: torch.randn(24, C, 4, 4, generator=g) * 0.1}
    model.set_support_dict(support)
    return model, cfg


def cpu_baseline(model, img, budget_s=12.0):
    """Oracle (CPU restatement of the reference) on the host cores; bounded sample of the same workload."""
    from oracle import decode as odec
    from oracle import ref_model as R
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # a 1-GPU box's CPU share is 16 cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    support = {k: model.support_dict[k][0].cpu() for k in ("p3", "p4", "p5")}

    def one():
        with torch.no_grad():
            o = R.eval_dense(img, sd, support)
        hms = [h[0, 0].numpy() for h in o["hm"]]
        regs = [r[0].permute(1, 2, 0).contiguous().numpy() for r in o["reg"]]
        return odec.decode_nms(hms, regs, (8, 16, 32), 1e-5, 1000, 0.6, 256)

    for _ in range(2):
        one()
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    return {"value": round(n / el, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} images of the same 640x640 bs=1 eval workload (oracle/ref_model.py + oracle/ref_decode.c, "
                      f"torch {torch.__version__} CPU, {el:.1f} s)"}


def train_leg(device, steps=8, warmup=4, size=640, shots=24, batch=1, graph=True):
    """SURVEY 8d metric (ii), single GPU: forward + backward + clip/SGD of finetune_vovnet.yaml on one query + 24 support crops
    (tools/bench_train.py is the stand-alone / multi-GPU version).  Reported beside the headline, never as `value`."""
    from detectron2.structures import Boxes, Instances
    from fewx.solver import build_lr_scheduler, build_optimizer
    model, cfg = build_model(device)
    model.train()
    model.train_graph = graph       # the shape-static dense part (fwd + bwd) replays as two hipGraphs; falls back to eager if capture fails
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                   # second-stage weights: small, so the synthetic losses stay finite
        for n, p in model.named_parameters():
            if n.startswith("roi_heads.") and p.dim() > 1:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).to(device) * (p[0].numel() ** -0.5))
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    items = []
    for b in range(batch):
        wh = torch.rand(17, 2, generator=g) * 120 + 30
        ctr = torch.rand(17, 2, generator=g) * (size - wh) + wh / 2
        inst = Instances((size, size))
        inst.gt_boxes = Boxes(torch.cat([ctr - wh / 2, ctr + wh / 2], 1).to(device))
        inst.gt_classes = torch.zeros(17, dtype=torch.int64, device=device)
        sup = torch.stack([synth_image(100 + 50 * b + i, 240, 240) for i in range(shots)]).to(device)
        side = torch.rand(shots, 2, generator=g) * 120 + 80
        c = torch.rand(shots, 2, generator=g) * (240 - side) + side / 2
        items.append({"image": synth_image(7 + b, size, size).to(device), "instances": inst, "support_images": sup,
                      "support_bboxes": torch.cat([c - side / 2, c + side / 2], 1).numpy()})

    def step():
        losses = model(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()
        return losses

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"images_per_s": round(batch * steps / el, 2), "batch_per_gpu": batch, "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "warmup": warmup, "dtype": "f32",
            "workload": "finetune_vovnet.yaml train step on 1 GPU: %d x (1 query %dx%d + %d support 240x240), fwd + bwd (HIP backward kernels) + "
                        "flat-bucket clip/SGD, FREEZE_AT=3" % (batch, size, size, shots),
            "dense_part_hipgraph": bool(graph) and model.__dict__.get("_ore_train_graph_error") is None,
            "exchanged_bytes_per_step_if_dp": 4 * opt.bucket.size, "loss_sum": round(float(sum(v.detach() for v in losses.values())), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the single-GPU train-step measurement printed as \"train_step\"")
    ap.add_argument("--profile-passes", type=int, default=20)
    ap.add_argument("--conv-operands", choices=("fp32", "bf16"), default="fp32",
                    help="fp32 = the reference's precision (the headline).  bf16 = BASELINE configs[4]: MFMA conv operands rounded to bf16, "
                         "fp32 storage / accumulation / NMS (include/ore_hip.h ORE_CONV_BF16); reported with dtype \"bf16\", never the default")
    ap.add_argument("--fold-streams", type=int, default=4, help="engine passes kept in flight in the folded-serving leg")
    ap.add_argument("--fold", type=int, default=8,
                    help="also time the same requests folded F at a time into one engine pass (\"folded_serving\" in the output); 0/1 = skip")
    ap.add_argument("--inflight", type=int, default=4,
                    help="images kept in flight per GPU, each a bs=1 forward on its own engine + HIP stream (bs=1 leaves most of "
                         "the 256 CUs idle in the small pyramid layers; independent images fill them).  1 = strictly one image "
                         "at a time; that rate is always reported too (\"sequential\").")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    device = torch.device("cuda", local_rank)
    model, cfg = build_model(device)
    model.conv_operands = args.conv_operands
    bf16 = args.conv_operands == "bf16"
    # each rank owns its shard of images (pure data parallel); a handful of distinct images is cycled
    imgs = [synth_image(rank * 1000 + i).to(device) for i in range(4)]
    eng = model.engine()
    use_graph = not args.no_graph
    engines = [eng] + [model.make_engine() for _ in range(max(args.inflight, 1) - 1)]
    prio = os.environ.get("ORE_BENCH_STREAM_PRIO")          # experiment knob: comma list of stream priorities (e.g. "0,-1")
    if len(engines) > 1:
        pl = [int(v) for v in prio.split(",")] if prio else [0]
        streams = [torch.cuda.Stream(device, priority=pl[i % len(pl)]) for i in range(len(engines))]
    else:
        streams = [None]

    def run_step(i):
        e, s = engines[i % len(engines)], streams[i % len(engines)]
        if s is None:
            e.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
        else:
            with torch.cuda.stream(s):
                e.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(max(args.warmup, len(engines))):
        run_step(i)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    # the same K steps strictly one image at a time on one stream (what --inflight 1 measures)
    seq_elapsed = elapsed
    if len(engines) > 1:
        sync_all()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eng.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
        sync_all()
        seq_elapsed = time.perf_counter() - t1
        # every concurrent engine must reproduce engine 0 bit for bit on the same image
        ref = None
        for e, st in zip(engines, streams):
            with torch.cuda.stream(st):
                e.eval_forward(imgs[1], use_graph=use_graph)
            torch.cuda.synchronize()
            got = [t.clone() for t in e.proposals()] + ([t.clone() for t in e.detections()] if getattr(e, "has_roi", False) else [])
            if ref is None:
                ref = got
            else:
                assert len(got) == len(ref) and all(torch.equal(a, b) for a, b in zip(got, ref)), "concurrent engines disagree"
    def comm_max(v):
        from detectron2.utils import comm as _c
        return _c.max_over_ranks(v, device)

    # folded serving: the SAME single-image requests, F at a time through ONE engine pass (dense stages batched over the F images,
    # detection tail and second stage per image).  Reported beside the headline, never as `value`.
    folded = None
    if args.fold > 1:
        S = max(args.fold_streams, 1)
        efs = [model.make_engine(max_batch=args.fold) for _ in range(S)]
        fstreams = [torch.cuda.Stream(device) for _ in range(S)]
        packs = [torch.stack([imgs[(j + i) % len(imgs)] for i in range(args.fold)]).contiguous() for j in range(len(imgs))]
        nb = max(args.steps // args.fold, S)

        def fstep(j):
            with torch.cuda.stream(fstreams[j % S]):
                efs[j % S].eval_forward_batch(packs[j % len(packs)], use_graph=use_graph)
        for j in range(max(args.warmup // args.fold, 2 * S)):
            fstep(j)
        sync_all()
        tf0 = time.perf_counter()
        for j in range(nb):
            fstep(j)
        sync_all()
        f_el = comm_max(time.perf_counter() - tf0)
        folded = {"images_per_s": round(world * nb * args.fold / f_el, 2), "requests_per_pass": args.fold, "passes_in_flight": S,
                  "ms_per_pass": round(f_el / nb * 1e3, 4),
                  "note": "the same bs=1 requests, %d folded into one engine pass (ore_engine_eval_batch_fwd), %d passes in flight on "
                          "separate streams: the dense stages run batched, so a CU fetches each layer's weights once per pass instead "
                          "of once per image; per-image results are checked against the bs=1 engine in tests/test_hip_parity.py"
                          % (args.fold, S)}
        for ef in efs:
            ef.close()
    from detectron2.utils import comm
    elapsed = comm.max_over_ranks(elapsed, device)          # the slowest rank defines the job time
    seq_elapsed = comm.max_over_ranks(seq_elapsed, device)
    total_images = int(comm.sum_over_ranks(args.steps, device))
    n_prop = int(eng.buffer("counts")[1, 0].item())
    n_det = int(eng.buffer("det_count")[0, 0].item()) if getattr(eng, "has_roi", False) else None

    # per-image latency with a host sync after every image (the reference's inference_on_dataset protocol)
    lat = []
    for i in range(min(args.steps, 50)):
        torch.cuda.synchronize()
        a = time.perf_counter()
        eng.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - a)
    lat.sort()

    roof = None
    if rank == 0:
        # roofline leg: eager passes with HIP events around every conv launch, on the launch stream
        eng.set_profiling(True)
        for i in range(3):
            eng.eval_forward(imgs[0], use_graph=False)
        eng.read_profile()
        for i in range(args.profile_passes):
            eng.eval_forward(imgs[i % len(imgs)], use_graph=False)
        ms_raw, fl, nl = eng.read_profile()
        eng.set_profiling(False)
        import orehip
        # `achieved` uses the RAW event time (conservative: each bracket also contains the dispatch latency of its launch, ~10 % over
        # the durations rocprofv3 reports for the same kernels, profiles/r01_conv_layers.txt).  The calibrated figure subtracts what
        # an event pair adds around an empty launch beyond back-to-back issue (2*T(1) - T(2)) and is printed beside it.
        ev_us = orehip.event_pair_overhead_us(300)
        ms = ms_raw
        ms_cal = max(ms_raw - nl * ev_us * 1e-3, 1e-6)
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None                                  # HBM bytes of the conv launches of one image, from the committed PMC passes
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tf):
            try:
                with open(tf) as f:
                    traffic = float(json.load(f)["conv_hbm_bytes_per_image"])
            except Exception:
                traffic = None
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_note": "HBM-side bytes per image over the same conv launches: (2*FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 "
                                "--pmc passes (tools/pmc_pass.py -> profiles/r01_pmc_traffic.json); ~0.7 TB/s, far below the 8 TB/s roof",
                "kernel": ("k_conv_igemm / k_conv3x3_patch / k_conv3x3_ws with bf16 operands (v_mfma_f32_16x16x16_bf16, fp32 accumulate); 28 convs "
                           "+ the fp32 ROI fc GEMM" if bf16 else
                           "k_conv_igemm (fp32 v_mfma_f32_16x16x4_f32 implicit-GEMM conv incl. in-kernel split-K; 28 convs + the ROI fc GEMM)"),
                "launches_per_image": nl // max(args.profile_passes, 1),
                "gflop_per_image": round(fl / max(args.profile_passes, 1) / 1e9, 3),
                "kernel_ms_per_image": round(ms / max(args.profile_passes, 1), 4),
                "kernel_ms_per_image_calibrated": round(ms_cal / max(args.profile_passes, 1), 4), "event_pair_overhead_us": round(ev_us, 3),
                "achieved_calibrated": round(fl / (ms_cal * 1e-3) / 1e12, 2),
                "rocprof_note": "rocprofv3 --kernel-trace of the same launches (fp32): 0.665 ms per image = 51.1 TFLOP/s = 0.325 of peak (profiles/r01_conv_layers.txt)",
                "note": "per-kernel figure from isolated (one image at a time) launches; with images in flight the conv FLOP rate "
                        "end to end is value x gflop_per_image"}
        roof["end_to_end_tflops"] = round(total_images / elapsed * roof["gflop_per_image"] / 1e3, 2)

    if rank == 0:
        out = {
            "metric": "images/sec at 640x640 25-shot (eval FPS, bs=1 per GPU)",
            "value": round(total_images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": ("bf16 MFMA conv operands + fp32 NMS (BASELINE configs[4] precision) on " if bf16 else "") +
                                   "finetune_vovnet.yaml 25-shot eval-only bs=1 640x640 (BASELINE configs[1]): "
                                   "preprocess+VoVNet-19-slim-eSE+FPN -> correlation -> CenterNet head -> top-k/NMS proposals -> ROIAlign + cascade ROI head -> NMS -> detections",
                       "parallelism": f"dp{world} (images sharded, no data-path collective)", "hipgraph": use_graph,
                       "images_in_flight_per_gpu": len(engines),
                       "proposals_last_image": n_prop, "detections_last_image": n_det},
            "sequential": {"images_per_s": round(total_images / seq_elapsed, 2), "ms_per_image": round(seq_elapsed / args.steps * 1e3, 4),
                           "note": "same K steps, one image at a time on one stream (images_in_flight_per_gpu = 1)"},
            "latency_ms_host_sync": {"p50": round(lat[len(lat) // 2] * 1e3, 4), "min": round(lat[0] * 1e3, 4)},
            "folded_serving": folded,
            "roofline": roof,
        }
        if not args.no_train_leg and world == 1:
            try:
                out["train_step"] = train_leg(device)
            except Exception as ex:                         # the headline line must survive a failure of the side measurement
                out["train_step"] = {"error": repr(ex)[:300]}
            try:                                            # BASELINE configs[2]: 16 query images (+ 16 x 24 support crops) per GPU per step
                out["train_step_bs16"] = train_leg(device, steps=6, warmup=3, batch=16, graph=False)
            except Exception as ex:
                out["train_step_bs16"] = {"error": repr(ex)[:300]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, imgs[0].cpu())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
