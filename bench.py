#!/usr/bin/env python3
"""Faster-OreFSDet on MI355X -- headline benchmark (BASELINE.json: images/s at 640x640, 25-shot).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`value` follows the reference's own FPS protocol (d2z:evaluation/evaluator.py:138-161, SURVEY 8d metric (i)): one step =
`outputs = model([{"image": img, "height": 640, "width": 640}])` followed by `torch.cuda.synchronize()`, one image at a time,
>= 5 warm-up calls, finetune_vovnet.yaml, 25-shot cached support, bs = 1 (BASELINE configs[1]).  The call is the registered
meta-architecture's forward: support-cache check, the whole detector (fused preprocess + VoVNet-19-slim-eSE + FPN -> correlation ->
CenterNet head -> top-k/decode/NMS -> ROIAlign + cascade ROI head -> NMS -> top-100) as one replayed hipGraph, the count read-back,
`Instances`, `detector_postprocess`.  The image tensor is resident in HBM when the timed region starts (the tier rule for `value`);
the same protocol fed a host uint8 image (H2D copy inside the step, what a dataloader hands over) is reported as
"protocol_host_image".  The K steps are repeated until the timed region is >= --min-time seconds (default 1 s) so a small --steps
still gives a stable figure; `value` = images of all repeats / their time.
Beside it, never as `value`: "engine_sequential" (the C-ABI engine call alone, no per-image sync), "in_flight" (4 bs=1 forwards in
flight on 4 streams), "folded_serving", "train_step" / "train_step_bs16", and "gpu_idle_us_per_image" (a protocol step minus the same
graph replayed back to back = what the per-image sync leaves idle on the GPU).
N>1: pure data parallel.  Eval: every rank runs the protocol on its own images, no data-path collective (weak scaling); RCCL
(backend "nccl") carries the barrier and the max-over-ranks.  Training (BASELINE configs[3]): 16 query images per GPU per step under
FlatDataParallel, the gradient bucket all-reduced over RCCL from backward hooks -- "train_step" then carries the global rate, the
exchange time alone and the fraction of it hidden behind backward.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernels = the fp32 MFMA conv launches: algorithmic FLOPs of all conv launches of one image / their summed
               duration, measured live with HIP events on the launch stream (eager passes after the timed region), against the
               157.3 TFLOP/s fp32 matrix peak of gfx950.
  cpu_baseline the CPU oracle (oracle/: plain-PyTorch fp32 + C decode/NMS restatement of the reference, both stages) timed on this
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


def _pool_out(h):
    return -((h - 3) // -2) + 1                                   # max-pool 3x3 s2, ceil_mode (d2z:modeling/backbone/vovnet.py:310-332)


def layer_table_macs(H, W, fpn_ch=128):
    """MACs of ONE image of HxW through VoVNet-19-slim-eSE + FPN, layer by layer (SURVEY Appendix A restated as a function of the input
    size): returns (frozen part = stem + stage 2 + stage 3, trainable part = stage 4 + stage 5 + FPN, rows of p3 / p4 / p5, and the
    MACs of the trainable convs whose INPUT is a frozen map -- their data gradient is never needed)."""
    Hp, Wp = -(-H // 32) * 32, -(-W // 32) * 32
    h1, w1 = Hp // 2, Wp // 2
    h2, w2 = (h1 - 1) // 2 + 1, (w1 - 1) // 2 + 1
    frozen = h1 * w1 * 27 * 64 + h1 * w1 * 9 * 64 * 64 + h2 * w2 * 9 * 64 * 128
    spec = ((64, 112), (80, 256), (96, 384), (112, 512))           # (3x3 width, concat output) of stages 2..5
    cin, h, w = 128, h2, w2
    stage, first_in = [], []
    rows = {}
    for k, (c, oc) in enumerate(spec):
        if k > 0:
            h, w = _pool_out(h), _pool_out(w)
        m = h * w * 9 * (cin * c + 2 * c * c) + h * w * (cin + 3 * c) * oc + oc * oc
        stage.append(m)
        first_in.append(h * w * 9 * cin * c + h * w * cin * oc)     # layers.0 and the concat's slice that reads the stage input
        rows[k + 2] = h * w
        cin = oc
    frozen += stage[0] + stage[1]
    fpn = sum(rows[k] * (cc * fpn_ch + 9 * fpn_ch * fpn_ch) for k, cc in ((3, 256), (4, 384), (5, 512)))
    trainable = stage[2] + stage[3] + fpn
    no_dgrad = first_in[2] + rows[3] * 256 * fpn_ch                # stage 4's reads of stage 3's (frozen) output + lateral 3
    return frozen, trainable, (rows[3], rows[4], rows[5]), no_dgrad


def train_step_gflop_expected(batch, size, shots, support_hw=240, rois=128, fpn_ch=128):
    """Algorithmic GFLOP of one training step from the layer table (BASELINE.md section 2 / SURVEY 8d): forward of the conv stack on
    `batch` query images and batch*shots support crops (padded to /32), data + weight gradients of stage 4 / 5 / FPN on both, the
    correlation's conv3 + CenterNet head (forward, data, weight gradient) on the query pyramid, SM_Block's three Linear(128,128) per
    level on the support maps, and the second stage on `rois` sampled ROIs per image.  MAC = 2 FLOP."""
    fq, tq, rq, ndq = layer_table_macs(size, size, fpn_ch)
    fs, ts, _, nds = layer_table_macs(support_hw, support_hw, fpn_ch)
    n_sup = batch * shots
    conv = batch * (fq + 3 * tq - ndq) + n_sup * (fs + 3 * ts - nds)
    rows_q = sum(rq)
    heads = batch * rows_q * (2 * fpn_ch * fpn_ch + 9 * fpn_ch * fpn_ch + 9 * fpn_ch * 5) * 3
    sm = n_sup * (32 * 32 + 16 * 16 + 8 * 8) * 3 * fpn_ch * fpn_ch * 3
    roi = batch * rois * (64 * 2 * fpn_ch * fpn_ch + 2 * 64 * fpn_ch * (fpn_ch // 2) + 64 * fpn_ch * 128 + 128 * 6) * 3
    return 2.0 * (conv + heads + sm + roi) / 1e9


def _profile_summary(name):
    """profiles/r*_<name>.json written by tools/bench_train.py --rocprof-summary (launches and kernel time of one step by rocprofv3), only if
    it was taken with this library version."""
    import glob
    import orehip
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s.json" % name)), reverse=True):
        try:
            with open(tf) as f:
                tj = json.load(f)
            if int(tj.get("ore_version", -1)) == int(orehip.lib().ore_version()):
                return dict(tj, source=os.path.basename(tf))
        except Exception:
            pass
    return None


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
    rcnn_8 = model.support_dict["rcnn_8"][0].cpu()
    H, W = img.shape[-2:]

    def one():
        with torch.no_grad():
            o = R.eval_dense(img, sd, support)
            hms = [h[0, 0].numpy() for h in o["hm"]]
            regs = [r[0].permute(1, 2, 0).contiguous().numpy() for r in o["reg"]]
            d = odec.decode_nms(hms, regs, (8, 16, 32), 1e-5, 1000, 0.6, 256)
            # second stage too (ROIAlign + DSA mix + fc1 + predictor + NMS 0.9 + top-100): the GPU step includes it
            feats = [o["features"][k] for k in ("p3", "p4", "p5")]
            return R.roi_head_eval(feats, torch.from_numpy(d["boxes"]), rcnn_8, sd, (H, W), 0.0, 0.9, 100, compiled_roi_align=True)

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
            "sample": f"{n} images of the same 640x640 bs=1 eval workload, both stages (oracle/ref_model.py + oracle/ref_decode.c incl. its ROIAlign, "
                      f"torch {torch.__version__} CPU, {el:.1f} s)"}


def train_leg(device, steps=8, warmup=4, size=640, shots=24, batch=1, graph=True, world=1, rank=0, min_time=0.0, precision="fp32"):
    """SURVEY 8d metric (ii): forward + backward (+ gradient exchange) + clip/SGD of finetune_vovnet.yaml, `batch` query images with 24
    support crops each per GPU per step.  world > 1 (BASELINE configs[3]): the detector is wrapped in FlatDataParallel, every rank runs
    the same per-GPU batch on its own images and the flat gradient bucket is all-reduced over RCCL from the backward hooks; reported:
    the global rate, the exchange alone, and how much of it backward hides.  Reported beside the headline, never as `value`."""
    import torch.distributed as dist
    from detectron2.structures import Boxes, Instances
    from detectron2.utils import comm
    from fewx.solver import FlatDataParallel, build_lr_scheduler, build_optimizer
    import orehip
    # precision "bf16" = BASELINE configs[4] ("bf16 MFMA conv path + fp32 NMS"): the frozen stem / stage 2 / stage 3 keep bf16 maps and
    # weights (bf16-storage kernels), every trainable conv / linear multiplies bf16-rounded operands on the bf16 matrix cores in its
    # forward, data gradient and weight gradient (fp32 accumulation); normalisation, correlation, losses, NMS and the optimizer stay fp32
    prev_precision = orehip.set_conv_precision(precision)
    model, cfg = build_model(device)
    model.train()
    # graph = True: the shape-static dense part (fwd + bwd) replays as two hipGraphs; graph = "step": the WHOLE iteration (forward, losses,
    # backward, clip + SGD) is captured once and replayed (fewx.solver.GraphedTrainStep, single process); either falls back to eager if
    # the capture fails (reported below)
    whole = graph == "step" and world == 1
    model.train_graph = bool(graph) and not whole
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():                                   # second-stage weights: small, so the synthetic losses stay finite
        for n, p in model.named_parameters():
            if n.startswith("roi_heads.") and p.dim() > 1:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1).to(device) * (p[0].numel() ** -0.5))
    dp = FlatDataParallel(model, cfg) if world > 1 else None
    net = dp if dp is not None else model
    opt = build_optimizer(cfg, net)
    sched = build_lr_scheduler(cfg, opt)
    g = torch.Generator().manual_seed(1 + 7919 * rank)       # every rank trains on its own images
    items = []
    for b in range(batch):
        wh = torch.rand(17, 2, generator=g) * 120 + 30
        ctr = torch.rand(17, 2, generator=g) * (size - wh) + wh / 2
        inst = Instances((size, size))
        inst.gt_boxes = Boxes(torch.cat([ctr - wh / 2, ctr + wh / 2], 1).to(device))
        inst.gt_classes = torch.zeros(17, dtype=torch.int64, device=device)
        sup = torch.stack([synth_image(100 + 50 * b + i + 5000 * rank, 240, 240) for i in range(shots)]).to(device)
        side = torch.rand(shots, 2, generator=g) * 120 + 80
        c = torch.rand(shots, 2, generator=g) * (240 - side) + side / 2
        items.append({"image": synth_image(7 + b + 5000 * rank, size, size).to(device), "instances": inst, "support_images": sup,
                      "support_bboxes": torch.cat([c - side / 2, c + side / 2], 1).numpy()})

    stepper = None
    if whole:
        from fewx.solver import GraphedTrainStep
        stepper = GraphedTrainStep(model, opt)

    def step():
        if stepper is not None and model.__dict__.get("_ore_count_flops") is None:
            losses = stepper(items)
            sched.step()
            return losses
        losses = net(items)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()
        return losses

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n):
            last = step()
        sync_all()
        return comm.max_over_ranks(time.perf_counter() - t0, device), last

    for _ in range(warmup):
        step()
    # algorithmic FLOPs of one step, counted by the library as the per-op conv calls are made (ore_flop_counter_read): one EAGER step (a
    # replayed hipGraph makes no calls), outside the timed region
    model.train_graph = False
    model.__dict__["_ore_count_flops"] = True               # (this one step runs eagerly whatever the mode)
    orehip.flop_counter(reset=True)
    last = step()
    step_flops, conv_calls = orehip.flop_counter(reset=True)
    del last, model.__dict__["_ore_count_flops"]
    model.train_graph = bool(graph) and not whole
    step()
    el, losses = timed(steps)
    n_steps = steps
    while el < min_time and n_steps < 64 * steps:             # every rank sees the same max-over-ranks time: same decision
        e2, losses = timed(steps)
        el, n_steps = el + e2, n_steps + steps
    out = {"images_per_s": round(world * batch * n_steps / el, 2), "n_gpus": world, "batch_per_gpu": batch, "global_batch": world * batch,
           "ms_per_step": round(el / n_steps * 1e3, 3), "steps": n_steps, "warmup": warmup, "dtype": "bf16" if precision == "bf16" else "f32",
           "workload": "finetune_vovnet.yaml train step: %d x (1 query %dx%d + %d support 240x240) per GPU, fwd + bwd (HIP backward kernels) + "
                       "%sflat-bucket clip/SGD, FREEZE_AT=3%s%s" % (batch, size, size, shots, "RCCL all-reduce of the gradient bucket + " if world > 1 else "",
                                                                   "; the whole iteration replayed as one hipGraph" if stepper is not None and stepper.error is None else "",
                                                                   "; bf16: frozen stages in bf16 storage, bf16 MFMA operands in the trainable convs' forward / "
                                                                   "data / weight gradients, fp32 accumulation, fp32 NMS / losses / optimizer" if precision == "bf16" else ""),
           "dense_part_hipgraph": bool(graph) and not whole and model.__dict__.get("_ore_train_graph_error") is None,
           "whole_step_hipgraph": bool(stepper is not None and stepper.error is None and stepper.replays > 0),
           "whole_step_hipgraph_error": None if stepper is None else stepper.error,
           "exchanged_bytes_per_step": 4 * opt.bucket.size if world > 1 else 0, "bucket_bytes": 4 * opt.bucket.size,
           "loss_sum": round(float(sum(v.detach() for v in losses.values())), 4)}
    peak = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else PEAK_FP32_MFMA_TFLOPS
    ach = step_flops / (el / n_steps) / 1e12
    expected = train_step_gflop_expected(batch, size, shots)
    out["roofline"] = {
        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
        "gflop_per_step": round(step_flops / 1e9, 2), "gflop_per_query_image": round(step_flops / 1e9 / batch, 2),
        "gflop_per_step_layer_table": round(expected, 2), "counted_over_layer_table": round(step_flops / 1e9 / expected, 4),
        "conv_calls_per_step": conv_calls,
        "note": "END-TO-END: algorithmic (direct-convolution) FLOPs of every conv / linear / weight-gradient call of one step (counted by the "
                "library at its C-ABI entry points during one eager step; gflop_per_step_layer_table is the same sum from the layer table) "
                "/ wall time of the step (all kernels, host included), against the dense MFMA peak of the leg's dtype"
                + ("; only the trainable convs multiply in bf16 -- frozen stages run bf16 STORAGE kernels, normalisation / correlation / "
                   "losses / optimizer stay fp32 -- so this fraction prices a mixed step against the pure-bf16 peak" if precision == "bf16" else "")}
    prof = _profile_summary("train_step_bs%d_%s_summary" % (batch, "bf16" if precision == "bf16" else "fp32"))
    if prof is not None:
        out["roofline"].update({"launches_per_step": prof.get("launches_per_step"), "kernel_ms_per_step": prof.get("kernel_ms_per_step"),
                                "rocprof_source": prof.get("source")})
    if world > 1:
        # the exchange alone: the same slices, same order, nothing else running
        bk = opt.bucket
        def exchange():
            ws = [dist.all_reduce(bk.grads[b_:e_], op=dist.ReduceOp.SUM, async_op=True) for b_, e_, _, _ in bk.slices]
            for w in ws:
                w.wait()
        for _ in range(3):
            exchange()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(20):
            exchange()
        sync_all()
        ar_ms = comm.max_over_ranks(time.perf_counter() - t0, device) / 20 * 1e3
        bk.zero_grad()
        # the same step with the exchange switched off (hooks see world = 1): what backward + SGD cost without it
        dp.world, saved_scale = 1, getattr(bk, "grad_scale", 1.0)
        for _ in range(2):
            step()
        el0, _ = timed(steps)
        dp.world, bk.grad_scale = world, saved_scale
        step_ms, base_ms = el / n_steps * 1e3, el0 / steps * 1e3
        exposed = max(step_ms - base_ms, 0.0)
        out.update({"allreduce_ms": round(ar_ms, 3), "allreduce_slices": len(bk.slices),
                    "allreduce_algbw_GBps": round(4 * bk.size / (ar_ms * 1e-3) / 1e9, 2),
                    "ms_per_step_without_exchange": round(base_ms, 3), "exposed_exchange_ms": round(exposed, 3),
                    "overlap_frac": round(min(max(1.0 - exposed / ar_ms, 0.0), 1.0), 3) if ar_ms > 0 else None,
                    "rccl_ranks_seen": int(comm.sum_over_ranks(1, device)), "backend": dist.get_backend()})
    orehip.set_conv_precision(prev_precision)
    return out


def self_launch(n):
    """One fresh process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the same argv.  Children are started as
    subprocesses of the same interpreter (no fork of a GPU-initialised process, no exec from one)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    try:
        # poll ALL ranks: whichever dies first (bad device, OOM, rendezvous error) is seen at once, the others -- which would sit in
        # init_process_group or a collective until the backend's timeout -- are terminated, killed after a grace period, and the
        # first failing exit code is returned
        live = list(procs)
        while live and worst == 0:
            for q in list(live):
                rc = q.poll()
                if rc is None:
                    continue
                live.remove(q)
                if rc != 0:
                    worst = rc
                    break
            else:
                time.sleep(0.05)
        if worst != 0:
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            deadline = time.time() + 5.0
            while time.time() < deadline and any(q.poll() is None for q in procs):
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
        for q in procs:
            q.wait()
    return worst if 0 < worst < 256 else (1 if worst != 0 else 0)


def dry_run(args, world, rank):
    """--dry: everything around the GPU legs (rendezvous, barrier, max-over-ranks, the one JSON line from rank 0) on gloo / CPU."""
    import torch.distributed as dist
    from detectron2.utils import comm
    if os.environ.get("ORE_BENCH_DRY_FAIL_RANK") == str(rank):       # test hook: a rank that dies before the rendezvous
        sys.exit(7)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("ORE_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    K = max(args.steps, 1)
    comm.synchronize()
    t0 = time.perf_counter()
    time.sleep(0.001 * K * (1 + rank))                          # ranks differ: the slowest one defines the job time
    comm.synchronize()
    el = comm.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    seen = int(comm.sum_over_ranks(1, torch.device("cpu")))
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU legs)", "value": round(world * K / el, 2), "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / K * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry": True, "ranks_seen": seen,
                          "config": {"workload": "dry", "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the K timed steps until the timed region is at least this long (s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the train-step measurements printed as \"train_step\"")
    ap.add_argument("--no-extras", action="store_true", help="skip engine_sequential / in_flight / folded_serving / latency legs")
    ap.add_argument("--profile-passes", type=int, default=20)
    ap.add_argument("--conv-operands", choices=("fp32", "bf16", "bf16s"), default="fp32",
                    help="fp32 = the reference's precision (the headline).  bf16s = BASELINE configs[4] as a byte-saving path: bf16 activations "
                         "and weights in HBM and LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulation / statistics / head outputs / NMS "
                         "(include/ore_hip.h ORE_CONV_BF16S).  bf16 = the operand-rounding form (fp32 tensors, ORE_CONV_BF16).  Both are "
                         "reported with dtype \"bf16\", never the default")
    ap.add_argument("--fold-streams", type=int, default=4, help="engine passes kept in flight in the folded-serving leg")
    ap.add_argument("--fold", type=int, default=8,
                    help="also time the same requests folded F at a time into one engine pass (\"folded_serving\" in the output); 0/1 = skip")
    ap.add_argument("--inflight", type=int, default=4,
                    help="\"in_flight\" leg: images kept in flight per GPU, each a bs=1 forward on its own engine + HIP stream")
    ap.add_argument("--probe-step-graph", action="store_true",
                    help="(internal) child mode: capture and replay one whole training iteration (bs 1) and exit 0 if that worked")
    ap.add_argument("--dry", action="store_true",
                    help="rehearse the launch / rendezvous / max-over-ranks / JSON plumbing only (no GPU legs; CPU test of --gpus N)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (ref:fsod_train_net.py:108-118 does the same through
        # d2z:engine/launch.py:27-82).  Nothing in this process has touched the GPU yet (import torch does not), and it never will: it
        # only waits for its children and returns the worst exit code; rank 0's JSON line goes to the inherited stdout.
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry:
        return dry_run(args, world, rank)
    if args.probe_step_graph:
        torch.cuda.set_device(0)
        out = train_leg(torch.device("cuda", 0), steps=2, warmup=4, batch=1, graph="step")
        sys.exit(0 if out.get("whole_step_hipgraph") else 3)
    # The train legs replay the whole iteration as one hipGraph.  A capture that goes wrong inside the runtime takes its process down
    # (seen once during development: a graph that depended on an uncaptured stream segfaulted in hipGraphInstantiate), and the headline
    # line must not depend on that: the capture is rehearsed in a CHILD first -- started before this process has touched the GPU -- and
    # the legs fall back to the dense-part graph / eager step if the child did not come back clean.
    step_probe = None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if under_profiler:
        step_probe = "skipped: a profiler's preloaded library has initialised the GPU in this process already (no child may be started)"
    elif world == 1 and not args.no_train_leg:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe-step-graph"], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, timeout=300)
            step_probe = "ok" if r.returncode == 0 else "child exit %d" % r.returncode
        except Exception as ex:                              # noqa: BLE001
            step_probe = repr(ex)[:120]
    step_graph = step_probe == "ok"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    # ORE_BENCH_BACKEND=gloo: rehearsal of the N-rank code path on a box with fewer GPUs than ranks (ranks share the cards; the
    # exchange then runs over gloo, so its timings say nothing about RCCL).  The driver's runs use the default, nccl = RCCL.
    backend = os.environ.get("ORE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    from detectron2.utils import comm

    device = torch.device("cuda", local_rank)
    model, cfg = build_model(device)
    model.conv_operands = args.conv_operands
    bf16 = args.conv_operands != "fp32"
    bf16s = args.conv_operands == "bf16s"
    # each rank owns its shard of images (pure data parallel); a handful of distinct images is cycled
    host_imgs = [synth_image(rank * 1000 + i) for i in range(4)]
    imgs = [t.to(device) for t in host_imgs]
    use_graph = not args.no_graph
    K, W = max(args.steps, 1), max(args.warmup, 5)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region(step_fn, k, min_time):
        """K steps between two barrier + synchronize brackets, repeated until >= min_time in total.  Returns (steps, seconds)."""
        n, el = 0, 0.0
        while True:
            sync_all()
            t0 = time.perf_counter()
            for i in range(k):
                step_fn(n + i)
            sync_all()
            el += comm.max_over_ranks(time.perf_counter() - t0, device)     # the slowest rank defines the job time
            n += k
            if el >= min_time or n >= 1000 * k:
                return n, el

    # ---- headline: the reference's FPS protocol -- model(batched_inputs) + device sync, one image at a time
    requests = [[{"image": t, "height": 640, "width": 640}] for t in imgs]
    out_last = [None]

    def proto_step(i):
        out_last[0] = model(requests[i % len(requests)])
        torch.cuda.synchronize()

    for i in range(W):
        proto_step(i)
    n_steps, elapsed = timed_region(proto_step, K, args.min_time)
    total_images = world * n_steps
    n_det_proto = len(out_last[0][0]["instances"])
    eng = model.engine()

    extras = {}
    if not args.no_extras:
        # the same protocol fed HOST images (uint8 CHW, as a dataloader hands them over): H2D copy inside the step
        host_requests = [[{"image": t, "height": 640, "width": 640}] for t in host_imgs]

        def host_step(i):
            model(host_requests[i % len(host_requests)])
            torch.cuda.synchronize()
        for i in range(5):
            host_step(i)
        n_h, el_h = timed_region(host_step, K, min(args.min_time, 0.5))
        extras["protocol_host_image"] = {"images_per_s": round(world * n_h / el_h, 2), "ms_per_image": round(el_h / n_h * 1e3, 4),
                                         "note": "same protocol, image handed over as a host uint8 tensor (PCIe copy inside the step)"}
        # the C-ABI engine call alone: one image at a time on one stream, no per-image host sync, no Instances / postprocess
        def eng_step(i):
            eng.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
        for i in range(5):
            eng_step(i)
        n_e, el_e = timed_region(eng_step, K, min(args.min_time, 0.5))
        extras["engine_sequential"] = {"images_per_s": round(world * n_e / el_e, 2), "ms_per_image": round(el_e / n_e * 1e3, 4),
                                       "note": "ore_engine_eval_fwd (one hipGraph replay per image) back to back on one stream, no per-image sync"}
        # per-image latency of the engine call with a host sync after every image
        lat = []
        for i in range(50):
            torch.cuda.synchronize()
            a = time.perf_counter()
            eng.eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - a)
        lat.sort()
        extras["engine_latency_ms_host_sync"] = {"p50": round(lat[len(lat) // 2] * 1e3, 4), "min": round(lat[0] * 1e3, 4)}
        extras["protocol_overhead_us"] = round((elapsed / n_steps - lat[len(lat) // 2]) * 1e6, 1)
        # what the per-image sync of the protocol leaves idle on the GPU: a protocol step against the same graph replayed back to back
        # (sync wake-up + Python + graph launch + input copy)
        extras["gpu_idle_us_per_image"] = round((elapsed / n_steps - el_e / n_e) * 1e6, 1)
        # several bs=1 forwards in flight per GPU, each on its own engine + HIP stream
        if args.inflight > 1:
            engines = [eng] + [model.make_engine() for _ in range(args.inflight - 1)]
            streams = [torch.cuda.Stream(device) for _ in engines]

            def fl_step(i):
                with torch.cuda.stream(streams[i % len(engines)]):
                    engines[i % len(engines)].eval_forward(imgs[i % len(imgs)], use_graph=use_graph)
            for i in range(2 * len(engines)):
                fl_step(i)
            n_f, el_f = timed_region(fl_step, K, min(args.min_time, 0.5))
            ref = None                                       # every concurrent engine must reproduce engine 0 bit for bit
            for e, st in zip(engines, streams):
                with torch.cuda.stream(st):
                    e.eval_forward(imgs[1], use_graph=use_graph)
                torch.cuda.synchronize()
                got = [t.clone() for t in e.proposals()] + ([t.clone() for t in e.detections()] if getattr(e, "has_roi", False) else [])
                if ref is None:
                    ref = got
                else:
                    assert len(got) == len(ref) and all(torch.equal(a_, b_) for a_, b_ in zip(got, ref)), "concurrent engines disagree"
            for e in engines[1:]:
                e.close()
            extras["in_flight"] = {"images_per_s": round(world * n_f / el_f, 2), "images_in_flight_per_gpu": len(engines),
                                   "ms_per_image": round(el_f / n_f * 1e3, 4),
                                   "note": "bs=1 forwards of independent images on %d engines / HIP streams; results bit-identical to "
                                           "the sequential engine (asserted here)" % len(engines)}
        # folded serving: the SAME single-image requests, F at a time through ONE engine pass
        if args.fold > 1 and not bf16s:                        # (a bf16-storage engine takes one image per pass)
            S = max(args.fold_streams, 1)
            efs = [model.make_engine(max_batch=args.fold) for _ in range(S)]
            fstreams = [torch.cuda.Stream(device) for _ in range(S)]
            packs = [torch.stack([imgs[(j + i) % len(imgs)] for i in range(args.fold)]).contiguous() for j in range(len(imgs))]

            def fstep(j):
                with torch.cuda.stream(fstreams[j % S]):
                    efs[j % S].eval_forward_batch(packs[j % len(packs)], use_graph=use_graph)
            for j in range(2 * S):
                fstep(j)
            nb, f_el = timed_region(fstep, max(K // args.fold, S), min(args.min_time, 0.5))
            extras["folded_serving"] = {"images_per_s": round(world * nb * args.fold / f_el, 2), "requests_per_pass": args.fold,
                                        "passes_in_flight": S, "ms_per_pass": round(f_el / nb * 1e3, 4),
                                        "note": "the same bs=1 requests, %d folded into one engine pass (ore_engine_eval_batch_fwd), %d passes "
                                                "in flight; per-image results are checked against the bs=1 engine in tests/test_hip_parity.py"
                                                % (args.fold, S)}
            for ef in efs:
                ef.close()

    n_prop = int(eng.buffer("counts")[1, 0].item())
    n_det = int(eng.buffer("det_count")[0, 0].item()) if getattr(eng, "has_roi", False) else None

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
        fl_exec = eng.profile_executed_flops()
        eng.set_profiling(False)
        import glob
        import orehip
        # `achieved` uses the RAW event time (conservative: each bracket also contains the dispatch latency of its launch, ~10 % over
        # the durations rocprofv3 reports for the same kernels; the rocprofv3-based figure of the committed profile is `frac_rocprof`)
        ms = ms_raw
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM bytes of the conv launches of one image, from the committed PMC passes -- only if they were taken with THIS library version
        traffic, traffic_src, lib_ver = None, None, int(orehip.lib().ore_version())
        pmc_pat = "r*_pmc_traffic_bf16s.json" if bf16s else ("" if bf16 else "r*_pmc_traffic.json")     # one file per engine mode
        for tf in (sorted(glob.glob(os.path.join(ROOT, "profiles", pmc_pat)), reverse=True) if pmc_pat else []):
            try:
                with open(tf) as f:
                    tj = json.load(f)
                if int(tj.get("ore_version", -1)) != lib_ver:
                    traffic_src = "%s is from ore_version %s, library is %d: stale, not reported" % (os.path.basename(tf), tj.get("ore_version"), lib_ver)
                    break
                traffic = float(tj["conv_hbm_bytes_per_image"])
                traffic_src = os.path.basename(tf)
                break
            except Exception:
                traffic = None
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        npp = max(args.profile_passes, 1)
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_note": "HBM-side bytes per image over the same conv launches: (2*FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 "
                                "--pmc passes (tools/pmc_pass.py -> profiles/%s)" % traffic_src,
                "kernel": ("the DMA-fed conv kernels on bf16 tensors (k_conv_gs / k_conv_kw, v_mfma_f32_16x16x32_bf16, fp32 accumulate) + the fp32 ROI fc GEMM" if bf16s else
                           "the MFMA conv kernels with bf16 operands (v_mfma_f32_16x16x16_bf16, fp32 accumulate) + the fp32 ROI fc GEMM" if bf16 else
                           "the fp32 MFMA conv kernels (v_mfma_f32_16x16x4_f32: Winograd F(2x2,3x3) k_conv3x3_wino on the large-M 3x3 layers, the descriptor-addressed "
                           "LDS-DMA implicit GEMMs k_conv_gd / k_conv_kd on the rest) + the second-stage GEMM (k_conv_gd, K split); FLOPs are the ALGORITHMIC (direct-convolution) "
                           "count, so the Winograd layers, which execute 2.25x fewer multiplies, can exceed the MFMA peak"),
                "launches_per_image": nl // npp, "gflop_per_image": round(fl / npp / 1e9, 3),
                "kernel_ms_per_image": round(ms / npp, 4),
                # what the matrix cores actually executed: a layer on the Winograd F(2x2,3x3) kernel runs its algorithmic count / 2.25
                "gflop_executed_per_image": round(fl_exec / npp / 1e9, 3),
                "mfma_executed_frac": round(fl_exec / (ms * 1e-3) / 1e12 / peak, 4) if ms > 0 else None,
                "note": "FLOP-weighted over all conv launches of one image (different shapes), isolated one-image-at-a-time launches; "
                        "profiles/ holds the rocprofv3 --kernel-trace --stats summary of the same command"}
        roof["end_to_end_tflops"] = round(total_images / elapsed * roof["gflop_per_image"] / 1e3, 2)
        cl = _profile_summary("conv_layers_bf16s" if bf16s else "conv_layers") if not (bf16 and not bf16s) else None
        if cl is not None:        # the same launches by rocprofv3 durations (tools/conv_layers_table.py --json on the committed kernel trace)
            roof.update({"kernel_us_per_image_rocprof": cl.get("conv_us_per_image"), "rocprof_source": cl.get("source")})
            if not bf16s:         # (the table prices against the fp32 MFMA peak; the bf16-storage line is HBM-priced below)
                roof.update({"frac_rocprof": cl.get("frac"), "mfma_executed_frac_rocprof": cl.get("mfma_executed_frac")})
        if bf16s:
            # bf16 storage: at 16x the fp32 MFMA rate the conv stack is bound by the bytes it moves, so the roofline is priced against
            # HBM: algorithmic bytes = every conv layer reads its input and writes its output once in bf16 (half of the 336 MB fp32
            # figure of SURVEY 8d / DESIGN 3) + the bf16 weights once, over the summed duration of the same conv launches
            alg_bytes = 336e6 / 2 + 2.0 * 5.06e6
            gbs = alg_bytes / (ms / npp * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                         "algorithmic_bytes_per_image": alg_bytes, "mfma_tflops": round(ach, 2), "mfma_frac_of_bf16_peak": round(ach / PEAK_BF16_MFMA_TFLOPS, 4)})

    # BASELINE configs[4]'s precision as an EVAL leg of the default (fp32) run: a second detector whose engine keeps bf16 activations and
    # weights (ORE_CONV_BF16S), timed on the same protocol, with its own dtype and its own HBM-priced roofline -- reported beside the
    # fp32 line, never as `value` (the bf16 numbers must be driver-visible)
    bf16_leg = {}
    if not bf16 and not args.no_extras and rank == 0:
        try:
            m16, _ = build_model(device)
            m16.conv_operands = "bf16s"

            def p16(i):
                m16(requests[i % len(requests)])
                torch.cuda.synchronize()
            for i in range(8):
                p16(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n16 = 0
            while time.perf_counter() - t0 < 0.6:
                p16(n16)
                n16 += 1
            el16 = time.perf_counter() - t0
            e16 = m16.engine()
            e16.set_profiling(True)
            for i in range(3):
                e16.eval_forward(imgs[0], use_graph=False)
            e16.read_profile()
            for i in range(10):
                e16.eval_forward(imgs[i % len(imgs)], use_graph=False)
            ms16, fl16, nl16 = e16.read_profile()
            e16.set_profiling(False)
            alg_bytes = 336e6 / 2 + 2.0 * 5.06e6              # every conv layer round-trips once in bf16 + the bf16 weights (see --conv-operands bf16s)
            gbs = alg_bytes / (ms16 / 10 * 1e-3) / 1e9
            t16, t16_src = None, None
            import glob as _g
            for tf in sorted(_g.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_bf16s.json")), reverse=True):
                try:
                    with open(tf) as f:
                        tj = json.load(f)
                    if int(tj.get("ore_version", -1)) == int(__import__("orehip").lib().ore_version()):
                        t16, t16_src = float(tj["conv_hbm_bytes_per_image"]), os.path.basename(tf)
                    break
                except Exception:
                    pass
            bf16_leg["eval_bf16s"] = {
                "images_per_s": round(n16 / el16, 2), "ms_per_image": round(el16 / n16 * 1e3, 4), "dtype": "bf16",
                "workload": "the headline protocol on an engine in the bf16 STORAGE mode (bf16 activations / weights in HBM and LDS, "
                            "v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 top-k / NMS / second stage): BASELINE configs[4]'s precision",
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                             "traffic": t16, "traffic_src": t16_src, "algorithmic_bytes_per_image": alg_bytes,
                             "kernel_ms_per_image": round(ms16 / 10, 4), "launches_per_image": nl16 // 10,
                             "mfma_tflops": round(fl16 / (ms16 * 1e-3) / 1e12, 2)}}
        except Exception as ex:                             # a side measurement must not take the headline line down
            bf16_leg["eval_bf16s_error"] = repr(ex)[:300]

    train = {}
    if not args.no_train_leg:
        try:
            if world == 1:
                train["train_step"] = train_leg(device, min_time=0.3, graph="step" if step_graph else True)
                train["train_step"]["whole_step_probe"] = step_probe
                train["train_step_bs16"] = train_leg(device, steps=6, warmup=3, batch=16, graph="step" if step_graph else False, min_time=0.8)   # BASELINE configs[2]
                # BASELINE configs[4] ("bf16 MFMA conv path + fp32 NMS") on one GPU: its own dtype, never mixed into `value`
                train["train_step_bs16_bf16"] = train_leg(device, steps=6, warmup=3, batch=16, graph="step" if step_graph else False, min_time=0.8,
                                                          precision="bf16")
            else:                                           # BASELINE configs[3]: 16 per GPU, gradients over RCCL
                train["train_step"] = train_leg(device, steps=6, warmup=3, batch=16, graph=False, world=world, rank=rank, min_time=0.8)
        except Exception as ex:                             # the headline line must survive a failure of the side measurement
            train["train_step_error"] = repr(ex)[:300]
            if world > 1:
                raise

    if rank == 0:
        out = {
            "metric": "images/sec at 640x640 25-shot (eval FPS: model(inputs) + device sync per image, bs=1 per GPU)",
            "value": round(total_images / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": W, "ms_per_step": round(elapsed / n_steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": ("bf16 storage (bf16 activations / weights in HBM and LDS) + bf16 MFMA + fp32 NMS (BASELINE configs[4] precision) on " if bf16s else
                                    "bf16 MFMA conv operands + fp32 NMS (BASELINE configs[4] precision) on " if bf16 else "") +
                                   "finetune_vovnet.yaml 25-shot eval-only bs=1 640x640 (BASELINE configs[1]), the reference's FPS protocol: "
                                   "model([{image,height,width}]) + torch.cuda.synchronize() per image; preprocess+VoVNet-19-slim-eSE+FPN -> "
                                   "correlation -> CenterNet head -> top-k/NMS proposals -> ROIAlign + cascade ROI head -> NMS -> detections",
                       "parallelism": f"dp{world} (images sharded, no data-path collective)",
                       "hipgraph": True,                        # model() -> ore_engine_detect_fwd always replays the captured graph
                       "extra_legs_hipgraph": use_graph,       # --no-graph only affects engine_sequential / in_flight / folded_serving
                       "images_in_flight_per_gpu": 1, "timed_steps": n_steps, "timed_region_s": round(elapsed, 3),
                       "input": "uint8 BGR CHW image resident in HBM when the timed region starts (the tier rule for `value`); the same protocol fed the "
                                "dataloader's host tensor, PCIe copy inside the step, is \"protocol_host_image\"", "proposals_last_image": n_prop, "detections_last_image": n_det,
                       "detections_returned": n_det_proto},
            **extras, "roofline": roof, **bf16_leg, **train,
        }
        if not args.no_cpu_baseline:                           # rank 0 only; the other ranks wait in the final barrier meanwhile
            out["cpu_baseline"] = cpu_baseline(model, host_imgs[0], budget_s=12.0 if world == 1 else 6.0)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
