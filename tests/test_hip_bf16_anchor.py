"""BASELINE configs[4] ("bf16 MFMA conv path + fp32 NMS") anchored to the REFERENCE RUN.

The reference is fp32 only (ref:fewx/modeling/fsod/fsod_cen.py:417-535, ref:log/fsod_finetune_stone_vovnet_25_test_log.txt:13), so the bf16
modes cannot be pinned bit for bit; tests/test_hip_bf16.py pins them to the oracle's restatement OF THE MODE.  This file states what the
modes cost against the reference's own fp32 arithmetic, as explicit error budgets on the reference-run fixtures (tests/golden/*.npz,
produced by oracle/refrun/gen_golden.py from the executed reference):

  (a) eval, bf16 STORAGE engine: FPN maps of `backbone_fpn_96x128.npz` (rms), detections of `eval_end_to_end.npz` on the two demo images
      (matched fraction inside a px / score budget, IoU agreement);
  (b) training, bf16 operand mode: one iteration against `train_iter_ref_small.npz` -- the five losses, the reference's gradient
      DIRECTION per parameter (cosine against the reference's own sampled gradients);
  (c) 200 optimizer steps from the reference's initialisers on the reference's schedule, fp32 vs bf16 from the same seed: the loss curves
      stay inside a band, both descend, the parameter displacements point the same way.

Budgets are 1.3-2x what tools/r05_probe.py measured on MI355X (numbers in the docstrings); the networks carry seeded RANDOM weights
(no checkpoint exists), which is an ill-conditioned detector: many near-duplicate candidates at similar scores, so a 1e-2 feature
perturbation re-orders NMS decisions -- the fp32 path matches the same fixtures to 1e-3 px (tests/test_hip_parity.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import PKG

from oracle import ref_model as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ore():
    import orehip
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    orehip.lib()
    return orehip


def _rms_rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30))


def _detector(shots, extra=()):
    from fewx.config import get_cfg
    from detectron2.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(PKG, "configs", "fsod", "finetune_vovnet.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "INPUT.FS.SUPPORT_SHOT", shots, *extra])
    cfg.freeze()
    torch.manual_seed(0)
    return build_model(cfg), cfg


def _iou_matrix(a, b):
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-12)


# ---- (a) eval ------------------------------------------------------------------------------------------------------------------------
def test_bf16_storage_backbone_vs_reference_run(ore, golden):
    """bf16 STORAGE engine against the executed reference's backbone + FPN (d2z:modeling/backbone/vovnet.py:471-481, fpn.py:113-154;
    tests/golden/backbone_fpn_96x128.npz, two 96x128 images of zero-mean noise with sigma 50): rms error relative to the map's rms.
    Measured on MI355X (round 5): p3 6.9e-2, p4 9.8e-2, p5 1.55e-1 (worst channel 0.23 / 0.40 / 0.39).  Budget: 0.12 / 0.16 / 0.25.
    Where it comes from (tools/bf16s_error_by_stage.py, profiles/r05_bf16s_error_by_stage.txt): the error grows by ~1.5e-3 per layer in
    quadrature through the stages (stem_1 1.7e-3 ... stage 5 2.2e-2 on a 640x640 image) and then triples at the laterals, because the eSE
    gates of this seeded random network sit in the linear region of hsigmoid with pre-activations of order 10: a 1 % systematic error of a
    channel mean (rounded weights err the same way at every pixel) moves a gate by several percent, and the gate multiplies the whole
    channel.  On a smooth 640x640 image the same engine is within 2.5e-2 / 3.0e-2 / 3.6e-2 of the fp32 engine.  The fp32 engine meets 1e-4
    max-norm on this fixture (test_engine_backbone_golden)."""
    from conftest import chan_err
    g = golden("backbone_fpn_96x128")
    prev = ore.set_conv_precision("bf16s")
    try:
        e = ore.Engine(max_batch=1, max_h=int(g["x"].shape[2]), max_w=int(g["x"].shape[3]))     # the mode takes one image per pass
    finally:
        ore.set_conv_precision(prev)
    e.load_state_dict(R.synth_state_dict(0))
    e.set_support(R.synth_support(0))
    e.finalize()
    x = torch.from_numpy(g["x"]) + torch.tensor(R.PIXEL_MEAN).view(1, 3, 1, 1)
    outs = []
    for b in range(x.shape[0]):
        o = e.backbone(x[b:b + 1].contiguous().cuda())
        outs.append({k: o[k].float().cpu().numpy().copy() for k in ("p3", "p4", "p5")})
    seen = {}
    for k in ("p3", "p4", "p5"):
        got = np.concatenate([o[k] for o in outs], 0)
        assert got.shape == g[k].shape
        seen[k] = (_rms_rel(got, g[k]), chan_err(got, g[k]))
    print("bf16-storage FPN maps vs reference run (rms rel, worst channel):", {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in seen.items()})
    for k, bound in (("p3", 0.12), ("p4", 0.16), ("p5", 0.25)):
        rms, ch = seen[k]
        assert rms <= bound, (k, rms)
        assert ch <= 0.6, (k, ch)
        assert rms >= 1e-4, (k, rms, "this is not the bf16 engine")         # the mode really ran (fp32 sits at 2e-6)
    e.close()


def test_bf16_storage_eval_end_to_end_vs_reference_run(ore, golden, tmp_path):
    """The whole eval path with `model.conv_operands = "bf16s"` (bf16 activations / weights in HBM and LDS, fp32 accumulation, fp32
    top-k / NMS / second stage) against the reference's own init_model + inference executed end to end on the two demo images
    (tests/golden/eval_end_to_end.npz, ref:fewx/modeling/fsod/fsod_cen.py:309-408,417-535; ~100 detections per image on a 300x300 output).
    Budget (measured on MI355X: matched within 2 px / 5e-2 relative score 0.818 / 0.823; within 1 px / 2e-2 0.566 / 0.542; median box
    delta 0.78 / 0.86 px; median relative score delta 4.9e-3 / 2.2e-3; detection counts 100 / 98 vs 99 / 96):
      * >= 70 % of the reference's detections have a partner within 2 px and 5e-2 relative score, >= 40 % within 1 px and 2e-2;
      * median nearest-partner distance <= 1.5 px, median relative score difference <= 1.5e-2;
      * >= 85 % of the reference's detections are overlapped (IoU >= 0.7) by one of ours and vice versa -- the same objects are found;
      * detection counts within 5.
    The fp32 path matches ALL of them to 0.05 px / 1e-3 (test_eval_end_to_end_vs_reference_run)."""
    from oracle import ref_train as RT
    from test_oracle_golden import eval_end_to_end_state
    g = golden("eval_end_to_end")
    shots = int(g["shots"])
    m, _ = _detector(shots)
    m.eval()
    missing, unexpected = m.load_state_dict(eval_end_to_end_state(), strict=False)
    assert not missing and not unexpected
    m.conv_operands = "bf16s"
    m.init_model(support_file=str(tmp_path / "support_dir" / "support_feature.pkl"), support_df=RT.eval_support_df(shots),
                 read_image=RT.eval_support_crop)
    imgs = golden("demo_images_320")["images"]
    for i in range(2):
        for _ in range(2):                                                       # second call = hipGraph replay
            out = m([{"image": torch.from_numpy(imgs[i]), "height": 300, "width": 300}])[0]["instances"]
        ob, osc = out.pred_boxes.tensor.cpu().numpy(), out.scores.cpu().numpy()
        rb, rs = g[f"img{i}_boxes"], g[f"img{i}_scores"]
        d = np.abs(ob[None] - rb[:, None]).max(2)
        j = d.argmin(1)
        dd = d[np.arange(len(rb)), j]
        ds = np.abs(osc[j] - rs) / rs
        iou = _iou_matrix(rb, ob)
        stats = {"n": (len(ob), len(rb)), "2px": float(((dd <= 2.0) & (ds <= 5e-2)).mean()), "1px": float(((dd <= 1.0) & (ds <= 2e-2)).mean()),
                 "med_px": float(np.median(dd)), "med_score": float(np.median(ds)), "iou_ref": float((iou.max(1) >= 0.7).mean()),
                 "iou_ours": float((iou.max(0) >= 0.7).mean())}
        print(f"bf16-storage detections vs reference run, image {i}:", stats)
        assert abs(len(ob) - len(rb)) <= 5, stats
        assert stats["2px"] >= 0.70 and stats["1px"] >= 0.40, stats
        assert stats["med_px"] <= 1.5 and stats["med_score"] <= 1.5e-2, stats
        assert stats["iou_ref"] >= 0.85 and stats["iou_ours"] >= 0.85, stats
        assert (out.pred_classes == 0).all() and stats["med_px"] > 1e-3, "the bf16 engine did not run"


# ---- (b) one training iteration ------------------------------------------------------------------------------------------------------
def _bf16_train_model(ore, shots):
    m, cfg = _detector(shots)
    sd = R.synth_roi_state(R.synth_state_dict(0), 0)
    sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
    sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    m.train()
    for lvl in (3, 4, 5):
        getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0
    return m


def test_bf16_train_iteration_vs_reference_run(ore, golden):
    """BASELINE configs[4]'s training form (5-shot: SUPPORT_SHOT 4; frozen stages in bf16 storage, bf16 operands in the trainable convs'
    forward / data gradient / weight gradient, fp32 everything else) against the EXECUTED reference's iteration
    (tests/golden/train_iter_ref_small.npz, ref:fewx/modeling/fsod/fsod_cen.py:151-308), same sample, the reference's sampled ROIs.
    Measured on MI355X: positive indices identical; losses off by cls 4.2e-2, box_reg 4.7e-3, loc 1.6e-5, agn_pos 1.7e-3, agn_neg 1.1e-2
    relative; per-parameter cosine against the reference's sampled gradient: min 0.902, 10th percentile 0.932, median 0.9949 over the 72
    live parameters; norm ratios 0.90-1.17.
    Budget: losses within 8e-2 / 2e-2 / 1e-3 / 1e-2 / 4e-2; cosine >= 0.80 for every parameter, >= 0.88 at the 10th percentile, >= 0.98
    at the median; gradient norm within [0.75, 1.35] of the reference's.  (fp32: 1e-4 losses, per-parameter bounds of
    test_train_iteration_vs_reference_run.)"""
    from detectron2.structures import Boxes, Instances
    from fewx.modeling.fsod.train_forward import train_forward
    from oracle import ref_train as T
    g = golden("train_iter_ref_small")
    shots, hw = int(g["shots"]), tuple(int(v) for v in g["hw"])
    prev = ore.set_conv_precision("bf16")
    try:
        m = _bf16_train_model(ore, shots)
        img, gt, sup, sbox = T.synth_train_inputs(int(g["input_seed"]), hw, n_gt=int(g["n_gt"]), shots=shots, support_hw=int(g["support_hw"]))
        inst = Instances(hw)
        inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
        item = {"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}
        over = {"boxes": torch.from_numpy(g["roi_boxes"]), "labels": torch.from_numpy(g["roi_labels"]), "gt": torch.from_numpy(g["roi_gt"])}
        losses, aux = train_forward(m, [item], return_aux=True, roi_override=over)
        n = len(g["pos_inds"])
        assert int(aux["pos_count"].item()) == n and np.array_equal(aux["pos_inds"][:n].cpu().numpy(), g["pos_inds"])   # fp32 targets
        budget = {"loss_cls_stage0": 8e-2, "loss_box_reg_stage0": 2e-2, "loss_centernet_loc": 1e-3, "loss_centernet_agn_pos": 1e-2,
                  "loss_centernet_agn_neg": 4e-2}
        seen = {}
        for k, b in budget.items():
            want = float(g["loss/" + k])
            seen[k] = abs(float(losses[k].detach()) - want) / max(abs(want), 1e-3)
        print("bf16 training iteration vs reference run, relative loss error:", {k: f"{v:.2e}" for k, v in seen.items()})
        for k, b in budget.items():
            assert seen[k] <= b, (k, seen[k], b)
        assert max(seen.values()) > 1e-5, "this is not the bf16 mode"
        sum(losses.values()).backward()
    finally:
        ore.set_conv_precision(prev)
    named = dict(m.named_parameters())
    rows = []
    for key in g:
        if not key.startswith("gs/"):
            continue
        k = key[3:]
        ref = g[key].astype(np.float64)
        if not np.any(ref):                                                       # a live parameter whose gradient is exactly 0 on this sample
            assert float(named[k].grad.abs().max()) == 0.0, k                      # (scales.2: no p5 positives)
            continue
        f = named[k].grad.reshape(-1)
        smp = f[:: max(1, f.numel() // 1024)][:1024].cpu().numpy().astype(np.float64)
        cos = float((smp * ref).sum() / np.sqrt((smp ** 2).sum() * (ref ** 2).sum()))
        rows.append((cos, float(np.sqrt((smp ** 2).sum() / (ref ** 2).sum())), k))
    rows.sort()
    cs = np.array([r[0] for r in rows])
    nr = np.array([r[1] for r in rows])
    print(f"gradient cosine vs the reference's gradients over {len(rows)} parameters: min {cs.min():.4f} p10 {np.quantile(cs, 0.1):.4f} "
          f"median {np.median(cs):.5f}; norm ratio {nr.min():.3f}..{nr.max():.3f}; lowest: {[(round(c, 4), k) for c, _, k in rows[:4]]}")
    assert len(rows) == 72
    assert cs.min() >= 0.80 and np.quantile(cs, 0.1) >= 0.88 and np.median(cs) >= 0.98, rows[:6]
    assert nr.min() >= 0.75 and nr.max() <= 1.35, (nr.min(), nr.max())
    for k in g["dead"]:
        p = named[str(k)]
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


# ---- (c) 200 steps ---------------------------------------------------------------------------------------------------------------------
def _trajectory(ore, mode, steps, sampler_seed):
    """`steps` optimizer steps of the detector on four repeating synthetic samples (256x320 query, 4 support crops of 96x96): the
    reference's head / FPN / second-stage initialisers (build_model), a seeded He-initialised frozen backbone with non-trivial FrozenBN
    statistics, the reference's optimizer and schedule (SGD 0.9, BASE_LR 1e-3, 500 warm-up iterations from 2.5e-4 of it, value clip 1.0:
    ref:configs/fsod/finetune_vovnet.yaml, ref:fewx/solver/build.py)."""
    from detectron2.structures import Boxes, Instances
    from fewx.solver import build_lr_scheduler, build_optimizer
    from oracle import ref_train as T
    shots = 4
    prev = ore.set_conv_precision(mode)
    try:
        m, cfg = _detector(shots)
        m.load_state_dict({k: v for k, v in R.synth_state_dict(0).items() if k.startswith("backbone.bottom_up.")}, strict=False)
        m.train()
        for lvl in (3, 4, 5):
            getattr(m, f"vip_p{lvl}").reweighting.drop.p = 0.0
        opt = build_optimizer(cfg, m)
        sched = build_lr_scheduler(cfg, opt)
        batches = []
        for i in range(4):
            img, gt, sup, sbox = T.synth_train_inputs(30 + i, (256, 320), n_gt=5, shots=shots, support_hw=96)
            inst = Instances((256, 320))
            inst.gt_boxes, inst.gt_classes = Boxes(gt), torch.zeros(len(gt), dtype=torch.int64)
            batches.append([{"image": img, "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}])
        torch.manual_seed(sampler_seed)                                           # the ROI sampler's draws
        live = [p for p in m.parameters() if p.requires_grad]
        p0 = torch.cat([p.detach().reshape(-1).clone() for p in live])
        rec = []
        for it in range(steps):
            losses = m(batches[it % len(batches)])
            opt.zero_grad()
            sum(losses.values()).backward()
            opt.step()
            sched.step()
            rec.append(torch.stack([v.detach().float() for v in losses.values()]).sum())
        curve = torch.stack(rec).cpu().numpy().astype(np.float64)
        disp = (torch.cat([p.detach().reshape(-1) for p in live]) - p0).cpu().double()
    finally:
        ore.set_conv_precision(prev)
    return curve, disp


def test_bf16_vs_fp32_training_trajectory_200_steps(ore):
    """Does the bf16 mode TRAIN like fp32?  (ref:fsod_train_net.py:103-105 -> d2z:engine/train_loop.py:133-159 is a 12 000-iteration loop;
    one iteration says nothing about drift.)  200 steps from the same seed in fp32 and in the bf16 mode, plus a second fp32 run that only
    differs in the ROI sampler's seed: the noise floor two correct fp32 runs already have (the sampled ROIs decide the second-stage
    losses).  Window = mean total loss over 20 consecutive steps (5 passes
    over the 4 samples).
    Measured on MI355X, two runs: total loss 3.4 -> 1.3-1.6 in every run; the curve has one steep descent (windows 6-8, when the warm-up
    learning rate lets the second stage fit) whose timing moves by a window between runs, so window-for-window the bf16 curve is within
    0.4-11 % (run 1) / 0.7-33 % (run 2: the descent came one window later) of fp32's, the two fp32 runs within 0.03-13 % of each other;
    the mean over all 200 steps differs by 5 %, the last two windows by <= 2 %.  Parameter displacement after 200 steps (reported):
    |fp32| 0.83-0.89, |bf16| 0.95-1.37, cosine between them 0.26-0.32 -- the same as between the two fp32 runs (0.29-0.34): at this
    length the displacement is dominated by the sampler's noise, not by the precision.
    Budget: every window within 40 % of fp32's and within 25 % of the nearer of fp32's neighbouring windows; mean over the 200 steps
    within 10 %; last two windows within 10 %; both runs end below 70 % of their first window; displacement norm within [0.5, 2] of
    fp32's and cosine >= 0.1."""
    steps, W = 200, 20
    c32, d32 = _trajectory(ore, "fp32", steps, 1)
    cbf, dbf = _trajectory(ore, "bf16", steps, 1)
    c32b, d32b = _trajectory(ore, "fp32", steps, 2)
    assert np.isfinite(c32).all() and np.isfinite(cbf).all()
    w32, wbf, w32b = (c.reshape(-1, W).mean(1) for c in (c32, cbf, c32b))
    dev, noise = np.abs(wbf - w32) / w32, np.abs(w32b - w32) / w32
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))             # noqa: E731
    print("window means fp32:", np.round(w32, 3).tolist())
    print("window means bf16:", np.round(wbf, 3).tolist())
    print(f"bf16 vs fp32 per window: {np.round(dev, 3).tolist()}; fp32 seed noise: {np.round(noise, 3).tolist()}")
    print(f"displacement |fp32| {float(d32.norm()):.4f} |bf16| {float(dbf.norm()):.4f} rel distance {float((dbf - d32).norm() / d32.norm()):.3f} "
          f"cos {cos(d32, dbf):.3f}; fp32 vs fp32(other sampler seed): rel distance {float((d32b - d32).norm() / d32.norm()):.3f} cos {cos(d32, d32b):.3f}")
    nb = np.array([min(abs(wbf[k] - w32[j]) / w32[j] for j in (k - 1, k, k + 1) if 0 <= j < len(w32)) for k in range(len(wbf))])
    print(f"allowing a one-window shift: {np.round(nb, 3).tolist()}; mean over all steps fp32 {c32.mean():.4f} bf16 {cbf.mean():.4f}")
    assert dev.max() <= 0.40 and nb.max() <= 0.25, (dev, nb)
    assert abs(cbf.mean() - c32.mean()) <= 0.10 * c32.mean(), (cbf.mean(), c32.mean())
    assert dev[-2:].max() <= 0.10, dev
    assert w32[-1] <= 0.7 * w32[0] and wbf[-1] <= 0.7 * wbf[0], (w32, wbf)
    r = float(dbf.norm() / d32.norm())
    assert 0.5 <= r <= 2.0 and cos(d32, dbf) >= 0.1, (r, cos(d32, dbf))
    assert not np.array_equal(c32, cbf), "the bf16 mode did not run"
