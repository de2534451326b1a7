"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN FILES (this container only).

    python -m oracle.refrun.gen_golden            # writes tests/golden/

Each fixture = seeded synthetic inputs (+ the seed of oracle.ref_model.synth_state_dict for the
weights) -> outputs produced by the reference modules loaded through oracle/refrun/shims.py.
Fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model as R  # noqa: E402
from oracle.refrun import shims  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    desc = ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items() if not k.startswith(("gs/", "gn/", "gc/")))
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  {desc}")


def sub_sd(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(8)
    ns = shims.setup()
    sd = R.synth_state_dict(SEED)
    g = torch.Generator().manual_seed(1234)

    # ---- backbone + FPN (d2z vovnet.py + fpn.py executed) ------------------------------------
    bb = ns.vovnet.build_fcos_vovnet_fpn_backbone(shims.vovnet_cfg(), ns.layers.ShapeSpec(channels=3))
    missing = bb.load_state_dict(sub_sd(sd, "backbone."), strict=True)
    bb.eval()
    with torch.no_grad():
        x = torch.randn(2, 3, 96, 128, generator=g) * 50.0
        o = bb(x)
        save("backbone_fpn_96x128", seed=np.int64(SEED), x=np_(x), p3=np_(o["p3"]), p4=np_(o["p4"]), p5=np_(o["p5"]))
        # single conv+FrozenBN+ReLU blocks and one OSA stage on an odd size (ceil_mode pooling)
        st = bb.bottom_up.stem
        x2 = torch.randn(1, 64, 24, 40, generator=g)
        y2 = st[5](st[4](st[3](x2)))  # stem_2 conv/norm/relu
        save("conv_stem2", seed=np.int64(SEED), x=np_(x2), y=np_(y2))
        x3 = torch.randn(1, 64, 25, 37, generator=g)
        y3 = st[8](st[7](st[6](x3)))  # stem_3, stride 2, odd size
        save("conv_stem3_odd", seed=np.int64(SEED), x=np_(x3), y=np_(y3))
        x4 = torch.relu(torch.randn(1, 112, 21, 21, generator=g))
        y4 = bb.bottom_up.stage3(x4)
        save("osa_stage3_odd", seed=np.int64(SEED), x=np_(x4), y=np_(y4))
        x5 = torch.relu(torch.randn(1, 384, 10, 10, generator=g))
        y5 = bb.bottom_up.stage5(x5)
        save("osa_stage5", seed=np.int64(SEED), x=np_(x5), y=np_(y5))
        # preprocess via the reference ImageList (pad to /32) on a non-divisible size
        img = R.synth_image(SEED, 75, 100)
        norm = (img - sd["pixel_mean"]) / sd["pixel_std"]
        il = ns.image_list.ImageList.from_tensors([norm], 32)
        save("preprocess_75x100", image=np_(img), x=np_(il.tensor))

    # ---- SM_Block (ref fsod_cen.py executed) ------------------------------------------------------
    cen = shims.load_fsod_cen()
    with torch.no_grad():
        for lvl, seg in ((3, 32), (5, 8)):
            blk = cen.SM_Block(128, seg).eval()
            blk.load_state_dict(sub_sd(sd, f"vip_p{lvl}."), strict=True)
            xs = torch.randn(2, seg, seg, 128, generator=g)
            ys = blk(xs)
            proto = ys.permute(0, 3, 2, 1).mean(0, True)
            save(f"sm_block_p{lvl}", seed=np.int64(SEED), x=np_(xs), y0=np_(ys[:1]), proto=np_(proto))

    # ---- correlation: CenterNet2Detector.inference executed with stub backbone / heads -----------
    D = cen.CenterNet2Detector
    det = D.__new__(D)
    torch.nn.Module.__init__(det)
    det.register_buffer("pixel_mean", sd["pixel_mean"].clone())
    det.register_buffer("pixel_std", sd["pixel_std"].clone())
    det.support_pool_1x1 = torch.nn.AdaptiveAvgPool2d((1, 1))
    det.support_pool_1x3 = torch.nn.AdaptiveAvgPool2d((1, 3))
    det.support_pool_3x1 = torch.nn.AdaptiveAvgPool2d((3, 1))
    det.conv3 = torch.nn.Conv2d(256, 128, 1)
    det.conv3.load_state_dict(sub_sd(sd, "conv3."))
    feats = {"p3": torch.randn(1, 128, 20, 24, generator=g), "p4": torch.randn(1, 128, 10, 12, generator=g),
             "p5": torch.randn(1, 128, 5, 6, generator=g)}
    support = R.synth_support(SEED)

    class _BB(torch.nn.Module):
        size_divisibility = 32

        def forward(self, x):
            return feats

    captured = {}

    class _PG(torch.nn.Module):
        def forward(self, images, pos, gt):
            captured.update(pos)
            return [None], {}

    class _RH(torch.nn.Module):
        def forward(self, images, f, s, p, gt):
            return [None], {}

    det.backbone, det.proposal_generator, det.roi_heads = _BB(), _PG(), _RH()
    det.support_dict = {k: {0: support[k]} for k in ("p3", "p4", "p5")}
    det.support_dict.update(rcnn_8={0: torch.zeros(1)}, rcnn_4={0: torch.zeros(1)})
    det.eval()
    with torch.no_grad():
        det.inference([{"image": torch.zeros(3, 160, 192)}], do_postprocess=False)
    save("correlation", seed=np.int64(SEED),
         **{f"q_{k}": np_(v) for k, v in feats.items()}, **{f"s_{k}": np_(v) for k, v in support.items()},
         **{f"out_{k}": np_(v) for k, v in captured.items()})

    # ---- CenterNet head (ref centernet_head.py executed) -------------------------------------------
    Head = ns.centernet_head.CenterNetHead
    head = Head(in_channels=128, num_levels=3, num_classes=1, with_agn_hm=True, only_proposal=True, norm="GN",
                num_cls_convs=1, num_box_convs=1, num_share_convs=0, use_deformable=False, prior_prob=0.01).eval()
    head.load_state_dict(sub_sd(sd, "proposal_generator.centernet_head."), strict=True)
    with torch.no_grad():
        hx = [torch.relu(torch.randn(1, 128, 20, 24, generator=g)), torch.relu(torch.randn(1, 128, 10, 12, generator=g)),
              torch.relu(torch.randn(1, 128, 5, 6, generator=g))]
        clss, regs, hms = head(hx)
    assert all(c is None for c in clss)
    save("cn_head", seed=np.int64(SEED), **{f"x{l}": np_(hx[l]) for l in range(3)},
         **{f"reg{l}": np_(regs[l]) for l in range(3)}, **{f"hm{l}": np_(hms[l]) for l in range(3)})

    # ---- CenterNet inference (ref fsod_rpn.py executed; head output injected) ----------------------
    rpn = ns.fsod_rpn
    Instances = ns.instances.Instances

    class _FakeHead(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.out = None

        def forward(self, feats):
            return self.out

    def run_infer(hm_logits, regs, training=False):
        fh = _FakeHead()
        cn = rpn.CenterNet(in_channels=128, num_classes=1, in_features=("p3", "p4", "p5"), strides=(8, 16, 32),
                           score_thresh=1e-5, with_agn_hm=True, only_proposal=True, not_norm_reg=True,
                           pre_nms_topk_train=4000, pre_nms_topk_test=1000, post_nms_topk_train=2000,
                           post_nms_topk_test=256, nms_thresh_train=0.9, nms_thresh_test=0.6,
                           pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85, reg_weight=1.0,
                           sizes_of_interest=[[0, 64], [48, 192], [128, 1000000]], centernet_head=fh)
        cn.eval()
        fh.out = ([None] * 3, regs, hm_logits)
        cap = {}
        orig = rpn.ml_nms

        def spy(boxlist, thr, *a, **k):
            cap["pre_boxes"] = boxlist.pred_boxes.tensor.clone()
            cap["pre_scores"] = boxlist.scores.clone()
            r = orig(boxlist, thr, *a, **k)
            cap["nms_boxes"] = r.pred_boxes.tensor.clone()
            return r

        rpn.ml_nms = spy
        try:
            images = types.SimpleNamespace(image_sizes=[(640, 640)])
            feats_d = {k: torch.zeros(1, 128, s, s) for k, s in (("p3", 80), ("p4", 40), ("p5", 20))}
            with torch.no_grad():
                props, _ = cn(images, feats_d, None)
        finally:
            rpn.ml_nms = orig
        p = props[0]
        return cap, p.proposal_boxes.tensor, p.objectness_logits

    for tag, shift in (("sparse", -15.0), ("dense", 2.0)):
        hm_l, reg_l = [], []
        for s in (80, 40, 20):
            # smooth blobs + noise so neighbouring boxes overlap (exercises NMS); reg in stride units
            base = torch.nn.functional.interpolate(torch.randn(1, 1, s // 4, s // 4, generator=g), size=(s, s),
                                                   mode="bilinear", align_corners=False)
            hm_l.append(base * 2.5 + torch.randn(1, 1, s, s, generator=g) * 0.7 + shift)
            reg_l.append(torch.relu(torch.randn(1, 4, s, s, generator=g) * 1.5 + 2.5))
        cap, boxes, scores = run_infer(hm_l, reg_l)
        save(f"cn_infer_640_{tag}",
             **{f"hm{l}": np_(hm_l[l][0, 0]) for l in range(3)},
             **{f"reg{l}": np_(reg_l[l][0].permute(1, 2, 0).contiguous()) for l in range(3)},
             pre_boxes=np_(cap["pre_boxes"]), pre_scores=np_(cap["pre_scores"]),
             nms_boxes=np_(cap["nms_boxes"]), boxes=np_(boxes), scores=np_(scores))
    # ---- CenterNet training targets + losses (ref fsod_rpn.py _get_ground_truth / losses executed), B=2 ----------------------
    fh = _FakeHead()
    cn = rpn.CenterNet(in_channels=128, num_classes=1, in_features=("p3", "p4", "p5"), strides=(8, 16, 32),
                       score_thresh=1e-5, with_agn_hm=True, only_proposal=True, not_norm_reg=True,
                       pre_nms_topk_train=4000, pre_nms_topk_test=1000, post_nms_topk_train=2000,
                       post_nms_topk_test=256, nms_thresh_train=0.9, nms_thresh_test=0.6,
                       pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85, reg_weight=1.0, hm_focal_alpha=0.25,
                       sizes_of_interest=[[0, 64], [48, 192], [128, 1000000]], centernet_head=fh)
    cn.train()
    Boxes = ns.boxes.Boxes
    shapes = [(40, 48), (20, 24), (10, 12)]                       # a 320 x 384 image
    gts = []
    for n_obj in (17, 9):
        ctr = torch.rand(n_obj, 2, generator=g) * torch.tensor([384.0, 320.0])
        wh = torch.exp(torch.rand(n_obj, 2, generator=g) * 3.2 + 2.3)      # 10 .. 245 px
        b = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).clamp(min=0)
        b[:, 2].clamp_(max=384); b[:, 3].clamp_(max=320)
        inst = Instances((320, 384))
        inst.gt_boxes = Boxes(b)
        inst.gt_classes = torch.zeros(n_obj, dtype=torch.int64)
        gts.append(inst)
    feats_t = [torch.zeros(2, 128, h, w) for h, w in shapes]
    grids = cn.compute_grids(feats_t)
    spl = grids[0].new_tensor(shapes)
    with torch.no_grad():
        pos_inds, labels, reg_targets, flattened_hms = cn._get_ground_truth(grids, spl, gts)
        M2 = reg_targets.shape[0]
        reg_pred = torch.relu(torch.randn(M2, 4, generator=g) * 1.5 + 2.5)
        hm_logit = torch.randn(M2, generator=g) * 2.0 - 3.0
        losses = cn.losses(pos_inds, labels, reg_targets, flattened_hms, None, reg_pred, hm_logit.clone())
    save("cn_train_targets", gt0=np_(gts[0].gt_boxes.tensor), gt1=np_(gts[1].gt_boxes.tensor), shapes=np.array(shapes, np.int64),
         pos_inds=np_(pos_inds), reg_targets=np_(reg_targets), hms=np_(flattened_hms[:, 0]), reg_pred=np_(reg_pred), hm_logit=np_(hm_logit),
         loss_loc=np_(losses["loss_centernet_loc"]), loss_pos=np_(losses["loss_centernet_agn_pos"]),
         loss_neg=np_(losses["loss_centernet_agn_neg"]))
    # ---- second-stage training pieces that ARE pure Python in the vendored detectron2 (executed from the 7z extract): pairwise_iou,
    #      Matcher, subsample_labels, Box2BoxTransform.get_deltas  (d2z:structures/boxes.py, modeling/matcher.py, sampling.py, box_regression.py)
    D2 = shims.d2_root()
    mt = shims.load("d2real_matcher", D2 + "/modeling/matcher.py")
    sp = shims.load("d2real_sampling", D2 + "/modeling/sampling.py")
    br = shims.load("d2real_box_regression", D2 + "/modeling/box_regression.py")
    n_p, n_g = 300, 9
    ctr = torch.rand(n_g, 2, generator=g) * 500 + 60
    wh = torch.rand(n_g, 2, generator=g) * 120 + 30
    gtb = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    jit = torch.cat([gtb[torch.randint(0, n_g, (120,), generator=g)] + torch.randn(120, 4, generator=g) * 6.0,
                     torch.cat([torch.rand(n_p - 120, 2, generator=g) * 500, torch.rand(n_p - 120, 2, generator=g) * 500 + 20], 1)], 0)
    jit[:, 2:] = torch.max(jit[:, 2:], jit[:, :2] + 4.0)
    allb = torch.cat([jit, gtb], 0)                                        # proposal_append_gt
    iou = ns.boxes.pairwise_iou(Boxes(gtb), Boxes(allb))
    matcher = mt.Matcher([0.6], [0, 1], allow_low_quality_matches=False)
    midx, mlab = matcher(iou)
    gt_classes = torch.zeros(n_g, dtype=torch.int64)[midx]
    gt_classes[mlab == 0] = 1                                              # background = num_classes = 1
    torch.manual_seed(123)
    fg_i, bg_i = sp.subsample_labels(gt_classes, 128, 0.5, 1)
    sampled = torch.cat([fg_i, bg_i], 0)
    b2b = br.Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    fg_rows = sampled[gt_classes[sampled] == 0]
    deltas = b2b.get_deltas(allb[fg_rows], gtb[midx[fg_rows]])
    save("roi_train_pieces", gt=np_(gtb), boxes=np_(allb), iou=np_(iou), matched_idx=np_(midx), labels=np_(gt_classes),
         sampled=np_(sampled), fg_rows=np_(fg_rows), deltas=np_(deltas), seed=np.array(123))
    print("missing/unexpected:", missing)

    gen_roi_stage(ns, sd)
    gen_train_iteration(sd)
    gen_state_dict_layout()
    gen_demo_images()
    gen_generate_support()
    gen_dataset_split()
    gen_eval_end_to_end(sd)
    gen_cn_indices()


def gen_cn_indices():
    """SURVEY Appendix E (`cn_infer_640_*`: "per-level selected flat indices ... NMS keep indices int64"), VERDICT r03 missing #4.

        python -m oracle.refrun.gen_golden indices        # writes tests/golden/cn_infer_640_{sparse,dense,train}_idx.npz only

    The reference's predict_instances / predict_single_level / nms_and_topK (ref:fewx/modeling/fsod/fsod_rpn.py:1101-1210) executed on
    the head outputs the existing cn_infer_640_* fixtures hold, in eval mode (1e-5 / 1000 / 0.6 / 256) and -- on the dense maps -- with
    the TRAINING thresholds (4000 / 0.9 / 2000; the call CenterNet.forward makes in its training branch, :674-679).  The index tensors
    never leave predict_single_level, so torch.Tensor.nonzero / topk and the module's batched NMS are watched while it runs: per level
    `per_candidate_inds.nonzero()[:, 0]` (the flat locations above the threshold) and, where a level holds more than pre_topk, the
    `top_k_indices` into them.  Stored:
      sel{l}            int64, SORTED flat indices the reference selected at level l (topk(sorted=False) order is implementation-defined);
      pre_level/pre_loc the reference's own pre-NMS order (level, flat index) -- row r of the stored pre_boxes / pre_scores;
      nms_keep          int64 keep list of ml_nms in the CANONICAL pre order (levels concatenated, ascending flat index inside a level;
                        what oracle/ref_decode.c and the HIP path emit), in the reference's order (descending score);
      post_keep         the same after the post-NMS `score >= kth` filter = rows of the returned proposals.
    The NMS arithmetic itself is the restated torchvision nms (un-vendored); everything else is the reference's code."""
    ns = shims.setup()
    rpn = ns.fsod_rpn

    class _FakeHead(torch.nn.Module):
        def forward(self, feats):
            raise RuntimeError("head outputs are injected")

    def build():
        return rpn.CenterNet(in_channels=128, num_classes=1, in_features=("p3", "p4", "p5"), strides=(8, 16, 32),
                             score_thresh=1e-5, with_agn_hm=True, only_proposal=True, not_norm_reg=True,
                             pre_nms_topk_train=4000, pre_nms_topk_test=1000, post_nms_topk_train=2000,
                             post_nms_topk_test=256, nms_thresh_train=0.9, nms_thresh_test=0.6,
                             pos_weight=0.5, neg_weight=0.5, ignore_high_fp=0.85, reg_weight=1.0,
                             sizes_of_interest=[[0, 64], [48, 192], [128, 1000000]], centernet_head=_FakeHead())

    def run(hm_l, reg_l, training):
        cn = build()
        cn.train(training)
        log = []                                                    # ("nonzero", tensor) / ("topk", indices) in call order
        o_nz, o_tk, o_nms = torch.Tensor.nonzero, torch.Tensor.topk, rpn.ml_nms
        cap = {}

        def nz(self, *a, **k):
            r = o_nz(self, *a, **k)
            if self.dtype == torch.bool and r.dim() == 2 and r.shape[1] == 2:
                log.append(("nonzero", r.clone()))
            return r

        def tk(self, *a, **k):
            r = o_tk(self, *a, **k)
            log.append(("topk", r[1].clone()))
            return r

        def spy_nms(boxlist, thr, *a, **k):
            cap["pre_boxes"], cap["pre_scores"] = boxlist.pred_boxes.tensor.clone(), boxlist.scores.clone()
            ob = o_nms.__globals__["batched_nms"]

            def bn(boxes, scores, labels, t):
                kp = ob(boxes, scores, labels, t)
                cap["keep"] = kp.clone()
                return kp
            o_nms.__globals__["batched_nms"] = bn
            try:
                return o_nms(boxlist, thr, *a, **k)
            finally:
                o_nms.__globals__["batched_nms"] = ob

        torch.Tensor.nonzero, torch.Tensor.topk = nz, tk
        rpn.ml_nms = spy_nms
        try:
            feats = [torch.zeros(1, 128, s, s) for s in (80, 40, 20)]
            grids = cn.compute_grids(feats)
            with torch.no_grad():
                props = cn.predict_instances(grids, [x.sigmoid() for x in hm_l], reg_l, [(640, 640)], [None, None, None])
        finally:
            torch.Tensor.nonzero, torch.Tensor.topk = o_nz, o_tk
            rpn.ml_nms = o_nms
        # replay the log: one nonzero per level, followed by a topk when the level was cut
        sel, i = [], 0
        for l in range(3):
            assert log[i][0] == "nonzero", [e[0] for e in log]
            loc = log[i][1][:, 0]
            i += 1
            if i < len(log) and log[i][0] == "topk":
                loc = loc[log[i][1]]
                i += 1
            sel.append(loc)
        # the `cls_scores >= image_thresh` filter calls torch.nonzero (the function), not Tensor.nonzero on a 2-d mask: nothing else logged
        assert i == len(log), [e[0] for e in log]
        pre_level = torch.cat([torch.full((len(s_),), l, dtype=torch.int64) for l, s_ in enumerate(sel)])
        pre_loc = torch.cat(sel)
        assert len(pre_loc) == len(cap["pre_scores"])
        # canonical pre order: (level, flat index) ascending
        key = pre_level * (1 << 20) + pre_loc
        order = torch.argsort(key)
        canon_of_ref = torch.empty_like(order)
        canon_of_ref[order] = torch.arange(len(order))
        keep_ref = cap["keep"]
        nms_keep = canon_of_ref[keep_ref]
        p = props[0]
        boxes_out = p.pred_boxes.tensor if p.has("pred_boxes") else p.proposal_boxes.tensor
        scores_out = p.scores
        # rows of the returned proposals among the NMS survivors: the reference filters result[keep] with a mask, order preserved
        kept_scores = cap["pre_scores"][keep_ref]
        if len(keep_ref) > (cn.post_nms_topk_train if training else cn.post_nms_topk_test):
            k_ = cn.post_nms_topk_train if training else cn.post_nms_topk_test
            thr = torch.kthvalue(kept_scores.float(), len(keep_ref) - k_ + 1)[0]
            post_keep = nms_keep[kept_scores >= thr]
        else:
            post_keep = nms_keep
        assert len(post_keep) == len(scores_out) and torch.equal(cap["pre_scores"][order][post_keep], scores_out)
        return dict(**{f"sel{l}": np_(torch.sort(sel[l])[0]) for l in range(3)}, pre_level=np_(pre_level), pre_loc=np_(pre_loc),
                    pre_boxes=np_(cap["pre_boxes"]), pre_scores=np_(cap["pre_scores"]), nms_keep=np_(nms_keep), post_keep=np_(post_keep),
                    boxes=np_(boxes_out), scores=np_(scores_out))

    for tag in ("sparse", "dense"):
        f = dict(np.load(os.path.join(OUT, f"cn_infer_640_{tag}.npz")))
        hm_l = [torch.from_numpy(f[f"hm{l}"])[None, None] for l in range(3)]
        reg_l = [torch.from_numpy(f[f"reg{l}"]).permute(2, 0, 1)[None].contiguous() for l in range(3)]
        out = run(hm_l, reg_l, training=False)
        # the same run the box fixture came from: identical pre lists and outputs, bit for bit
        assert np.array_equal(out["pre_boxes"], f["pre_boxes"]) and np.array_equal(out["pre_scores"], f["pre_scores"])
        assert np.array_equal(out["boxes"], f["boxes"]) and np.array_equal(out["scores"], f["scores"])
        save(f"cn_infer_640_{tag}_idx", **{k: v for k, v in out.items() if k not in ("pre_boxes", "pre_scores", "boxes", "scores")})
    # training thresholds (4000 / 0.9 / 2000) on maps of their own: large smooth boxes so that neighbours overlap by more than 0.9
    g = torch.Generator().manual_seed(4321)
    hm_l, reg_l = [], []
    for s_ in (80, 40, 20):
        base = torch.nn.functional.interpolate(torch.randn(1, 1, s_ // 4, s_ // 4, generator=g), size=(s_, s_), mode="bilinear", align_corners=False)
        hm_l.append(base * 2.5 + torch.randn(1, 1, s_, s_, generator=g) * 0.7 + 2.0)
        rb = torch.nn.functional.interpolate(torch.randn(1, 4, s_ // 8, s_ // 8, generator=g), size=(s_, s_), mode="bilinear", align_corners=False)
        reg_l.append(torch.relu(rb * 2.0 + 9.0 + torch.randn(1, 4, s_, s_, generator=g) * 0.15))
    out_t = run(hm_l, reg_l, training=True)
    save("cn_infer_640_train_idx", **{f"hm{l}": np_(hm_l[l][0, 0]) for l in range(3)},
         **{f"reg{l}": np_(reg_l[l][0].permute(1, 2, 0).contiguous()) for l in range(3)}, **out_t)


def reference_detector(sd, shots, device_cfg="cpu"):
    """The reference's complete CenterNet2Detector built by ITS OWN __init__ / from_config chain from ITS OWN logged resolved config
    (ref:fewx/modeling/fsod/fsod_cen.py:45-78), with the builders it imports pointed at the reference's real classes."""
    ns = shims.load_roi_heads()
    cen = shims.load_fsod_cen()
    cfg = shims.logged_cfg()
    cfg.INPUT.FS.SUPPORT_SHOT = shots
    cfg.MODEL.DEVICE = device_cfg
    SS = ns.layers.ShapeSpec
    cen.build_backbone = lambda c: ns.vovnet.build_fcos_vovnet_fpn_backbone(c, SS(channels=len(c.MODEL.PIXEL_MEAN)))
    cen.build_proposal_generator = lambda c, shape: ns.fsod_rpn.CenterNet(c, shape)
    cen.build_roi_heads = ns.roi_heads.build_roi_heads
    det = cen.CenterNet2Detector(cfg)
    missing = det.load_state_dict(sd, strict=True)
    return det, cfg, ns


def gen_roi_stage(ns, sd0):
    """f1: CustomCascadeROIHeads (ref:fewx/modeling/fsod/fsod_roi_heads.py:380-520), CustomFastRCNNOutputLayers
    (ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:51-170), FastRCNNOutputLayers.box_reg_loss / predict_boxes and
    fast_rcnn_inference (d2z:modeling/roi_heads/fast_rcnn.py:118-171,490-620), ROIPooler (d2z:modeling/poolers.py), executed."""
    ns = shims.load_roi_heads()
    cfg = shims.logged_cfg()
    SS = ns.layers.ShapeSpec
    Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances
    shape = {k: SS(channels=128, stride=st) for k, st in (("p3", 8), ("p4", 16), ("p5", 32))}
    rh = ns.roi_heads.CustomCascadeROIHeads(cfg, shape)
    sd = R.synth_roi_state(sd0, SEED)
    rh.load_state_dict(sub_sd(sd, "roi_heads."), strict=True)
    g = torch.Generator().manual_seed(77)
    H, W = 160, 192
    feats = {f"p{l}": torch.randn(1, 128, H >> l, W >> l, generator=g) for l in (3, 4, 5)}
    sup8 = torch.randn(3, 128, 8, 8, generator=g) * 0.5
    sup4 = torch.randn(3, 128, 4, 4, generator=g) * 0.5
    n = 72                                                          # sizes 6..600 px: all three pyramid levels, many outside the image
    wh = torch.exp(torch.rand(n, 1, generator=g) * 4.6 + 1.8) * torch.exp(torch.randn(n, 2, generator=g) * 0.25)
    ctr = torch.rand(n, 2, generator=g) * torch.tensor([W + 20.0, H + 20.0]) - 10.0
    pb = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    pb[:8] = pb[8:16] + torch.randn(8, 4, generator=g) * 1.5        # near-duplicates: the 0.9 NMS has something to suppress
    ps = torch.rand(n, generator=g)
    cap = {}
    rh.box_pooler.register_forward_hook(lambda m, i, o: cap.__setitem__("box_features", o.detach().clone()))
    rh.box_head[0].register_forward_hook(lambda m, i, o: cap.__setitem__("h", o.detach().clone()))
    rh.box_predictor[0].register_forward_hook(lambda m, i, o: cap.__setitem__("pred", (o[0].detach().clone(), o[1].detach().clone())))
    # ---- eval
    rh.eval()
    prop = Instances((H, W))
    prop.proposal_boxes, prop.objectness_logits = Boxes(pb.clone()), ps.clone()
    prop.scores, prop.pred_classes = ps.clone(), torch.zeros(n, dtype=torch.int64)       # as CenterNet.inference leaves them
    with torch.no_grad():
        pred, _ = rh(None, feats, [sup8, sup4], [prop])
    pi = pred[0]
    sub = torch.tensor([0, 5, 17, 33, 41, 58, 71])                  # a few pooled ROIs in full (all levels), the rest through h
    save("roi_stage_eval", seed=np.int64(SEED), **{k: np_(v) for k, v in feats.items()}, sup8=np_(sup8), sup4=np_(sup4),
         proposals=np_(pb), proposal_scores=np_(ps), image_hw=np.array([H, W]), sub=np_(sub), box_features_sub=np_(cap["box_features"][sub]),
         h=np_(cap["h"]), cls_logits=np_(cap["pred"][0]), deltas=np_(cap["pred"][1]),
         pred_boxes=np_(pi.pred_boxes.tensor), scores=np_(pi.scores), pred_classes=np_(pi.pred_classes))
    # ---- train: label_and_sample_proposals (d2z:modeling/roi_heads/roi_heads.py:181-295) + the two second-stage losses
    rh.train()
    n_g = 7
    gwh = torch.rand(n_g, 2, generator=g) * 60 + 20
    gctr = torch.rand(n_g, 2, generator=g) * (torch.tensor([float(W), float(H)]) - gwh) + gwh / 2
    gtb = torch.cat([gctr - gwh / 2, gctr + gwh / 2], 1)
    tp = torch.cat([gtb[torch.randint(0, n_g, (150,), generator=g)] + torch.randn(150, 4, generator=g) * 3.0, pb], 0)
    tp[:, 2:] = torch.max(tp[:, 2:], tp[:, :2] + 2.0)
    prop = Instances((H, W))
    prop.proposal_boxes, prop.objectness_logits = Boxes(tp.clone()), torch.rand(len(tp), generator=g)
    tgt = Instances((H, W))
    tgt.gt_boxes, tgt.gt_classes = Boxes(gtb.clone()), torch.zeros(n_g, dtype=torch.int64)
    for pth in rh.parameters():
        pth.grad = None
    fl = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    torch.manual_seed(31)                                            # seeds subsample_labels' two randperm calls
    sampled, losses = rh(None, fl, [sup8, sup4], [prop], [tgt])
    (losses["loss_cls_stage0"] + losses["loss_box_reg_stage0"]).backward()
    sp = sampled[0]
    named = dict(rh.named_parameters())
    save("roi_stage_train", seed=np.int64(SEED), randperm_seed=np.int64(31), proposals=np_(tp), gt=np_(gtb),
         roi_boxes=np_(sp.proposal_boxes.tensor), roi_labels=np_(sp.gt_classes), roi_gt=np_(sp.gt_boxes.tensor),
         loss_cls=np_(losses["loss_cls_stage0"]), loss_box_reg=np_(losses["loss_box_reg_stage0"]),
         cls_logits=np_(cap["pred"][0]), deltas=np_(cap["pred"][1]),
         g_cls_w=np_(named["box_predictor.0.cls_score.weight"].grad), g_box_w=np_(named["box_predictor.0.bbox_pred.weight"].grad),
         g_fc1_b=np_(named["box_head.0.fc1.bias"].grad), g_conv3_w=np_(named["conv3.weight"].grad),
         g_conv1_b=np_(named["conv1.bias"].grad), **{f"g_{k}_sum": np_(v.grad.sum((0, 2, 3))) for k, v in fl.items() if v.grad is not None},
         dead=np.array([k for k, v in named.items() if v.grad is None]))


def grad_sample(t):
    """Fixed strided sample (<= 1024 values) of a gradient + its (l2, max|.|): what the fixtures keep of a large tensor."""
    f = t.detach().reshape(-1)
    return f[:: max(1, f.numel() // 1024)][:1024].clone(), torch.stack([f.double().norm().float(), f.abs().max()])


def gen_train_iteration(sd0):
    """The reference's COMPLETE training forward (ref:fewx/modeling/fsod/fsod_cen.py:151-308: two backbone passes, support pooling,
    SM blocks, correlation, CenterNet losses + train-mode proposals, label/sample, second stage) + backward, executed on the seeded
    synthetic sample of oracle.ref_train.synth_train_inputs.  Dropout(0.1) of SM_Block.reweighting is set to p = 0 (the one knob
    turned: it is stochastic); subsample_labels' randperm is seeded by a forward-pre-hook on roi_heads."""
    from oracle import ref_train as T
    for tag, shots, hw, n_gt, shw, in_seed, rp_seed in (("small", 4, (320, 384), 9, 112, 0, 11), ("full", 24, (640, 640), 17, 240, 4, 21)):
        sd = R.synth_roi_state(sd0, SEED)
        sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
        sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
        det, cfg, ns = reference_detector(sd, shots)
        Boxes, Instances = ns.boxes.Boxes, ns.instances.Instances
        det.train()
        for m in det.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        img, gt, sup, sbox = T.synth_train_inputs(in_seed, hw, n_gt=n_gt, shots=shots, support_hw=shw)
        inst = Instances(hw)
        inst.gt_boxes, inst.gt_classes = Boxes(gt.clone()), torch.zeros(len(gt), dtype=torch.int64)
        cap = {}
        det.roi_heads.register_forward_pre_hook(lambda m, a: (torch.manual_seed(rp_seed), cap.__setitem__("proposals", a[3][0]))[0] and None)
        orig = det.roi_heads.label_and_sample_proposals

        def spy(proposals, targets):
            out = orig(proposals, targets)
            cap["sampled"] = out[0]
            return out
        det.roi_heads.label_and_sample_proposals = spy
        cap_t = {}
        og = det.proposal_generator._get_ground_truth

        def spy_gt(*a, **k):
            out = og(*a, **k)
            cap_t["pos_inds"] = out[0].clone()
            return out
        det.proposal_generator._get_ground_truth = spy_gt
        torch.set_num_threads(8)
        losses = det([{"image": img.float(), "instances": inst, "support_images": sup, "support_bboxes": sbox.numpy()}])
        sum(losses.values()).backward()
        out = {f"loss/{k}": np_(v) for k, v in losses.items()}
        dead = []
        for k, p_ in det.named_parameters():
            if not T.is_trainable(k):
                continue
            if p_.grad is None:
                dead.append(k)
                continue
            smp, nrm = grad_sample(p_.grad)
            out["gs/" + k], out["gn/" + k] = np_(smp), np_(nrm)
        sp = cap["sampled"]
        pr = cap["proposals"]
        # Conditioning of this sample, measured on the REFERENCE itself with the sampled ROIs held fixed: (i) the same code in fp64,
        # (ii) 10 fp32 runs with every parameter multiplied by (1 + 1e-6 N(0,1)) -- rounding-sized perturbations, the size of the
        # difference between two correct fp32 implementations (or two CPUs' conv kernels).  Hard decisions (ReLU masks, min/max picks,
        # the 1e-4 heat-map cut, ignore_high_fp) flip under them; gc/<name> = the largest resulting change of that parameter's gradient
        # sample relative to its maximum.  The GPU tests bound each parameter with it.
        def rerun(dtype, sd_run):
            d2, _, _ = reference_detector(sd_run, shots)
            d2.train().to(dtype)
            for m in d2.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            sp2 = Instances(hw)
            sp2.proposal_boxes, sp2.gt_boxes = Boxes(sp.proposal_boxes.tensor.to(dtype)), Boxes(sp.gt_boxes.tensor.to(dtype))
            sp2.gt_classes, sp2.objectness_logits = sp.gt_classes.clone(), sp.objectness_logits.to(dtype)
            d2.roi_heads.label_and_sample_proposals = lambda proposals, targets: [sp2]
            i2 = Instances(hw)
            i2.gt_boxes, i2.gt_classes = Boxes(gt.to(dtype)), torch.zeros(len(gt), dtype=torch.int64)
            l2 = d2([{"image": img.to(dtype), "instances": i2, "support_images": sup.to(dtype), "support_bboxes": sbox.to(dtype).numpy()}])
            sum(l2.values()).backward()
            return l2, {k: grad_sample(p_.grad) for k, p_ in d2.named_parameters() if "gs/" + k in out}

        def spread(gsamples):
            for k, (s2, n2) in gsamples.items():
                d_ = float(np.abs(out["gs/" + k].astype(np.float64) - np_(s2).astype(np.float64)).max() / max(float(out["gn/" + k][1]), 1e-30))
                out["gc/" + k] = np.float32(max(float(out.get("gc/" + k, 0.0)), d_))
        l64, g64 = rerun(torch.float64, sd)
        spread(g64)
        out.update({f"loss64/{k}": np_(v) for k, v in l64.items()})
        gp = torch.Generator().manual_seed(99)
        for trial in range(10):
            sdp = {k: (v * (1.0 + 1e-6 * torch.randn(v.shape, generator=gp)) if v.is_floating_point() and "running_" not in k and not k.startswith("pixel_")
                       else v) for k, v in sd.items()}
            _, gpert = rerun(torch.float32, sdp)
            spread(gpert)
        save(f"train_iter_ref_{tag}", seed=np.int64(SEED), input_seed=np.int64(in_seed), randperm_seed=np.int64(rp_seed), shots=np.int64(shots),
             hw=np.array(hw), n_gt=np.int64(n_gt), support_hw=np.int64(shw), proposals=np_(pr.proposal_boxes.tensor),
             proposal_scores=np_(pr.objectness_logits), pos_inds=np_(cap_t["pos_inds"]), roi_boxes=np_(sp.proposal_boxes.tensor),
             roi_labels=np_(sp.gt_classes), roi_gt=np_(sp.gt_boxes.tensor), dead=np.array(dead), **out)


def gen_state_dict_layout():
    """Key -> shape of the reference detector's state_dict, from the executed reference modules."""
    det, cfg, ns = reference_detector(R.synth_roi_state(R.synth_state_dict(SEED), SEED), 24)
    keys = list(det.state_dict().keys())
    shapes = [",".join(str(int(d)) for d in v.shape) for v in det.state_dict().values()]
    params = {k for k, _ in det.named_parameters()}
    save("state_dict_layout", keys=np.array(keys), shapes=np.array(shapes), is_param=np.array([k in params for k in keys]),
         n_params=np.int64(sum(p.numel() for p in det.parameters())))


def gen_demo_images():
    """BASELINE configs[0]'s inputs: ref:directory/0000{0,1}.png (the reference's only shipped images; 300x300 RGB), decoded and
    resized as the reference's predictor does (ref:predictor.py: read BGR, ResizeShortestEdge(MIN_SIZE_TEST = 320) = PIL bilinear,
    `log:805`), stored as uint8 BGR CHW.  Data only; the R-50-C4 model of that config is out of scope, the VoVNet path
    runs on these inputs instead."""
    from PIL import Image
    out = []
    for n in ("00000", "00001"):
        rgb = np.asarray(Image.open(os.path.join(shims.REF, "directory", n + ".png")).convert("RGB"))
        bgr = np.ascontiguousarray(rgb[:, :, ::-1])
        h, w = bgr.shape[:2]
        scale = 320.0 / min(h, w)
        nh, nw = int(h * scale + 0.5), int(w * scale + 0.5)
        res = np.asarray(Image.fromarray(bgr).resize((nw, nh), Image.BILINEAR))
        out.append(np.ascontiguousarray(res.transpose(2, 0, 1)))
    save("demo_images_320", images=np.stack(out), orig_hw=np.array([300, 300]))


def synth_support_df(seed=7, n_img=30, per_img=3):
    """A stand-in for datasets/coco/*_shot_support_df.pkl (the ore dataset is not shipped): the columns generate_support reads."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows, aid = [], 1000
    for img in range(n_img):
        for k in range(per_img):
            x0, y0 = rng.uniform(10, 90, 2)
            rows.append({"id": aid, "image_id": 500 + img, "category_id": 1 + (img % 3),
                         "file_path": f"support/{aid}.jpg", "support_box": [float(x0), float(y0), float(x0 + rng.uniform(40, 120)), float(y0 + rng.uniform(40, 120))]})
            aid += 1
    return pd.DataFrame(rows)


def synth_crop(path):
    """Deterministic 240x240x3 uint8 'support crop' for a file path (what utils.read_image would decode)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(path.encode()))
    return rng.integers(0, 256, (240, 240, 3), dtype=np.uint8)


def gen_generate_support():
    """f3: DatasetMapperWithSupport.generate_support (ref:fewx/data/dataset_mapper.py:198-269) executed on a synthetic support
    dataframe: which support annotations are drawn (pandas .sample(random_state=annotation id), exclusion of the query image and of
    already used ids), their order, boxes and class flags, for 1-way and 2-way."""
    dm = shims.load_dataset_mapper(lambda path, format=None: synth_crop(path))
    df = synth_support_df()
    out = {}
    for tag, way, shot, q in (("w1", 1, 9, 4), ("w2", 2, 4, 17)):
        M = dm.DatasetMapperWithSupport
        mp = M.__new__(M)
        mp.support_way, mp.support_shot, mp.img_format, mp.support_df = way, shot, "BGR", df
        row = df.iloc[q]
        dd = {"annotations": [{"id": int(a)} for a in df.loc[df["image_id"] == row["image_id"], "id"].tolist()[:2]]}
        dd["annotations"][0]["id"] = int(row["id"])
        data, boxes, cls = mp.generate_support(dd)
        out[f"{tag}_query_ids"] = np.array([a["id"] for a in dd["annotations"]], np.int64)
        out[f"{tag}_boxes"] = boxes
        out[f"{tag}_cls"] = np.array(cls, np.int64)
        out[f"{tag}_pix"] = data[:, :, ::60, ::60].copy()                 # a 4x4 sample of every crop identifies it
        out[f"{tag}_way_shot"] = np.array([way, shot], np.int64)
    save("generate_support", seed=np.int64(7), **out)


def synth_dataset_dicts(seed=11, n_img=9):
    """A small detectron2-format dataset (list of dicts) with several categories per image, crowd annotations and an image whose
    annotations are all crowd: the input of the reference's per-category split."""
    rng = np.random.default_rng(seed)
    out, aid = [], 100
    for i in range(n_img):
        anns = []
        for _ in range(int(rng.integers(1, 6))):
            x, y, w, h = [float(v) for v in rng.integers(5, 200, 4)]
            anns.append({"id": aid, "bbox": [x, y, w, h], "bbox_mode": 1, "category_id": int(rng.integers(0, 3)),
                         "iscrowd": int(rng.random() < 0.2), "segmentation": [[x, y, x + w, y, x + w, y + h]], "keypoints": [1, 2, 3]})
            aid += 1
        if i == 4:
            for a in anns:
                a["iscrowd"] = 1
        out.append({"file_name": f"img_{i}.png", "height": 300 + i, "width": 400 + i, "image_id": 1000 + i, "annotations": anns})
    return out


def gen_dataset_split():
    """b/f3: the reference's own `fsod_get_detection_dataset_dicts` (ref:fewx/data/build.py:27-106) executed on a synthetic registered
    dataset: for a name containing 'train' every image record is split into one record per category (image_id dropped, segmentation /
    keypoints popped, crowd-only records filtered twice); any other name passes through.  Stored as JSON inside the npz."""
    import copy
    import json
    D2 = shims.d2_root()
    shims.setup()
    shims.mod("termcolor", colored=lambda s, *a, **k: s)
    shims.mod("fvcore.common", file_io=None)
    shims.mod("fvcore.common.file_io", PathManager=None)
    shims.mod("detectron2.utils.file_io", PathManager=None)
    sys.modules["detectron2.utils.logger"]._log_api_usage = lambda *a, **k: None
    sys.modules["detectron2.utils.env"].seed_all_rng = None
    shims.pkg("detectron2.data", D2 + "/data")
    cat = shims.load("detectron2.data.catalog", D2 + "/data/catalog.py")
    shims.mod("detectron2.data.common", AspectRatioGroupedDataset=None, DatasetFromList=None, MapDataset=None)
    shims.mod("detectron2.data.dataset_mapper", DatasetMapper=None)
    shims.mod("detectron2.data.detection_utils", check_metadata_consistency=lambda *a, **k: None)
    shims.mod("detectron2.data.samplers", InferenceSampler=None, RepeatFactorTrainingSampler=None, TrainingSampler=None)
    shims.load("detectron2.data.build", D2 + "/data/build.py")
    shims.pkg("fewx.data", shims.REF + "/fewx/data")
    fb = shims.load("fewx.data.build", shims.REF + "/fewx/data/build.py")
    dicts = synth_dataset_dicts()
    cat.DatasetCatalog.register("synth_ore_train", lambda: copy.deepcopy(dicts))
    cat.DatasetCatalog.register("synth_ore_val", lambda: copy.deepcopy(dicts))
    tr = fb.fsod_get_detection_dataset_dicts(["synth_ore_train"], filter_empty=True)
    va = fb.fsod_get_detection_dataset_dicts(["synth_ore_val"], filter_empty=False)
    print(f"  dataset split: {len(dicts)} images -> {len(tr)} per-category training records, {len(va)} test records")
    save("dataset_split", input_json=np.array(json.dumps(dicts)), train_json=np.array(json.dumps(tr)), test_json=np.array(json.dumps(va)))


def gen_eval_end_to_end(sd0, shots=5):
    """The reference's eval path END TO END, its own code only: `CenterNet2Detector.init_model` (ref:fewx/modeling/fsod/
    fsod_cen.py:309-408) executed in a scratch working directory holding a synthetic ./datasets/coco/10_shot_support_df.pkl (it walks
    the dataframe, computes the support features, writes ./support_dir/support_feature.pkl and calls sys.exit), then
    `CenterNet2Detector.inference` (ref:...:417-535, incl. `_postprocess` -> d2 detector_postprocess) on the two shipped demo images
    (BASELINE configs[0]'s inputs, 320x320 -> 300x300 output) with that pickle as `support_dict`.  Only utils.read_image (no image
    files) and MetadataCatalog (unused result) are stand-ins."""
    import pickle
    import tempfile
    sd = R.synth_roi_state(sd0, SEED)
    sd["roi_heads.box_head.0.fc1.weight"] = sd["roi_heads.box_head.0.fc1.weight"] * 0.02
    sd["proposal_generator.centernet_head.agn_hm.bias"] = torch.full((1,), -2.0)
    det, cfg, ns = reference_detector(sd, shots)
    cen = shims.load_fsod_cen()
    sys.modules["detectron2.structures"].ROIMasks = type("ROIMasks", (), {})
    pp = shims.load("d2z.modeling.postprocessing", shims.d2_root() + "/modeling/postprocessing.py")
    cen.detector_postprocess = pp.detector_postprocess
    from oracle import ref_train as T
    cen.utils = types.SimpleNamespace(read_image=T.eval_support_crop)
    cen.MetadataCatalog = types.SimpleNamespace(get=lambda name: None)
    det.eval()
    df = T.eval_support_df(shots)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "datasets", "coco"))
        df.to_pickle(os.path.join(tmp, "datasets", "coco", "10_shot_support_df.pkl"))
        os.chdir(tmp)
        try:
            with torch.no_grad():
                try:
                    det.init_model()
                    raise RuntimeError("init_model was expected to exit after writing the pickle")
                except SystemExit:
                    pass
            with open(os.path.join(tmp, "support_dir", "support_feature.pkl"), "rb") as f:
                support = pickle.load(f)
        finally:
            os.chdir(cwd)
    det.support_dict = support
    out = {f"support_{k}": np_(v[1]) for k, v in support.items()}
    imgs = np.load(os.path.join(OUT, "demo_images_320.npz"))["images"]
    # what never leaves `inference`: the first stage's proposals (a forward hook on the reference's CenterNet module) and the second
    # stage's per-proposal rows as the reference hands them to fast_rcnn_inference (boxes after apply_deltas, class probabilities) --
    # the inputs of the two NMS decisions, so a test can tell a near-tie flip from a wrong value
    seen = {}
    hook = det.proposal_generator.register_forward_hook(lambda mod, inp, o: seen.__setitem__("prop", o[0][0]))
    rh_glob = type(det.roi_heads)._forward_box.__globals__         # the namespace the reference's method resolves the name in
    fri = rh_glob["fast_rcnn_inference"]

    def spy(boxes, scores, *a, **k):
        seen["s2_boxes"], seen["s2_scores"] = boxes[0].detach().clone(), scores[0].detach().clone()
        return fri(boxes, scores, *a, **k)
    rh_glob["fast_rcnn_inference"] = spy
    for i in range(2):
        with torch.no_grad():
            res = det.inference([{"image": torch.from_numpy(imgs[i]), "height": 300, "width": 300}])[0]["instances"]
        out[f"img{i}_prop_boxes"] = np_(seen["prop"].proposal_boxes.tensor)
        out[f"img{i}_prop_scores"] = np_(seen["prop"].objectness_logits)
        out[f"img{i}_stage2_boxes"] = np_(seen["s2_boxes"])
        out[f"img{i}_stage2_scores"] = np_(seen["s2_scores"])
        out[f"img{i}_boxes"] = np_(res.pred_boxes.tensor)
        out[f"img{i}_scores"] = np_(res.scores)
        out[f"img{i}_classes"] = np_(res.pred_classes)
        print(f"  image {i}: {len(res)} detections, scores {res.scores[:4].tolist()}")
    hook.remove()
    rh_glob["fast_rcnn_inference"] = fri
    save("eval_end_to_end", shots=np.int64(shots), df_seed=np.int64(23), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "split":
        gen_dataset_split()
    elif len(sys.argv) > 1 and sys.argv[1] == "eval":
        torch.manual_seed(0)
        gen_eval_end_to_end(R.synth_state_dict(SEED))
    elif len(sys.argv) > 1 and sys.argv[1] == "indices":
        gen_cn_indices()
    else:
        main()
