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
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items()))


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


if __name__ == "__main__":
    main()
