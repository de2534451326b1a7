"""CPU restatement of ONE training iteration of the reference detector -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path never does.
torch fp32 on the CPU + torch.autograd; every stage cites the reference lines it follows.  Pinning: the backbone/FPN, SM_Block,
correlation, CenterNet head and the CenterNet targets/losses used here are the functions of oracle/ref_model.py that
tests/test_oracle_golden.py pins against outputs of the executed reference files.  The second-stage TRAINING helpers
that are pure Python in the vendored detectron2 -- pairwise_iou, Matcher([0.6],[0,1]), subsample_labels, Box2BoxTransform.get_deltas
-- are pinned bit-exactly by tests/golden/roi_train_pieces.npz (those files executed by oracle/refrun/gen_golden.py).  ROIAlign
(torchvision C++) and fvcore's smooth_l1_loss (beta = 0, i.e. L1) are un-vendored: restated from the published algorithm,
"parity unpinned" exactly like the eval second stage.

    ref:fewx/modeling/fsod/fsod_cen.py:151-308   CenterNet2Detector.forward (training branch)
    ref:fewx/modeling/fsod/fsod_rpn.py:644-700   CenterNet.forward (training: losses + proposals with the *_TRAIN thresholds)
    d2z:modeling/roi_heads/roi_heads.py:181-295  label_and_sample_proposals / _sample_proposals
    d2z:modeling/matcher.py:60-103, d2z:modeling/sampling.py:9-54, d2z:modeling/proposal_generator/proposal_utils.py:140-201
    ref:fewx/modeling/fsod/fsod_roi_heads.py:404-520  _forward_box / _run_stage (second definition: the live one)
    ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:52-81,131-157; d2z:modeling/roi_heads/fast_rcnn.py:490-530
    d2z:modeling/box_regression.py:41-75 (get_deltas)
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_model as R

Tensor = torch.Tensor

TRAINABLE_PREFIXES = ("backbone.fpn_", "backbone.bottom_up.stage4.", "backbone.bottom_up.stage5.", "proposal_generator.",
                      "roi_heads.", "vip_p", "conv1.", "conv2.", "conv3.")


def is_trainable(name: str) -> bool:
    """FREEZE_AT = 3 freezes stem + stage2 + stage3 (d2z:modeling/backbone/vovnet.py:440-461); FrozenBN has buffers only."""
    return name.startswith(TRAINABLE_PREFIXES) and not any(s in name for s in ("running_", "/norm."))


def pairwise_iou(a: Tensor, b: Tensor) -> Tensor:
    """d2z:structures/boxes.py:286-310."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b - inter), torch.zeros(1))


def label_proposals(proposal_boxes: Tensor, gt_boxes: Tensor, iou_thresh: float = 0.6) -> Tuple[Tensor, Tensor, Tensor]:
    """proposal_append_gt + Matcher([0.6], [0, 1]) for one class.  Returns (all boxes [N+G,4], matched gt index [N+G],
    label [N+G]: 0 = foreground class 0, 1 = background (num_classes))."""
    boxes = torch.cat([proposal_boxes, gt_boxes], 0)
    if gt_boxes.shape[0] == 0:
        return boxes, torch.zeros(len(boxes), dtype=torch.int64), torch.ones(len(boxes), dtype=torch.int64)
    q = pairwise_iou(gt_boxes, boxes)
    vals, idx = q.max(0)
    labels = torch.where(vals >= iou_thresh, torch.zeros_like(idx), torch.ones_like(idx))
    return boxes, idx, labels


def sample_labels(labels: Tensor, batch: int, positive_fraction: float, perm: Callable[[int], Tensor]) -> Tensor:
    """subsample_labels (sampling.py:9-54): `perm(n)` stands in for torch.randperm(n)."""
    pos = torch.nonzero(labels == 0).squeeze(1)
    neg = torch.nonzero(labels == 1).squeeze(1)
    n_pos = min(pos.numel(), int(batch * positive_fraction))
    n_neg = min(neg.numel(), batch - n_pos)
    return torch.cat([pos[perm(pos.numel())[:n_pos]], neg[perm(neg.numel())[:n_neg]]], 0)


def get_deltas(src: Tensor, tgt: Tensor, weights=(10.0, 10.0, 5.0, 5.0)) -> Tensor:
    """Box2BoxTransform.get_deltas (box_regression.py:41-75)."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), 1)


def preprocess_batch(imgs: Tensor, div: int = 32) -> Tensor:
    """(x - mean)/std then zero-pad to a multiple of 32 (fsod_cen.py:540-551).  imgs [N,3,H,W] BGR."""
    mean = torch.tensor(R.PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(R.PIXEL_STD).view(1, 3, 1, 1)
    x = (imgs.float() - mean) / std
    H, W = x.shape[2:]
    return F.pad(x, (0, (W + div - 1) // div * div - W, 0, (H + div - 1) // div * div - H))


def second_stage_losses(feats: Sequence[Tensor], roi_boxes: Tensor, roi_labels: Tensor, roi_gt: Tensor, sup8: Tensor,
                        sd: Dict[str, Tensor], prefix: str = "roi_heads.") -> Dict[str, Tensor]:
    """Second stage in training mode for one image, on already sampled ROIs (fsod_roi_heads.py:404-520 `_forward_box`/`_run_stage`,
    the live second definition; custom_fast_rcnn.py:52-81,131-157 `losses`; d2z fast_rcnn.py:490-530 `box_reg_loss`): ROIAlign 8x8 of
    the query pyramid, DSA mix with the mean support feature, fc1, predictor, softmax cross-entropy (mean over the ROIs) and
    smooth-L1 with beta 0 (= L1) over the foreground rows, normalised by the number of ROIs.  `_ScaleGradient` is the identity
    for one cascade stage.  Pinned by tests/golden/roi_stage_train.npz (the reference classes executed)."""
    box_feat = R.roi_pool_levels(feats, roi_boxes, 8)
    h = R.roi_head_features(box_feat, sup8, sd, prefix)
    p = prefix + "box_predictor.0."
    scores = R.dense_linear(h, sd[p + "cls_score.weight"], sd[p + "cls_score.bias"])
    deltas = R.dense_linear(h, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"])
    loss_cls = F.cross_entropy(scores, roi_labels, reduction="mean")
    fg = torch.nonzero(roi_labels == 0).squeeze(1)
    tgt = get_deltas(roi_boxes[fg], roi_gt[fg])
    loss_box = (deltas[fg] - tgt).abs().sum() / max(roi_labels.numel(), 1)
    return {"box_features": box_feat, "h": h, "scores": scores, "deltas": deltas, "loss_cls": loss_cls, "loss_box_reg": loss_box}


def train_iteration(sd: Dict[str, Tensor], image: Tensor, gt_boxes: Tensor, support_images: Tensor, support_boxes: Tensor,
                    perm: Callable[[int], Tensor], num_gpus: int = 1, batch_per_image: int = 128,
                    positive_fraction: float = 0.5, iou_thresh: float = 0.6, pre_topk: int = 4000, nms_thresh: float = 0.9,
                    post_topk: int = 2000, score_thresh: float = 1e-5, roi_override: Optional[Dict[str, Tensor]] = None) -> Dict[str, object]:
    """One query image [3,H,W] + its support set ([N,3,h,w], [N,4]); batch size 1 per process as the reference trains.
    `sd` holds leaf tensors (requires_grad where trainable); returns the 5 losses (graph attached) and the intermediates the
    parity tests compare (proposals, sampled indices, targets)."""
    from . import decode as odec
    H, W = image.shape[1:]
    x = preprocess_batch(image[None])
    feats = R.backbone_fpn(x, sd)                                                 # query p3..p5
    sfeats = R.backbone_fpn(preprocess_batch(support_images), sd)                 # support p3..p5 [N,128,32|16|8,..]
    levels = ("p3", "p4", "p5")
    # --- support prototypes (fsod_cen.py:216-227) and correlation (:229-275)
    protos = {k: R.support_prototype(sfeats[k], sd, 3 + i) for i, k in enumerate(levels)}
    pos = [R.correlation(feats[k], protos[k], sd["conv3.weight"], sd["conv3.bias"]) for k in levels]
    # --- CenterNet head, targets, losses (fsod_rpn.py:644-779)
    regs, hms = R.centernet_head(pos, sd)
    shapes = [tuple(r.shape[2:]) for r in regs]
    pos_inds, reg_targets, hm_targets = R.centernet_targets([gt_boxes], shapes)
    reg_flat = torch.cat([r.permute(0, 2, 3, 1).reshape(-1, 4) for r in regs], 0)
    hm_flat = torch.cat([h.permute(0, 2, 3, 1).reshape(-1) for h in hms], 0)
    cn = R.centernet_losses(reg_flat, hm_flat, pos_inds, reg_targets, hm_targets, num_gpus=num_gpus)
    # --- proposals with the training thresholds (fsod_rpn.py:677-692, 1185-1210); detached
    with torch.no_grad():
        d = odec.decode_nms([h[0, 0].numpy() for h in hms], [r[0].permute(1, 2, 0).numpy() for r in regs], (8, 16, 32),
                            score_thresh, pre_topk, nms_thresh, post_topk)
        proposals = torch.from_numpy(d["boxes"])
        boxes, matched, labels = label_proposals(proposals, gt_boxes, iou_thresh)
        sampled = sample_labels(labels, batch_per_image, positive_fraction, perm)
        roi_boxes, roi_labels, roi_gt = boxes[sampled], labels[sampled], gt_boxes[matched[sampled]] if len(gt_boxes) else boxes[sampled]
        if roi_override is not None:      # conditioning runs: the sampled ROIs of another run (fp64 vs fp32 decide NMS ties differently)
            roi_boxes, roi_labels, roi_gt = (roi_override[k] for k in ("boxes", "labels", "gt"))
    # --- second stage
    flist = [feats[k] for k in levels]
    sup8 = torch.cat([R.roi_pool_levels([sfeats[k][n:n + 1] for k in levels], support_boxes[n:n + 1], 8)
                      for n in range(support_images.shape[0])], 0)
    st = second_stage_losses(flist, roi_boxes, roi_labels, roi_gt, sup8, sd)
    box_feat, h, scores, deltas, loss_cls, loss_box = (st[k] for k in ("box_features", "h", "scores", "deltas", "loss_cls", "loss_box_reg"))
    losses = {"loss_cls_stage0": loss_cls, "loss_box_reg_stage0": loss_box,
              "loss_centernet_loc": cn["loss_centernet_loc"], "loss_centernet_agn_pos": cn["loss_centernet_agn_pos"],
              "loss_centernet_agn_neg": cn["loss_centernet_agn_neg"]}
    return {"losses": losses, "proposals": proposals, "proposal_scores": torch.from_numpy(d["scores"]), "sampled": sampled,
            "roi_boxes": roi_boxes, "roi_labels": roi_labels, "roi_gt": roi_gt, "pos_inds": pos_inds, "reg_targets": reg_targets,
            "hm_targets": hm_targets, "features": feats, "support_features": sfeats, "prototypes": protos, "pos_features": pos,
            "reg": regs, "hm": hms, "box_features": box_feat, "support_8": sup8, "h": h, "scores": scores, "deltas": deltas}


def synth_train_inputs(seed: int = 0, hw: Tuple[int, int] = (640, 640), n_gt: int = 17, shots: int = 24, support_hw: int = 240):
    """SURVEY 8d synthetic training sample: query image, 15-20 gt boxes of side 30..150 px, `shots` support crops with a box each."""
    g = torch.Generator().manual_seed(seed + 500)
    H, W = hw
    img = R.synth_image(seed, H, W)
    wh = torch.rand(n_gt, 2, generator=g) * 120 + 30
    ctr = torch.rand(n_gt, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh) + wh / 2
    gt = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    sup = torch.stack([R.synth_image(seed + 10 + i, support_hw, support_hw).float() for i in range(shots)], 0)
    side = torch.rand(shots, 2, generator=g) * 120 + 80
    c = torch.rand(shots, 2, generator=g) * (support_hw - side) + side / 2
    sbox = torch.cat([c - side / 2, c + side / 2], 1)
    return img, gt, sup, sbox


def leaf_state(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Clone a state dict into autograd leaves (requires_grad on the trainable parameters)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and is_trainable(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# ---- inputs of tests/golden/eval_end_to_end.npz (regenerated from the seed by the generator and by the tests) ----------------
def eval_support_df(shots, extra=2, seed=23):
    """One-category support dataframe for the end-to-end eval fixture (shots + `extra` rows: init_model keeps the first `shots`)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    rows = []
    for k in range(shots + extra):
        x0, y0 = rng.uniform(10, 60, 2)
        rows.append({"id": 7000 + k, "image_id": 300 + k, "category_id": 1, "file_path": f"support/{7000 + k}.jpg",
                     "support_box": [float(x0), float(y0), float(x0 + rng.uniform(90, 160)), float(y0 + rng.uniform(90, 160))]})
    return pd.DataFrame(rows)


def eval_support_crop(path, format=None):
    """Deterministic smooth 240x240x3 uint8 BGR 'support crop' for a path (low-frequency pattern + noise, so features are not flat)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(os.path.basename(path).encode()))
    yy, xx = np.mgrid[0:240, 0:240].astype(np.float32)
    img = np.zeros((240, 240, 3), np.float32)
    for c in range(3):
        fx, fy, ph = rng.uniform(0.01, 0.08, 3)
        img[:, :, c] = 120 + 70 * np.sin(fx * xx + fy * yy + ph * 40) + rng.normal(0, 12, (240, 240))
    return np.clip(img, 0, 255).astype(np.uint8)
