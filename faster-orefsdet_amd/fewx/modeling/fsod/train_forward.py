"""Training forward of CenterNet2Detector: returns the reference's loss dict with an autograd tape whose
FLOP-carrying nodes are libore_hip.so kernels (orehip.autograd).

Reference flow restated here (the reference trains with IMS_PER_BATCH = 1 per GPU; for B > 1 the two backbone passes, conv3 and the
head run batched as the reference's do, the per-image stages loop, and the losses are the mean over the images -- SURVEY App. C.1):
    ref:fewx/modeling/fsod/fsod_cen.py:151-308     CenterNet2Detector.forward, training branch
    ref:fewx/modeling/fsod/fsod_rpn.py:644-700     CenterNet.forward: head -> targets -> 3 losses -> proposals (*_TRAIN thresholds)
    d2z:modeling/roi_heads/roi_heads.py:181-295    label_and_sample_proposals (append gt, IoU matcher 0.6, 128 samples, <= 50 % fg)
    ref:fewx/modeling/fsod/fsod_roi_heads.py:404-520  _forward_box / _run_stage (the second, live definition)
    ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:52-81,131-157   losses
What runs where: convs / linears (forward, data gradient, weight gradient), ROIAlign fwd/bwd, CenterNet targets + losses + their
gradient, the depthwise correlation fwd/bwd, GroupNorm fwd/bwd, the eSE scale and its gradient, max-pool fwd/bwd, the FPN top-down
add and its gradient, top-k / decode / NMS are HIP kernels.  Still torch tensor ops on the device this round (small, listed in
DESIGN.md): the [B,C]-sized gate algebra of eSE, SM_Block pointwise math, proposal matching/sampling, the two ROI losses.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

from detectron2.layers import nhwc_view

LEVELS = ("p3", "p4", "p5")


def _normalise_pad(imgs: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, div: int) -> torch.Tensor:
    """(x - mean) / std, then zero-pad bottom/right to a multiple of `div` (fsod_cen.py:540-551).  imgs [N,3,H,W]."""
    x = (imgs.float() - mean.view(1, -1, 1, 1)) / std.view(1, -1, 1, 1)
    H, W = x.shape[-2:]
    return F.pad(x, (0, (W + div - 1) // div * div - W, 0, (H + div - 1) // div * div - H)).contiguous()


def pairwise_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """d2z:structures/boxes.py:286-310."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b - inter), torch.zeros(1, device=a.device))


def get_deltas(src: torch.Tensor, tgt: torch.Tensor, weights) -> torch.Tensor:
    """Box2BoxTransform.get_deltas (d2z:modeling/box_regression.py:41-75)."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), 1)


def head_train(head, feats_nhwc: List[torch.Tensor]) -> List[torch.Tensor]:
    """CenterNetHead.forward with gradients: per level [1,H,W,128] -> [1,H,W,16] (0..3 ltrb after Scale+ReLU, 4 logit, rest 0).
    ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:141-161."""
    from orehip import autograd as A
    tower, gn = head.bbox_tower[0], head.bbox_tower[1]
    w5 = torch.cat([head.bbox_pred.weight, head.agn_hm.weight], 0)
    b5 = torch.cat([head.bbox_pred.bias, head.agn_hm.bias], 0)
    outs = []
    for l, x in enumerate(feats_nhwc):
        t = A.conv(x, tower.weight, tower.bias)
        t = A.group_norm_relu(t, gn.weight, gn.bias, gn.num_groups, gn.eps, True)       # statistics per image, one launch per kernel
        o = A.conv(t, w5, b5)
        reg = F.relu(o[..., :4] * head.scales[l].scale)
        outs.append(torch.cat([reg, o[..., 4:5], torch.zeros(*o.shape[:3], 11, device=o.device)], -1))
    return outs


def dense_part(model, xq: torch.Tensor, xs: torch.Tensor):
    """The shape-static part of a training iteration for B query images at once: query and support pyramids (two batched backbone
    passes, fsod_cen.py:165,179), per-image support prototypes and correlation (:197-275), conv3 and the head batched over B.
    xq [B,3,H,W], xs [B*N,3,h,w] normalised + padded, image b owning support rows b*N..(b+1)*N.
    Returns (q3, q4, q5 [B,..], s3, s4, s5 [B*N,..] as NHWC, head3, head4, head5 [B,H,W,16])."""
    from orehip import autograd as A
    B = xq.shape[0]
    N = xs.shape[0] // B
    feats = model.backbone(xq)
    sfeats = model.backbone(xs)
    pos = []
    for i, k in enumerate(LEVELS):
        size = (32, 16, 8)[i]
        sf = sfeats[k]
        if sf.shape[-2:] != (size, size):
            sf = F.adaptive_avg_pool2d(sf, (size, size))
        # support prototypes: avg-pool to 32/16/8, SM_Block, the reference's H<->W swapping permute, mean over the image's shots
        v = getattr(model, f"vip_p{3 + i}")(nhwc_view(sf)).permute(0, 3, 2, 1)
        proto = v.mean(0, True) if B == 1 else v.reshape(B, N, *v.shape[1:]).mean(1)
        k11 = F.adaptive_avg_pool2d(proto, (1, 1))[:, :, 0, 0]                  # support kernels (fsod_cen.py:229-231), [B,C]
        k13 = F.adaptive_avg_pool2d(proto, (1, 3))[:, :, 0, :]                  # [B,C,3]
        k31 = F.adaptive_avg_pool2d(proto, (3, 1))[:, :, :, 0]
        # [B,H,W,2C] = [attn | q], every image correlated with its own support kernels in one launch
        cat = A.correlation_cat(nhwc_view(feats[k]), k11, k13, k31)
        pos.append(A.conv(cat, model.conv3.weight, model.conv3.bias, None, None, True))
    heads = head_train(model.proposal_generator.centernet_head, pos)
    return tuple(nhwc_view(feats[k]) for k in LEVELS) + tuple(nhwc_view(sfeats[k]) for k in LEVELS) + tuple(heads)


class _DensePart(torch.nn.Module):
    """dense_part as a Module so torch.cuda.make_graphed_callables can capture its forward AND backward into two hipGraphs
    (one replay each per iteration instead of ~1000 launches).  The weight repacks are forced inside the capture so the replay
    always repacks from the current parameters."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, xq, xs):
        from orehip import autograd as A
        A.weights_changed()
        return dense_part(self.model, xq, xs)


def graphed_dense_part(model, xq, xs):
    """Cached hipGraph capture of dense_part for these input shapes (opt-in: model.train_graph = True).  Falls back to the eager
    path (and remembers why) if the capture fails."""
    key = (tuple(xq.shape), tuple(xs.shape))
    cache = model.__dict__.setdefault("_ore_train_graphs", {})
    if key not in cache:
        try:
            mod = _DensePart(model)
            cache[key] = torch.cuda.make_graphed_callables(mod, (xq.clone(), xs.clone()), num_warmup_iters=3, allow_unused_input=True)
        except Exception as ex:                                   # noqa: BLE001 -- capture is an optimisation, never a requirement
            cache[key] = None
            model.__dict__["_ore_train_graph_error"] = repr(ex)[:500]
    g = cache[key]
    return g(xq, xs) if g is not None else dense_part(model, xq, xs)


def first_stage(pg, heads: List[torch.Tensor], gt_boxes: torch.Tensor):
    """CenterNet.forward, training branch, after the head (ref:fewx/modeling/fsod/fsod_rpn.py:658-700): ground truth, the three
    losses, and the proposals with the *_TRAIN thresholds, WITHOUT a host sync.  heads[l] [1,H,W,16] (0..3 ltrb after Scale+ReLU,
    4 heatmap logit).  Returns (detect outputs with the device-side counts, losses dict, targets dict)."""
    import orehip
    from orehip import autograd as A
    dev = heads[0].device
    shapes = [tuple(h.shape[1:3]) for h in heads]
    rows = torch.cat([h.reshape(-1, h.shape[-1]) for h in heads], 0)
    tg = orehip.centernet_targets([gt_boxes], shapes, pg.strides, pg.sizes_of_interest, pg.hm_min_overlap, pg.min_radius, device=dev)
    hp = dict(gamma=pg.loss_gamma, beta=pg.hm_focal_beta, sigmoid_clamp=pg.sigmoid_clamp, ignore_high_fp=pg.ignore_high_fp,
              alpha=pg.hm_focal_alpha, pos_weight=pg.pos_weight, neg_weight=pg.neg_weight, reg_weight=pg.reg_weight)
    l3 = A.centernet_losses(rows, tg["reg_targets"], tg["hm_targets"], tg["pos_inds"], tg["pos_count"], hp)
    with torch.no_grad():
        o = orehip.detect([h[0].detach().contiguous() for h in heads], pg.strides, pg.score_thresh, pg.pre_nms_topk_train,
                          pg.nms_thresh_train, pg.post_nms_topk_train)
    losses = {"loss_centernet_loc": l3[0], "loss_centernet_agn_pos": l3[1], "loss_centernet_agn_neg": l3[2]}
    return o, losses, tg


def proposal_losses_and_proposals(pg, heads: List[torch.Tensor], gt_boxes: torch.Tensor):
    """first_stage for one image + the host read of the proposal count.  Returns (proposal boxes [n,4], scores [n], losses, targets)."""
    o, losses, tg = first_stage(pg, heads, gt_boxes)
    n = int(o["counts"][1].item())
    return o["out_boxes"][:n], o["out_scores"][:n], losses, tg


@torch.no_grad()
def label_and_sample(rh, proposals: torch.Tensor, gt_boxes: torch.Tensor, perm: Callable[[int], torch.Tensor]):
    """label_and_sample_proposals for one image and one foreground class (d2z:modeling/roi_heads/roi_heads.py:181-295):
    append the gt boxes, IoU matcher (>= IOUS[0] -> foreground), BATCH_SIZE_PER_IMAGE samples with <= POSITIVE_FRACTION
    foreground.  Returns (sampled indices, roi boxes, labels (0 fg / 1 bg), matched gt boxes)."""
    dev = proposals.device
    boxes = torch.cat([proposals, gt_boxes], 0) if rh.proposal_append_gt else proposals
    if gt_boxes.shape[0]:
        vals, midx = pairwise_iou(gt_boxes, boxes).max(0)
        labels = torch.where(vals >= rh.iou_threshold, torch.zeros_like(midx), torch.ones_like(midx))
    else:
        midx = torch.zeros(len(boxes), dtype=torch.int64, device=dev)
        labels = torch.ones(len(boxes), dtype=torch.int64, device=dev)
    p_idx = torch.nonzero(labels == 0).squeeze(1)
    n_idx = torch.nonzero(labels == 1).squeeze(1)
    n_pos = min(p_idx.numel(), int(rh.batch_size_per_image * rh.positive_fraction))
    n_neg = min(n_idx.numel(), rh.batch_size_per_image - n_pos)
    sampled = torch.cat([p_idx[perm(p_idx.numel()).to(dev)[:n_pos]], n_idx[perm(n_idx.numel()).to(dev)[:n_neg]]], 0)
    roi_gt = gt_boxes[midx[sampled]] if gt_boxes.shape[0] else boxes[sampled]
    return sampled, boxes[sampled].contiguous(), labels[sampled], roi_gt


def roi_stage_losses(rh, qf: List[torch.Tensor], sup8: torch.Tensor, roi_boxes, roi_labels, roi_gt, strides,
                     roi_image: Optional[torch.Tensor] = None, rois_per_image: Optional[List[int]] = None):
    """_run_stage + CustomFastRCNNOutputLayers.losses for the single cascade stage (ref:fewx/modeling/fsod/fsod_roi_heads.py:459-520,
    custom_fast_rcnn.py:52-81).  One image: qf[l] [H,W,C] query pyramid (NHWC), sup8 [N, P*P*C] pooled support features ordered
    [pos][c].  B images in one pass: qf[l] [B,H,W,C], sup8 [B,N,P*P*C], the ROIs of all images concatenated with roi_image [R]
    (int32 image of each ROI) and rois_per_image (host ints); each image's losses keep their own 1/R_b normaliser and the result is
    the mean over the images."""
    from orehip import autograd as A
    R = roi_boxes.shape[0]
    C = qf[0].shape[-1]
    P = rh.pooler_resolution
    if roi_image is None:
        x = A.roi_align(qf, roi_boxes, strides, P).reshape(R * P * P, C)                # rows ordered [roi][pos], channels last
        s = sup8.mean(0, True).reshape(P * P, C)
        s_exp = s.unsqueeze(0).expand(R, P * P, C).reshape(R * P * P, C)
        s2_exp = A.linear(s, rh.conv2.weight.flatten(1), rh.conv2.bias).unsqueeze(0).expand(R, P * P, C // 2).reshape(R * P * P, C // 2)
    else:
        B = sup8.shape[0]
        img = roi_image.long()
        x = A.roi_align_batched(qf, roi_boxes, roi_image, strides, P).reshape(R * P * P, C)
        s = sup8.mean(1).reshape(B * P * P, C)                                          # each image's own support prototype
        s_exp = s.reshape(B, P * P, C)[img].reshape(R * P * P, C)
        s2_exp = A.linear(s, rh.conv2.weight.flatten(1), rh.conv2.bias).reshape(B, P * P, C // 2)[img].reshape(R * P * P, C // 2)
    a = A.linear(torch.cat((x, s_exp), 1), rh.conv3.weight.flatten(1), rh.conv3.bias) + \
        torch.cat((A.linear(x, rh.conv1.weight.flatten(1), rh.conv1.bias), s2_exp), 1)
    a = a.reshape(R, P * P, C).permute(0, 2, 1).reshape(R, C * P * P)                   # NCHW flatten order of fc1's weight
    fc1 = rh.box_head[0].fc1
    h = A.linear(a.contiguous(), fc1.weight, fc1.bias, True)
    pr = rh.box_predictor[0]
    scores = A.linear(h, pr.cls_score.weight, pr.cls_score.bias)
    deltas = A.linear(h, pr.bbox_pred.weight, pr.bbox_pred.bias)
    fg = torch.nonzero(roi_labels == 0).squeeze(1)
    tgt = get_deltas(roi_boxes[fg], roi_gt[fg], rh.bbox_reg_weights)
    if roi_image is None:
        loss_cls = F.cross_entropy(scores, roi_labels, reduction="mean")
        loss_box = (deltas[fg] - tgt).abs().sum() / max(roi_labels.numel(), 1)           # smooth_l1, beta = 0
    else:
        w = torch.tensor([1.0 / (max(n, 1) * len(rois_per_image)) for n in rois_per_image], device=scores.device)[img]
        loss_cls = (F.cross_entropy(scores, roi_labels, reduction="none") * w).sum()
        loss_box = ((deltas[fg] - tgt).abs().sum(1) * w[fg]).sum()
    return {"loss_cls_stage0": loss_cls, "loss_box_reg_stage0": loss_box}, dict(scores=scores, deltas=deltas, h=h)


def train_forward(model, batched_inputs, perm: Optional[Callable[[int], torch.Tensor]] = None, return_aux: bool = False,
                  roi_override: Optional[Dict[str, torch.Tensor]] = None):
    """model: CenterNet2Detector in training mode.  batched_inputs[i]: image [3,H,W] (uint8/float BGR), instances (gt_boxes),
    support_images [N,3,h,w], support_bboxes [N,4].  Returns the 5 losses (mean over the images of the call).
    `perm(n)` replaces torch.randperm in the fg/bg subsampling; `roi_override` = {boxes, labels, gt} replaces the sampled set
    altogether (parity tests pin the second stage on the oracle's sample so a 1-ulp heatmap difference cannot change the batch)."""
    from orehip import autograd as A
    dev = model.device
    pg, rh = model.proposal_generator, model.roi_heads
    mean, std = model.pixel_mean.view(-1), model.pixel_std.view(-1)
    div = model.backbone.size_divisibility
    if perm is None:
        perm = lambda n: torch.randperm(n, device=dev)          # noqa: E731  (subsample_labels, d2z:modeling/sampling.py:49-50)
    acc: Dict[str, List[torch.Tensor]] = {}
    aux = {}
    B = len(batched_inputs)
    N = model.support_way * model.support_shot
    assert model.support_way == 1
    imgs = [item["image"].to(dev).float() for item in batched_inputs]
    Hm, Wm = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
    # ImageList.from_tensors semantics (d2z:structures/image_list.py:69-121): normalise each image, zero-pad bottom/right to the batch
    # maximum rounded up to the size divisibility; the whole batch goes through the backbone at once (fsod_cen.py:156,165)
    xq = torch.cat([F.pad(_normalise_pad(i[None], mean, std, 1), (0, Wm - i.shape[-1], 0, Hm - i.shape[-2])) for i in imgs], 0)
    xq = F.pad(xq, (0, (Wm + div - 1) // div * div - Wm, 0, (Hm + div - 1) // div * div - Hm)).contiguous()
    sups = [item["support_images"].to(dev) for item in batched_inputs]
    for s_ in sups:
        assert s_.shape[0] == N, "support_images must hold SUPPORT_WAY * SUPPORT_SHOT crops"
    xs = _normalise_pad(torch.cat(sups, 0) if B > 1 else sups[0], mean, std, div)
    outs = graphed_dense_part(model, xq, xs) if getattr(model, "train_graph", False) else dense_part(model, xq, xs)
    gts, sbx = [], []
    for item in batched_inputs:
        inst = item["instances"]
        gts.append((inst.gt_boxes.tensor if hasattr(inst.gt_boxes, "tensor") else inst.gt_boxes).to(dev).float())
        sbx.append(torch.as_tensor(item["support_bboxes"], dtype=torch.float32, device=dev))
    if B == 1:
        gt_boxes, qf, sf_levels, heads = gts[0], [f[0] for f in outs[0:3]], list(outs[3:6]), list(outs[6:9])
        # ---- first stage: ground truth, losses, proposals (no gradient through the proposals)
        proposals, _scores, l_rpn, tg = proposal_losses_and_proposals(pg, heads, gt_boxes)
        sampled, roi_boxes, roi_labels, roi_gt = label_and_sample(rh, proposals, gt_boxes, perm)
        if roi_override is not None:
            roi_boxes = roi_override["boxes"].to(dev).float().contiguous()
            roi_labels, roi_gt = roi_override["labels"].to(dev), roi_override["gt"].to(dev).float()
        # ---- second stage: support rcnn_8 features (one box per support crop), DSA mix, fc1, predictor, losses
        sup8 = A.roi_align_batched(sf_levels, sbx[0], torch.arange(N, dtype=torch.int32, device=dev), pg.strides, rh.pooler_resolution)
        l_roi, a2 = roi_stage_losses(rh, qf, sup8, roi_boxes, roi_labels, roi_gt, pg.strides)
        losses = {**l_roi, **l_rpn}
        if return_aux:
            aux = dict(proposals=proposals, sampled=sampled, roi_boxes=roi_boxes, roi_labels=roi_labels, pos_inds=tg["pos_inds"],
                       pos_count=tg["pos_count"], features={k: f.permute(2, 0, 1)[None] for k, f in zip(LEVELS, qf)}, heads=heads, **a2)
        return (losses, aux) if return_aux else losses
    # ---- B > 1.  Phase 1 (no host sync): per image ground truth, first-stage losses, proposals
    hd_b = [h.split(1, 0) for h in outs[6:9]]       # one split per level: its backward is ONE cat, not B zero-filled full-size adds
    stage1 = [first_stage(pg, [h[b] for h in hd_b], gts[b]) for b in range(B)]
    for _, l_rpn, _ in stage1:
        for k, v in l_rpn.items():
            acc.setdefault(k, []).append(v)
    counts = torch.stack([o["counts"][1] for o, _, _ in stage1]).tolist()             # ONE sync for the B proposal counts
    # ---- phase 2: label + sample per image (host-shaped, only tiny kernels queued behind its syncs)
    rb, rl, rg, per_image = [], [], [], []
    for b in range(B):
        _, roi_boxes, roi_labels, roi_gt = label_and_sample(rh, stage1[b][0]["out_boxes"][:counts[b]], gts[b], perm)
        if roi_override is not None:
            roi_boxes = roi_override["boxes"].to(dev).float().contiguous()
            roi_labels, roi_gt = roi_override["labels"].to(dev), roi_override["gt"].to(dev).float()
        rb.append(roi_boxes); rl.append(roi_labels); rg.append(roi_gt); per_image.append(int(roi_boxes.shape[0]))
    roi_image = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(per_image)).to(dev)
    # ---- phase 3: the second stage of all B images in one pass (ROIAlign of every support crop's own box, then the ROI head)
    sup8 = A.roi_align_batched(list(outs[3:6]), torch.cat(sbx, 0), torch.arange(B * N, dtype=torch.int32, device=dev), pg.strides,
                               rh.pooler_resolution)
    l_roi, a2 = roi_stage_losses(rh, list(outs[0:3]), sup8.reshape(B, N, -1), torch.cat(rb, 0), torch.cat(rl, 0), torch.cat(rg, 0),
                                 pg.strides, roi_image, per_image)
    if return_aux:
        aux = dict(roi_boxes=rb, roi_labels=rl, roi_gt=rg, rois_per_image=per_image, **a2)
    losses = {**l_roi, **{k: torch.stack(v).mean() for k, v in acc.items()}}
    return (losses, aux) if return_aux else losses
