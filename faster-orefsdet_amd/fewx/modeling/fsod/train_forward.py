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
add and its gradient, top-k / decode / NMS, the fg / bg subsample of the proposals (one launch) and the two ROI losses with their
gradients (one launch) are HIP kernels.  Still torch tensor ops on the device (small, listed in DESIGN.md): the [B,C]-sized gate
algebra of eSE, SM_Block pointwise math, the head's scale / ReLU / concat.
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


def _mean_std(model):
    """pixel mean / std as host floats, read from the device once per model."""
    tag = (model.pixel_mean._version, model.pixel_mean.data_ptr(), model.pixel_std._version, model.pixel_std.data_ptr())
    ms = model.__dict__.get("_ore_mean_std")
    if ms is None or ms[0] != tag:                     # (a checkpoint load rewrites the buffers in place: the version counter moves)
        ms = model.__dict__["_ore_mean_std"] = (tag, tuple(model.pixel_mean.view(-1).tolist()), tuple(model.pixel_std.view(-1).tolist()))
    return ms[1], ms[2]


def dense_part(model, xq: torch.Tensor, xs: torch.Tensor, raw: bool = False):
    """The shape-static part of a training iteration for B query images at once: query and support pyramids (two batched backbone
    passes, fsod_cen.py:165,179), per-image support prototypes and correlation (:197-275), conv3 and the head batched over B.
    xq [B,3,H,W], xs [B*N,3,h,w] normalised + padded (raw = False) or raw uint8 / fp32 images of one size each (raw = True), image b
    owning support rows b*N..(b+1)*N.
    Returns (q3, q4, q5 [B,..], s3, s4, s5 [B*N,..] as NHWC, head3, head4, head5 [B,H,W,16])."""
    from orehip import autograd as A
    B = xq.shape[0]
    N = xs.shape[0] // B
    if raw:                                             # raw image batches: normalisation + /32 zero padding fused into stem_1
        mean, std = _mean_std(model)
        div = model.backbone.size_divisibility
        pad = lambda n: (n + div - 1) // div * div     # noqa: E731
        feats = model.backbone(xq, raw_norm=(mean, std, pad(xq.shape[-2]), pad(xq.shape[-1])))
        sfeats = model.backbone(xs, raw_norm=(mean, std, pad(xs.shape[-2]), pad(xs.shape[-1])))
    else:
        feats = model.backbone(xq)
        sfeats = model.backbone(xs)
    pos = []
    for i, k in enumerate(LEVELS):
        size = (32, 16, 8)[i]
        # support prototypes: avg-pool to 32/16/8, SM_Block, mean over the image's shots -- all on the NHWC maps (HIP pooling kernels,
        # no NCHW round trip).  The reference pools `proto = v.permute(0, 3, 2, 1)` = [B,C,W,H] to (1,1) / (1,3) / (3,1)
        # (fsod_cen.py:226-231): in NHWC terms the (1,3) kernel is 3 bins over H and one over W, the (3,1) kernel the other way round.
        sf = nhwc_view(sfeats[k])
        if tuple(sf.shape[1:3]) != (size, size):
            sf = A.adaptive_avg_pool(sf, size, size)
        pn = A.group_mean(getattr(model, f"vip_p{3 + i}")(sf), B)               # [B,S,S,C]
        k11 = A.adaptive_avg_pool(pn, 1, 1)[:, 0, 0, :]                          # support kernels, [B,C]
        k13 = A.adaptive_avg_pool(pn, 3, 1)[:, :, 0, :].permute(0, 2, 1)         # [B,C,3]
        k31 = A.adaptive_avg_pool(pn, 1, 3)[:, 0, :, :].permute(0, 2, 1)
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
        A.prepack(list(self.model.parameters()))               # one captured launch repacks every weight the cache knows from the eager steps
        return dense_part(self.model, xq, xs)


def graphed_dense_part(model, xq, xs):
    """Cached hipGraph capture of dense_part for these input shapes (opt-in: model.train_graph = True).  Falls back to the eager
    path (and remembers why) if the capture fails."""
    key = (tuple(xq.shape), tuple(xs.shape))
    cache = model.__dict__.setdefault("_ore_train_graphs", {})
    if key not in cache:
        try:
            mod = _DensePart(model)
            # The capture runs on side streams and keeps its autograd graph: AccumulateGrad nodes of the parameters that exist
            # already (an eager step ran before) meet the capture's stream now, the ones the capture creates meet the training
            # stream in every later step.  Either way the engine orders the two streams with events -- gradient accumulation, the
            # post-accumulate hooks that issue the RCCL slices and the end-of-backward join run correctly
            # (tests/test_hip_train.py::test_rccl_one_rank_rehearsal pins a wrapped step to the plain one) -- so the mismatch is
            # intentional and torch's once-per-process warning about it is switched off.  (Creating the nodes up front on the
            # training stream instead would make the CAPTURE depend on the legacy default stream, which a capture may not.)
            _w = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if _w is not None:
                _w(False)
            # (the captured backward hands its gradients to autograd: its warm-up passes run torch.autograd.grad, whose side effects --
            # weight gradients accumulated in place -- would otherwise land in the bucket)
            from orehip import autograd as _A
            _A.DIRECT_GRAD_OFF = True
            try:
                cache[key] = torch.cuda.make_graphed_callables(mod, (xq.clone(), xs.clone()), num_warmup_iters=3, allow_unused_input=True)
            finally:
                _A.DIRECT_GRAD_OFF = False
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


def first_stage_batch(pg, heads: List[torch.Tensor], gts: Optional[List[torch.Tensor]], norm_avg: Optional[torch.Tensor] = None,
                      padded=None):
    """CenterNet.forward, training branch, after the head, for the B images of a call (ref:fewx/modeling/fsod/fsod_rpn.py:658-700) with
    NO host sync: ground truth of all images in one launch, the three losses over all rows in one (normalisers the reference's
    max(reduce_sum / num_gpus, 1) over the rank's whole batch -- orehip.autograd.CenterNetLossFn), the *_TRAIN proposals per image.
    heads[l] [B,H,W,16].  Returns (per-image detect outputs, losses dict, targets dict, this call's [reg rows, positives])."""
    import orehip
    from orehip import autograd as A
    dev = heads[0].device
    B = heads[0].shape[0]
    shapes = [tuple(h.shape[1:3]) for h in heads]
    rows = torch.cat([h.reshape(-1, h.shape[-1]) for h in heads], 0)       # level-major, the images of a level in order: the targets' rows
    # `padded` = (boxes [B,G,4], counts [B] int32) on the device: the fixed-shape form a captured step uses
    tg = orehip.centernet_targets(gts, shapes, pg.strides, pg.sizes_of_interest, pg.hm_min_overlap, pg.min_radius, device=dev, padded=padded)
    hp = dict(gamma=pg.loss_gamma, beta=pg.hm_focal_beta, sigmoid_clamp=pg.sigmoid_clamp, ignore_high_fp=pg.ignore_high_fp,
              alpha=pg.hm_focal_alpha, pos_weight=pg.pos_weight, neg_weight=pg.neg_weight, reg_weight=pg.reg_weight, images=B,
              norm_avg=norm_avg)
    l3, counts = A.centernet_losses(rows, tg["reg_targets"], tg["hm_targets"], tg["pos_inds"], tg["pos_count"], hp, with_counts=True)
    with torch.no_grad():
        outs = orehip.detect_batch([[h[b].detach() for h in heads] for b in range(B)], pg.strides, pg.score_thresh,
                                   pg.pre_nms_topk_train, pg.nms_thresh_train, pg.post_nms_topk_train)
    losses = {"loss_centernet_loc": l3[0], "loss_centernet_agn_pos": l3[1], "loss_centernet_agn_neg": l3[2]}
    return outs, losses, tg, counts


@torch.no_grad()
def sample_rois_device(rh, prop: torch.Tensor, prop_n: torch.Tensor, gtp: torch.Tensor, gt_n: torch.Tensor):
    """label_and_sample_proposals for B images at once, on the device, without a host sync: ONE launch (ore_sample_rois_fwd) behind the
    draw of the iid uniform keys.  Same contract and, for the same keys, the same sample as `sample_rois_torch` below (the element-wise
    form of rounds 3-5, kept as the statement of the semantics and for the test that pins the kernel to it)."""
    import orehip
    B, cap, _ = prop.shape
    G = gtp.shape[1]
    N = cap + (G if rh.proposal_append_gt else 0)
    R = int(rh.batch_size_per_image)
    if N > 12800 or G > 256:
        return sample_rois_torch(rh, prop, prop_n, gtp, gt_n)
    u = torch.rand(B, N, device=prop.device)
    return orehip.sample_rois(prop.contiguous(), prop_n, gtp.contiguous(), gt_n, u, R, int(R * rh.positive_fraction), float(rh.iou_threshold),
                              bool(rh.proposal_append_gt))


@torch.no_grad()
def sample_rois_torch(rh, prop: torch.Tensor, prop_n: torch.Tensor, gtp: torch.Tensor, gt_n: torch.Tensor):
    """label_and_sample_proposals (d2z:modeling/roi_heads/roi_heads.py:181-295, sampling.py:10-53) for B images at once, entirely on
    the device and without a host sync.  prop [B,cap,4] with prop_n [B] valid rows, gtp [B,G,4] with gt_n [B] valid rows.
    Candidates = proposals then ground truth (proposal_append_gt); IoU matcher (>= IOUS[0] -> foreground = class 0, else background
    = 1); BATCH_SIZE_PER_IMAGE samples with at most POSITIVE_FRACTION foreground, drawn uniformly: the n smallest of iid uniform
    keys among the foreground (background) candidates = `randperm(n)[:k]` of the reference in distribution.
    Returns boxes [B,R,4], labels [B,R], matched gt [B,R,4], valid [B,R] (rows beyond an image's sample count are padding)."""
    B, cap, _ = prop.shape
    G = gtp.shape[1]
    dev = prop.device
    R = int(rh.batch_size_per_image)
    P = int(R * rh.positive_fraction)
    pv = torch.arange(cap, device=dev)[None, :] < prop_n[:, None]
    gv = torch.arange(G, device=dev)[None, :] < gt_n[:, None]
    if rh.proposal_append_gt:
        cand, cv = torch.cat([prop, gtp], 1), torch.cat([pv, gv], 1)
    else:
        cand, cv = prop, pv
    N = cand.shape[1]
    area_g = (gtp[:, :, 2] - gtp[:, :, 0]) * (gtp[:, :, 3] - gtp[:, :, 1])                     # pairwise_iou, batched
    area_c = (cand[:, :, 2] - cand[:, :, 0]) * (cand[:, :, 3] - cand[:, :, 1])
    wh = (torch.min(gtp[:, :, None, 2:], cand[:, None, :, 2:]) - torch.max(gtp[:, :, None, :2], cand[:, None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    iou = torch.where(inter > 0, inter / (area_g[:, :, None] + area_c[:, None, :] - inter), torch.zeros((), device=dev))
    iou = torch.where(gv[:, :, None], iou, torch.full((), -1.0, device=dev))
    vals, midx = iou.max(1)                                                                    # [B,N]; no gt at all -> -1 -> background
    labels = torch.where(vals >= rh.iou_threshold, 0, 1)
    labels = torch.where(cv, labels, 2)
    u = torch.rand(B, N, device=dev)
    two = torch.full((), 2.0, device=dev)
    pos_pick = torch.topk(torch.where(labels == 0, u, two), min(P, N), dim=1, largest=False).indices
    neg_pick = torch.topk(torch.where(labels == 1, u, two), min(R, N), dim=1, largest=False).indices
    n_pos = (labels == 0).sum(1).clamp(max=P)
    n_neg = torch.minimum((labels == 1).sum(1), R - n_pos)
    j = torch.arange(R, device=dev)[None, :]
    pj = j.clamp(max=pos_pick.shape[1] - 1).expand(B, R)
    nj = (j - n_pos[:, None]).clamp(min=0, max=neg_pick.shape[1] - 1)
    pick = torch.where(j < n_pos[:, None], pos_pick.gather(1, pj), neg_pick.gather(1, nj))
    valid = j < (n_pos + n_neg)[:, None]
    boxes = cand.gather(1, pick[:, :, None].expand(B, R, 4))
    lab = labels.gather(1, pick)
    gt = gtp.gather(1, midx.gather(1, pick)[:, :, None].expand(B, R, 4))
    boxes = torch.where(valid[:, :, None], boxes, _pad_box(dev))
    lab = torch.where(valid, lab, 1)
    return boxes.contiguous(), lab, gt, valid


_ROI_IMAGE = {}
_PAD_BOX = {}


def _pad_box(dev) -> torch.Tensor:
    """The box of a padding row, uploaded once per device (a captured step may not copy from pageable host memory)."""
    k = str(dev)
    if k not in _PAD_BOX:
        _PAD_BOX[k] = torch.tensor([0.0, 0.0, 8.0, 8.0]).to(dev)
    return _PAD_BOX[k]



def _roi_image(B: int, R: int, dev) -> torch.Tensor:
    key = (B, R, str(dev))
    if key not in _ROI_IMAGE:
        _ROI_IMAGE[key] = torch.arange(B, dtype=torch.int32).repeat_interleave(R).to(dev)
    return _ROI_IMAGE[key]


def roi_stage_losses_padded(rh, qf: List[torch.Tensor], sup8: torch.Tensor, boxes, labels, gt, valid, strides):
    """_run_stage + CustomFastRCNNOutputLayers.losses for the single cascade stage (ref:fewx/modeling/fsod/fsod_roi_heads.py:459-520,
    custom_fast_rcnn.py:52-81) for B images in ONE pass, fixed shapes, no host sync: qf[l] [B,H,W,C] query pyramid (NHWC), sup8
    [B,N,P*P*C] pooled support features, boxes / gt [B,R,4], labels [B,R] (0 fg / 1 bg), valid [B,R].  Every image's losses keep their
    own 1/R_b normaliser (R_b = its valid rows) and the result is the mean over the images (= the gradient-averaged data-parallel
    result of B single-image ranks)."""
    from orehip import autograd as A
    B, R = labels.shape
    C = qf[0].shape[-1]
    P = rh.pooler_resolution
    dev = boxes.device
    rimg = _roi_image(B, R, dev)
    img = rimg.long()
    RT = B * R
    bx, lb, gtf, vf = boxes.reshape(RT, 4), labels.reshape(RT), gt.reshape(RT, 4), valid.reshape(RT)
    x = A.roi_align_batched(qf, bx, rimg, strides, P).reshape(RT * P * P, C)            # rows ordered [roi][pos], channels last
    s = sup8.mean(1).reshape(B * P * P, C)                                              # each image's own support prototype
    # the ROIs of image b are rows b*R .. (b+1)*R: broadcasting its prototype is an expand (backward = one sum over R), not a gather
    s_exp = s.reshape(B, 1, P * P, C).expand(B, R, P * P, C).reshape(RT * P * P, C)
    s2_exp = A.linear(s, rh.conv2.weight.flatten(1), rh.conv2.bias).reshape(B, 1, P * P, C // 2).expand(B, R, P * P, C // 2) \
        .reshape(RT * P * P, C // 2)
    a = A.linear(torch.cat((x, s_exp), 1), rh.conv3.weight.flatten(1), rh.conv3.bias) + \
        torch.cat((A.linear(x, rh.conv1.weight.flatten(1), rh.conv1.bias), s2_exp), 1)
    # fc1 reads the NCHW flatten [c][pos] of the reference; the rows here are [pos][c]: permute the 4 MB weight, not the activations
    fc1 = rh.box_head[0].fc1
    w1 = fc1.weight.reshape(fc1.weight.shape[0], C, P * P).permute(0, 2, 1).reshape(fc1.weight.shape[0], P * P * C)
    h = A.linear(a.reshape(RT, P * P * C), w1, fc1.bias, True)
    pr = rh.box_predictor[0]
    scores = A.linear(h, pr.cls_score.weight, pr.cls_score.bias)
    deltas = A.linear(h, pr.bbox_pred.weight, pr.bbox_pred.bias)
    # both losses and their gradients from ONE launch (ore_roi_losses_fwd): w_i = valid_i / (n_b * B), n_b = an image's valid rows;
    # cross-entropy over the two classes; |deltas - get_deltas(box, gt)| on the foreground rows (smooth_l1 with beta = 0; background and
    # padding rows have no target) -- what ~50 element-wise launches and ~40 more in their backward computed
    loss_cls, loss_box = A.roi_losses(scores, deltas, bx, gtf, lb, vf, B, R, rh.bbox_reg_weights)
    return {"loss_cls_stage0": loss_cls, "loss_box_reg_stage0": loss_box}, dict(scores=scores, deltas=deltas, h=h)


def _pad_stack(ts: List[torch.Tensor], width: int, dev):
    """[n_i, 4] tensors -> ([B, max(n,1), 4] zero padded on `dev`, counts [B] int64 on `dev`) without a blocking copy."""
    import orehip
    n = [int(t.shape[0]) for t in ts]
    m = max(1, max(n))
    if any(t.is_cuda for t in ts):
        out = torch.zeros(len(ts), m, width, device=dev)
        for i, t in enumerate(ts):
            if n[i]:
                out[i, :n[i]] = orehip.to_device(t.float(), dev)
    else:
        out = torch.zeros(len(ts), m, width)
        for i, t in enumerate(ts):
            out[i, :n[i]] = t.float()
        out = orehip.to_device(out, dev)
    return out, orehip.to_device(torch.tensor(n, dtype=torch.int64), dev)


def train_forward(model, batched_inputs, perm: Optional[Callable[[int], torch.Tensor]] = None, return_aux: bool = False,
                  roi_override: Optional[Dict[str, torch.Tensor]] = None, cn_norm_avg: Optional[torch.Tensor] = None,
                  fused_preprocess: bool = True):
    """model: CenterNet2Detector in training mode.  batched_inputs[i]: image [3,H,W] (uint8/float BGR), instances (gt_boxes),
    support_images [N,3,h,w], support_bboxes [N,4].  Returns the 5 losses of the call: B images on one rank give what B data-parallel
    single-image ranks of the reference give after gradient averaging (CenterNet losses: sums over all images / the all-image
    normalisers; second-stage losses: mean over the images).
    No host sync between the first kernel and the last (the fg/bg subsample runs on the device, sample_rois_device), unless the caller
    asks for it: `perm(n)` replaces torch.randperm in the subsampling (host-shaped legacy sampling, tests); `roi_override` = {boxes,
    labels, gt} replaces the sampled set of every image (parity tests pin the second stage on the reference's sample so a 1-ulp heatmap
    difference cannot change the batch); `cn_norm_avg` = the two averaged CenterNet normalisers of a larger virtual batch;
    `fused_preprocess=False` forces the generic normalise-then-pad input path (otherwise used for mixed sizes and graph capture)."""
    st = stage_inputs(model, batched_inputs)
    return train_core(model, st, perm=perm, return_aux=return_aux, roi_override=roi_override, cn_norm_avg=cn_norm_avg,
                      fused_preprocess=fused_preprocess)


def stage_inputs(model, batched_inputs, gt_capacity: Optional[int] = None) -> Dict:
    """Every host -> device upload of a call (pinned, asynchronous), so nothing in the core waits for the stream to drain:
    imgs / sups (lists, as given), gts (list of [n_i,4]), gtp [B,G,4] + gt_n [B] int64 (zero padded to the batch maximum, or to
    `gt_capacity` rows: the fixed-shape form of a captured step), sbx [B*N,4]."""
    import orehip
    dev = model.device
    N = model.support_way * model.support_shot
    assert model.support_way == 1
    gts, sbx = [], []
    for item in batched_inputs:
        inst = item["instances"]
        gts.append(orehip.to_device((inst.gt_boxes.tensor if hasattr(inst.gt_boxes, "tensor") else inst.gt_boxes).float(), dev))
        sbx.append(torch.as_tensor(item["support_bboxes"], dtype=torch.float32))
    sbx = orehip.to_device(torch.cat(sbx, 0), dev)
    gtp, gt_n = _pad_stack(gts, 4, dev)
    if gt_capacity is not None:
        assert gtp.shape[1] <= gt_capacity
        if gtp.shape[1] < gt_capacity:
            gtp = F.pad(gtp, (0, 0, 0, gt_capacity - gtp.shape[1]))
    imgs = [orehip.to_device(item["image"], dev) for item in batched_inputs]
    sups = [orehip.to_device(item["support_images"], dev) for item in batched_inputs]
    for s_ in sups:
        assert s_.shape[0] == N, "support_images must hold SUPPORT_WAY * SUPPORT_SHOT crops"
    return dict(imgs=imgs, sups=sups, gts=gts, gtp=gtp.contiguous(), gt_n=gt_n, sbx=sbx)


def train_core(model, st: Dict, perm=None, return_aux: bool = False, roi_override=None, cn_norm_avg=None, fused_preprocess: bool = True,
               static: bool = False):
    """train_forward behind its uploads.  `static`: the inputs are the fixed buffers of a captured step (st["xq"] / st["xs"] stacked raw
    images, gtp / gt_n / sbx; no per-image lists) -- nothing in here may then touch host memory or depend on a host-known count."""
    import orehip
    from orehip import autograd as A
    dev = model.device
    pg, rh = model.proposal_generator, model.roi_heads
    mean, std = model.pixel_mean.view(-1), model.pixel_std.view(-1)
    div = model.backbone.size_divisibility
    aux = {}
    N = model.support_way * model.support_shot
    gtp, gt_n, sbx = st["gtp"], st["gt_n"], st["sbx"]
    gts = st.get("gts")
    graph = getattr(model, "train_graph", False) and not static
    if static:
        B = int(st["xq"].shape[0])
        # inside a captured step: the packed copies are rebuilt by launches that belong to the graph (as in _DensePart)
        A.weights_changed()
        A.prepack(list(model.parameters()))
        outs = dense_part(model, st["xq"], st["xs"], raw=True)
    else:
        imgs, sups = st["imgs"], st["sups"]
        B = len(imgs)
        if not graph:
            # every packed weight the optimizer step has made stale -- both layouts of every trainable conv / linear -- in ONE launch
            # (orehip.autograd.prepack) instead of ~86 small ones scattered over the step; a captured dense part does the same inside its graph
            A.prepack(list(model.parameters()))
        same = all(i.shape == imgs[0].shape and i.dtype == imgs[0].dtype for i in imgs) and all(s_.shape == sups[0].shape for s_ in sups)
        if same and not graph and fused_preprocess:
            # one size per batch (the usual case): hand the RAW images to stem_1, which normalises and pads on the fly
            xq = torch.stack(imgs) if B > 1 else imgs[0][None]
            xs = torch.cat(sups, 0) if B > 1 else sups[0]
            outs = dense_part(model, xq.contiguous(), xs.contiguous(), raw=True)
        else:
            imgs = [i.float() for i in imgs]
            Hm, Wm = max(i.shape[-2] for i in imgs), max(i.shape[-1] for i in imgs)
            # ImageList.from_tensors semantics (d2z:structures/image_list.py:69-121): normalise each image, zero-pad bottom/right to the
            # batch maximum rounded up to the size divisibility; the whole batch goes through the backbone at once (fsod_cen.py:156,165)
            xq = torch.cat([F.pad(_normalise_pad(i[None], mean, std, 1), (0, Wm - i.shape[-1], 0, Hm - i.shape[-2])) for i in imgs], 0)
            xq = F.pad(xq, (0, (Wm + div - 1) // div * div - Wm, 0, (Hm + div - 1) // div * div - Hm)).contiguous()
            xs = _normalise_pad(torch.cat(sups, 0) if B > 1 else sups[0], mean, std, div)
            outs = graphed_dense_part(model, xq, xs) if graph else dense_part(model, xq, xs)
    qf, sf_levels, heads = list(outs[0:3]), list(outs[3:6]), list(outs[6:9])
    # ---- first stage: ground truth, losses, proposals (no gradient through the proposals)
    dets, l_rpn, tg, cn_counts = first_stage_batch(pg, heads, gts, cn_norm_avg, padded=(gtp, gt_n.to(torch.int32)) if static else None)
    # ---- fg/bg sample per image
    counts_host = None
    if roi_override is not None:
        rb = roi_override["boxes"].to(dev).float()
        boxes = rb[None].expand(B, *rb.shape).contiguous()
        labels = roi_override["labels"].to(dev)[None].expand(B, -1).contiguous()
        roi_gt = roi_override["gt"].to(dev).float()[None].expand(B, *rb.shape).contiguous()
        valid = torch.ones(B, rb.shape[0], dtype=torch.bool, device=dev)
    elif perm is not None:                                   # legacy host-shaped sampling with an injected permutation (tests)
        counts_host = torch.stack([o["counts"][1] for o in dets]).tolist()
        per = [label_and_sample(rh, dets[b]["out_boxes"][:counts_host[b]], gts[b], perm)[1:] for b in range(B)]
        R = max(1, max(int(p[0].shape[0]) for p in per))
        boxes = torch.tensor([0.0, 0.0, 8.0, 8.0], device=dev).repeat(B, R, 1)
        labels = torch.ones(B, R, dtype=torch.int64, device=dev)
        roi_gt = torch.zeros(B, R, 4, device=dev)
        valid = torch.zeros(B, R, dtype=torch.bool, device=dev)
        for b, (bx, lb, rg) in enumerate(per):
            n = int(bx.shape[0])
            boxes[b, :n], labels[b, :n], roi_gt[b, :n], valid[b, :n] = bx, lb, rg, True
    else:
        prop = torch.stack([o["out_boxes"] for o in dets])
        prop_n = torch.stack([o["counts"][1] for o in dets]).long()
        boxes, labels, roi_gt, valid = sample_rois_device(rh, prop, prop_n, gtp, gt_n)
    # ---- second stage of all B images in one pass: support rcnn_8 features (one box per support crop), DSA mix, fc1, predictor, losses
    sup8 = A.roi_align_batched(sf_levels, sbx, _roi_image(B * N, 1, dev), pg.strides, rh.pooler_resolution)
    l_roi, a2 = roi_stage_losses_padded(rh, qf, sup8.reshape(B, N, -1), boxes, labels, roi_gt, valid, pg.strides)
    losses = {**l_roi, **l_rpn}
    if return_aux:                                           # (host syncs from here on: only for callers that ask)
        nv = valid.sum(1).tolist()
        if counts_host is None:
            counts_host = torch.stack([o["counts"][1] for o in dets]).tolist()
        aux = dict(roi_boxes=[boxes[b, :nv[b]] for b in range(B)], roi_labels=[labels[b, :nv[b]] for b in range(B)],
                   roi_gt=[roi_gt[b, :nv[b]] for b in range(B)], rois_per_image=nv, cn_counts=cn_counts, heads=heads,
                   pos_inds=tg["pos_inds"], pos_count=tg["pos_count"], valid=valid, **a2)
        if B == 1:
            aux.update(proposals=dets[0]["out_boxes"][:counts_host[0]], proposal_scores=dets[0]["out_scores"][:counts_host[0]],
                       detect=dets[0], roi_boxes=aux["roi_boxes"][0], roi_labels=aux["roi_labels"][0],
                       roi_gt=aux["roi_gt"][0], features={k: f[0].permute(2, 0, 1)[None] for k, f in zip(LEVELS, qf)})
    return (losses, aux) if return_aux else losses
