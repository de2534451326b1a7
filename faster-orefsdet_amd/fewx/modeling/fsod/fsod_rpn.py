"""Proposal generators registered under the reference's names (ref:fewx/modeling/fsod/fsod_rpn.py:149, :491).

`CenterNet` -- anchor-free CenterNet2 proposal generator.  Eval path (this round): head -> on-device
sigmoid / threshold / per-level top-k / decode / NMS / post-NMS top-k in libore_hip.so (ore_detect_fwd), no host sync
until the caller reads the count.  `FsodRPN` is the legacy attention-RPN of the R50-C4 FewX model: name kept, not built."""
from typing import Dict, List

import torch
from torch import nn

from detectron2.layers import ShapeSpec, nhwc_view, _require_gpu
from detectron2.modeling import PROPOSAL_GENERATOR_REGISTRY
from detectron2.structures import Boxes, Instances

from .centernet_head import CenterNetHead


@PROPOSAL_GENERATOR_REGISTRY.register()
class CenterNet(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        c = cfg.MODEL.CENTERNET
        self.in_features = list(c.IN_FEATURES)
        self.strides = list(c.FPN_STRIDES)
        self.score_thresh = c.INFERENCE_TH
        self.pre_nms_topk_train, self.pre_nms_topk_test = c.PRE_NMS_TOPK_TRAIN, c.PRE_NMS_TOPK_TEST
        self.post_nms_topk_train, self.post_nms_topk_test = c.POST_NMS_TOPK_TRAIN, c.POST_NMS_TOPK_TEST
        self.nms_thresh_train, self.nms_thresh_test = c.NMS_TH_TRAIN, c.NMS_TH_TEST
        self.only_proposal, self.with_agn_hm, self.not_nms = c.ONLY_PROPOSAL, c.WITH_AGN_HM, c.NOT_NMS
        if not (self.only_proposal and self.with_agn_hm) or self.not_nms or c.CENTER_NMS or c.MORE_POS:
            raise NotImplementedError("CenterNet: only ONLY_PROPOSAL + WITH_AGN_HM with NMS is built (finetune_vovnet.yaml)")
        # training-side hyper-parameters (fewx/modeling/fsod/train_forward.py)
        self.hm_focal_alpha, self.hm_focal_beta, self.loss_gamma = c.HM_FOCAL_ALPHA, c.HM_FOCAL_BETA, c.LOSS_GAMMA
        self.reg_weight, self.not_norm_reg, self.pos_weight, self.neg_weight = c.REG_WEIGHT, c.NOT_NORM_REG, c.POS_WEIGHT, c.NEG_WEIGHT
        self.sigmoid_clamp, self.ignore_high_fp, self.min_radius = c.SIGMOID_CLAMP, c.IGNORE_HIGH_FP, c.MIN_RADIUS
        self.sizes_of_interest, self.no_reduce = c.SOI, c.NO_REDUCE
        self.hm_min_overlap = c.HM_MIN_OVERLAP
        self.delta = (1 - c.HM_MIN_OVERLAP) / (1 + c.HM_MIN_OVERLAP)
        shapes = [input_shape[f] for f in self.in_features]
        self.centernet_head = CenterNetHead(**CenterNetHead.from_config(cfg, shapes))

    def forward(self, images, features_dict, gt_instances=None):
        if self.training:
            # reference protocol (fsod_rpn.py:644-700): (proposals, losses) for ONE query image and its gt instances
            from .train_forward import head_train, proposal_losses_and_proposals
            assert gt_instances is not None and len(gt_instances) == 1 and len(images.image_sizes) == 1, "one query image per call"
            feats = [features_dict[f] for f in self.in_features]
            for f in feats:
                _require_gpu(f, "CenterNet")
            heads = head_train(self.centernet_head, [nhwc_view(f) for f in feats])
            gt = gt_instances[0].gt_boxes
            gt = (gt.tensor if hasattr(gt, "tensor") else gt).to(feats[0].device).float()
            boxes, scores, losses, _ = proposal_losses_and_proposals(self, heads, gt)
            return [make_proposals(images.image_sizes[0], boxes, scores)], losses
        feats = [features_dict[f] for f in self.in_features]
        for f in feats:
            _require_gpu(f, "CenterNet")
        assert feats[0].shape[0] == len(images.image_sizes) == 1, "inference is per image (ref fsod_cen.py:438-439)"
        heads = self.centernet_head.forward_nhwc([nhwc_view(f) for f in feats])
        return self.inference(images, heads), {}

    @torch.no_grad()
    def inference(self, images, heads: List[torch.Tensor]):
        import orehip
        o = orehip.detect([h[0] for h in heads], self.strides, self.score_thresh, self.pre_nms_topk_test, self.nms_thresh_test,
                          self.post_nms_topk_test)
        n = int(o["counts"][1].item())  # the one host sync of the detection tail
        return [make_proposals(images.image_sizes[0], o["out_boxes"][:n], o["out_scores"][:n])]


def make_proposals(image_size, boxes, scores):
    """ref fsod_rpn.py:1084-1088: proposal_boxes/objectness_logits added, scores and pred_classes kept."""
    p = Instances(image_size)
    p.proposal_boxes = Boxes(boxes)
    p.objectness_logits = scores
    p.scores = scores
    p.pred_classes = torch.zeros(len(scores), dtype=torch.int64, device=scores.device)
    return p


@PROPOSAL_GENERATOR_REGISTRY.register()
class FsodRPN(nn.Module):
    def __init__(self, cfg, input_shape):
        raise NotImplementedError("FsodRPN (legacy attention-RPN of the R50-C4 FewX model) is outside the built path; "
                                  "every shipped finetune config selects MODEL.PROPOSAL_GENERATOR.NAME=CenterNet")
