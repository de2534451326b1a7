"""fewx-local ROI heads registry (ref:fewx/modeling/fsod/fsod_roi_heads.py:33,45-50).

CustomCascadeROIHeads is the second stage of Faster-OreFSDet ("next"): this round provides the module
tree with the reference's parameter names/shapes (so checkpoints load and the DP gradient bucket has the right layout);
its eval forward (ROIAlign + support-guided mixing + FC + box decode + NMS) runs on the HIP kernels of csrc/ore_roi.hip."""
import torch
from torch import nn

from detectron2.utils.registry import Registry

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


class _FC(nn.Sequential):
    pass


@ROI_HEADS_REGISTRY.register()
class CustomCascadeROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.in_features = list(cfg.MODEL.ROI_HEADS.IN_FEATURES)
        C = input_shape[self.in_features[0]].channels
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        res2 = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION2
        fc_dim = cfg.MODEL.ROI_BOX_HEAD.FC_DIM // 8           # d2z:modeling/roi_heads/box_head.py:70 (FC_DIM / 8)
        n_stage = len(cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS)
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk = cfg.TEST.DETECTIONS_PER_IMAGE
        self.pooler_resolution, self.pooler_resolution2 = res, res2
        self.batch_size_per_image = cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION
        self.iou_threshold = cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS[0]
        self.proposal_append_gt = cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT
        self.bbox_reg_weights = tuple(cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS[0])
        heads, preds = [], []
        for _ in range(n_stage):
            h = nn.Sequential()
            h.add_module("fc1", nn.Linear(C * res * res, fc_dim))
            heads.append(h)
            p = nn.Module()
            p.cls_score = nn.Linear(fc_dim, self.num_classes + 1)
            p.bbox_pred = nn.Linear(fc_dim, 4)
            nn.init.normal_(p.cls_score.weight, std=0.01)
            nn.init.normal_(p.bbox_pred.weight, std=0.001)
            nn.init.constant_(p.cls_score.bias, 0)
            nn.init.constant_(p.bbox_pred.bias, 0)
            preds.append(p)
        self.box_head = nn.ModuleList(heads)
        self.box_predictor = nn.ModuleList(preds)
        self.fc2 = nn.Linear(C * res2 * res2, fc_dim)      # dead branch in the reference; kept for checkpoints
        self.fc3 = nn.Linear(2 * fc_dim, fc_dim)
        self.conv1 = nn.Conv2d(C, C // 2, 1)
        self.conv2 = nn.Conv2d(C, C // 2, 1)
        self.conv3 = nn.Conv2d(2 * C, C, 1)

    def composed(self, support_8):
        """(W', b') of the linear chain DSA-mix -> flatten -> fc1, cached per (parameters, support features)."""
        import orehip
        key = tuple(p._version for p in self.parameters()) + (support_8.data_ptr(), support_8._version, str(support_8.device))
        if getattr(self, "_comp", None) is None or self._comp[0] != key:
            Wp, bp = orehip.compose_roi_head({"roi_heads." + k: v for k, v in self.state_dict().items()}, support_8)
            dev = next(self.parameters()).device
            self._comp = (key, Wp.to(dev), bp.to(dev))
        return self._comp[1], self._comp[2]

    def forward(self, images, features, support_box_features, proposals, targets=None, perm=None):
        if self.training:
            return self._forward_train(features, support_box_features, proposals, targets, perm)
        with torch.no_grad():
            return self._forward_eval(images, features, support_box_features, proposals)

    def _forward_train(self, features, support_box_features, proposals, targets, perm=None):
        """Reference protocol (ref fsod_roi_heads.py:247-262, :404-520): label_and_sample_proposals, then the single cascade stage and
        its losses.  support_box_features[0] = rcnn_8 features of the support crops, [N,C,8,8] as the reference pools them."""
        from detectron2.layers import nhwc_view
        from .train_forward import label_and_sample, roi_stage_losses
        assert targets is not None and len(proposals) == 1 and len(targets) == 1 and len(self.box_head) == 1
        props = proposals[0]
        dev = props.proposal_boxes.tensor.device
        gt = targets[0].gt_boxes
        gt = (gt.tensor if hasattr(gt, "tensor") else gt).to(dev).float()
        if perm is None:
            perm = lambda n: torch.randperm(n, device=dev)      # noqa: E731
        _, roi_boxes, roi_labels, roi_gt = label_and_sample(self, props.proposal_boxes.tensor.detach(), gt, perm)
        qf = [nhwc_view(features[f])[0] for f in self.in_features]
        s8 = support_box_features[0]
        sup8 = s8.permute(0, 2, 3, 1).reshape(s8.shape[0], -1)            # [N,C,8,8] -> [N, pos*C + c]
        losses, _ = roi_stage_losses(self, qf, sup8, roi_boxes, roi_labels, roi_gt, [8, 16, 32][: len(qf)])
        return proposals, losses

    def _forward_eval(self, images, features, support_box_features, proposals, targets=None):
        """Eval second stage for one image on the HIP kernels: ROIAlign 8x8 over p3..p5 -> pre-composed [8192->128] GEMM + ReLU
        -> cls/box -> softmax, apply_deltas, clip, score filter, NMS, keep[:topk].  (ref fsod_roi_heads.py:374-457; the second
        `_forward_box` definition shadows the first, so MULT_PROPOSAL_SCORE is NOT applied -- SURVEY 8f row 1.)"""
        import orehip
        from detectron2.layers import nhwc_view
        from detectron2.structures import Boxes, Instances
        assert len(proposals) == 1 and len(self.box_head) == 1, "one image, one cascade stage (finetune_vovnet.yaml)"
        props = proposals[0]
        boxes = props.proposal_boxes.tensor
        feats = [nhwc_view(features[f]) for f in self.in_features]
        strides = [8, 16, 32][: len(feats)]
        Wp, bp = self.composed(support_box_features[0])
        n = boxes.shape[0]
        x = orehip.roi_align(feats, boxes, strides, self.pooler_resolution, cap=max(n, 1))
        h = orehip.conv2d(x.view(1, 1, x.shape[0], x.shape[1]), Wp, Wp.shape[0], 1, shift=bp, relu_cout=Wp.shape[0])
        h = h.view(x.shape[0], Wp.shape[0])
        pr = self.box_predictor[0]
        det = orehip.roi_predict(h, pr.cls_score.weight.detach().contiguous(), pr.cls_score.bias.detach().contiguous(),
                                 pr.bbox_pred.weight.detach().contiguous(), pr.bbox_pred.bias.detach().contiguous(), boxes,
                                 self.bbox_reg_weights, props.image_size, self.test_score_thresh, self.test_nms_thresh, self.test_topk)
        k = int(det["count"].item())
        res = Instances(props.image_size)
        res.pred_boxes = Boxes(det["boxes"][:k])
        res.scores = det["scores"][:k]
        res.pred_classes = torch.zeros(k, dtype=torch.int64, device=boxes.device)
        return [res], {}


@ROI_HEADS_REGISTRY.register()
class CustomROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        raise NotImplementedError("CustomROIHeads is not selected by any shipped config; outside the built path")


@ROI_HEADS_REGISTRY.register()
class FsodRes5ROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        raise NotImplementedError("FsodRes5ROIHeads belongs to the legacy FsodRCNN (R50-C4) model, outside the built path")
