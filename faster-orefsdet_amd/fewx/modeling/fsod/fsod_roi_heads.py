"""fewx-local ROI heads registry (ref:fewx/modeling/fsod/fsod_roi_heads.py:33,45-50).

CustomCascadeROIHeads is the second stage of Faster-OreFSDet ("next"): this round provides the module
tree with the reference's parameter names/shapes (so checkpoints load and the DP gradient bucket has the right layout);
its forward (ROIAlign + support-guided mixing + FC + box decode + NMS) is the next row to be built in HIP."""
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

    def forward(self, images, features, support_box_features, proposals, targets=None):
        raise NotImplementedError("CustomCascadeROIHeads.forward (SURVEY 8f row 1: ROIAlign + DSA + FC + NMS) is the next row "
                                  "to be built in HIP; use CenterNet2Detector.inference_proposals() for the built hot path")


@ROI_HEADS_REGISTRY.register()
class CustomROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        raise NotImplementedError("CustomROIHeads is not selected by any shipped config; outside the built path")


@ROI_HEADS_REGISTRY.register()
class FsodRes5ROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        raise NotImplementedError("FsodRes5ROIHeads belongs to the legacy FsodRCNN (R50-C4) model, outside the built path")
