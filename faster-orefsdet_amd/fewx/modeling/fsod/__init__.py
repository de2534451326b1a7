from .centernet_head import CenterNetHead, Scale
from .fsod_rpn import CenterNet, FsodRPN
from .fsod_roi_heads import (ROI_HEADS_REGISTRY, CustomCascadeROIHeads, CustomROIHeads, FsodRes5ROIHeads, build_roi_heads)
from .fsod_fast_rcnn import FsodFastRCNNOutputLayers
from .fsod_rcnn import FsodRCNN
from .fsod_cen import MLP, CenterNet2Detector, SM_Block
