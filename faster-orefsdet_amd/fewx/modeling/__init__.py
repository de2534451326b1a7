"""fewx.modeling: importing this package registers the few-shot architectures by the reference's names
(ref:fewx/modeling/__init__.py:2): meta-archs CenterNet2Detector / FsodRCNN, proposal generators CenterNet / FsodRPN,
ROI heads CustomCascadeROIHeads / CustomROIHeads / FsodRes5ROIHeads."""
from .fsod import (CenterNet, CenterNet2Detector, CenterNetHead, FsodFastRCNNOutputLayers, FsodRCNN, FsodRes5ROIHeads,
                   FsodRPN, SM_Block)

__all__ = ["CenterNet2Detector", "CenterNet", "CenterNetHead", "SM_Block", "FsodRCNN", "FsodRes5ROIHeads",
           "FsodFastRCNNOutputLayers", "FsodRPN"]
