from abc import ABCMeta, abstractmethod

import torch.nn as nn

from detectron2.layers import ShapeSpec
from detectron2.utils.registry import Registry

BACKBONE_REGISTRY = Registry("BACKBONE")


class Backbone(nn.Module, metaclass=ABCMeta):
    """d2z:modeling/backbone/backbone.py: forward -> dict[str, Tensor]; output_shape(); size_divisibility."""

    @abstractmethod
    def forward(self, x):
        ...

    @property
    def size_divisibility(self) -> int:
        return 0

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone


from .vovnet import VoVNet, build_vovnet_backbone, build_fcos_vovnet_fpn_backbone, build_vovnet_fpn_backbone  # noqa: E402,F401
from .fpn import FPN  # noqa: E402,F401
