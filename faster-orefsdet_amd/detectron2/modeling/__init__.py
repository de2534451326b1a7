"""Registries + builders the reference resolves by name from the yaml."""
import torch

from detectron2.layers import ShapeSpec
from detectron2.utils.registry import Registry

from .backbone import BACKBONE_REGISTRY, Backbone, build_backbone  # noqa: F401

META_ARCH_REGISTRY = Registry("META_ARCH")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)


def build_model(cfg):
    """d2z:modeling/meta_arch/build.py: META_ARCH_REGISTRY.get(name)(cfg).to(cfg.MODEL.DEVICE)."""
    import fewx.modeling  # noqa: F401  (registers the few-shot architectures)
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
