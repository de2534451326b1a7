from torch import nn

from detectron2.modeling import META_ARCH_REGISTRY


@META_ARCH_REGISTRY.register()
class FsodRCNN(nn.Module):
    """Legacy FewX attention-RPN meta-arch (ref:fewx/modeling/fsod/fsod_rcnn.py:36).  The registry name resolves (Base-FSOD-C4.yaml
    names it) but every finetune_*.yaml overrides it with CenterNet2Detector; its compute is outside the built path."""

    def __init__(self, cfg):
        raise NotImplementedError("FsodRCNN (R50-C4 attention-RPN) is outside the built path; use MODEL.META_ARCHITECTURE=CenterNet2Detector")
