from torch import nn


class FsodFastRCNNOutputLayers(nn.Module):
    """Name kept for `from fewx.modeling import FsodFastRCNNOutputLayers`; belongs to the legacy FsodRCNN model."""

    def __init__(self, *a, **k):
        raise NotImplementedError("FsodFastRCNNOutputLayers belongs to the legacy FsodRCNN (R50-C4) model, outside the built path")
