"""FPN (d2z:modeling/backbone/fpn.py:17-154) on the HIP conv kernel: lateral 1x1 + bias with the nearest-2x
top-down sum fused into its epilogue, then the 3x3 output conv.  Module names `fpn_lateral{3,4,5}` / `fpn_output{3,4,5}`."""
import math

import torch
import torch.nn as nn

from detectron2.layers import Conv2d, nhwc_view, _require_gpu
from . import Backbone


def c2_xavier_fill(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert isinstance(bottom_up, Backbone) and in_features
        assert norm == "" and top_block is None and fuse_type == "sum", "only the configuration on the path is built"
        shapes = bottom_up.output_shape()
        strides = [shapes[f].stride for f in in_features]
        chans = [shapes[f].channels for f in in_features]
        for i in range(1, len(strides)):
            assert strides[i] == 2 * strides[i - 1], strides
        lat, out = [], []
        for s, c in zip(strides, chans):
            l = Conv2d(c, out_channels, kernel_size=1, bias=True)
            o = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
            c2_xavier_fill(l)
            c2_xavier_fill(o)
            stage = int(math.log2(s))
            self.add_module(f"fpn_lateral{stage}", l)
            self.add_module(f"fpn_output{stage}", o)
            lat.append(l)
            out.append(o)
        self.lateral_convs, self.output_convs = lat[::-1], out[::-1]
        self.top_block = None
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in strides}
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x, raw_norm=None):
        _require_gpu(x, "FPN")
        feats = self.bottom_up(x) if raw_norm is None else self.bottom_up(x, raw_norm=raw_norm)
        results, prev = [], None
        if torch.is_grad_enabled() and any(p.requires_grad for c in self.lateral_convs + self.output_convs for p in c.parameters()):
            from orehip import autograd as A
            for idx, (lateral, output) in enumerate(zip(self.lateral_convs, self.output_convs)):
                f = nhwc_view(feats[self.in_features[-idx - 1]])
                # lateral 1x1 + bias with F.interpolate(scale_factor=2, "nearest") + sum (fpn.py:136-141) fused in its epilogue;
                # the backward of the add is a 2x2 sum-pool of the lateral's gradient (ore_sumpool2x2_fwd)
                prev = A.conv(f, lateral.weight, lateral.bias, None, None, False, prev)
                results.insert(0, A.conv(prev, output.weight, output.bias).permute(0, 3, 1, 2))
            return dict(zip(self._out_features, results))
        for idx, (lateral, output) in enumerate(zip(self.lateral_convs, self.output_convs)):
            f = nhwc_view(feats[self.in_features[-idx - 1]])
            prev = lateral.forward_nhwc(f, add=prev)          # conv + bias (+ nearest2x(prev)) in one launch
            results.insert(0, output.forward_nhwc(prev).permute(0, 3, 1, 2))
        return dict(zip(self._out_features, results))
