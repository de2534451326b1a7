"""Layer wrappers with the reference's names (d2z:layers/{wrappers,batch_norm,shape_spec,nms}.py), computing on the
hand-written HIP kernels (orehip -> libore_hip.so) whenever the tensor lives on the GPU.  A CPU tensor is rejected:
there is no CPU fallback in the product path."""
from __future__ import annotations

from collections import namedtuple
from typing import List, Optional

import torch
from torch import nn


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


def cat(tensors: List[torch.Tensor], dim: int = 0):
    assert isinstance(tensors, (list, tuple))
    return tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)


def _require_gpu(x: torch.Tensor, what: str):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: the product path runs on the MI355X HIP kernels only (got a {x.device} tensor); "
                           "the CPU restatement lives under oracle/ and is test infrastructure")


def nhwc_view(x_nchw: torch.Tensor) -> torch.Tensor:
    """Logical NCHW -> contiguous NHWC fp32 (zero copy if the tensor is already channels_last)."""
    y = x_nchw.permute(0, 2, 3, 1)
    return y if (y.is_contiguous() and y.dtype == torch.float32) else y.contiguous().float()


class FrozenBatchNorm2d(nn.Module):
    """Fixed statistics + affine (d2z:layers/batch_norm.py:13-66).  On the HIP path it is folded into the producing
    conv's epilogue (scale = w*rsqrt(var+eps), shift = b - mean*scale); standalone forward applies the same affine."""
    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        """(scale, shift) of the frozen affine, computed once per state of the four buffers (their version counters and addresses): a
        training step used to spend five element-wise launches per FrozenBN and backbone pass on these constants."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b._version, b.data_ptr()) for b in bufs)
        c = self.__dict__.get("_ore_scale_shift")
        if c is None or c[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                shift = self.bias - self.running_mean * scale
            c = self.__dict__["_ore_scale_shift"] = (key, scale, shift)
        return c[1], c[2]

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)

    def __repr__(self):
        return f"FrozenBatchNorm2d(num_features={self.num_features}, eps={self.eps})"


def get_norm(norm, out_channels):
    if norm is None or (isinstance(norm, str) and len(norm) == 0):
        return None
    if isinstance(norm, str):
        norm = {"FrozenBN": FrozenBatchNorm2d, "GN": lambda c: nn.GroupNorm(32, c), "BN": nn.BatchNorm2d,
                "SyncBN": nn.SyncBatchNorm}[norm]
    return norm(out_channels)


class _PackedWeight:
    """Cache of the MFMA-packed copy of a conv weight, refreshed when the parameter changes."""

    def __init__(self):
        self.key = None
        self.packed = None

    def get(self, w: torch.Tensor):
        import orehip
        key = (w.data_ptr(), w._version, str(w.device))
        if key != self.key:
            self.packed = orehip.pack_conv_weight(w)
            self.key = key
        return self.packed


class Conv2d(nn.Conv2d):
    """nn.Conv2d + optional norm + activation (d2z:layers/wrappers.py:48-91) on the implicit-GEMM MFMA kernel.
    Supported on the HIP path: groups=1, dilation=1, square kernel, Cin % 16 == 0, norm in {None, FrozenBN},
    activation in {None, relu}."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation
        self._pw = _PackedWeight()

    def hip_params(self):
        """(packed_w, scale, shift) for the fused epilogue."""
        w = self._pw.get(self.weight)
        if isinstance(self.norm, FrozenBatchNorm2d):
            scale, shift = self.norm.scale_shift()
            if self.bias is not None:
                shift = shift + self.bias * scale
            return w, scale.contiguous(), shift.contiguous()
        assert self.norm is None, "only FrozenBN can be folded into the HIP conv epilogue"
        return w, None, (self.bias.detach() if self.bias is not None else None)

    def forward_nhwc(self, x_nhwc, **kw):
        import orehip
        assert self.groups == 1 and self.dilation == (1, 1) and self.kernel_size[0] == self.kernel_size[1]
        w, scale, shift = self.hip_params()
        relu = self.activation is not None
        if relu:
            assert self.activation in (torch.relu, torch.nn.functional.relu, torch.relu_) or isinstance(self.activation, nn.ReLU)
        return orehip.conv2d(x_nhwc, w, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0], scale=scale,
                             shift=shift, relu_cout=self.out_channels if relu else 0, **kw)

    def forward(self, x):
        _require_gpu(x, "Conv2d")
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # training: the autograd bindings (forward + data/weight-gradient kernels); stride-1 'same' convolutions only
            from orehip import autograd as A
            k = self.kernel_size[0]
            if not (self.groups == 1 and self.dilation == (1, 1) and self.stride == (1, 1) and self.padding == (k // 2, k // 2)
                    and self.kernel_size[1] == k and self.in_channels % 16 == 0):
                raise NotImplementedError("the HIP backward kernels cover stride-1 'same' convolutions with Cin % 16 == 0 "
                                          "(every trainable conv of finetune_vovnet.yaml)")
            relu = self.activation is not None
            if isinstance(self.norm, FrozenBatchNorm2d):
                assert self.bias is None, "conv + FrozenBN has no bias in the reference"
                sc, sh = self.norm.scale_shift()
                y = A.conv(nhwc_view(x), self.weight, None, sc.contiguous(), sh.contiguous(), relu)
            else:
                assert self.norm is None, "only FrozenBN can be folded into the HIP conv epilogue"
                y = A.conv(nhwc_view(x), self.weight, self.bias, None, None, relu)
            return y.permute(0, 3, 1, 2)
        return self.forward_nhwc(nhwc_view(x)).permute(0, 3, 1, 2)


def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float):
    """d2z:layers/nms.py:10-30 on the HIP bitmask NMS: class-offset trick then one NMS."""
    import orehip
    _require_gpu(boxes, "batched_nms")
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return orehip.nms(boxes.float() + offsets[:, None], scores, float(iou_threshold))


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float):
    import orehip
    _require_gpu(boxes, "nms")
    return orehip.nms(boxes, scores, float(iou_threshold))


__all__ = ["ShapeSpec", "cat", "FrozenBatchNorm2d", "get_norm", "Conv2d", "batched_nms", "nms", "nhwc_view"]
