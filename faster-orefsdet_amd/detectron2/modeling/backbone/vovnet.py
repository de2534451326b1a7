"""VoVNet-eSE backbone on the HIP kernels (module tree / parameter names of d2z:modeling/backbone/vovnet.py so
reference checkpoints load: `stem.stem_1/conv.weight`, `stage3.OSA3_1.layers.0.OSA3_1_0/norm.running_var`, ...).

Compute is NHWC: each OSA block owns ONE concat buffer, its 3x3 layers write channel slices of it (no torch.cat),
the eSE gate is folded into the consumers where possible, the max-pool is ceil-mode like the reference."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn

from detectron2.layers import FrozenBatchNorm2d, ShapeSpec, get_norm, nhwc_view, _require_gpu
from . import BACKBONE_REGISTRY, Backbone

# d2z:modeling/backbone/vovnet.py:28-96 (non-depthwise bodies; the *-dw-* bodies are not on the path)
_STAGE_SPECS = {
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                          layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 2, 2]),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 4, 3]),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 3, 9, 3]),
}


def _conv_bn_relu(name, cin, cout, k, stride, norm):
    return [(f"{name}/conv", nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)),
            (f"{name}/norm", get_norm(norm, cout)), (f"{name}/relu", nn.ReLU(inplace=True))]


class _Unit(nn.Sequential):
    """conv (no bias) + FrozenBN + ReLU as ONE fused HIP launch."""

    def __init__(self, name, cin, cout, k, stride, norm):
        super().__init__(OrderedDict(_conv_bn_relu(name, cin, cout, k, stride, norm)))
        self._n = name
        self._cache = None
        self._cache_bf16 = None

    def parts(self):
        return self._modules[self._n + "/conv"], self._modules[self._n + "/norm"]

    def hip(self, x_nhwc, **kw):
        import orehip
        conv, bn = self.parts()
        assert isinstance(bn, FrozenBatchNorm2d), "the HIP path folds FrozenBN only (MODEL.VOVNET.NORM=FrozenBN)"
        key = (conv.weight.data_ptr(), conv.weight._version, bn.weight._version, bn.running_var._version, str(conv.weight.device))
        if self._cache is None or self._cache[0] != key:
            sc, sh = bn.scale_shift()
            w = orehip.pack_conv_weight(conv.weight)
            # 3x3 stride-1 layers with 64 / 128 input channels also get the Winograd form of their weights: the library then runs the
            # large-M launches (stem_2, stage 2; every frozen layer of a training step) on k_conv3x3_wino
            wino_ok = conv.kernel_size[0] == 3 and conv.stride[0] == 1 and orehip.winograd_covers(conv.out_channels, conv.in_channels)
            U = orehip.winograd_weight(w, conv.out_channels, conv.in_channels) if wino_ok and w.is_cuda else None
            self._cache = (key, w, sc.contiguous(), sh.contiguous(), U)
        _, w, sc, sh, U = self._cache
        if x_nhwc.dtype == torch.bfloat16:              # bf16 STORAGE: bf16 map in, bf16 map out, bf16 weights, fp32 FrozenBN epilogue
            if self._cache_bf16 is None or self._cache_bf16[0] != key:
                self._cache_bf16 = (key, orehip.pack_conv_weight_bf16(conv.weight))
            return orehip.conv2d(x_nhwc, self._cache_bf16[1], conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0],
                                 scale=sc, shift=sh, relu_cout=conv.out_channels, **kw)
        return orehip.conv2d(x_nhwc, w, conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0], scale=sc,
                             shift=sh, relu_cout=conv.out_channels, w_wino=U, **kw)


class eSEModule(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.fc = nn.Conv2d(channel, channel, kernel_size=1, padding=0)

    def gate(self, x_nhwc):
        import orehip
        return orehip.ese_gate(x_nhwc, self.fc.weight.detach().contiguous(), self.fc.bias.detach().contiguous())


class _OSA_module(nn.Module):
    def __init__(self, in_ch, stage_ch, concat_ch, layer_per_block, module_name, identity, norm):
        super().__init__()
        self.identity = identity
        self.in_ch, self.stage_ch, self.n = in_ch, stage_ch, layer_per_block
        self.layers = nn.ModuleList()
        c = in_ch
        for i in range(layer_per_block):
            self.layers.append(_Unit(f"{module_name}_{i}", c, stage_ch, 3, 1, norm))
            c = stage_ch
        self.cat_ch = in_ch + layer_per_block * stage_ch
        self.concat = _Unit(f"{module_name}_concat", self.cat_ch, concat_ch, 1, 1, norm)
        self.ese = eSEModule(concat_ch)

    def hip(self, cat_buf):
        """cat_buf [B,H,W,cat_ch] with the block input already in channels [0,in_ch). Returns (pre-gate out, gate)."""
        src, dst = 0, self.in_ch
        for i, layer in enumerate(self.layers):
            layer.hip(cat_buf, in_coff=src, Cin=self.in_ch if i == 0 else self.stage_ch, out=cat_buf, out_coff=dst)
            src, dst = dst, dst + self.stage_ch
        y = self.concat.hip(cat_buf)
        return y, self.ese.gate(y)


class _OSA_stage(nn.Sequential):
    def __init__(self, in_ch, stage_ch, concat_ch, block_per_stage, layer_per_block, stage_num, norm):
        super().__init__()
        if stage_num != 2:
            self.add_module("Pooling", nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True))
        self.add_module(f"OSA{stage_num}_1", _OSA_module(in_ch, stage_ch, concat_ch, layer_per_block, f"OSA{stage_num}_1", False, norm))
        for i in range(block_per_stage - 1):
            name = f"OSA{stage_num}_{i + 2}"
            self.add_module(name, _OSA_module(concat_ch, stage_ch, concat_ch, layer_per_block, name, True, norm))
        self.pool = stage_num != 2

    def blocks(self):
        return [m for m in self.children() if isinstance(m, _OSA_module)]


class VoVNet(Backbone):
    def __init__(self, cfg, input_ch, out_features=None):
        super().__init__()
        norm = cfg.MODEL.VOVNET.NORM
        spec = _STAGE_SPECS[cfg.MODEL.VOVNET.CONV_BODY]
        self.spec = spec
        stem_ch = spec["stem"]
        self._out_features = out_features
        stem = _conv_bn_relu("stem_1", input_ch, stem_ch[0], 3, 2, norm)
        stem += _conv_bn_relu("stem_2", stem_ch[0], stem_ch[1], 3, 1, norm)
        stem += _conv_bn_relu("stem_3", stem_ch[1], stem_ch[2], 3, 2, norm)
        self.add_module("stem", nn.Sequential(OrderedDict(stem)))
        self._stem_units = None
        stride = 4
        self._out_feature_strides = {"stem": stride, "stage2": stride}
        self._out_feature_channels = {"stem": stem_ch[2]}
        in_ch = [stem_ch[2]] + spec["stage_out_ch"][:-1]
        self.stage_names = []
        for i in range(4):
            name = f"stage{i + 2}"
            self.stage_names.append(name)
            self.add_module(name, _OSA_stage(in_ch[i], spec["stage_conv_ch"][i], spec["stage_out_ch"][i],
                                             spec["block_per_stage"][i], spec["layer_per_block"], i + 2, norm))
            self._out_feature_channels[name] = spec["stage_out_ch"][i]
            if i != 0:
                stride *= 2
                self._out_feature_strides[name] = stride
        self._freeze_backbone(cfg.MODEL.BACKBONE.FREEZE_AT)

    def _freeze_backbone(self, freeze_at):
        """d2z vovnet.py:455-468: stage 0 = stem, stage k = stage(k+1)."""
        for idx in range(max(freeze_at, 0)):
            m = self.stem if idx == 0 else getattr(self, f"stage{idx + 1}")
            for p in m.parameters():
                p.requires_grad = False

    # ---- HIP forward -------------------------------------------------------------------------------------
    # "bf16": the FROZEN stages of a training forward keep their maps and weights in bf16 (BASELINE configs[4]: "bf16 MFMA conv path"):
    # stem_1 writes bf16, every frozen conv runs on the bf16-storage kernels (v_mfma_f32_16x16x32_bf16), max-pool and eSE read bf16;
    # FrozenBN, the eSE pool / gate and all accumulation stay fp32.  What crosses into the trainable stages is converted to fp32 once.
    # "auto" (default): bf16 storage exactly when the process-wide conv precision is the bf16 mode (orehip.set_conv_precision("bf16"),
    # which is how a bf16 training run is selected); "fp32" / "bf16" force it.
    frozen_storage = "auto"

    def stem_hip(self, x_nchw, raw_norm=None, out_bf16=False):
        """Normalised, padded NCHW input -> stem_1 output (the 3-channel kernel).  raw_norm = (mean, std, Hp, Wp): x_nchw is the RAW
        image batch (uint8 or fp32, all of one size) and the kernel fuses (x - mean) / std and the zero padding to Hp x Wp
        (ref:fewx/modeling/fsod/fsod_cen.py:540-555 + ImageList.from_tensors) -- no normalised copy of the batch is ever written."""
        import orehip
        s = self.stem
        c1, b1 = s._modules["stem_1/conv"], s._modules["stem_1/norm"]
        sc, sh = b1.scale_shift()
        B, _, H, W = x_nchw.shape
        if raw_norm is not None:
            mean, std, Hp, Wp = raw_norm
            x = x_nchw.contiguous()
            if x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            return orehip.stem1(x, Hp, Wp, mean, std, c1.weight.detach().contiguous(), sc.contiguous(), sh.contiguous(), out_bf16=out_bf16)
        y = orehip.stem1(x_nchw.contiguous().float(), H, W, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), c1.weight.detach().contiguous(),
                         sc.contiguous(), sh.contiguous(), out_bf16=out_bf16)
        return y

    def _unit(self, name):
        if self._stem_units is None:
            self._stem_units = {}
        if name not in self._stem_units:
            conv, bn = self.stem._modules[name + "/conv"], self.stem._modules[name + "/norm"]
            u = _Unit.__new__(_Unit)
            nn.Module.__init__(u)
            u._n, u._cache, u._cache_bf16 = name, None, None
            u._modules[name + "/conv"], u._modules[name + "/norm"] = conv, bn
            self._stem_units[name] = u
        return self._stem_units[name]

    def forward(self, x, raw_norm=None):
        """Frozen stages run as plain HIP launches; stages with trainable parameters (FREEZE_AT < stage) run through the autograd
        bindings (orehip.autograd.OSAFn: forward + data/weight-gradient kernels) when gradients are enabled.
        raw_norm: see stem_hip (the fused-preprocess entry used by the training forward)."""
        import orehip
        _require_gpu(x, "VoVNet")
        grad_on = torch.is_grad_enabled()
        outputs = {}
        # bf16 storage for the frozen stages: only when something trainable follows (a training forward) -- eval goes through the engine
        want16 = self.frozen_storage == "bf16" or (self.frozen_storage == "auto" and orehip.get_conv_precision() == "bf16")
        st16 = want16 and grad_on and not any(p.requires_grad for p in self.stem.parameters())
        fdt = torch.bfloat16 if st16 else torch.float32

        def new_cat(B_, H_, W_, ch):
            # bf16 rows carry 64 bytes of zeroed slack: a 64-byte K chunk that starts inside the last 16 channels of the last row reads
            # past it against zero weights (include/ore_hip.h, ORE_ST_BF16)
            if not st16:
                return torch.empty(B_, H_, W_, ch, device=x.device, dtype=torch.float32)
            # zeroed: a chunk that starts inside the last 16 channels of a slice also reads the first channels of the NEXT slice -- which
            # the block has not written yet -- against zero weights, and 0 x (a NaN bit pattern of fresh memory) would be NaN
            flat = torch.zeros(B_ * H_ * W_ * ch + 32, device=x.device, dtype=torch.bfloat16)
            return flat[:-32].view(B_, H_, W_, ch)
        with torch.no_grad():
            y = self.stem_hip(x, raw_norm, out_bf16=st16)
            y = self._unit("stem_2").hip(y)
            first = getattr(self, "stage2").blocks()[0]
            B, H, W, _ = y.shape
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            cat = new_cat(B, Ho, Wo, first.cat_ch)
            self._unit("stem_3").hip(y, out=cat, out_coff=0)
        if "stem" in self._out_features:
            outputs["stem"] = cat[..., : first.in_ch].float().permute(0, 3, 1, 2)
        prev, gate, full = None, None, None      # prev = pre-gate block output (frozen path), full = gated output (train path)
        for name in self.stage_names:
            stage = getattr(self, name)
            blocks = stage.blocks()
            train = grad_on and any(p.requires_grad for p in stage.parameters())
            if train:
                if prev is not None and prev.dtype == torch.bfloat16:      # the frozen / trainable seam: fp32 from here on
                    with torch.no_grad():
                        if stage.pool:
                            prev, gate = orehip.maxpool3x3s2(prev, gate).float(), None     # round(max(x) * g) once, as a stored bf16 map
                            stage_pooled = True
                        else:
                            prev = prev.float()
                            stage_pooled = False
                    full = self._stage_train(stage, blocks, None, prev, gate, None, pooled=stage_pooled)
                    prev, gate = None, None
                    if name in self._out_features:
                        outputs[name] = full.permute(0, 3, 1, 2)
                    continue
                full = self._stage_train(stage, blocks, cat if prev is None and full is None else None, prev, gate, full)
                prev, gate = None, None
                if name in self._out_features:
                    outputs[name] = full.permute(0, 3, 1, 2)
                continue
            assert full is None, "a frozen stage after a trainable one is not a configuration of the reference (FREEZE_AT)"
            with torch.no_grad():
                for bi, blk in enumerate(blocks):
                    if prev is not None:
                        if bi == 0 and stage.pool:
                            B = prev.shape[0]
                            H, W = orehip.maxpool3x3s2_out_hw(prev.shape[1], prev.shape[2])
                            cat = new_cat(B, H, W, blk.cat_ch)
                            # gate folded (max commutes with the positive scale); pooled straight into the concat buffer's first slice
                            orehip.maxpool3x3s2(prev, gate, out=cat, out_coff=0)
                            ident = None
                        else:
                            assert prev.dtype == torch.float32, "identity blocks (V-39 / V-57) stay on the fp32 path"
                            fullp = orehip.scale_channels(prev, gate) if gate is not None else prev
                            cat = torch.empty(*fullp.shape[:3], blk.cat_ch, device=x.device, dtype=torch.float32)
                            cat[..., : blk.in_ch] = fullp
                            ident = fullp
                    y, g = blk.hip(cat)
                    if blk.identity:
                        y = orehip.scale_channels(y, g) + ident
                        g = None
                    prev, gate = y, g
                if name in self._out_features:
                    if prev.dtype == torch.bfloat16:               # the gated map leaves the frozen part as fp32: bf16 value x fp32 gate
                        fullp = prev.float() * gate[:, None, None, :] if gate is not None else prev.float()
                    else:
                        fullp = orehip.scale_channels(prev, gate) if gate is not None else prev
                    outputs[name] = fullp.permute(0, 3, 1, 2)
        return outputs

    def _stage_train(self, stage, blocks, cat0, prev, gate, full, pooled=False):
        """One trainable stage: max-pool, OSA blocks through OSAFn, eSE.  The input comes either from a frozen stage
        (prev, gate: no gradient needed) or from the previous trainable stage (`full`, carries gradient)."""
        import orehip
        import torch.nn.functional as F
        from orehip import autograd as A
        for bi, blk in enumerate(blocks):
            if bi == 0:
                if full is not None:
                    x_in = A.maxpool(full) if stage.pool else full
                elif prev is not None and pooled:
                    x_in = prev                                    # already pooled (and gated) on the bf16 side of the seam
                elif prev is not None:
                    with torch.no_grad():
                        x_in = orehip.maxpool3x3s2(prev, gate) if stage.pool else (orehip.scale_channels(prev, gate) if gate is not None else prev)
                else:
                    x_in = cat0[..., : blk.in_ch].contiguous()
            else:
                x_in = full
            layers = []
            for unit in list(blk.layers) + [blk.concat]:
                conv, bn = unit.parts()
                sc, sh = bn.scale_shift()
                layers.append((conv.weight, sc.contiguous(), sh.contiguous()))
            y = A.osa_block(x_in, layers)
            out = A.ese(y, blk.ese.fc.weight, blk.ese.fc.bias)                                   # eSE (vovnet.py:238-260)
            full = out + x_in if blk.identity else out
        return full


@BACKBONE_REGISTRY.register()
def build_vovnet_backbone(cfg, input_shape):
    return VoVNet(cfg, input_shape.channels, out_features=cfg.MODEL.VOVNET.OUT_FEATURES)


@BACKBONE_REGISTRY.register()
def build_fcos_vovnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    """d2z vovnet.py:527-555; only TOP_LEVELS=0 (no p6/p7) is on the path."""
    from .fpn import FPN
    assert cfg.MODEL.FCOS.TOP_LEVELS == 0, "MODEL.FCOS.TOP_LEVELS > 0 (p6/p7) is outside the built path"
    bottom_up = build_vovnet_backbone(cfg, input_shape)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=None, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)


@BACKBONE_REGISTRY.register()
def build_vovnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    raise NotImplementedError("build_vovnet_fpn_backbone (LastLevelMaxPool top block) is outside the built path; "
                              "use build_fcos_vovnet_fpn_backbone")
